#!/bin/bash
# two PMC passes (FETCH_SIZE, WRITE_SIZE: they cannot share a pass on gfx950) over the bench command in the given mode, eager launches
# usage: tools/pmc_x3.sh MODE OUTNAME     → gpurun_out/OUTNAME.csv + profiles/dominant_gemm[_x3]_traffic.json (copy both back)
mode="$1"; name="$2"
root="$(pwd)"; out="$root/gpurun_out/pmc_$name"
mkdir -p "$out"; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/$c" -o p -- python3 "$root/bench.py" --precision "$mode" --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > "$out/$c.log" 2>&1 ) || exit 1
done
f=$(find "$out/FETCH_SIZE" -name "p_counter_collection.csv" | head -1)
w=$(find "$out/WRITE_SIZE" -name "p_counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$f" "$w" "gpurun_out/$name.csv" "$mode" || exit 1
cp profiles/dominant_gemm*_traffic.json gpurun_out/ 2>/dev/null
rm -rf "$out/FETCH_SIZE" "$out/WRITE_SIZE"
