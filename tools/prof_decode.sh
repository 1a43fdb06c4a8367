#!/bin/bash
# rocprofv3 kernel trace + stats of the greedy-decode bench (config 5) → gpurun_out/<name>/kernel_stats.csv, bench.log
name="$1"; shift
root="$(pwd)"; out="$root/gpurun_out/$name"
mkdir -p "$out"; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 "$root/bench.py" --decode "$@" > "$out/bench.log" 2>&1
rc=$?
cd "$root"
stats=$(find "$out" -name "p_kernel_stats.csv" | head -1)
[ -n "$stats" ] && cp "$stats" "$out/kernel_stats.csv"
find "$out" -name "p_*" -delete; find "$out" -type d -empty -delete
tail -n 2 "$out/bench.log" | cut -c1-400
exit $rc
