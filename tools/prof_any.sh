#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python command → gpurun_out/<name>/kernel_stats.csv     usage: tools/prof_any.sh NAME script.py args…
name="$1"; shift
root="$(pwd)"; out="$root/gpurun_out/$name"
mkdir -p "$out"; export TMPDIR=/tmp
script="$root/$1"; shift
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 "$script" "$@" > "$out/run.log" 2>&1
rc=$?
cd "$root"
stats=$(find "$out" -name "p_kernel_stats.csv" | head -1)
[ -n "$stats" ] && cp "$stats" "$out/kernel_stats.csv"
find "$out" -name "p_*" -delete; find "$out" -type d -empty -delete
grep -E "attn|gemm" "$out/kernel_stats.csv" | cut -d, -f1-4 | cut -c1-160
exit $rc
