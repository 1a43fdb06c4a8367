#!/bin/bash
# rocprofv3 kernel trace of a bench.py command → per-kernel stats csv + the step summary of tools/step_profile.py, under gpurun_out/<name>/
# usage: tools/prof_step.sh NAME [bench.py arguments…]        (run from the repo root on the GPU box)
name="$1"; shift
root="$(pwd)"
out="$root/gpurun_out/$name"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 "$root/bench.py" "$@" > "$out/bench.log" 2>&1
rc=$?
cd "$root"
trace=$(find "$out" -name "p_kernel_trace.csv" | head -1)
stats=$(find "$out" -name "p_kernel_stats.csv" | head -1)
if [ -n "$trace" ]; then
    python3 tools/step_profile.py "$trace" 70 > "$out/step_profile.txt" 2>&1
    python3 tools/step_profile.py "$trace" 0 --seq > "$out/seq.txt" 2>&1
    python3 tools/dominant_avg.py "$trace" > "$out/dominant_avg.txt" 2>&1
    cp "$stats" "$out/kernel_stats.csv" 2>/dev/null
    find "$out" -name "p_*" -delete          # the raw trace is tens of MB: keep the summaries only
    find "$out" -type d -empty -delete
fi
tail -n 3 "$out/bench.log"
exit $rc
