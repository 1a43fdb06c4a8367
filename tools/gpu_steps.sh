#!/bin/bash
# Run GPU steps one after another on the gpurun box; each step under its own `timeout -k 10`, output to gpurun_out/<name>.log.
# A step that fails is recorded and the next one still runs; a step that TIMES OUT (or is killed) ends the session: after a
# hung GPU step nothing else is started.      usage: tools/gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in "$@"; do
    name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "=== $name (limit ${secs}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out: stopping the session"; exit 1; fi
done
exit 0
