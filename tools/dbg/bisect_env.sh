T="tests/test_clip_graphs_gpu.py::test_bucketed_clip_counts_and_the_dp_reducer"
run() { echo "== $1"; env $1 timeout -k 10 200 python -m pytest $T -x -q 2>&1 | tail -1; }
run "SVPC_NOP=1"
run "SVPC_Q1R=0"
run "SVPC_L32_SKINNY_TS=32"
run "SVPC_LN_WIDE=0"
run "SVPC_LN_PARAM_GROUPS=1024"
