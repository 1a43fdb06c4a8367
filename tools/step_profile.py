"""Summarise one replayed training step from a rocprofv3 kernel trace: per-kernel totals, phase timeline.
usage: python tools/step_profile.py gpurun_out/prof_x/NAME_kernel_trace.csv [top_n]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("opt_adam")]
a, b = idx[-6], idx[-5]                      # a graph replay inside the timed region (the last 3 steps are instrumented eager ones)
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernels %d  span %.3f ms  sum %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(dur(r) for r in step) / 1e6))
g = collections.defaultdict(lambda: [0, 0])
for r in step:
    k = (r["Kernel_Name"].replace("void ", "")[:72], r["Grid_Size_X"])
    g[k][0] += dur(r); g[k][1] += 1
for k, v in sorted(g.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%8.1f us %4d x %7.1f  grid %8s  %s" % (v[0] / 1e3, v[1], v[0] / 1e3 / v[1], k[1], k[0]))
marks = [("clip-encoder fwd", "ln_fwd_kernelILi12"), ("step/sim/decoder fwd", "sim_recur_fwd"), ("pointer+loss", "ptr_attn_fwd"),
         ("reconstruct fwd", "gumbel"), ("backward starts", "sim_recur_bwd"), ("pointer bwd", "ptr_attn_bwd"),
         ("visual simulator bwd", "sim_recur_bwd"), ("clip-encoder bwd", "attn_mfma_bwd_kernelILi64EDF16bLi128"), ("video-embed LN bwd", "ln_bwd_kernelILi12"),
         ("optimizer", "opt_sumsq")]
seen = collections.Counter()
for r in step:
    for label, m in marks:
        if m in r["Kernel_Name"]:
            seen[(label, m)] += 1
            want = 2 if label == "visual simulator bwd" else 1
            if seen[(label, m)] == want and not (label == "backward starts" and False):
                print("t=%7.3f ms  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, label))
