"""Summarise one replayed training step from a rocprofv3 kernel trace: per-kernel totals, phase timeline.
usage: python tools/step_profile.py gpurun_out/prof_x/NAME_kernel_trace.csv [top_n]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("opt_adam")]
a, b = idx[-6], idx[-5]                      # a graph replay inside the timed region (the last 3 steps are instrumented eager ones)
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernels %d  span %.3f ms  sum %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(dur(r) for r in step) / 1e6))
gaps = [int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"]) for i in range(len(step) - 1)]
pos = sorted(g_ for g_ in gaps if g_ > 0) or [0]
print("gaps between consecutive kernels: total %.3f ms, median %.2f us, p90 %.2f us, max %.1f us; kernels shorter than 6 us: %d (sum %.3f ms)"
      % (sum(pos) / 1e6, pos[len(pos) // 2] / 1e3, pos[int(len(pos) * 0.9)] / 1e3, pos[-1] / 1e3,
         sum(1 for r in step if dur(r) < 6000), sum(dur(r) for r in step if dur(r) < 6000) / 1e6))
if "--seq" in sys.argv:      # dump the launch sequence (time, duration, gap before, grid, name)
    for i, r in enumerate(step):
        print("  %8.3f ms %7.1f us gap %5.1f  grid %7s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, dur(r) / 1e3,
              (gaps[i - 1] / 1e3 if i else 0.0), r["Grid_Size_X"], r["Kernel_Name"].replace("void ", "")[:90]))
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
