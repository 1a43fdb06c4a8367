"""ragged steps (clip-encoder / decoder graphs on unless GRAPHS=0) for a kernel trace: rocprofv3 --kernel-trace -- python3 tools/dbg/ragged_trace.py
then: python3 tools/dbg/ragged_trace.py --gaps TRACE.csv"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--gaps":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("opt_adam")]
    for a, b in zip(idx[-5:-1], idx[-4:]):
        step = rows[a + 1:b + 1]
        t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
        print("step: %d kernels, span %.3f ms, busy %.3f ms" % (len(step), (t1 - t0) / 1e6, busy / 1e6))
        for i in range(len(step) - 1):
            g = int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"])
            if g > 40000:
                print("   gap %7.1f us at %7.3f ms after %-50s before %s" % (g / 1e3, (int(step[i]["End_Timestamp"]) - t0) / 1e6,
                      step[i]["Kernel_Name"].replace("void ", "")[:50], step[i + 1]["Kernel_Name"].replace("void ", "")[:50]))
    import collections
    g = collections.defaultdict(lambda: [0, 0])
    for r in step:
        k = r["Kernel_Name"].replace("void ", "")[:80]
        g[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); g[k][1] += 1
    print("last step by kernel:")
    for k, v in sorted(g.items(), key=lambda kv: -kv[1][0])[:45]:
        print("%8.1f us %4d x %7.1f  %s" % (v[0] / 1e3, v[1], v[0] / 1e3 / v[1], k))
    sys.exit(0)
import torch
import bench
from svpc_amd import ops, synthetic as syn, clip_graphs
from svpc_amd.optim import FusedBertAdam
from svpc_amd.graph import backward_all
args = bench.parse_args([])
dev = torch.device("cuda:0")
ops.set_precision("bf16x3")
cfg, model = bench.build(args, dev, model_type="vivt")
model.train()
rb, clips = bench.ragged_batches(cfg, args, dev, 8)
rargs = [syn.forward_args(b) for b in rb]
opt = FusedBertAdam(list(model.named_parameters()), lr=1e-4, warmup=0.1, t_total=100000, weight_decay=0.01, grad_clip=1.0)
st = torch.cuda.Stream()
def step(k):
    model._plans.clear(); model._ptr_plans.clear(); model._span_cache.clear()
    opt.zero_grad(); loss = model(*rargs[k % 8])[0]; backward_all(model, loss); opt.step(); return loss
with torch.cuda.stream(st):
    for k in range(3): step(k)
    if os.environ.get("GRAPHS", "1") != "0":
        clip_graphs.enable(model)
    for k in range(8): step(k)
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(16): step(k)
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print("ragged: host enqueue %.2f ms/step, wall %.2f ms/step, clips %s" % ((t1 - t0) / 16 * 1e3, (t2 - t0) / 16 * 1e3, clips))
