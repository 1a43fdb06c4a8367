"""Every GEMM launch of one eager training step by (M, N, K, layout, dtypes): count and event time (development tool)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from svpc_amd import ops, synthetic as syn
from svpc_amd.optim import FusedBertAdam
from svpc_amd.graph import backward_all

class A: pass
args = A(); args.__dict__.update(dict(batch=16, clips=12, layers=None, precision="bf16", model_type="vivt"))
import argparse
ap_defaults = bench.__dict__.get("DEFAULTS")
sys.argv = ["bench.py"]
device = torch.device("cuda", 0)
ops.set_precision("bf16")
# reuse bench's builders through its argparse defaults
import importlib
ns = argparse.Namespace(batch=16, clips=12, model_type="vivt", precision="bf16")
for k, v in dict(hidden=768, layers=6, heads=12, seed=0).items():
    setattr(ns, k, v)
try:
    cfg, model = bench.build(ns, device)
    batch = bench.device_batch(cfg, ns, device, seed=2019)
except Exception as e:
    print("adapt args:", e); raise
model.train()
fargs = syn.forward_args(batch)
opt = FusedBertAdam(list(model.named_parameters()), lr=1e-4, warmup=0.1, t_total=100000, grad_clip=1.0, ema_decay=-1.0)
def step():
    opt.zero_grad(); loss = model(*fargs)[0]; backward_all(model, loss); opt.ensure_built(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
ops.GEMM_TIMER = ops.KernelTimer()
step(); torch.cuda.synchronize()
t, ops.GEMM_TIMER = ops.GEMM_TIMER, None
agg = collections.defaultdict(lambda: [0, 0.0])
for w, e0, e1, d in t.records:
    agg[d][0] += 1; agg[d][1] += e0.elapsed_time(e1) * 1e3
print("launches", len(t.records))
for d, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%5d x %8.1f us (avg %6.1f)  M=%6d N=%5d K=%6d a_kc=%d b_kc=%d dt=%s" % (n, us, us / n, d[0], d[1], d[2], d[3], d[4], d[5:]))
