import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import test_model_gpu as T
from svpc_amd import ops, synthetic as syn
cfg, model, batch = T._stream_model(n_videos=4, steps=4, max_t_len=8)
names = [n for n, p in model.named_parameters()]
res = {}
for mode, stream in (("fp32", False), ("bf16", False), ("bf16", True)):
    ops.set_precision(mode); ops.BF16_STREAM = stream
    model.zero_grad()
    loss = model(*syn.forward_args(batch))[0]
    loss.backward()
    res[(mode, stream)] = (loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
ops.set_precision("fp32"); ops.BF16_STREAM = True
ref = res[("fp32", False)]
print("loss", [res[k][0] for k in res])
for n in ref[1]:
    r = ref[1][n]
    a = res[("bf16", False)][1][n]; b = res[("bf16", True)][1][n]
    ea = ((a - r).norm() / (r.norm() + 1e-12)).item(); eb = ((b - r).norm() / (r.norm() + 1e-12)).item()
    ma = ((a - r).abs().max() / (r.abs().max() + 1e-12)).item(); mb = ((b - r).abs().max() / (r.abs().max() + 1e-12)).item()
    if eb > 0.05 or mb > 0.08:
        print("%-55s fro %.3f -> %.3f   max %.3f -> %.3f" % (n, ea, eb, ma, mb))
