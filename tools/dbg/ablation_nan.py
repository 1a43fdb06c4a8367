"""debug: the bwd_ablation row (headline, seed 7, BWD_EXACT) whose final parameters were not finite — which tensor, which step"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import bwd_ablation as B
from svpc_amd import ops, synthetic as syn
from svpc_amd.graph import backward_all, ops_stream
from svpc_amd.optim import FusedBertAdam
import copy
cm = B.build("headline", 7)
cfg, model_cpu, batch, noise = cm
ops.set_precision("bf16x3"); ops.BWD_EXACT = True
model = copy.deepcopy(model_cpu).to("cuda:0"); model.eval()
model.gumbel_noise = [n.to("cuda:0") for n in noise]
fargs = syn.forward_args(B.to_dev(batch))
opt = FusedBertAdam(list(model.named_parameters()), lr=B.LR, warmup=B.WARMUP, t_total=B.T_TOTAL, weight_decay=B.WD, grad_clip=1.0)
with torch.cuda.stream(ops_stream()):
    for k in range(20):
        opt.zero_grad(); loss = model(*fargs)[0]; backward_all(model, loss); ops.join_side(); torch.cuda.synchronize()
        badg = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        opt.step(); torch.cuda.synchronize()
        badw = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        print(k, float(loss), "bad grads", badg[:5], len(badg), "bad weights", badw[:5], len(badw), flush=True)
        if badw: break
