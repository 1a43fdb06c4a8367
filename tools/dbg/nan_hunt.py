"""debug: which eager training step of the c1 model first produces a non-finite loss / gradient / weight"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_model
from svpc_amd import synthetic as syn, ops
from svpc_amd.graph import backward_all, ops_stream
from svpc_amd.optim import FusedBertAdam
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 2e-3
z, cfg, batch, model = build_model("c1", "vivt", os.path.join(ROOT, "tests", "golden"), "cuda:0")
args = syn.forward_args(batch)
with torch.cuda.stream(ops_stream()):
    opt = FusedBertAdam(list(model.named_parameters()), lr=lr, weight_decay=0.0, grad_clip=1.0, ema_decay=0.9)
    model.train()
    for it in range(6):
        opt.zero_grad()
        loss = model(*args)[0]
        if it == int(os.environ.get("ANOMALY_AT", "-1")):
            good = [n for n, p in model.named_parameters() if p.grad is not None]
            with torch.autograd.detect_anomaly():
                try:
                    backward_all(model, loss)
                except RuntimeError as e:
                    print("ANOMALY:", str(e)[:600], flush=True)
                    break
        else:
            backward_all(model, loss)
        ops.join_side()
        torch.cuda.synchronize()
        bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        opt.step()
        torch.cuda.synchronize()
        bad_w = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        gn = float(opt.grad_norm()) if opt.arena is not None else -1
        print("step", it, "loss", float(loss), "gnorm", gn, "bad grads", bad_g[:6], len(bad_g), "bad weights", bad_w[:6], len(bad_w), flush=True)
