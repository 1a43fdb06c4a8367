import os, sys, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from svpc_amd import ops, synthetic as syn
from svpc_amd.optim import FusedBertAdam
from svpc_amd.graph import backward_all
ns = argparse.Namespace(batch=16, clips=12, model_type="vivt", precision="bf16", hidden=768, layers=6, heads=12, seed=0)
device = torch.device("cuda", 0)
ops.set_precision("bf16")
cfg, model = bench.build(ns, device)
batch = bench.device_batch(cfg, ns, device, seed=2019)
model.train()
fargs = syn.forward_args(batch)
opt = FusedBertAdam(list(model.named_parameters()), lr=1e-4, warmup=0.1, t_total=100000, grad_clip=1.0, ema_decay=-1.0)
for i in range(2):
    opt.zero_grad(); loss = model(*fargs)[0]; backward_all(model, loss); opt.ensure_built(); opt.step()
    torch.cuda.synchronize()
    print("step", i, "loss", float(loss), "sink stats", ops.SINK_STATS, "USE", ops.USE_RES_SINK)
