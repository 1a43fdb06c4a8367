"""where the host time of an eager step over freshly-structured (ragged) batches goes: cProfile over 8 steps, plan caches cleared"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from svpc_amd import ops, synthetic as syn
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
if os.environ.get("GRAPHS", "1") != "0":
    from svpc_amd import clip_graphs
    clip_graphs.enable(model)
def step(k):
    model._plans.clear(); model._ptr_plans.clear(); model._span_cache.clear()
    opt.zero_grad(); loss = model(*rargs[k % 8])[0]; backward_all(model, loss); opt.step(); return loss
with torch.cuda.stream(st):
    for k in range(8): step(k)
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(8): step(k)
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print("ragged eager: host enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) / 8 * 1e3, (t2 - t0) / 8 * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for k in range(8): step(k)
    pr.disable(); torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(30); ps.sort_stats("cumtime").print_stats(40)
