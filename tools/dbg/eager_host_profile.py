"""where the host time of an EAGER training step goes (cProfile over 5 steps of the headline workload, no graph)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from svpc_amd import ops, synthetic as syn, make_batch
from svpc_amd.optim import FusedBertAdam
from svpc_amd.graph import backward_all
args = bench.parse_args([])
dev = torch.device("cuda:0")
ops.set_precision("bf16x3")
cfg, model = bench.build(args, dev, model_type="vivt")
model.train()
b = make_batch(cfg, n_videos=16, max_steps=12, n_ingr=10, n_oov=0, seed=7, full_clips=True, device=dev)
fargs = syn.forward_args(b)
opt = FusedBertAdam(list(model.named_parameters()), lr=1e-4, warmup=0.1, t_total=100000, weight_decay=0.01, grad_clip=1.0)
st = torch.cuda.Stream()
def step():
    opt.zero_grad(); loss = model(*fargs)[0]; backward_all(model, loss); opt.step(); return loss
with torch.cuda.stream(st):
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5): step()
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print("eager: host enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): step()
    pr.disable(); torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(28)
