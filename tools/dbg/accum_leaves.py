"""Which parameters still receive their gradient through autograd's AccumulateGrad (an ATen add kernel each) instead of an
in-place arena write?  Prints their names for the bench model in bf16 mode; also lists the ATen/rocclr kernels of one step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from svpc_amd import ops, synthetic as syn
from svpc_amd.graph import backward_all
from svpc_amd.optim import FusedBertAdam

args = bench.parse_args(sys.argv[1:])
dev = torch.device("cuda:0")
ops.set_precision(args.precision)
cfg, model = bench.build(args, dev)
model.train()
batch = bench.device_batch(cfg, args, dev, seed=2019)
fargs = syn.forward_args(batch)
opt = FusedBertAdam(list(model.named_parameters()), lr=1e-4, warmup=0.1, t_total=100000, grad_clip=1.0)
for _ in range(2):
    opt.zero_grad(); loss = model(*fargs)[0]; backward_all(model, loss); opt.step()
fired = []
hs = [p.register_post_accumulate_grad_hook((lambda n: (lambda p_: fired.append(n)))(n)) for n, p in model.named_parameters()]
opt.zero_grad(); loss = model(*fargs)[0]; backward_all(model, loss); opt.step()
torch.cuda.synchronize()
print("AccumulateGrad fired for %d parameter tensors:" % len(fired))
for n in fired:
    print("   ", n, tuple(dict(model.named_parameters())[n].shape))
for h in hs:
    h.remove()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    opt.zero_grad(); loss = model(*fargs)[0]; backward_all(model, loss); opt.step()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0 and not e.key.startswith("svpc")]
rows.sort(key=lambda r: -r[2])
print("non-svpc device kernels / ops of one eager step:")
for k, c, t in rows[:40]:
    print("  %6d x %9.1f us  %s" % (c, t, k[:110]))
