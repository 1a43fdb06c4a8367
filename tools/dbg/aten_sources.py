"""Where do the remaining ATen device kernels of one training step come from?  Wraps the torch ops that launch kernels in a
TorchDispatchMode, records (op, shapes, innermost svpc_amd source line, forward|backward) for every call that touches a CUDA
tensor, prints them grouped.  Run on the GPU box: python tools/dbg/aten_sources.py"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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

NOKERNEL = ("view", "reshape", "as_strided", "detach", "alias", "empty", "slice", "select", "squeeze", "unsqueeze", "expand",
            "transpose", "permute", "t.default", "_unsafe_view", "narrow", "split", "unbind", "record_stream", "is_", "sym_",
            "_local_scalar", "size", "stride", "lift_fresh", "_to_copy_noop", "resize_", "set_", "result_type", "chunk")
seen = collections.Counter()
phase = ["fwd"]


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        short = name.replace("aten.", "")
        if any(short.startswith(k) for k in NOKERNEL):
            return out
        ts = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in ts):
            return out
        where = "?"
        for fr in reversed(traceback.extract_stack()):
            if "/svpc_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                break
        shp = ",".join("x".join(map(str, t.shape)) + ("b" if t.dtype == torch.bfloat16 else "") for t in ts[:3])
        seen[(phase[0], short, where, shp)] += 1
        return out


with Spy():
    opt.zero_grad()
    loss = model(*fargs)[0]
    phase[0] = "bwd"
    backward_all(model, loss)
    phase[0] = "opt"
    opt.step()
torch.cuda.synchronize()
print("%d kernel-launching ATen calls in one eager step (autograd-engine internals such as AccumulateGrad adds are listed under "
      "bwd with where='?')" % sum(seen.values()))
for (ph, op, where, shp), c in sorted(seen.items(), key=lambda kv: (kv[0][0], kv[0][2])):
    print("  %s %3d x %-28s %-46s %s" % (ph, c, op, where, shp))
