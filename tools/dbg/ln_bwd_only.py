"""LayerNorm backward at the clip-encoder shape, a few launches (for rocprofv3 kernel-trace / --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svpc_amd import ops as O
O.set_precision("bf16")
dev = torch.device("cuda")
rng = O.make_rng(dev)
R, D = 19200, 768
x = torch.randn(R, D, device=dev).bfloat16().requires_grad_(True)
res = torch.randn(R, D, device=dev).bfloat16().requires_grad_(True)
g = torch.ones(D, device=dev, requires_grad=True); b = torch.zeros(D, device=dev, requires_grad=True)
for p in (0.1, 0.0):
    for _ in range(5):
        x.grad = res.grad = g.grad = b.grad = None
        y = O.layernorm(x, g, b, 1e-12, residual=res, pre_drop=(p, rng, rng.site()) if p > 0 else None)
        y.backward(torch.randn_like(y))
torch.cuda.synchronize()
