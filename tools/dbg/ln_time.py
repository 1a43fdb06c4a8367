"""Time LayerNorm forward / backward at the clip-encoder shape (19200 × 768, bf16 stream, residual, hidden dropout)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svpc_amd import ops as O
O.set_precision("bf16")
dev = torch.device("cuda")
rng = O.make_rng(dev)
for R, dt in ((19200, torch.bfloat16), (4224, torch.bfloat16), (4224, torch.float32), (192, torch.float32)):
    D = 768
    x = torch.randn(R, D, device=dev).to(dt).requires_grad_(True)
    res = torch.randn(R, D, device=dev).to(dt).requires_grad_(True)
    g = torch.ones(D, device=dev, requires_grad=True); b = torch.zeros(D, device=dev, requires_grad=True)
    for p in (0.1, 0.0):
        drop = (p, rng, rng.site()) if p > 0 else None
        y = O.layernorm(x, g, b, 1e-12, residual=res, pre_drop=drop)
        dy = torch.randn_like(y)
        y.backward(dy)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        n = 20
        for _ in range(n):
            x.grad = res.grad = g.grad = b.grad = None
            e[0].record()
            y = O.layernorm(x, g, b, 1e-12, residual=res, pre_drop=drop)
            e[1].record()
            y.backward(dy)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        print("R=%5d %s p=%.1f  fwd %.1f us  bwd %.1f us (host-inclusive)" % (R, str(dt).split(".")[-1], p, 1e3 * tf / n, 1e3 * tb / n))
