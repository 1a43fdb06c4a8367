"""Time the clip-encoder attention core (192 sequences × 100 tokens × 12 heads × 64) forward / backward, with and without dropout."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svpc_amd import ops as O
from svpc_amd.ops_common import SeqInfo
O.set_precision("bf16")
dev = torch.device("cuda")
T, L, D, H = 192, 100, 768, 12
seq = SeqInfo.uniform(T, L, L, dev)
rng = O.make_rng(dev)
for dt in (torch.bfloat16, torch.float32):
    qkv = torch.randn(T * L, 3 * D, device=dev).to(dt).requires_grad_(True)
    km = torch.ones(T * L, device=dev)
    for p in (0.1, 0.0):
        drop = (p, rng, rng.site()) if p > 0 else None
        out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=drop)
        g = torch.randn_like(out)
        out.backward(g)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 20
        tf = tb = 0.0
        for _ in range(n):
            qkv.grad = None
            e[0].record()
            out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=drop)
            e[1].record()
            out.backward(g)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        print("%s p=%.1f  fwd %.1f us  bwd %.1f us" % (str(dt).split(".")[-1], p, 1e3 * tf / n, 1e3 * tb / n))
