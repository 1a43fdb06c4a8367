"""Clip-encoder attention backward, a few launches (for rocprofv3 --pmc passes)."""
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
qkv = torch.randn(T * L, 3 * D, device=dev).bfloat16().requires_grad_(True)
km = torch.ones(T * L, device=dev)
for _ in range(5):
    qkv.grad = None
    out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=(0.1, rng, rng.site()))
    out.backward(torch.randn_like(out))
torch.cuda.synchronize()
