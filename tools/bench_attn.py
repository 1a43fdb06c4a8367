"""Clip-encoder attention (192 sequences x 100 rows x 12 heads x 64) forward / backward timings, with and without dropout.
usage (GPU box): python tools/bench_attn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svpc_amd import ops

dev = torch.device("cuda:0")
MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ops.set_precision(MODE)
B, L, H, dh = 192, 100, 12, 64
D = H * dh
torch.manual_seed(0)
if MODE == "bf16x3":
    qkv = ops.new_split(B * L, 3 * D, dev)
    src = 0.5 * torch.randn(B * L, 3 * D, device=dev)
    qkv.copy_(src)
    torch.as_strided(qkv, qkv.shape, qkv.stride(), qkv.storage_offset() + 3 * D).copy_(src - qkv.float())
    qkv.requires_grad_(True)
else:
    qkv = (0.5 * torch.randn(B * L, 3 * D, device=dev)).bfloat16().requires_grad_(True)
seq = ops.SeqInfo.uniform(B, L, L, dev)
km = torch.ones(B * L, device=dev)
rng = ops.default_rng(dev)
go = torch.randn(B * L, D, device=dev).bfloat16()


def run(p, iters=30):
    drop = (p, rng, 7) if p > 0 else None
    out = ops.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=drop)
    out.backward(go)
    torch.cuda.synchronize()
    ef = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(iters):
        qkv.grad = None
        ef[0].record()
        out = ops.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=drop)
        ef[1].record()
        out.backward(go)
        ef[2].record()
        torch.cuda.synchronize()
        tf += ef[0].elapsed_time(ef[1]); tb += ef[1].elapsed_time(ef[2])
    return tf / iters * 1e3, tb / iters * 1e3


for p in (0.0, 0.1, 0.0, 0.1):
    f, b = run(p)
    print("p_drop %.1f: fwd %.1f us  bwd %.1f us (event-bracketed, includes the autograd glue of one node)" % (p, f, b))
