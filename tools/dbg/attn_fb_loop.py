"""clip-encoder attention forward (bf16x3) + backward (hi planes) through the C-ABI in a short loop — for counter passes.
usage: python tools/dbg/attn_fb_loop.py [n]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svpc_amd import _lib, ops
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0"); H, dh = 12, 64; D = H * dh
rng = ops.default_rng(dev); st = torch.cuda.current_stream().cuda_stream
B, L = 192, 100
seq = ops.SeqInfo.uniform(B, L, L, dev); n = B * L; W = 3 * D
torch.manual_seed(1)
x = torch.randn(n, W, device=dev)
hi = x.bfloat16()
qkv = torch.cat([hi, (x - hi.float()).bfloat16()], 1).contiguous()
out = torch.zeros(n, 2 * D, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, L, device=dev)
km = torch.ones(n, device=dev)
dO = torch.randn(n, D, device=dev).bfloat16(); dqkv = torch.empty(n, W, device=dev, dtype=torch.bfloat16)
sc = 1 / math.sqrt(dh)
for _ in range(n_it):
    _lib.call("attn_x3_fwd", qkv.data_ptr(), 2 * W, W, qkv.data_ptr() + 2 * D, 2 * W, W, qkv.data_ptr() + 4 * D, 2 * W, W, out.data_ptr(), 2 * D, D,
              lse.data_ptr(), seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, sc, 0.1, 7, rng.seed.data_ptr(), st)
    _lib.call("attn_mfma_bwd_t", qkv.data_ptr(), 2 * W, qkv.data_ptr() + 2 * D, 2 * W, qkv.data_ptr() + 4 * D, 2 * W, out.data_ptr(), 2 * D, 1,
              lse.data_ptr(), dO.data_ptr(), D, dqkv.data_ptr(), W, dqkv.data_ptr() + 2 * D, W, dqkv.data_ptr() + 4 * D, W,
              seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, sc, 0.1, 7, rng.seed.data_ptr(), st)
torch.cuda.synchronize()
