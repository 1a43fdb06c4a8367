"""the clip-encoder attention forward in a loop for counter passes: python tools/dbg/pipe_loop.py [x3|bf16] [pipe 0/1] [dbg] [p_drop] [n]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svpc_amd import _lib, ops
x3 = (sys.argv[1] if len(sys.argv) > 1 else "x3") == "x3"
pipe = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dbg_bits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
n_it = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0"); H, dh = 12, 64; D = H * dh
lib = _lib.load(); rng = ops.default_rng(dev); st = torch.cuda.current_stream().cuda_stream
lib.svpc_attn_pipe_enable(pipe); lib.svpc_attn_pipe_debug(dbg_bits, None)
B, L = 192, 100
seq = ops.SeqInfo.uniform(B, L, L, dev); n = B * L; W = 3 * D
torch.manual_seed(1)
x = torch.randn(n, W, device=dev)
hi = x.bfloat16(); lo = (x - hi.float()).bfloat16()
qkv = torch.cat([hi, lo], 1).contiguous() if x3 else hi.contiguous()
out = torch.zeros(n, (2 if x3 else 1) * D, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, L, device=dev)
km = torch.ones(n, device=dev)
sc = 1 / math.sqrt(dh)
for _ in range(n_it):
    if x3:
        _lib.call("attn_x3_fwd", qkv.data_ptr(), 2 * W, W, qkv.data_ptr() + 2 * D, 2 * W, W, qkv.data_ptr() + 4 * D, 2 * W, W, out.data_ptr(), 2 * D, D,
                  lse.data_ptr(), seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, sc, p, 7, rng.seed.data_ptr(), st)
    else:
        _lib.call("attn_mfma_fwd_t", qkv.data_ptr(), W, qkv.data_ptr() + 2 * D, W, qkv.data_ptr() + 4 * D, W, out.data_ptr(), D, 1, lse.data_ptr(),
                  seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, sc, p, 7, rng.seed.data_ptr(), st)
torch.cuda.synchronize()
