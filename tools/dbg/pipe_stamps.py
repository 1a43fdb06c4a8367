"""per-iteration cycle stamps of the pipelined attention forward (SVPC_PP_DBG=16): where the loader and compute wave 0 of a workgroup
spend a pair's time.   usage: python tools/dbg/pipe_stamps.py [x3|bf16] [extra dbg bits]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svpc_amd import _lib, ops
x3 = (sys.argv[1] if len(sys.argv) > 1 else "x3") == "x3"
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0"); H, dh = 12, 64; D = H * dh
lib = _lib.load(); rng = ops.default_rng(dev); st = torch.cuda.current_stream().cuda_stream
B, L = 192, 100
seq = ops.SeqInfo.uniform(B, L, L, dev); n = B * L; W = 3 * D
torch.manual_seed(1)
x = torch.randn(n, W, device=dev)
hi = x.bfloat16(); lo = (x - hi.float()).bfloat16()
qkv = torch.cat([hi, lo], 1).contiguous() if x3 else hi.contiguous()
out = torch.zeros(n, (2 if x3 else 1) * D, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, L, device=dev)
km = torch.ones(n, device=dev)
buf = torch.zeros(16 * 128, dtype=torch.int64, device=dev)
sc = 1 / math.sqrt(dh)
def go():
    if x3:
        _lib.call("attn_x3_fwd", qkv.data_ptr(), 2 * W, W, qkv.data_ptr() + 2 * D, 2 * W, W, qkv.data_ptr() + 4 * D, 2 * W, W, out.data_ptr(), 2 * D, D,
                  lse.data_ptr(), seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, sc, 0.1, 7, rng.seed.data_ptr(), st)
    else:
        _lib.call("attn_mfma_fwd_t", qkv.data_ptr(), W, qkv.data_ptr() + 2 * D, W, qkv.data_ptr() + 4 * D, W, out.data_ptr(), D, 1, lse.data_ptr(),
                  seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, sc, 0.1, 7, rng.seed.data_ptr(), st)
lib.svpc_attn_pipe_debug(extra, None)
for _ in range(5): go()
lib.svpc_attn_pipe_debug(16 | extra, buf.data_ptr())
go(); torch.cuda.synchronize()
s = buf.view(16, 16, 8).cpu()
names = ["L:after barrier", "L:issued", "L:landed", "C:at barrier", "C:after barrier", "C:compute end", "C:Q waited", "C:stored"]
for wg in (0, 5):
    t0 = int(s[wg, 0, 3])
    print("workgroup", wg, "(cycles relative to compute wave 0's first barrier arrival)")
    for it in range(10):
        print("  it %d: " % it + "  ".join("%s %6d" % (names[k].split(":")[1][:10], int(s[wg, it, k]) - t0) for k in (3, 4, 5, 6, 7)) +
              "   | loader: " + "  ".join("%s %6d" % (names[k].split(":")[1][:8], int(s[wg, it, k]) - t0) for k in (2, 0, 1)))
