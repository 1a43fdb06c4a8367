"""gemm_p8t (stream dgrad, k-strided weights) with / without the residual addend R and the activation-backward factor G, kernel
durations by rocprofv3 (run under tools/prof_any.sh): python tools/bench_p8t.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svpc_amd import _lib
dev = "cuda:0"
M, N, K = 19200, 768, 768
A = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(K, N, device=dev) / 28).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
G = torch.randn(M, N, device=dev).bfloat16()
R = torch.randn(M, N, device=dev).bfloat16()
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(256 << 20, device=dev, dtype=torch.float32)
def go(g, gact, r):
    _lib.call("gemm_p8t", A.data_ptr(), K, W.data_ptr(), N, C.data_ptr(), N, g.data_ptr() if g is not None else None, gact,
              r.data_ptr() if r is not None else None, M, N, K, st)
for rep in range(10):
    flush.fill_(float(rep)); go(None, 0, None)            # plain, cold
    flush.fill_(float(rep)); go(G, 2, None)               # GELU' factor, cold
    flush.fill_(float(rep)); go(None, 0, R)               # R cold
    flush.fill_(float(rep)); R.mul_(1.0); go(None, 0, R)  # R just written
    flush.fill_(float(rep)); go(None, 0, C)               # R == C (in place), cold
torch.cuda.synchronize()
