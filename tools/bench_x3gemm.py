"""Times the split-bf16 (bf16x3) forward projections at the decoder's and the clip encoder's shapes: python tools/bench_x3gemm.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svpc_amd import ops as O
O.set_precision("bf16x3")
DEV = "cuda:0"
SHAPES = [(4224, 768, 768), (4224, 2304, 768), (576, 9216, 768), (19200, 768, 768), (19200, 2304, 768)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for M, N, K in SHAPES:
    x = O.to_split(torch.randn(M, K, device=DEV))
    w = torch.randn(N, K, device=DEV) / math.sqrt(K)
    b = torch.randn(N, device=DEV) * 0.1
    w16 = O._transient_split(w)
    with torch.no_grad():
        for _ in range(3):
            y = O.linear(x, w, b, w16=w16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = O.linear(x, w, b, w16=w16)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"M={M:6d} N={N:5d} K={K:5d}  {us:7.1f} us  {6.0 * M * N * K / us / 1e6:7.1f} TFLOP/s issued")
