"""decode-step GEMM shapes (M = 768 sentences), bf16x3 products on fp32 storage: python tools/dbg/dec_gemm_time.py"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svpc_amd import ops as O
O.set_precision("bf16x3")
DEV = "cuda:0"
SPLIT = len(sys.argv) > 1 and sys.argv[1] == "split"
RD = False
for M, N, K in [(768, 768, 768), (768, 2304, 768), (768, 951, 768), (192, 768, 768), (264, 768, 768), (192, 2304, 768)]:
    x = torch.randn(M, K, device=DEV)
    x0 = x
    w16 = None
    if SPLIT:
        x = O.to_split(x)
    w = torch.randn(N, K, device=DEV) / math.sqrt(K); b = torch.randn(N, device=DEV)
    if SPLIT or RD:
        w16 = O._transient_split(w)
    with torch.no_grad():
        for _ in range(5):
            y = O.linear(x, w, b, w16=w16)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(100):
                y = O.linear(x, w, b, w16=w16)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay(); g.replay()
        e1.record(); torch.cuda.synchronize()
    y = O.to_f32(y) if SPLIT else y
    ref = x0.double() @ w.double().t() + b.double()
    print("M=%4d N=%5d K=%4d  %6.2f us   err %.2e" % (M, N, K, e0.elapsed_time(e1) * 5, (y.double() - ref).abs().max().item() / ref.abs().max().item()), flush=True)
