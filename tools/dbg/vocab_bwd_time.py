"""How fast would the vocabulary projection's backward be on 4-aligned shapes?  (DESIGN §10.7 "known inefficiency": V = 951)
dgrad  dx (R x 768) = dz (R x Vp) . W (Vp x 768)   [a k-contiguous, b k-strided]
wgrad  dW (Vp x 768) += dz^T . x                     [grouped fp32 wgrad, one problem]
against the same two products at V = 951 as the step runs them (generic split-K kernel).  python tools/dbg/vocab_bwd_time.py (GPU box)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from svpc_amd import ops, _lib
DEV = "cuda:0"
ops.set_precision("bf16x3")
R, D = 2852, 768
def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for V in (951, 960):
    dz = torch.randn(R, V, device=DEV)
    W = torch.randn(V, D, device=DEV) * 0.05
    x = torch.randn(R, D, device=DEV)
    dx = torch.empty(R, D, device=DEV)
    dw = torch.zeros(V, D, device=DEV)
    t_d = timeit(lambda: ops._gemm(dz, V, 1, W, D, 0, dx, R, D, V))
    t_w = timeit(lambda: ops._gemm(dz, V, 0, x, D, 0, dw, V, D, R, accumulate=1))
    print("V = %d: dgrad %.1f us, wgrad (_gemm) %.1f us" % (V, t_d, t_w))
    if V % 4 == 0:
        class P(ctypes.Structure):
            _fields_ = [("dz", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("db", ctypes.c_void_p), ("n_out", ctypes.c_int),
                        ("n_in", ctypes.c_int), ("rows", ctypes.c_int), ("ld_dz", ctypes.c_int), ("ld_x", ctypes.c_int), ("ld_dw", ctypes.c_int)]
        pr = (P * 1)()
        pr[0] = P(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), None, V, D, R, V, D, D)
        t_g = timeit(lambda: _lib.call("gemm_group_wgrad", ctypes.addressof(pr), 1, torch.cuda.current_stream().cuda_stream))
        print("         grouped fp32 wgrad, one problem: %.1f us" % t_g)
