"""Micro-benchmark of the GEMM kernels on the hot-path shapes (development tool; run on the GPU box)."""
import os, sys, time, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svpc_amd import ops as O

def bench(M, N, K, a_kc, b_kc, prec, iters=20, dts=(torch.float32, torch.float32, torch.float32)):
    O.set_precision(prec)
    A = torch.randn((M, K) if a_kc else (K, M), device="cuda").to(dts[0])
    B = torch.randn((N, K) if b_kc else (K, N), device="cuda").to(dts[1])
    C = torch.empty(M, N, device="cuda", dtype=dts[2])
    for _ in range(3):
        O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6

if __name__ == "__main__":
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    shapes = [("enc fwd  NT", 19200, 768, 768, 1, 1), ("enc qkv  NT", 19200, 2304, 768, 1, 1), ("vid emb  NT", 19200, 768, 3072, 1, 1),
              ("enc dgrad NN", 19200, 768, 768, 1, 0), ("enc wgrad TN", 768, 768, 19200, 0, 0), ("vid wgrad TN", 768, 3072, 19200, 0, 0),
              ("dec fwd  NT", 4224, 768, 768, 1, 1), ("head     NT", 4224, 951, 768, 1, 1), ("step     NT", 192, 768, 768, 1, 1)]
    bf, f = torch.bfloat16, torch.float32
    for name, M, N, K, a, b in shapes:
        us, tf = bench(M, N, K, a, b, prec)
        line = "%-14s M=%6d N=%5d K=%6d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, us, tf)
        if prec == "bf16" and M % 128 == 0 and N % 128 == 0 and M >= 768:
            dts = (bf, bf, f) if (not a and not b) else (bf, f, bf)
            us2, tf2 = bench(M, N, K, a, b, prec, dts=dts)
            line += "   | bf16 storage: %8.1f us  %7.1f TFLOP/s" % (us2, tf2)
            dts3 = (bf, bf, f) if (not a and not b) else (bf, bf, bf)
            us3, tf3 = bench(M, N, K, a, b, prec, dts=dts3)
            line += "   | glds (bf16 W): %8.1f us  %7.1f TFLOP/s" % (us3, tf3)
        print(line, flush=True)
