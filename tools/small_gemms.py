"""Time the small GEMM shapes of the step-level / text-side path (serialized launches on one stream, as in the step)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svpc_amd import ops as O
O.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
shapes = [  # M, N, K, a_kc, b_kc, accumulate
    (768, 768, 192, 0, 0, 1), (768, 768, 192, 0, 0, 0), (3072, 768, 16, 0, 0, 1), (768, 768, 4224, 0, 0, 1), (2304, 768, 4224, 0, 0, 1),
    (192, 768, 768, 1, 1, 0), (192, 2304, 768, 1, 1, 0), (192, 768, 768, 1, 0, 0),
    (4224, 768, 768, 1, 1, 0), (4224, 2304, 768, 1, 1, 0), (4224, 768, 768, 1, 0, 0), (4224, 951, 768, 1, 1, 0),
    (16, 3072, 768, 1, 1, 0), (16, 768, 3072, 1, 0, 0), (576, 1536, 768, 1, 1, 0),
]
def timed(M, N, K, a_kc, b_kc, acc, A, B, C):
    for _ in range(3):
        O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K, accumulate=acc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K, accumulate=acc)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 50


for M, N, K, a_kc, b_kc, acc in shapes:
    A = torch.randn((M, K) if a_kc else (K, M), device="cuda")
    B = torch.randn((N, K) if b_kc else (K, N), device="cuda")
    C = torch.zeros(M, N, device="cuda")
    O.USE_L32 = False
    t_old = timed(M, N, K, a_kc, b_kc, acc, A, B, C)
    O.USE_L32 = True
    t_new = timed(M, N, K, a_kc, b_kc, acc, A, B, C)
    print("M=%5d N=%5d K=%5d %s acc=%d : register-staged %7.1f us | direct-to-LDS %7.1f us  %6.1f TFLOP/s" % (
        M, N, K, "TN"[a_kc] + "TN"[b_kc], acc, t_old, t_new, 2.0 * M * N * K / t_new / 1e6))
