"""Run one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svpc_amd import ops as O
M, N, K, a_kc, b_kc = [int(v) for v in sys.argv[1:6]]
O.set_precision(sys.argv[6] if len(sys.argv) > 6 else "bf16")
T = {"f": torch.float32, "b": torch.bfloat16}
dts = sys.argv[7] if len(sys.argv) > 7 else "fff"
A = torch.randn((M, K) if a_kc else (K, M), device="cuda").to(T[dts[0]])
B = torch.randn((N, K) if b_kc else (K, N), device="cuda").to(T[dts[1]])
C = torch.empty(M, N, device="cuda", dtype=T[dts[2]])
for _ in range(5):
    O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K)
torch.cuda.synchronize()
