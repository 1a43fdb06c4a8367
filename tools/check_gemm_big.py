"""Check the 256x256 ping-pong GEMM against torch on stream-sized shapes (development tool; run on the GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svpc_amd import ops as O
from svpc_amd import _lib
O.set_precision("bf16")
torch.manual_seed(0)
bf = torch.bfloat16
bad = 0
for (M, N, K, a_kc, b_kc) in [(19200, 768, 768, 1, 1), (19200, 2304, 768, 1, 1), (19200, 768, 3072, 1, 1), (19200, 768, 768, 1, 0),
                              (19200, 768, 2304, 1, 0), (18000, 760, 768, 1, 1), (15608, 1000, 96, 1, 0)]:
    for rep in range(3):
        A = torch.randn((M, K) if a_kc else (K, M), device="cuda").to(bf)
        B = (torch.randn((N, K) if b_kc else (K, N), device="cuda") * 0.05).to(bf)
        C = torch.empty(M, N, device="cuda", dtype=bf)
        O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K)
        ref = (A.float() if a_kc else A.float().t()) @ (B.float().t() if b_kc else B.float())
        err = (C.float() - ref).abs().max().item() / ref.abs().max().item()
        ok = err < 1e-2
        bad += not ok
        print(M, N, K, a_kc, b_kc, "rel err %.2e" % err, "ok" if ok else "BAD", flush=True)
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
