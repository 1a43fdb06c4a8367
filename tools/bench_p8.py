"""A/B of the forward-projection GEMM forms in ONE process, interleaved rounds, random operands (cdna_hip_programming.md §5.4 rules
24, 25): the 8-phase kernel (svpc_gemm_p8), the two-group ping-pong kernel it replaces (SVPC_P8=0 path of svpc_gemm_glds) and the
vendor GEMM (torch.matmul = hipBLASLt).  Also checks p8 against fp64 of the same bf16 operands."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svpc_amd import _lib, ops as O

lib = _lib.load()
dev = "cuda:0"
st = lambda: torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
shapes = [(19200, 768, 768, 0), (19200, 768, 768, 2), (19200, 2304, 768, 0), (19200, 1536, 768, 0), (19200, 768, 3072, 1), (18000, 768, 768, 0), (4224, 2304, 768, 0)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
for M, N, K, act in shapes:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(bf).to(dev)
    B = (0.05 * torch.randn(N, K, generator=g)).to(bf).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(M, N, device=dev, dtype=bf)
    Z = torch.empty(M, N, device=dev, dtype=bf) if act == 2 else None
    ws = O._ws(torch.device(dev))
    def p8():
        _lib.call("gemm_p8", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, Z.data_ptr() if Z is not None else None, M, N, K,
                  bias.data_ptr(), act, st())
    def pp():
        _lib.call("gemm_glds_r", A.data_ptr(), K, 1, B.data_ptr(), K, 1, C.data_ptr(), 1, N, Z.data_ptr() if Z is not None else None, None, M, N, K,
                  bias.data_ptr(), act, 0.0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, st())
    def vendor():
        torch.matmul(A, B.t(), out=C)
    # correctness of p8
    C.zero_(); p8(); torch.cuda.synchronize()
    zr = A.double() @ B.double().t() + bias.double()
    ref = torch.relu(zr) if act == 1 else torch.nn.functional.gelu(zr) if act == 2 else zr
    scale = max(1.0, zr.abs().max().item())
    err = (C.double() - ref).abs().max().item() / scale
    zerr = ((Z.double() - zr).abs().max().item() / scale) if Z is not None else 0.0
    times = {"p8": [], "pp": [], "hipblaslt": []}
    os.environ["SVPC_P8"] = "0"      # (read once by the library: set before its first svpc_gemm_glds call)
    for f in (p8, pp, vendor):
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    for r in range(rounds):
        for name, f in (("p8", p8), ("pp", pp), ("hipblaslt", vendor)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 100.0)      # us per launch
    fl = 2.0 * M * N * K
    line = "M=%5d N=%4d K=%4d act=%d  err %.2e zerr %.2e |" % (M, N, K, act, err, zerr)
    for name in ("p8", "pp", "hipblaslt"):
        t = sorted(times[name]); med = t[len(t) // 2]
        line += "  %s %6.1f us (min %6.1f) %6.0f TF" % (name, med, t[0], fl / med / 1e6)
    print(line, flush=True)
