"""dgrad of a stream projection (dx = dz·W, W k-strided) with and without the residual-gradient addend R, interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svpc_amd import ops as O
O.set_precision("bf16")
dev = "cuda:0"; bf = torch.bfloat16
for M, N, K in ((19200, 768, 768), (19200, 2304, 768)):
    dz = torch.randn(M, N, device=dev).to(bf); W = (0.05 * torch.randn(N, K, device=dev)).to(bf)
    R = torch.randn(M, K, device=dev).to(bf); dx = torch.empty(M, K, device=dev, dtype=bf)
    # a second set of operands so that consecutive launches do not find everything in the 256-MiB infinity cache
    big = [torch.randn(M, N, device=dev).to(bf) for _ in range(8)]
    def run(r, i): O._gemm(big[i % 8], N, 1, W, K, 0, dx, M, K, N, R=(R if r else None))
    for r in (0, 1):
        for i in range(3): run(r, i)
    torch.cuda.synchronize()
    res = {0: [], 1: []}
    for rnd in range(7):
        for r in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(8): run(r, i)
            e1.record(); torch.cuda.synchronize()
            res[r].append(e0.elapsed_time(e1) * 125.0)
    print("M=%d N(contraction)=%d K(out)=%d: plain %.1f us   with addend R %.1f us" % (M, N, K, sorted(res[0])[3], sorted(res[1])[3]))
