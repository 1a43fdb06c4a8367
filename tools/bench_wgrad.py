"""Microbenchmark of the grouped bf16 weight-gradient kernels on uniform tables (one chip-wide round of whole tiles):
python tools/bench_wgrad.py    → µs per launch, TFLOP/s, µs per 64-row k-tile of a 256×256 tile."""
import ctypes, sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svpc_amd import ops as O, _lib

DEV = "cuda:0"
CASES = [("28x 768x768 K19200 ld=n", 28, 19200, 768, 768, 1), ("28x 768x768 K19200 ld=2n", 28, 19200, 768, 768, 2),
         ("9x 2304x768 K19200", 9, 19200, 2304, 768, 1), ("7x 768x3072 K19200", 7, 19200, 768, 3072, 1),
         ("28x 768x768 K4224", 28, 4224, 768, 768, 1), ("4x 768x768 K19200 (36 tiles)", 4, 19200, 768, 768, 1)]
if len(sys.argv) > 1:
    CASES = [CASES[int(i)] for i in sys.argv[1].split(",")]
ws = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
for label, n, rows, n_out, n_in, mul in CASES:
    dz = [torch.randn(rows, n_out, device=DEV).to(torch.bfloat16) for _ in range(min(n, 6))]
    xs = [torch.randn(rows, mul * n_in, device=DEV).to(torch.bfloat16)[:, :n_in] for _ in range(min(n, 6))]
    dws = [torch.zeros(n_out, n_in, device=DEV) for _ in range(n)]
    probs = (O._WgradProblem * n)()
    for i in range(n):
        d, x = dz[i % len(dz)], xs[i % len(xs)]
        probs[i] = O._WgradProblem(d.data_ptr(), x.data_ptr(), dws[i].data_ptr(), None, n_out, n_in, rows, d.stride(0), x.stride(0), dws[i].stride(0))
    tiles = n * ((n_out + 255) // 256) * ((n_in + 255) // 256)
    for name in ("gemm_group_wgrad_bf16_p8", "gemm_group_wgrad_bf16_ws"):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            _lib.call(name, ctypes.addressof(probs), n, ws.data_ptr(), ws.numel() * 4, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _lib.call(name, ctypes.addressof(probs), n, ws.data_ptr(), ws.numel() * 4, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2.0 * n * rows * n_out * n_in
        print(f"{label:34s} {name[-3:]:4s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  tiles {tiles:4d}  {us / ((rows + 63) // 64) / max(1, (tiles + 255) // 256):6.2f} us/k-tile")
