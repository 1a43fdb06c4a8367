"""Fixed cost of the 256x256 ping-pong GEMM: time vs K, with/without the epilogue (development tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_gemm import bench
bf = torch.bfloat16
for K in (32, 96, 384, 768, 1536):
    us, tf = bench(19200, 768, K, 1, 1, "bf16", iters=50, dts=(bf, bf, bf))
    print("K=%5d  %7.1f us  %7.1f TF" % (K, us, tf), flush=True)
# an empty-ish launch for reference: a 64-element elementwise op
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): x.add_(1.0)
e1.record(); torch.cuda.synchronize()
print("tiny kernel back-to-back: %.2f us" % (e0.elapsed_time(e1) * 1e3 / 200))
