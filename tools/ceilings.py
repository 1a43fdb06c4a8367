#!/usr/bin/env python
"""Same-box ceilings for the two rooflines SURVEY §8(d) names (measurement tooling — nothing here is on the product path):

* the vendor GEMM (``torch.matmul`` on bf16 = hipBLASLt / rocBLAS on ROCm) on the dominant shapes of the clip-encoder stream —
  the yardstick for the hand-written ``gemm_glds_pp_kernel``;
* a device-to-device copy (``Tensor.copy_``: 16-byte vector loads/stores) — the achievable HBM bandwidth next to the 8 TB/s
  specification figure, the yardstick for attention / LayerNorm / optimizer kernels.

  python tools/ceilings.py [--out profiles/r02_ceilings.json]

``measure(device, quick)`` is what bench.py calls (after its timed region) to print ``ceilings`` in the JSON line.
"""
import argparse
import json
import sys

import torch

# (M, N, K, what) — C[M,N] = A[M,K]·W[N,K]ᵀ, the forward projections of one step at the headline config (bench.py's roofline kernel)
SHAPES = [
    (19200, 2304, 768, "Q/K/V projection (5 per step)"),
    (19200, 1536, 768, "K/V projection of the [CLS]-only last layer (1)"),
    (19200, 768, 768, "attention-out / FFN-in / FFN-out (15)"),
    (19200, 768, 3072, "video embedding (1)"),
    (4224, 2304, 768, "decoder Q/K/V (6)"),
]
WEIGHTS = [5, 1, 15, 1, 0]      # launches per step of the dominant kernel symbol (decoder shape listed for reference only)


def _time(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in evs:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return ts[len(ts) // 2] * 1e-3


def measure(device="cuda:0", quick=False):
    dev = torch.device(device)
    iters = 10 if quick else 50
    out = {"hipblaslt": [], "note": "torch.matmul bf16 (vendor GEMM) and Tensor.copy_ on this box, median of %d timed launches each" % iters}
    tot_flop = tot_t = 0.0
    for (M, N, K, what), wgt in zip(SHAPES, WEIGHTS):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = _time(lambda: torch.matmul(a, w.t(), out=c), iters)
        fl = 2.0 * M * N * K
        out["hipblaslt"].append({"M": M, "N": N, "K": K, "what": what, "us": t * 1e6, "tflops": fl / t / 1e12})
        tot_flop += wgt * fl
        tot_t += wgt * t
        del a, w, c
    out["hipblaslt_tflops"] = tot_flop / tot_t / 1e12      # launch-weighted mean over the dominant kernel's 22 launches per step
    n = (1 << 30) // 2 if quick else (1 << 30)
    src = torch.empty(n, device=dev, dtype=torch.float32)
    dst = torch.empty_like(src)
    src.normal_()
    t = _time(lambda: dst.copy_(src), iters)
    out["copy_tbps"] = 2.0 * n * 4 / t / 1e12              # bytes read + bytes written
    out["copy_bytes"] = 2 * n * 4
    del src, dst
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    res = measure("cuda:0", quick=args.quick)
    js = json.dumps(res, indent=1)
    print(js)
    if args.out:
        with open(args.out, "w") as f:
            f.write(js + "\n")


if __name__ == "__main__":
    sys.exit(main())
