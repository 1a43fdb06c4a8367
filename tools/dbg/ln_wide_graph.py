"""first LayerNorm (3072 wide) eager vs inside a captured graph vs the generic kernel (SVPC_LN_WIDE=0 in a second process)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from svpc_amd import ops
from svpc_amd.graph import ops_stream, capturing
DEV = "cuda:0"
ops.set_precision("bf16x3")
R, D = 1600, 3072
g = torch.Generator().manual_seed(1)
x = torch.randn(R, D, generator=g).to(DEV)
x[5::7] = 0
gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV).requires_grad_(True)
beta = (0.1 * torch.randn(D, generator=g)).to(DEV).requires_grad_(True)
rng = ops.make_rng(DEV, seed=3)
with torch.cuda.stream(ops_stream()):
    def f():
        return ops.layernorm(x, gamma, beta, 1e-12, post_drop=(0.1, rng, 5), out_bf16=True)
    y0 = f()
    v0 = (y0.float() + ops._lo_view(y0).float()).clone()
    gr = torch.cuda.CUDAGraph()
    with capturing(gr, stream=torch.cuda.current_stream(), capture_error_mode="thread_local"):
        y1 = f()
    x2 = x.clone()
    gr.replay()
    torch.cuda.synchronize()
    v1 = y1.float() + ops._lo_view(y1).float()
    print("eager vs replay max diff", float((v0 - v1).abs().max()), "max", float(v0.abs().max()))
    out = os.environ.get("SAVE")
    if out:
        torch.save(v0.cpu(), out)
    ref = os.environ.get("REF")
    if ref:
        r = torch.load(ref).to(DEV)
        print("vs generic: eager", float((v0 - r).abs().max()), "replay", float((v1 - r).abs().max()))
