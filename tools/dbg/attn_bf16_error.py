"""error statistics of the bf16 clip-encoder attention forward (pipelined vs one-workgroup-per-pair) against an fp64 reference on the same
bf16 inputs: rms, maximum and signed mean (bias) of O - ref, no dropout"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svpc_amd import _lib, ops
dev = torch.device("cuda:0"); H, dh = 12, 64; D = H * dh
lib = _lib.load(); rng = ops.default_rng(dev); st = torch.cuda.current_stream().cuda_stream
B, L = 48, 100
seq = ops.SeqInfo.uniform(B, L, L, dev); n = B * L; W = 3 * D
for scale_in in (1.0, 3.0):
    torch.manual_seed(2)
    qkv = (scale_in * torch.randn(n, W, device=dev)).bfloat16().contiguous()
    xv = qkv.double()
    q = xv[:, :D].view(B, L, H, dh).permute(0, 2, 1, 3); k = xv[:, D:2 * D].view(B, L, H, dh).permute(0, 2, 1, 3); v = xv[:, 2 * D:].view(B, L, H, dh).permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1) @ v).permute(0, 2, 1, 3).reshape(n, D)
    out = torch.zeros(n, D, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, L, device=dev); km = torch.ones(n, device=dev)
    for on in (0, 1):
        lib.svpc_attn_pipe_enable(on)
        _lib.call("attn_mfma_fwd_t", qkv.data_ptr(), W, qkv.data_ptr() + 2 * D, W, qkv.data_ptr() + 4 * D, W, out.data_ptr(), D, 1, lse.data_ptr(),
                  seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, 1 / math.sqrt(dh), 0.0, 7, rng.seed.data_ptr(), st)
        torch.cuda.synchronize()
        e = out.double() - ref
        e_round = ref.bfloat16().double() - ref       # the unavoidable rounding of the output itself
        print("input scale %.0f pipe %d: rms %.3e  max %.3e  bias %.3e   (output rounding alone: rms %.3e)  rms|ref| %.3e" %
              (scale_in, on, e.pow(2).mean().sqrt(), e.abs().max(), e.mean(), e_round.pow(2).mean().sqrt(), ref.pow(2).mean().sqrt()))
lib.svpc_attn_pipe_enable(1)
