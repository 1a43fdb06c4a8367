"""attention_pipe.hip timing experiments: the pipelined forward with parts switched off (SVPC_PP_DBG bits: 1 no DMA, 2 no compute,
4 no O stores, 8 no Q loads), warm and cold, HIP events.    usage: python tools/dbg/pipe_phases.py [x3|bf16]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svpc_amd import _lib, ops
x3 = (sys.argv[1] if len(sys.argv) > 1 else "x3") == "x3"
pd = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
dev = torch.device("cuda:0"); H, dh = 12, 64; D = H * dh
lib = _lib.load(); rng = ops.default_rng(dev); st = torch.cuda.current_stream().cuda_stream
B, L = 192, 100
seq = ops.SeqInfo.uniform(B, L, L, dev); n = B * L; W = 3 * D
torch.manual_seed(1)
x = torch.randn(n, W, device=dev)
hi = x.bfloat16(); lo = (x - hi.float()).bfloat16()
qkv = torch.cat([hi, lo], 1).contiguous() if x3 else hi.contiguous()
out = torch.zeros(n, (2 if x3 else 1) * D, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, L, device=dev)
km = torch.ones(n, device=dev)
sc = 1 / math.sqrt(dh)
if x3:
    go = lambda: _lib.call("attn_x3_fwd", qkv.data_ptr(), 2 * W, W, qkv.data_ptr() + 2 * D, 2 * W, W, qkv.data_ptr() + 4 * D, 2 * W, W, out.data_ptr(), 2 * D, D,
                           lse.data_ptr(), seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, sc, pd, 7, rng.seed.data_ptr(), st)
else:
    go = lambda: _lib.call("attn_mfma_fwd_t", qkv.data_ptr(), W, qkv.data_ptr() + 2 * D, W, qkv.data_ptr() + 4 * D, W, out.data_ptr(), D, 1, lse.data_ptr(),
                           seq.table.data_ptr(), B, H, dh, L, L, km.data_ptr(), 0, sc, pd, 7, rng.seed.data_ptr(), st)
flush = torch.empty(256 << 20, device=dev, dtype=torch.float32)
names = {0: "everything", 1: "no DMA", 2: "no compute", 4: "no O stores", 8: "no Q loads", 9: "no DMA, no Q loads (compute + stores)", 13: "compute only",
         6: "no compute, no stores (loads only)", 14: "DMA only", 7: "Q loads only"}
for dbg, nm in names.items():
    if only is not None and dbg not in only: continue
    lib.svpc_attn_pipe_debug(dbg, None)
    for _ in range(3): go()
    torch.cuda.synchronize()
    cold = []
    for r in range(8):
        flush.fill_(float(r))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    print("dbg %2d %-42s cold %.1f us   back-to-back %.1f us" % (dbg, nm, sorted(cold)[len(cold) // 2], e0.elapsed_time(e1) * 1e3 / 20))
lib.svpc_attn_pipe_debug(0, None)
