"""s_memtime stamps of the ping-pong GEMM main loop (development tool; SVPC_PP_DBG=8 SVPC_GLDS_BIG=1)."""
import os, sys, torch
os.environ.setdefault("SVPC_PP_DBG", "8"); os.environ.setdefault("SVPC_GLDS_BIG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svpc_amd import ops as O
O.set_precision("bf16")
bf = torch.bfloat16
M, N, K = 19200, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 768
A = torch.randn(M, K, device="cuda").to(bf); B = torch.randn(N, K, device="cuda").to(bf); C = torch.empty(M, N, device="cuda", dtype=bf)
for _ in range(3):
    O._gemm(A, K, 1, B, K, 1, C, M, N, K)
torch.cuda.synchronize()
ws = O._ws(C.device).view(torch.int64)[:2 * 6 * 64].cpu().view(2, 6, 64)
nk = min(K // 32, 64)
names = ["top", "reads done", "after bar1", "mfma issued", "vm waited", "after bar2"]
for g in range(2):
    s = ws[g]
    t0 = int(s[0, 0])
    print("group", g, "(stamps relative to its first; columns: top→reads done→bar1→mfma issued→vm waited→bar2)")
    for t in range(nk):
        r = [int(s[i, t]) - t0 for i in range(6)]
        d = [r[i + 1] - r[i] for i in range(5)]
        nxt = (int(s[0, t + 1]) - int(s[5, t])) if t + 1 < nk else 0
        print("  t=%2d  start %6d | read %4d  bar1 %4d  mfma %4d  vmwait %4d  bar2 %4d | tile %5d" % (t, r[0], d[0], d[1], d[2], d[3], d[4], r[5] - r[0] + nxt))
print("group1 top - group0 top per tile:", [int(ws[1, 0, t]) - int(ws[0, 0, t]) for t in range(min(nk, 8))])
