import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svpc_amd import ops as O, _lib
n = 19200 * 768
for dt, tdt in ((1, torch.bfloat16), (0, torch.float32)):
    dy = torch.randn(n, device="cuda").to(tdt); z = torch.randn(n, device="cuda").to(tdt); dz = torch.empty_like(dy)
    for off, name in ((0, "aligned"), (1, "unaligned (scalar kernel)")):
        a, b, c = dy[off:off + n - 8], z[off:off + n - 8], dz[off:off + n - 8]
        for act in (2, 1):
            for _ in range(3):
                _lib.call("act_bwd_t", a.data_ptr(), b.data_ptr(), c.data_ptr(), dt, a.numel(), act, 0.0, 0, None, O._stream())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _lib.call("act_bwd_t", a.data_ptr(), b.data_ptr(), c.data_ptr(), dt, a.numel(), act, 0.0, 0, None, O._stream())
            e1.record(); torch.cuda.synchronize()
            print("dt", dt, name, "act", act, "%.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
