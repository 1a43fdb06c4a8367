"""gemm_p8t (stream dgrad with the weight matrix read k-strided through ds_read_b64_tr_b16, gemm_p8t.hip) through the C-ABI against an fp64
torch statement of C = (A·B) ⊙ act'(G) + R on the same seeded bf16 operands.  Tolerance: one bf16 rounding of the result (2⁻⁸ of its
magnitude) plus the fp32 accumulation.  reference op: the input gradient of nn.Linear (model.py:195-197, :230, :259, :281)."""
import math

import pytest
import torch

from svpc_amd import _lib
from svpc_amd.ops_common import ACT_GELU, ACT_NONE, ACT_RELU

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed * 1000 + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


@pytest.mark.parametrize("M,N,K,gact,has_r", [(512, 768, 768, ACT_NONE, False), (1000, 768, 768, ACT_NONE, True), (777, 776, 2304, ACT_GELU, False),
                                              (1300, 3072, 768, ACT_RELU, True), (19200, 768, 768, ACT_GELU, True), (5, 8, 64, ACT_NONE, False),
                                              (300, 1544, 1536, ACT_NONE, True)])
def test_p8t_vs_fp64(M, N, K, gact, has_r):
    A = _rand(M, K, seed=1).to(torch.bfloat16)
    W = _rand(K, N, seed=2, scale=1.0 / math.sqrt(K)).to(torch.bfloat16)          # (out = K, in = N): the weight as stored
    G = _rand(M, N, seed=3).to(torch.bfloat16) if gact != ACT_NONE else None
    R = _rand(M, N, seed=4).to(torch.bfloat16) if has_r else None
    guard = 3
    Cbuf = torch.full((M + guard, N), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.call("gemm_p8t", A.data_ptr(), K, W.data_ptr(), N, Cbuf.data_ptr(), N, G.data_ptr() if G is not None else None, gact,
              R.data_ptr() if R is not None else None, M, N, K, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = A.double() @ W.double()
    if gact == ACT_GELU:
        ref = ref * _gelu_grad(G.double())
    elif gact == ACT_RELU:
        ref = ref * (G.double() > 0).double()
    if has_r:
        ref = ref + R.double()
    err = float((Cbuf[:M].double() - ref).abs().max())
    assert err <= 6e-3 * float(ref.abs().max()), (err, float(ref.abs().max()))
    assert bool((Cbuf[M:] == 7.0).all())


@pytest.mark.parametrize("M,N,K,has_r", [(4224, 768, 768, False), (4224, 768, 2304, True), (576, 768, 9216, True), (300, 776, 192, True),
                                          (5, 8, 64, False), (1000, 2304, 768, True), (129, 136, 128, False)])
def test_s4t_vs_fp64(M, N, K, has_r):
    """gemm_s4t (the decoder's dgrads: 128² tiles, 8 waves, k-strided weights through ds_read_b64_tr_b16, 2 or 4 stages) through the C-ABI:
    C = A·B + R against fp64 on the same bf16 operands; rows past M are not written"""
    A = _rand(M, K, seed=11).to(torch.bfloat16)
    W = _rand(K, N, seed=12, scale=1.0 / math.sqrt(K)).to(torch.bfloat16)
    R = _rand(M, N, seed=14).to(torch.bfloat16) if has_r else None
    assert _lib.load().svpc_gemm_s4t_supported(K, N, N, M, N, K) == 1
    Cbuf = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.call("gemm_s4t", A.data_ptr(), K, W.data_ptr(), N, Cbuf.data_ptr(), N, R.data_ptr() if R is not None else None, M, N, K,
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = A.double() @ W.double()
    if has_r:
        ref = ref + R.double()
    err = float((Cbuf[:M].double() - ref).abs().max())
    assert err <= 6e-3 * float(ref.abs().max()), (err, float(ref.abs().max()))
    assert bool((Cbuf[M:] == 7.0).all())


def test_decoder_dgrad_takes_s4t_and_matches_the_old_kernel():
    """ops.linear's backward at the decoder's shape (4,224 sentence rows): with and without gemm_s4t, same gradients to bf16 rounding"""
    from svpc_amd import ops as O
    O.set_precision("bf16")
    try:
        M, N, K = 4224, 768, 768
        x = _rand(M, K, seed=15).to(torch.bfloat16).requires_grad_(True)
        w = _rand(N, K, seed=16, scale=1.0 / math.sqrt(K)).requires_grad_(True)
        g = _rand(M, N, seed=17).to(torch.bfloat16)
        outs = []
        for use in (True, False):
            O.USE_S4T = use
            x.grad = None
            O.linear(x, w, None).backward(g)
            O.join_side()
            outs.append(x.grad.float().clone())
        assert float((outs[0] - outs[1]).abs().max()) <= 1e-2 * float(outs[1].abs().max())
    finally:
        O.USE_S4T = True
        O.set_precision("fp32")


def test_stream_dgrad_takes_p8t_and_matches_the_old_kernel():
    """ops.linear's backward at a stream-sized shape: with and without gemm_p8t (env switch), same gradients to bf16 rounding"""
    from svpc_amd import ops as O
    O.set_precision("bf16")
    try:
        M, N, K = 19200, 768, 768
        x = _rand(M, K, seed=5).to(torch.bfloat16).requires_grad_(True)
        w = _rand(N, K, seed=6, scale=1.0 / math.sqrt(K)).requires_grad_(True)
        g = _rand(M, N, seed=7).to(torch.bfloat16)
        outs = []
        for use in (True, False):
            O.USE_P8T = use
            x.grad = None
            O.linear(x, w, None).backward(g)
            O.join_side()
            outs.append(x.grad.float().clone())
        assert float((outs[0] - outs[1]).abs().max()) <= 1e-2 * float(outs[1].abs().max())
    finally:
        O.USE_P8T = True
        O.set_precision("fp32")


def test_grouped_wgrad_p8_vs_fp64_and_the_old_kernel():
    """svpc_gemm_group_wgrad_bf16_p8 (gemm_p8w.hip: both operands k-strided through ds_read_b64_tr_b16) on a table like a step's —
    deep and shallow problems, a ragged row count, widths that are not multiples of the tile, strided operands (the hi planes of split
    rows), accumulation into non-zero gradients, enough deep tiles that some are cut into k-parts, bias gradients (db += Σ_rows dz) on
    whole and on cut tiles — against fp64, and against the round-1 kernel on the same table"""
    import ctypes
    from svpc_amd import ops as O
    specs = [(19200, 768, 768), (19200, 2304, 768), (19200, 768, 3072), (4224, 768, 768), (4224, 2304, 768), (1300, 776, 264), (576, 9216, 768),
             (19200, 768, 768), (19200, 768, 768), (19200, 2304, 768), (19200, 768, 768), (19200, 768, 768)]
    with_bias = [True, True, False, True, True, True, True, False, True, True, True, True]
    tens, probs = [], (O._WgradProblem * len(specs))()
    for i, (rows, n_out, n_in) in enumerate(specs):
        dz = _rand(rows, n_out, seed=20 + i).to(torch.bfloat16)
        xb = _rand(rows, 2 * n_in, seed=40 + i).to(torch.bfloat16)
        x = xb[:, :n_in]                                     # a strided view: row stride 2·n_in (as the hi plane of a split tensor)
        dw0 = _rand(n_out, n_in, seed=60 + i, scale=0.5)
        db0 = _rand(n_out, seed=80 + i, scale=2.0)
        tens.append((dz, x, dw0, db0))
    ws = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    outs = {}
    for name in ("gemm_group_wgrad_bf16_p8", "gemm_group_wgrad_bf16_ws"):
        dws = [t[2].clone() for t in tens]
        dbs = [t[3].clone() for t in tens]
        for i, ((dz, x, _, _), dw) in enumerate(zip(tens, dws)):
            db = dbs[i].data_ptr() if (with_bias[i] and name.endswith("_p8")) else None
            probs[i] = O._WgradProblem(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), db, dz.shape[1], x.shape[1], dz.shape[0], dz.stride(0),
                                       x.stride(0), dw.stride(0))
        if name.endswith("_p8"):
            assert _lib.load().svpc_gemm_group_wgrad_bf16_p8_ok(ctypes.addressof(probs), len(specs)) == 1
        _lib.call(name, ctypes.addressof(probs), len(specs), ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs[name] = (dws, dbs)
    for i, (dz, x, dw0, db0) in enumerate(tens):
        ref = dw0.double() + dz.double().t() @ x.double()
        scale = float(ref.abs().max())
        for name, (dws, dbs) in outs.items():
            err = float((dws[i].double() - ref).abs().max())
            assert err <= 2e-5 * scale * math.sqrt(dz.shape[0] / 1000.0) + 1e-4, (name, i, specs[i], err, scale)
        got = outs["gemm_group_wgrad_bf16_p8"][1][i].double()
        refb = db0.double() + (dz.double().sum(0) if with_bias[i] else 0.0)
        errb = float((got - refb).abs().max())
        assert errb <= 2e-5 * float(refb.abs().max()) * math.sqrt(dz.shape[0] / 1000.0) + 1e-4, ("db", i, specs[i], errb)
