"""The bf16x3 arithmetic mode (≤1e-4-parity throughput mode) on the MI355X: its kernels one by one against fp64 / fp32 torch
statements of the same op on the same seeded inputs, then the module end to end against the reference's goldens.

bf16x3: every forward contraction is a three-term split-bf16 product a_lo·b_hi + a_hi·b_lo + a_hi·b_hi with fp32 accumulation
(gemm_p8x3.hip, gemm_l32.hip X3 instances, attention_mfma.hip attn_stream_x3_fwd_kernel); the clip-encoder stream is stored as two
bf16 planes per row.  Stated tolerance of a product: 3e-5 of the output's magnitude (2⁻¹⁷ per operand, random over K); of the
module: north_star's loss ≤ 1e-4 relative.  The backward is the bf16 mode's (gradient tolerances as tests/test_ops_gpu.py's bf16 cases).
reference: src/rtransformer/model.py (:195-197 projections, :194-219 attention, :143-156 LayerNorm)."""
import math

import numpy as np
import pytest
import torch

import emul_ops as E
from helpers import build_model
from svpc_amd import _lib, ops as O, synthetic as syn
from svpc_amd.ops_common import ACT_GELU, ACT_NONE, ACT_RELU, SeqInfo

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _x3_mode():
    O.set_precision("bf16x3")
    yield
    O.set_precision("fp32")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed * 1000 + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _split(t):
    """fp32 (R, W) → split tensor (hi view tagged with the lo offset), and the value it represents (fp32, exact)"""
    R, W = t.shape
    s = O.new_split(R, W, t.device)
    lo = torch.as_strided(s, s.shape, s.stride(), s.storage_offset() + W)
    s.copy_(t)
    lo.copy_(t - s.float())
    return s, s.float() + lo.float()


def _value(s):
    lo = torch.as_strided(s, s.shape, s.stride(), s.storage_offset() + s._svpc_lo)
    return s.detach().float() + lo.detach().float()


def _act64(z, act):
    if act == ACT_RELU:
        return torch.relu(z)
    if act == ACT_GELU:
        return z * 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0)))
    return z


# ------------------------------------------------------------------------------------------------ split GEMM (gemm_p8x3.hip)
@pytest.mark.parametrize("M,N,K,act", [(256, 256, 64, ACT_NONE), (300, 768, 768, ACT_NONE), (1000, 2304, 768, ACT_NONE),
                                       (517, 832, 128, ACT_RELU), (640, 768, 3072, ACT_RELU), (1111, 768, 768, ACT_GELU),
                                       (19200, 768, 768, ACT_GELU), (3, 64, 64, ACT_NONE), (4224, 768, 768, ACT_NONE),
                                       (4224, 2304, 768, ACT_NONE), (576, 9216, 768, ACT_NONE), (4224, 768, 768, ACT_GELU)])
def test_split_gemm_vs_fp64(M, N, K, act):
    x, xv = _split(_rand(M, K, seed=1))
    w = _rand(N, K, seed=2, scale=1.0 / math.sqrt(K))
    b = _rand(N, seed=3, scale=0.1)
    w16 = O._transient_split(w)
    wv = w16.float() + torch.as_strided(w16, w16.shape, w16.stride(), w16.storage_offset() + w16._svpc_lo).float()
    y = O.linear(x, w, b, act=act, w16=w16)
    assert O.lo_off(y) == N and y.dtype == torch.bfloat16 and y.stride(0) == 2 * N
    ref = _act64(xv.double() @ wv.double().t() + b.double(), act)
    got = _value(y).double()
    err = float((got - ref).abs().max())
    scale = float(ref.abs().max())
    assert err <= 3e-5 * scale, (err, scale)
    # and the operands' own rounding is what the 3 terms remove: a one-term bf16 product of the same operands is ≈100× worse
    one = _act64(x.double() @ w16.double().t() + b.double(), act)
    assert float((one - ref).abs().max()) > 10 * err


@pytest.mark.parametrize("kernel", ["gemm_p8x3", "gemm_s4x3"])
def test_split_gemm_c_abi_edges(kernel):
    """svpc_gemm_p8x3 / svpc_gemm_s4x3 called directly: N % 8 == 0 (not a multiple of the tile), rows and columns past the edges are
    not written, the pre-activation copy is plain bf16"""
    M, N, K = 261, 776, 192
    x, xv = _split(_rand(M, K, seed=30))
    w = _rand(N, K, seed=31, scale=1.0 / math.sqrt(K))
    w16 = O._transient_split(w)
    wv = w16.float() + torch.as_strided(w16, w16.shape, w16.stride(), w16.storage_offset() + w16._svpc_lo).float()
    b = _rand(N, seed=32, scale=0.1)
    buf = torch.full((M + 3, 2 * N + 16), 7.0, dtype=torch.bfloat16, device=DEV)      # guard rows / columns around the output
    zbuf = torch.full((M + 3, N + 8), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.call(kernel, x.data_ptr(), x.stride(0), K, w16.data_ptr(), K, w16._svpc_lo, buf.data_ptr(), buf.stride(0), N + 8,
              zbuf.data_ptr(), zbuf.stride(0), M, N, K, b.data_ptr(), ACT_GELU, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    z64 = xv.double() @ wv.double().t() + b.double()
    ref = _act64(z64, ACT_GELU)
    got = buf[:M, :N].double() + buf[:M, N + 8:2 * N + 8].double()
    assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    assert float((zbuf[:M, :N].double() - z64).abs().max()) <= 6e-3 * float(z64.abs().max())
    assert bool((buf[M:] == 7.0).all()) and bool((buf[:M, N:N + 8] == 7.0).all()) and bool((buf[:M, 2 * N + 8:] == 7.0).all())
    assert bool((zbuf[M:] == 7.0).all()) and bool((zbuf[:M, N:] == 7.0).all())


def test_split_gemm_backward_is_the_bf16_backward():
    """dgrad / wgrad of a split projection: the bf16 kernels on the hi planes (same tolerances as the bf16 stream tests)"""
    M, N, K = 700, 768, 768
    xf = _rand(M, K, seed=4)
    w = _rand(N, K, seed=5, scale=1.0 / math.sqrt(K)).requires_grad_(True)
    b = _rand(N, seed=6, scale=0.1).requires_grad_(True)
    x, xv = _split(xf)
    x.requires_grad_(True)
    y = O.linear(x, w, b, act=ACT_GELU)
    g = _rand(M, N, seed=7).to(torch.bfloat16)
    y.backward(g)
    xr = xv.clone().requires_grad_(True)
    wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    E.linear(xr, wr, br, act=ACT_GELU).backward(g.float())
    for got, ref, nm in ((x.grad.float(), xr.grad, "dx"), (w.grad, wr.grad, "dw"), (b.grad, br.grad, "db")):
        err, scale = float((got - ref).abs().max()), float(ref.abs().max())
        assert err <= 2e-2 * scale, (nm, err, scale)


# ------------------------------------------------------------------------------------------------ fp32-storage GEMMs with x3 products
@pytest.mark.parametrize("M,N,K,trans", [(192, 768, 768, False), (192, 2304, 768, False), (4224, 768, 768, False), (4224, 951, 768, False),
                                         (192, 300, 384, True), (16, 3072, 768, False), (4224, 768, 300, False), (100, 33, 64, False),
                                         (192, 768, 300, False), (2112, 300, 44, False)])
def test_fp32_storage_gemm_x3_vs_fp64(M, N, K, trans):
    x = _rand(M, K, seed=8)
    w = _rand(K, N, seed=9, scale=1.0 / math.sqrt(K)) if trans else _rand(N, K, seed=9, scale=1.0 / math.sqrt(K))
    b = None if trans else _rand(N, seed=10, scale=0.1)
    y = O.linear(x, w, b, act=ACT_RELU, trans_w=trans)
    ref = torch.relu(x.double() @ (w.double() if trans else w.double().t()) + (0 if b is None else b.double()))
    err, scale = float((y.double() - ref).abs().max()), float(ref.abs().max())
    assert y.dtype == torch.float32 and err <= 3e-5 * scale, (err, scale)


# ------------------------------------------------------------------------------------------------ LayerNorm on split rows
@pytest.mark.parametrize("R,D,with_res,drop", [(19, 768, True, 0.0), (400, 768, True, 0.1), (257, 3072, False, 0.0), (64, 128, True, 0.0)])
def test_layernorm_split_vs_torch(R, D, with_res, drop):
    x, xv = _split(_rand(R, D, seed=11))
    res = resv = None
    if with_res:
        res, resv = _split(_rand(R, D, seed=12))
    gamma, beta = (1.0 + 0.1 * _rand(D, seed=13)), 0.1 * _rand(D, seed=14)
    rng = O.make_rng(DEV, seed=5)
    d = (drop, rng, 3) if drop > 0 else None
    y = O.layernorm(x, gamma, beta, 1e-12, residual=res, pre_drop=d)
    assert O.lo_off(y) == D
    ref = E.layernorm(xv, gamma, beta, 1e-12, residual=resv, pre_drop=d)
    err = float((_value(y) - ref).abs().max())
    assert err <= 3e-5 * float(ref.abs().max()), err


@pytest.mark.parametrize("R", [300, 1037, 4100])
def test_layernorm_fp32_gather_to_split_and_backward(R):
    """first LayerNorm of the video embedding: fp32 feature rows gathered → split rows; its backward reads the saved fp32 rows
    (R ≥ 512: the two-waves-per-row streaming forward and the long-workgroup parameter-gradient launch of round 5)"""
    D = 3072
    table = _rand(500, D, seed=15)
    rows = torch.randint(0, 500, (R,), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(DEV)
    gamma = (1.0 + 0.1 * _rand(D, seed=16)).requires_grad_(True)
    beta = (0.1 * _rand(D, seed=17)).requires_grad_(True)
    y = O.layernorm(table, gamma, beta, 1e-12, src_rows=rows, out_bf16=True)
    assert O.lo_off(y) == D
    g2, b2 = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    ref = E.layernorm(table, g2, b2, 1e-12, src_rows=rows)
    assert float((_value(y) - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    g = _rand(R, D, seed=18).to(torch.bfloat16)
    y.backward(g)
    O.join_side()
    ref.backward(g.float())
    for a, b_ in ((gamma.grad, g2.grad), (beta.grad, b2.grad)):
        assert float((a - b_).abs().max()) <= 2e-3 * float(b_.abs().max())


def test_layernorm_split_backward_reads_hi_planes():
    R, D = 333, 768
    x, xv = _split(_rand(R, D, seed=19))
    res, resv = _split(_rand(R, D, seed=20))
    x.requires_grad_(True); res.requires_grad_(True)
    gamma = (1.0 + 0.1 * _rand(D, seed=21)).requires_grad_(True)
    beta = (0.1 * _rand(D, seed=22)).requires_grad_(True)
    y = O.layernorm(x, gamma, beta, 1e-12, residual=res)
    g = _rand(R, D, seed=23).to(torch.bfloat16)
    y.backward(g)
    O.join_side()
    xr, rr = xv.clone().requires_grad_(True), resv.clone().requires_grad_(True)
    g2, b2 = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    E.layernorm(xr, g2, b2, 1e-12, residual=rr).backward(g.float())
    for a, b_, nm in ((x.grad.float(), xr.grad, "dx"), (res.grad.float(), rr.grad, "dres"), (gamma.grad, g2.grad, "dgamma"),
                      (beta.grad, b2.grad, "dbeta")):
        assert float((a - b_).abs().max()) <= 2e-2 * float(b_.abs().max()), nm


# ------------------------------------------------------------------------------------------------ attention on split rows
@pytest.mark.parametrize("n,L,H,dh,drop", [(3, 100, 12, 64, 0.0), (5, 100, 12, 64, 0.1), (2, 128, 4, 32, 0.0), (4, 37, 2, 64, 0.0)])
def test_attention_split_vs_torch(n, L, H, dh, drop):
    D = H * dh
    qkv, qkvv = _split(_rand(n * L, 3 * D, seed=24))
    km = (torch.rand(n * L, generator=torch.Generator().manual_seed(2)) > 0.15).float().to(DEV)
    km.view(n, L)[:, 0] = 1.0
    seq = SeqInfo.uniform(n, L, L, DEV)
    rng = O.make_rng(DEV, seed=9)
    d = (drop, rng, 5) if drop > 0 else None
    out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=d)
    assert O.lo_off(out) == D
    ref = E.attention(qkvv, qkvv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=d)
    err = float((_value(out) - ref).abs().max())
    assert err <= 5e-5 * float(ref.abs().max()), err
    # backward: the bf16 kernel on the hi planes
    qkv.requires_grad_(True)
    out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=d)
    g = _rand(n * L, D, seed=25).to(torch.bfloat16)
    out.backward(g)
    qr = qkvv.clone().requires_grad_(True)
    E.attention(qr, qr, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=d).backward(g.float())
    assert float((qkv.grad.float() - qr.grad).abs().max()) <= 3e-2 * float(qr.grad.abs().max())


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,lo,hi,drop", [(60, 33, 104, 0.1), (60, 100, 100, 0.0), (7, 64, 97, 0.1)])
def test_pipelined_attention_forward_vs_torch_and_vs_one_workgroup_per_pair(mode, n, lo, hi, drop):
    """attention_pipe.hip — the persistent forward of the clip encoder (a loader wave keeps the next two (sequence, head) pairs' K / V
    planes in flight by LDS-DMA, seven waves compute the current one): ragged sequences of 33-104 rows, 720 pairs on 256 / 512
    workgroups (three pairs each and more: every stage of the ring is reused), key-pad mask, dropout — against the fp32 reference (which
    applies the kernels' own dropout draws) and against the one-workgroup-per-pair kernels it replaces; the backward (which recomputes
    the probabilities from the forward's LSE and the same draws) against autograd."""
    from svpc_amd import _lib
    H, dh = 12, 64
    D = H * dh
    O.set_precision(mode)
    try:
        g = torch.Generator().manual_seed(5)
        lens = torch.randint(lo, hi + 1, (n,), generator=g).tolist()
        lens[0] = hi
        offs = [sum(lens[:i]) for i in range(n)]
        R = sum(lens)
        seq = SeqInfo(offs, lens, offs, lens, DEV)
        src = _rand(R, 3 * D, seed=60)
        if mode == "bf16x3":
            qkv, qkvv = _split(src)
        else:
            qkv = src.to(torch.bfloat16)
            qkvv = qkv.float()
        km = (torch.rand(R, generator=torch.Generator().manual_seed(6)) > 0.2).float().to(DEV)
        for o in offs:
            km[o] = 1.0
        rng = O.make_rng(DEV, seed=17)
        d = (drop, rng, 3) if drop > 0 else None
        ref = E.attention(qkvv, qkvv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=d)
        lib = _lib.load()
        outs = []
        for on in (0, 1):
            was = lib.svpc_attn_pipe_enable(on)
            try:
                out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=d)
                outs.append(_value(out) if mode == "bf16x3" else out.float())
            finally:
                lib.svpc_attn_pipe_enable(was)
        tol = 5e-5 if mode == "bf16x3" else 1.2e-2
        for o in outs:
            assert bool(torch.isfinite(o).all())
            assert float((o - ref).abs().max()) <= tol * float(ref.abs().max())
        assert float((outs[0] - outs[1]).abs().max()) <= tol * float(ref.abs().max())
        # backward through the pipelined forward's LSE
        assert lib.svpc_attn_pipe_enable(-1) == 1
        qkv.requires_grad_(True)
        out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=d)
        gr = _rand(R, D, seed=61).to(torch.bfloat16)
        out.backward(gr)
        qr = qkvv.clone().requires_grad_(True)
        E.attention(qr, qr, (0, D, 2 * D), D, H, seq, key_mask=km, causal=False, drop=d).backward(gr.float())
        assert float((qkv.grad.float() - qr.grad).abs().max()) <= 3e-2 * float(qr.grad.abs().max())
    finally:
        O.set_precision("fp32")


@pytest.mark.parametrize("scale_in", [1.0, 3.0])
def test_bf16_attention_forward_error_statistics(scale_in):
    """the bf16 clip-encoder forward against an fp64 reference on the same bf16 inputs: the kernel's own rounding (operands of the second
    product, fp32 accumulation) may add at most half of what the rounding of the bf16 OUTPUT alone costs, and no bias — a bound on
    the kernel itself, where the whole-model loss deviation of the bf16 mode is the net of millions of such roundings"""
    import math
    from svpc_amd import _lib
    H, dh, B, L = 12, 64, 48, 100
    D = H * dh
    seq = SeqInfo.uniform(B, L, L, DEV)
    qkv = (scale_in * _rand(B * L, 3 * D, seed=70)).to(torch.bfloat16).contiguous()
    xv = qkv.double()
    q, k, v = (xv[:, i * D:(i + 1) * D].view(B, L, H, dh).permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1) @ v).permute(0, 2, 1, 3).reshape(B * L, D)
    O.set_precision("bf16")
    try:
        for on in (1, 0):
            was = _lib.load().svpc_attn_pipe_enable(on)
            try:
                out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=None, causal=False, drop=None)
            finally:
                _lib.load().svpc_attn_pipe_enable(was)
            e = out.double() - ref
            floor = float((ref.to(torch.bfloat16).double() - ref).pow(2).mean().sqrt())
            assert float(e.pow(2).mean().sqrt()) <= 1.5 * floor, (on, float(e.pow(2).mean().sqrt()), floor)
            assert abs(float(e.mean())) <= 0.05 * floor, (on, float(e.mean()), floor)
    finally:
        O.set_precision("fp32")


@pytest.mark.parametrize("n,Lq,Lk,H,dh,causal,drop", [(7, 22, 22, 12, 64, True, 0.0), (7, 22, 22, 12, 64, True, 0.1), (9, 22, 3, 12, 64, False, 0.0),
                                                     (5, 32, 32, 4, 32, True, 0.0), (3, 6, 6, 4, 32, True, 0.0), (4, 22, 1, 12, 64, False, 0.0)])
def test_attention_short_sequences_split_vs_torch(n, Lq, Lk, H, dh, causal, drop):
    """the decoder's causal self-attention (packed Q/K/V) and its memory cross-attention (separate K|V rows) on split rows: one wave per
    (sequence, head); backward = the bf16 PERWAVE kernel on the hi planes"""
    D = H * dh
    same = Lq == Lk and causal
    if same:
        qt, qv = _split(_rand(n * Lq, 3 * D, seed=40))
        kt, kv, cols = qt, qv, (0, D, 2 * D)
    else:
        qt, qv = _split(_rand(n * Lq, D, seed=41))
        kt, kv = _split(_rand(n * Lk, 2 * D, seed=42))
        cols = (0, 0, D)
    km = torch.ones(n * Lk, device=DEV)
    if Lk > 3:
        km.view(n, Lk)[:, -2:] = 0.0
    seq = SeqInfo.uniform(n, Lq, Lk, DEV)
    rng = O.make_rng(DEV, seed=11)
    d = (drop, rng, 9) if drop > 0 else None
    qt.requires_grad_(True)
    if not same:
        kt.requires_grad_(True)
    out = O.attention(qt, kt, cols, D, H, seq, key_mask=km, causal=causal, drop=d)
    assert O.lo_off(out) == D
    qr = qv.clone().requires_grad_(True)
    kr = qr if same else kv.clone().requires_grad_(True)
    ref = E.attention(qr, kr, cols, D, H, seq, key_mask=km, causal=causal, drop=d)
    err = float((_value(out) - ref).abs().max())
    assert err <= 5e-5 * float(ref.abs().max()), err
    g = _rand(n * Lq, D, seed=43).to(torch.bfloat16)
    out.backward(g)
    ref.backward(g.float())
    if Lk > 1:       # (a single key: the softmax is the constant 1 and the query gradient is exactly zero — the bf16 backward leaves noise)
        assert float((qt.grad.float() - qr.grad).abs().max()) <= 3e-2 * float(qr.grad.abs().max()) + 1e-6
    if not same:
        assert float((kt.grad.float() - kr.grad).abs().max()) <= 3e-2 * float(kr.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("n,Lk,H,dh,drop,split", [(9, 100, 12, 64, 0.0, True), (9, 100, 12, 64, 0.1, True), (9, 100, 12, 64, 0.1, False),
                                                   (5, 128, 4, 32, 0.0, True), (3, 37, 4, 32, 0.0, False), (4, 1, 12, 64, 0.0, True),
                                                   (5, 128, 4, 64, 0.1, True), (3, 37, 4, 64, 0.0, False), (6, 64, 2, 64, 0.0, True),
                                                   (7, 65, 3, 64, 0.1, False)])
def test_one_query_attention_over_stream_rows(n, Lk, H, dh, drop, split):
    """the [CLS]-only clip-encoder layer: ONE fp32 query per clip against the stream's K | V rows in place — split rows (bf16x3 mode) or
    bf16 rows (bf16 mode); exact fp32 arithmetic on the stored values, forward and backward (attention_q1s.hip)"""
    D = H * dh
    O.set_precision("bf16x3" if split else "bf16")
    src = _rand(n * Lk, 2 * D, seed=50)
    if split:
        kv, kvv = _split(src)
    else:
        kv = src.to(torch.bfloat16)
        kvv = kv.float()
    q = _rand(n, D, seed=51).requires_grad_(True)
    kv.requires_grad_(True)
    km = (torch.rand(n * Lk, generator=torch.Generator().manual_seed(3)) > 0.1).float().to(DEV)
    km.view(n, Lk)[:, 0] = 1.0
    seq = SeqInfo(list(range(n)), [1] * n, [i * Lk for i in range(n)], [Lk] * n, DEV)
    rng = O.make_rng(DEV, seed=13)
    d = (drop, rng, 4) if drop > 0 else None
    out = O.attention(q, kv, (0, 0, D), D, H, seq, key_mask=km, causal=False, drop=d)
    assert out.dtype == torch.float32 and out.shape == (n, D)
    qr, kr = q.detach().clone().requires_grad_(True), kvv.clone().requires_grad_(True)
    ref = E.attention(qr, kr, (0, 0, D), D, H, seq, key_mask=km, causal=False, drop=d)
    assert float((out - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    g = _rand(n, D, seed=52)
    out.backward(g)
    ref.backward(g)
    if Lk > 1:
        assert float((q.grad - qr.grad).abs().max()) <= 1e-4 * float(qr.grad.abs().max()) + 1e-7
    assert kv.grad.dtype == torch.bfloat16 and kv.grad.shape == (n * Lk, 2 * D)
    assert float((kv.grad.float() - kr.grad).abs().max()) <= 6e-3 * float(kr.grad.abs().max()) + 1e-7


def test_untagged_bf16_tensor_fails_loudly_in_x3_mode():
    """a torch view / cast of a split tensor drops the lo-plane tag; in bf16x3 mode the forward ops refuse such a tensor instead of
    computing a one-term product on its hi plane (the mode's ≤ 1e-4 contract would break silently)"""
    from svpc_amd._lib import SvpcKernelError
    x, _ = _split(_rand(256, 128, seed=80))
    w, b = _rand(128, 128, seed=81), _rand(128, seed=82)
    g, be = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    assert O.lo_off(x) == 128
    O.linear(x, w, b)                                    # tagged: fine
    untagged = x[:, :]                                   # a plain torch view of the hi plane
    assert O.lo_off(untagged) is None
    with pytest.raises(SvpcKernelError, match="lo-plane tag"):
        O.linear(untagged, w, b)
    with pytest.raises(SvpcKernelError, match="lo-plane tag"):
        O.layernorm(untagged, g, be, 1e-12)
    seq = SeqInfo.uniform(2, 128, 128, DEV)
    qkv, _ = _split(_rand(256, 3 * 128, seed=83))
    with pytest.raises(SvpcKernelError, match="lo-plane tag"):
        O.attention(qkv[:, :], qkv[:, :], (0, 128, 256), 128, 2, seq)
    assert O.to_f32(x).dtype == torch.float32            # the explicit way out of the split domain


def test_split_cols_keeps_the_planes_and_gathers_gradients():
    wide, wv = _split(_rand(40, 6 * 128, seed=44))
    wide.requires_grad_(True)
    blocks = O.split_cols(wide, 3)
    assert all(O.lo_off(b) == 6 * 128 for b in blocks)
    for i, b in enumerate(blocks):
        assert torch.equal(O.to_f32(b), wv[:, i * 256:(i + 1) * 256])
    t, tv = O.to_split(wv[:, :128].contiguous()), wv[:, :128]
    assert O.lo_off(t) == 128 and float((O.to_f32(t) - tv).abs().max()) <= 1e-5 * float(tv.abs().max())


def test_attention_fp32_storage_is_exact_forward_with_mfma_backward():
    n, L, H, dh = 6, 22, 12, 64
    D = H * dh
    qkv = _rand(n * L, 3 * D, seed=26).requires_grad_(True)
    seq = SeqInfo.uniform(n, L, L, DEV)
    km = torch.ones(n * L, device=DEV)
    out = O.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=km, causal=True)
    qr = qkv.detach().clone().requires_grad_(True)
    ref = E.attention(qr, qr, (0, D, 2 * D), D, H, seq, key_mask=km, causal=True)
    assert float((out - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    g = _rand(n * L, D, seed=27)
    out.backward(g)
    ref.backward(g)
    assert float((qkv.grad - qr.grad).abs().max()) <= 2e-2 * float(qr.grad.abs().max())


def test_leaving_the_split_domain():
    t, tv = _split(_rand(50, 128, seed=28))
    assert torch.equal(O.to_f32(t), tv)
    idx = torch.tensor([3, 0, 49, 3], dtype=torch.int32, device=DEV)
    assert torch.equal(O.take_rows_f32(t, idx), tv[idx.long()])
    t.requires_grad_(True)
    O.take_rows_f32(t, idx).sum().backward()
    assert float(t.grad.float().sum()) == 4.0 * 128 and float(t.grad[3].float().sum()) == 2.0 * 128


# ------------------------------------------------------------------------------------------------ the module, end to end
@pytest.mark.parametrize("mt", ["v", "vi", "viv", "vivt"])
def test_tiny_forward_vs_reference_golden_x3(golden_dir, mt):
    """tiny config (D=32: no split stream — every projection is an fp32-storage x3 product): loss ≤ 1e-4 relative to the reference's
    own golden loss, probabilities to 3e-4"""
    z, cfg, batch, model = build_model("tiny", mt, golden_dir, DEV)
    loss, probs, ents, acts = model(*syn.forward_args(batch))
    ref = float(z["loss"])
    assert abs(loss.item() - ref) <= 1e-4 * abs(ref), (loss.item(), ref)
    for b, p in enumerate(probs):
        np.testing.assert_allclose(p.detach().cpu().numpy(), z["probs/%d" % b], rtol=3e-4, atol=1e-6)
    loss.backward()
    O.join_side()
    for name, p in model.named_parameters():
        k = "grad/" + name
        if k in z.files and float(np.abs(z[k]).max()) > 1e-4:
            g, r = p.grad.detach().double().cpu().reshape(-1), torch.from_numpy(z[k]).double().reshape(-1)
            cos = float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-300))
            assert cos >= 0.99, (name, cos)


@pytest.mark.parametrize("mt", ["v", "vivt"])
def test_c1_split_stream_vs_reference_golden_x3(golden_dir, mt):
    """config-1 shape (D=128, F=3072: the clip encoder runs on the split stream): loss ≤ 1e-4 vs the reference golden, argmax of
    the probabilities identical, gradient norms at the bf16 mode's accuracy"""
    z, cfg, batch, model = build_model("c1", mt, golden_dir, DEV)
    assert O.bf16_stream_ok(800, cfg.hidden_size, cfg.video_feature_size, cfg.intermediate_size)
    loss, probs, ents, acts = model(*syn.forward_args(batch))
    ref = float(z["loss"])
    assert abs(loss.item() - ref) <= 1e-4 * abs(ref), (loss.item(), ref)
    for b, p in enumerate(probs):
        p = p.detach().cpu().numpy()
        np.testing.assert_allclose(p[:, :, ::37], z["probs_slice/%d" % b], rtol=1e-3, atol=1e-6)
        assert (p.argmax(-1) == z["probs_argmax/%d" % b]).mean() > 0.999
    loss.backward()
    O.join_side()
    biggest = max(float(z[k]) for k in z.files if k.startswith("gradnorm/"))
    for k in z.files:
        if k.startswith("gradnorm/"):
            name = k[len("gradnorm/"):]
            g = dict(model.named_parameters())[name].grad
            refn = float(z[k])
            if refn < 1e-5:      # analytically zero gradients (key biases: softmax shift invariance; the single-key cross-attention of
                                 # MODEL_TYPE=v): what the bf16 backward leaves there is rounding noise — bounded against the real gradients
                assert float(g.double().norm()) <= 2e-2 * biggest, (name, float(g.double().norm()), biggest)
                continue
            assert abs(float(g.double().norm()) - refn) <= 6e-2 * refn + 2e-4, (name, float(g.double().norm()), refn)


@pytest.mark.parametrize("case,mt", [("tiny", "vivt"), ("c1", "v"), ("c1", "vivt")])
def test_greedy_decode_ids_vs_reference_x3(golden_dir, case, mt):
    """translate_batch in the bf16x3 mode against the reference Translator's token ids (bit-exact expected at these sizes: the
    mode's error is ≈1e-6, far below the logit gaps)"""
    from svpc_amd.translator import Translator
    z, cfg, batch, model = build_model(case, mt, golden_dir, DEV)
    tr = Translator(type("O", (), {"cuda": True})(), {"model_cfg": cfg, "model": model.state_dict()}, model=model)
    dec, _ = tr.translate_batch(syn.translate_inputs(batch))
    for b, d in enumerate(dec):
        np.testing.assert_array_equal(d.cpu().numpy(), z["decode/%d" % b])


def test_training_step_through_the_arena_x3(golden_dir):
    """the captured step's path in bf16x3 mode: weight store with a lo plane kept by the Adam kernel, gradients written in place;
    two optimizer steps keep hi + lo == bf16-split of the fp32 master weights"""
    from svpc_amd.optim import FusedBertAdam
    z, cfg, batch, model = build_model("c1", "vivt", golden_dir, DEV)
    model.train()
    opt = FusedBertAdam(list(model.named_parameters()), lr=1e-3, grad_clip=1.0)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = model(*syn.forward_args(batch))[0]
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)), losses
    st = opt.weights
    assert st.with_lo
    hi = st.shadow.float()
    assert torch.equal(st.shadow, st.flat.to(torch.bfloat16))
    assert torch.equal(st.shadow_lo, (st.flat - hi).to(torch.bfloat16))
