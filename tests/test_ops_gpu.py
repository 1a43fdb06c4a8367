"""Per-kernel parity: every HIP primitive (through the C-ABI, via svpc_amd.ops) against its plain-PyTorch fp32
statement (tests/emul_ops.py) on the same seeded inputs, forward and backward.  Tolerances are written per test;
index outputs are bit-exact.  Needs an MI355X: run with ``-m gpu``."""
import math

import pytest
import torch

import emul_ops as E
from svpc_amd import ops as O
from svpc_amd.ops_common import FIdx, Idx, SeqInfo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0, grad=True):
    g = torch.Generator().manual_seed(seed + sum(shape) * 7 + len(shape))
    t = (torch.randn(*shape, generator=g) * scale).to(DEV)
    return t.requires_grad_(grad)


def compare(fn_hip, fn_ref, inputs, rtol=2e-4, atol=2e-5, grad_rtol=5e-4, grad_atol=5e-5, name=""):
    """Run both on clones of `inputs` (list of tensors / None), compare outputs and input grads."""
    def run(fn):
        ins = [None if t is None else (t.detach().clone().requires_grad_(t.requires_grad) if t.dtype.is_floating_point else t)
               for t in inputs]
        out = fn(*ins)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        g = torch.Generator().manual_seed(99)
        loss = 0
        for o in outs:
            if o.dtype.is_floating_point:
                w = torch.randn(o.shape, generator=g).to(o.device)
                loss = loss + (o * w).sum()
        if any(t is not None and t.dtype.is_floating_point and t.requires_grad for t in ins):
            loss.backward()
        return outs, [None if (t is None or not t.dtype.is_floating_point) else t.grad for t in ins]
    o1, g1 = run(fn_hip)
    o2, g2 = run(fn_ref)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(o1, o2)):
        if a.dtype.is_floating_point:
            err = (a - b).abs().max().item()
            tol = atol + rtol * b.abs().max().item()
            assert err <= tol, "%s out[%d]: err %.3e > tol %.3e" % (name, i, err, tol)
        else:
            assert torch.equal(a, b), "%s out[%d] index mismatch" % (name, i)
    for i, (a, b) in enumerate(zip(g1, g2)):
        if b is None:
            continue
        assert a is not None, "%s grad[%d] missing" % (name, i)
        err = (a - b).abs().max().item()
        tol = grad_atol + grad_rtol * b.abs().max().item()
        assert err <= tol, "%s grad[%d]: err %.3e > tol %.3e" % (name, i, err, tol)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 5, 3), (64, 64, 16), (100, 33, 42), (192, 768, 768), (257, 129, 300),
                                   (1000, 951, 128), (2048, 768, 3072), (16, 3072, 768), (192, 3, 768), (192, 1, 300)])
@pytest.mark.parametrize("act", [E.ACT_NONE, E.ACT_RELU, E.ACT_GELU, E.ACT_SIGMOID])
def test_linear(M, N, K, act):
    if act != E.ACT_NONE and M * N * K > 5e7:
        pytest.skip("big shapes only with the plain epilogue")
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    compare(lambda x, w, b: O.linear(x, w, b, act=act), lambda x, w, b: E.linear(x, w, b, act=act), [x, w, b],
            rtol=3e-5 * math.sqrt(K) + 2e-4, grad_rtol=1e-3, name="linear")


def test_linear_wgrad_splitk_long_k():
    x, w = rnd(19200, 768, seed=4), rnd(768, 768, seed=5, scale=0.03)
    compare(lambda x, w: O.linear(x, w, None), lambda x, w: E.linear(x, w, None), [x, w], rtol=1e-3, grad_rtol=2e-3, name="splitk")


def test_linear_trans_w_and_strided_input():
    big = rnd(50, 96, seed=6)
    w = rnd(32, 40, seed=7)   # (K, N)
    compare(lambda big, w: O.linear(big[:, 32:64], w, None, trans_w=True), lambda big, w: E.linear(big[:, 32:64], w, None, trans_w=True),
            [big, w], name="trans_w")


def test_linear_dropout_mask_consistency():
    rng = O.make_rng(DEV, seed=5)
    x, w, b = rnd(200, 64, seed=8), rnd(48, 64, seed=9), rnd(48, seed=10)
    drop = (0.4, rng, 3)
    compare(lambda x, w, b: O.linear(x, w, b, act=E.ACT_RELU, drop=drop), lambda x, w, b: E.linear(x, w, b, act=E.ACT_RELU, drop=drop),
            [x, w, b], name="linear+dropout")
    m = rng.mask(3, 200 * 48, 0.4, DEV)
    assert 0.5 < m.mean().item() < 0.7 and set(m.unique().tolist()) <= {0.0, 1.0}


# ------------------------------------------------------------------------------------------------ LayerNorm family
@pytest.mark.parametrize("R,D", [(1, 4), (5, 20), (37, 32), (130, 300), (260, 768), (33, 3072), (9, 10), (7, 42)])
def test_layernorm_plain(R, D):
    x, g, b = rnd(R, D, seed=1, scale=2.0), rnd(D, seed=2), rnd(D, seed=3)
    compare(lambda x, g, b: O.layernorm(x, g, b, 1e-12), lambda x, g, b: E.layernorm(x, g, b, 1e-12), [x, g, b], name="ln")


def test_layernorm_zero_row_gives_beta():
    x = torch.zeros(3, 64, device=DEV)
    g, b = rnd(64, seed=1, grad=False), rnd(64, seed=2, grad=False)
    y = O.layernorm(x, g, b, 1e-12)
    assert torch.allclose(y, b.expand_as(y))


def test_layernorm_fused_everything():
    rng = O.make_rng(DEV, seed=11)
    R, D, T = 96, 128, 50
    table = rnd(T, D, seed=1)
    src = torch.randint(0, T, (R,), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(DEV)
    res, g, b = rnd(R, D, seed=2), rnd(D, seed=3), rnd(D, seed=4)
    pe = rnd(12, D, seed=5, grad=False)
    tt = rnd(4, D, seed=6)
    idx2 = torch.randint(0, 4, (R,), generator=torch.Generator().manual_seed(2)).to(torch.int32).to(DEV)
    pre, post = (0.1, rng, 1), (0.2, rng, 2)

    def hip(table, res, g, b, tt):
        return O.layernorm(table, g, b, 1e-12, residual=res, src_rows=src, pad_row=0, pre_drop=pre, post_drop=post, add1=pe,
                           add1_mod=12, add2=tt, add2_idx=idx2)

    def ref(table, res, g, b, tt):
        return E.layernorm(E.embedding_table(table, 0), g, b, 1e-12, residual=res, src_rows=src, pad_row=0, pre_drop=pre,
                           post_drop=post, add1=pe, add1_mod=12, add2=tt, add2_idx=idx2)
    compare(hip, ref, [table, res, g, b, tt], name="ln fused")


@pytest.mark.parametrize("R,D,with_res", [(300, 3072, False), (77, 128, True), (4100, 768, False), (1031, 3072, False), (5000, 3072, False)])
def test_layernorm_parameter_gradients_only(R, D, with_res):
    """No input needs a gradient (the frame-feature LayerNorm): the backward is the streaming two-output column-sum kernel;
    gathered rows, post-LayerNorm dropout and an (ungraded) residual are honoured."""
    rng = O.make_rng(DEV, seed=5)
    table = rnd(R + 9, D, seed=1, grad=False)
    src = torch.randint(0, R + 9, (R,), generator=torch.Generator().manual_seed(3)).to(torch.int32).to(DEV)
    res = rnd(R, D, seed=2, grad=False) if with_res else None
    g, b = rnd(D, seed=3), rnd(D, seed=4)
    post = (0.2, rng, 2)
    compare(lambda g, b: O.layernorm(table, g, b, 1e-12, residual=res, src_rows=src, pad_row=-1, post_drop=post),
            lambda g, b: E.layernorm(table, g, b, 1e-12, residual=res, src_rows=src, pad_row=-1, post_drop=post),
            [g, b], name="ln param grads")


# ------------------------------------------------------------------------------------------------ attention
def _attn_case(seq, D, H, causal, with_mask, drop, packed=True, cross=False, seed=0):
    Rq, Rk = seq.n_q_rows, seq.n_k_rows
    km = None
    if with_mask:
        km = (torch.rand(Rk, generator=torch.Generator().manual_seed(seed)) > 0.3).float().to(DEV)
        for o in seq.h_k_off:
            km[o] = 1.0
    if cross:
        qt, kvt = rnd(Rq, D, seed=seed + 1), rnd(Rk, 2 * D, seed=seed + 2)
        cols = (0, 0, D)
        compare(lambda qt, kvt: O.attention(qt, kvt, cols, D, H, seq, km, causal, drop),
                lambda qt, kvt: E.attention(qt, kvt, cols, D, H, seq, km, causal, drop), [qt, kvt], name="attn cross")
    else:
        qkv = rnd(Rq, 3 * D, seed=seed + 3)
        cols = (0, D, 2 * D)
        compare(lambda qkv: O.attention(qkv, qkv, cols, D, H, seq, km, causal, drop),
                lambda qkv: E.attention(qkv, qkv, cols, D, H, seq, km, causal, drop), [qkv], name="attn self")


def test_attention_encoder_shape():
    _attn_case(SeqInfo.uniform(6, 100, 100, DEV), 768, 12, False, True, None)


def test_attention_small_heads_and_causal():
    _attn_case(SeqInfo.uniform(5, 6, 6, DEV), 32, 4, True, True, None)
    _attn_case(SeqInfo.uniform(7, 22, 22, DEV), 128, 4, True, True, None, seed=3)


def test_attention_cross_few_keys():
    _attn_case(SeqInfo.uniform(9, 22, 3, DEV), 128, 4, False, False, None, cross=True)
    _attn_case(SeqInfo.uniform(4, 6, 1, DEV), 32, 4, False, False, None, cross=True, seed=5)


def test_attention_ragged_step_sequences():
    lens = [3, 1, 12, 7]
    off = [0, 3, 4, 16]
    _attn_case(SeqInfo(off, lens, off, lens, DEV), 64, 4, False, False, None)


def test_attention_long_keys_multi_chunk():
    _attn_case(SeqInfo.uniform(2, 130, 130, DEV), 64, 2, False, True, None, seed=7)


def test_attention_dropout():
    rng = O.make_rng(DEV, seed=3)
    _attn_case(SeqInfo.uniform(3, 20, 20, DEV), 64, 4, True, True, (0.1, rng, 4))


# ------------------------------------------------------------------------------------------------ rows / spans / losses
def test_span_mean_and_rownorm_softmax():
    x = rnd(40, 96, seed=1)
    starts, lens = Idx([0, 5, 6, 20, 33]), Idx([5, 1, 10, 13, 7])
    w = (torch.rand(40, generator=torch.Generator().manual_seed(3)) > 0.4).float().to(DEV)
    for s in starts.host:
        w[s] = 1.0
    pe = rnd(9, 96, seed=2, grad=False)
    aidx = Idx([3, 0, 8, 1, 1])
    compare(lambda x: O.span_mean(x, starts, lens, w, pe, aidx), lambda x: E.span_mean(x, starts, lens, w, pe, aidx), [x], name="span w")
    compare(lambda x: O.span_mean(x, starts, lens), lambda x: E.span_mean(x, starts, lens), [x], name="span")
    # spans that tile the rows (the backward then writes every row itself: no zero fill), with and without weights
    st2, ln2 = Idx([0, 5, 6, 16, 29]), Idx([5, 1, 10, 13, 11])
    for s in st2.host:
        w[s] = 1.0
    compare(lambda x: O.span_mean(x, st2, ln2, w, pe, aidx), lambda x: E.span_mean(x, st2, ln2, w, pe, aidx), [x], name="span tiling w")
    compare(lambda x: O.span_mean(x, st2, ln2), lambda x: E.span_mean(x, st2, ln2), [x], name="span tiling")
    a = torch.sigmoid(rnd(17, 384, seed=4)).detach().requires_grad_(True)
    compare(O.row_normalize, E.row_normalize, [a], name="row_normalize")
    compare(O.softmax_rows, E.softmax_rows, [rnd(50, 3, seed=5)], name="softmax3")
    compare(O.sum_all, E.sum_all, [rnd(1000, seed=6)], rtol=1e-4, name="sum")
    compare(O.add, E.add, [rnd(10, 33, seed=7), rnd(10, 33, seed=8)], name="add")


def test_losses():
    p = torch.sigmoid(rnd(13, 31, seed=1)).detach().requires_grad_(True)
    y = (torch.rand(13, 31, generator=torch.Generator().manual_seed(2)) < 0.2).float().to(DEV)
    widths = Idx([31, 5, 1, 10, 31, 7, 8, 9, 2, 3, 30, 29, 11])
    compare(lambda p: O.bce_rows(p, y, widths), lambda p: E.bce_rows(p, y, widths), [p], name="bce")
    pa = torch.sigmoid(rnd(13, 384, seed=3, scale=3.0)).detach().requires_grad_(True)
    ya = (torch.rand(13, 384, generator=torch.Generator().manual_seed(4)) < 0.02).float().to(DEV)
    ya[3] = 0
    act = O.row_any_eq1(ya)
    assert torch.equal(act, E.row_any_eq1(ya))
    compare(lambda p: O.asl_rows(p, ya, act), lambda p: E.asl_rows(p, ya, act), [pa], name="asl")
    lab = torch.tensor([3, -1, 60, 49, 50], dtype=torch.int32, device=DEV)
    assert torch.equal(O.clamp_labels(lab, 50, 6), E.clamp_labels(lab, 50, 6))


def test_lstm_cell():
    N, D = 5, 48
    gx, gh, c, h = rnd(N, 4 * D, seed=1), rnd(N, 4 * D, seed=2), rnd(N, D, seed=3), rnd(N, D, seed=4)
    active = torch.tensor([1, 1, 0, 1, 0], dtype=torch.float32, device=DEV)
    compare(lambda gx, gh, c, h: O.lstm_cell(gx, gh, c, h, active), lambda gx, gh, c, h: E.lstm_cell(gx, gh, c, h, active),
            [gx, gh, c, h], name="lstm")


# ------------------------------------------------------------------------------------------------ simulator / pointer / gumbel
@pytest.mark.parametrize("D,ent_len", [(32, [3, 31, 10]), (768, [3, 31, 10]), (768, [3, 10, 7]), (256, [12, 1, 5]), (512, [3, 24, 2])])
def test_sim_recur(D, ent_len):
    """entity-state recurrence fwd/bwd; the cases cover the three backward staging modes (state + upstream gradient by
    double-buffered LDS-DMA / both staged synchronously / state only) and 256…768-thread workgroups"""
    step_len = [3, 1, 12]
    step_off, ent_off = [0, 3, 4], [0, ent_len[0], ent_len[0] + ent_len[1]]
    T, NE, em = 16, sum(ent_len), max(ent_len)
    q = rnd(T, D, seed=1, scale=1 / math.sqrt(D))
    c = torch.softmax(rnd(T, 3, seed=2), -1).detach().requires_grad_(True)
    w4f = rnd(T, seed=3)
    E0 = rnd(NE, D, seed=4)
    args = (Idx(step_off), Idx(step_len), Idx(ent_off), Idx(ent_len), em)
    compare(lambda q, c, w, e0: O.sim_recur(q, c, w, e0, *args), lambda q, c, w, e0: E.sim_recur(q, c, w, e0, *args),
            [q, c, w4f, E0], rtol=5e-4, grad_rtol=2e-3, grad_atol=2e-4, name="sim_recur")


@pytest.mark.parametrize("T,D,Wd", [(192, 768, 300), (7, 128, 300), (33, 32, 20), (1, 64, 12)])
def test_sim_heads_forward_backward(T, D, Wd):
    """the simulator's choice softmax (D → 3) and verb scalar (W → 1) in one launch each way (round 5) against linear → softmax / linear,
    gradients through autograd and through arena-style direct targets (the table-driven finalizer)"""
    hh, fb = rnd(T, D, seed=1), rnd(T, Wd, seed=2)
    W3, b3, W4, b4 = rnd(3, D, seed=3, scale=0.2), rnd(3, seed=4), rnd(1, Wd, seed=5, scale=0.3), rnd(1, seed=6)
    compare(lambda *a: O.sim_heads(*a), lambda *a: E.sim_heads(*a), [hh, fb, W3, b3, W4, b4], grad_rtol=1e-3, grad_atol=1e-5, name="sim_heads")
    ps = [t.detach().clone().requires_grad_(True) for t in (W3, b3, W4, b4)]
    for t in ps:
        t.grad = torch.full_like(t, 0.25)
        t._svpc_direct = True
    c, w = O.sim_heads(hh.detach(), fb.detach(), *ps)
    gc_, gw = torch.randn(c.shape, generator=torch.Generator().manual_seed(3)).to(DEV), torch.randn(w.shape, generator=torch.Generator().manual_seed(4)).to(DEV)
    ((c * gc_).sum() + (w * gw).sum()).backward()
    O.join_side()
    rs = [t.detach().clone().requires_grad_(True) for t in (W3, b3, W4, b4)]
    cr, wr = E.sim_heads(hh.detach(), fb.detach(), *rs)
    ((cr * gc_).sum() + (wr * gw).sum()).backward()
    for a_, b_ in zip(ps, rs):
        assert float((a_.grad - 0.25 - b_.grad).abs().max()) <= 1e-5 + 1e-3 * float(b_.grad.abs().max())


def test_ptr_attn_entity_chunks_and_full_width():
    """pointer attention at the production row width with more entities than one staged chunk (16) holds, and lt = 1
    (incremental decoding)"""
    for T, lt, em, D, ne in ((3, 22, 20, 768, [20, 17, 3]), (4, 1, 5, 128, [5, 1, 4, 2])):
        step_ne = Idx(ne)
        dec, proj, bank = rnd(T * lt, D, seed=1, scale=0.2), rnd(T, em, D, seed=2, scale=0.2), rnd(T, em, D, seed=3)
        compare(lambda d, p, b: O.ptr_attn(d, p, b, step_ne, lt), lambda d, p, b: E.ptr_attn(d, p, b, step_ne, lt), [dec, proj, bank],
                name="ptr_attn wide")


@pytest.mark.parametrize("T,lt,nm,D,H,kind,p", [(5, 22, 3, 768, 12, "f32", 0.0), (5, 22, 3, 768, 12, "x3", 0.1), (3, 22, 2, 768, 12, "bf16", 0.0),
                                                (4, 6, 3, 128, 4, "f32", 0.1), (2, 22, 1, 768, 12, "x3", 0.0), (3, 9, 3, 256, 4, "x3", 0.1),
                                                (2, 24, 3, 512, 8, "bf16", 0.1)])
def test_cross_attn_ln_fused_vs_attention_then_layernorm(T, lt, nm, D, H, kind, p):
    """ops.cross_attn_ln (the decoder's cross-attention over the sentence's n_mem memory rows + residual + LayerNorm in one launch, forward
    and backward, round 5) against the plain statement of reference model.py:657-658 / :194-219 / :143-156 — outputs and the gradients of the
    query rows, the residual rows, [K | V], gamma, beta; split / bf16 / fp32 rows, with and without dropout of the probabilities (the same
    counter-based draw on both sides), gradients through the residual-gradient sink too."""
    import math
    dh = D // H
    Rm, R = T * nm, T * lt
    q, x1 = rnd(R, D, seed=1, scale=0.5), rnd(R, D, seed=7)
    kv = rnd(Rm, 2 * D, seed=2, scale=0.7)
    gamma, beta = (1.0 + 0.1 * rnd(D, seed=5, grad=False)).requires_grad_(True), rnd(D, seed=6, scale=0.1)
    rng = O.make_rng(DEV)
    site = 7
    drop = (p, rng, site) if p > 0 else None
    gout = torch.randn(R, D, generator=torch.Generator().manual_seed(9)).to(DEV)
    conv = {"x3": O.to_split, "bf16": lambda t: t.to(torch.bfloat16), "f32": lambda t: t}[kind]
    O.set_precision("bf16" if kind == "bf16" else ("bf16x3" if kind == "x3" else "fp32"))
    try:
        assert O.cross_attn_ln_usable(D, H, lt, nm)
        y = O.cross_attn_ln(conv(q), conv(x1), conv(kv), gamma, beta, 1e-12, H, lt, nm, drop=drop)
        yf = O.to_f32(y)
        (yf * gout).sum().backward()
        O.join_side()
        got = [t.grad.clone() for t in (q, x1, kv, gamma, beta)]
        for t in (q, x1, kv, gamma, beta):
            t.grad = None
    finally:
        O.set_precision("fp32")
    # reference in fp32 torch on the values the kernels saw (bf16 kinds: the rounded inputs)
    rd = (lambda t: t.detach().to(torch.bfloat16).float()) if kind == "bf16" else (lambda t: t.detach())
    qr, xr, kvr = (rd(t).requires_grad_(True) for t in (q, x1, kv))
    k = kvr[:, :D].view(T, nm, H, dh)
    v = kvr[:, D:].view(T, nm, H, dh)
    sc = torch.einsum("sthc,sjhc->shtj", qr.view(T, lt, H, dh), k) / math.sqrt(dh)
    pr = torch.softmax(sc, -1)
    if p > 0:
        pr = pr * rng.attn_mask(site, T * H * lt, nm, p, DEV).view(T, H, lt, nm) / (1.0 - p)
    o = torch.einsum("shtj,sjhc->sthc", pr, v).reshape(R, D)
    ref = torch.nn.functional.layer_norm(xr + o, (D,), gamma, beta, 1e-12)
    (ref * gout).sum().backward()
    want = [qr.grad, xr.grad, kvr.grad, gamma.grad, beta.grad]
    tol = 2e-2 if kind == "bf16" else 2e-5          # (bf16: the rounding of the stored output)
    assert float((yf.detach() - ref.detach()).abs().max()) <= tol * max(1.0, float(ref.detach().abs().max())), float((yf.detach() - ref.detach()).abs().max())
    gtol = 2e-2 if kind != "f32" else 1e-4          # (bf16 / split streams: the gradients arrive and leave as dense bf16 rows)
    for name, g, w in zip(("q", "x1", "kv", "gamma", "beta"), got, want):
        err = float((g.float() - w).abs().max())
        assert err <= gtol * float(w.abs().max()) + 1e-6, (name, err, float(w.abs().max()))


@pytest.mark.parametrize("kind,D,H,p", [("x3", 768, 12, 0.1), ("f32", 768, 12, 0.0), ("bf16", 128, 4, 0.1)])
def test_cross_attn_ln_ragged_rows_equal_the_padded_call(kind, D, H, p):
    """the fused decoder cross-attention over RAGGED sentences (the valid tokens only: rows = (row_off, row_len), round 5) gives, at every
    valid row, what the uniform call over the padded layout gives — outputs and the gradients of q, x1, [K | V], gamma, beta (the pad rows
    of the padded call receive a zero output gradient, as the loss gives them)"""
    T, lt, nm = 7, 22, 3
    lens = [22, 7, 13, 1, 22, 9, 16]
    off, acc = [], 0
    for n in lens:
        off.append(acc); acc += n
    R, Rp = T * lt, acc
    valid = torch.tensor([j * lt + t for j, n in enumerate(lens) for t in range(n)], dtype=torch.long, device=DEV)
    q, x1 = rnd(R, D, seed=1, scale=0.5, grad=False), rnd(R, D, seed=7, grad=False)
    kv0 = rnd(T * nm, 2 * D, seed=2, scale=0.7, grad=False)
    g0, b0 = (1.0 + 0.1 * rnd(D, seed=5, grad=False)), rnd(D, seed=6, scale=0.1, grad=False)
    rng = O.make_rng(DEV)
    drop = (p, rng, 7) if p > 0 else None
    gout = torch.randn(R, D, generator=torch.Generator().manual_seed(9)).to(DEV)
    gmask = torch.zeros(R, 1, device=DEV)
    gmask[valid] = 1.0
    conv = {"x3": O.to_split, "bf16": lambda t: t.to(torch.bfloat16), "f32": lambda t: t}[kind]
    O.set_precision("bf16" if kind == "bf16" else ("bf16x3" if kind == "x3" else "fp32"))
    res = []
    try:
        for ragged in (False, True):
            qq = (q[valid] if ragged else q).clone().requires_grad_(True)
            xx = (x1[valid] if ragged else x1).clone().requires_grad_(True)
            kv, ga, be = kv0.clone().requires_grad_(True), g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            rows = (torch.tensor(off, dtype=torch.int32, device=DEV), torch.tensor(lens, dtype=torch.int32, device=DEV)) if ragged else None
            y = O.to_f32(O.cross_attn_ln(conv(qq), conv(xx), conv(kv), ga, be, 1e-12, H, lt, nm, drop=drop, rows=rows))
            go = gout[valid] if ragged else gout * gmask
            (y * go).sum().backward()
            O.join_side()
            pick = (lambda t: t) if ragged else (lambda t: t[valid])
            res.append([pick(y.detach()), pick(qq.grad), pick(xx.grad), kv.grad, ga.grad, be.grad])
    finally:
        O.set_precision("fp32")
    tol = 1e-6 if kind == "f32" else 2e-2
    for name, a_, b_ in zip(("y", "dq", "dx1", "dkv", "dgamma", "dbeta"), res[0], res[1]):
        err = float((a_.float() - b_.float()).abs().max())
        assert err <= tol * float(a_.float().abs().max()) + 1e-6, (name, err)


@pytest.mark.parametrize("T,lt,em,D,ne", [(3, 22, 20, 768, [20, 17, 3]), (5, 6, 4, 64, [3, 3, 4, 2, 2]), (4, 1, 5, 128, [5, 1, 4, 2]),
                                          (2, 22, 31, 768, [31, 1]), (3, 7, 6, 200, [6, 2, 5])])
def test_ptr_attn_gate_forward_backward(T, lt, em, D, ne):
    """pointer attention + generation gate in one launch (training form, round 5) against ptr_attn → cat → linear → sigmoid: both outputs
    and the gradients of dec, proj, bank, the gate weight and bias — through autograd and through arena-style direct gradients"""
    step_ne = Idx(ne)
    dec, proj, bank = rnd(T * lt, D, seed=1, scale=0.2), rnd(T, em, D, seed=2, scale=0.2), rnd(T, em, D, seed=3)
    w, b = rnd(1, 2 * D, seed=4, scale=0.1), rnd(1, seed=5)
    compare(lambda d, p, bk, w_, b_: O.ptr_attn_gate(d, p, bk, step_ne, lt, w_, b_), lambda d, p, bk, w_, b_: E.ptr_attn_gate(d, p, bk, step_ne, lt, w_, b_),
            [dec, proj, bank, w, b], grad_rtol=1e-3, grad_atol=1e-4, name="ptr_attn_gate")
    # ragged sentences (the valid tokens only, round 5): the same rows through (row_off, row_len) as through the padded layout
    lens = [max(1, (lt * (j + 1)) // (T + 1) + (j % 2)) if lt > 1 else 1 for j in range(T)]
    lens = [min(lt, n) for n in lens]
    off, acc = [], 0
    for n in lens:
        off.append(acc); acc += n
    rows = (torch.tensor(off, dtype=torch.int32, device=DEV), torch.tensor(lens, dtype=torch.int32, device=DEV))
    valid = torch.tensor([j * lt + t for j, n in enumerate(lens) for t in range(n)], dtype=torch.long, device=DEV)
    dec_p = dec.detach()[valid].clone().requires_grad_(True)
    compare(lambda d, p, bk, w_, b_: O.ptr_attn_gate(d, p, bk, step_ne, lt, w_, b_, rows=rows),
            lambda d, p, bk, w_, b_: E.ptr_attn_gate(d, p, bk, step_ne, lt, w_, b_, rows=rows),
            [dec_p, proj, bank, w, b], grad_rtol=1e-3, grad_atol=1e-4, name="ptr_attn_gate ragged")
    # direct (arena-style) gate gradients through the deferred finalizer: accumulate onto what is already there
    wl, bl = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    wl.grad, bl.grad = torch.full_like(wl, 0.5), torch.full_like(bl, -0.25)
    wl._svpc_direct = bl._svpc_direct = True
    pi, g = O.ptr_attn_gate(dec.detach(), proj.detach(), bank.detach(), step_ne, lt, wl, bl)
    gw = torch.randn(g.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    (g * gw).sum().backward()
    O.join_side()
    wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    _, gr = E.ptr_attn_gate(dec.detach(), proj.detach(), bank.detach(), step_ne, lt, wr, br)
    (gr * gw).sum().backward()
    assert float((wl.grad - 0.5 - wr.grad).abs().max()) <= 1e-4 + 1e-3 * float(wr.grad.abs().max())
    assert float((bl.grad + 0.25 - br.grad).abs().max()) <= 1e-4 + 1e-3 * float(br.grad.abs().max())


def test_ptr_attn_and_mix_loss_and_gumbel():
    T, lt, em, D, V = 5, 6, 4, 64, 50
    step_ne = Idx([3, 3, 4, 2, 2])
    dec, proj, bank = rnd(T * lt, D, seed=1, scale=0.3), rnd(T, em, D, seed=2, scale=0.3), rnd(T, em, D, seed=3)
    compare(lambda d, p, b: O.ptr_attn(d, p, b, step_ne, lt), lambda d, p, b: E.ptr_attn(d, p, b, step_ne, lt), [dec, proj, bank],
            name="ptr_attn")
    R = T * lt
    logits = rnd(R, V, seed=4)
    g = torch.sigmoid(rnd(R, 1, seed=5)).detach().requires_grad_(True)
    pi = torch.softmax(rnd(R, em, seed=6), -1).detach().requires_grad_(True)
    row_vid = Idx([0] * 12 + [1] * 6 + [2] * 12)
    c_list = [V + 1, V, V + 2]
    row_c = Idx([c_list[b] for b in row_vid.host])
    csr_off, csr_ent = Idx([0, 4, 6, 9]), Idx([0, 0, 1, 2, 0, 1, 0, 1, 1])
    csr_id, csr_w = Idx([7, 50, 9, 7, 11, 12, 50, 51, 13]), FIdx([.5, .5, 1, 1, 1, 1, 1, .5, .5])
    labels = torch.randint(7, V, (R,), generator=torch.Generator().manual_seed(7)).to(torch.int32)
    labels[3] = -1; labels[5] = 50; labels[20] = 51; labels[21] = V - 1
    labels = labels.to(DEV)
    c_max = V + 2

    def mk(mod):
        return lambda logits, g, pi: mod.ptr_mix_loss(logits, g, pi, labels, row_c, row_vid, csr_off, csr_ent, csr_id, csr_w, c_max, 0.1)
    compare(mk(O), mk(E), [logits, g, pi], grad_rtol=1e-3, name="ptr_mix_loss")
    # label_smoothing == 0: cross-entropy of the probabilities, a mean per video (reference model.py:869-870)
    mk0 = lambda mod: (lambda logits, g, pi: mod.ptr_mix_loss(logits, g, pi, labels, row_c, row_vid, csr_off, csr_ent, csr_id, csr_w, c_max, 0.0))
    compare(mk0(O), mk0(E), [logits, g, pi], grad_rtol=1e-3, name="ptr_mix_loss (cross-entropy branch)")
    # MODEL_TYPE=v: plain softmax + loss
    empty = Idx([])
    mkv = lambda mod: (lambda logits: mod.ptr_mix_loss(logits, None, None, labels.clamp(max=V - 1), Idx([V] * R), row_vid, Idx([0, 0, 0, 0]),
                                                       empty, empty, FIdx([]), V, 0.1))
    compare(mkv(O), mkv(E), [logits], grad_rtol=1e-3, name="softmax loss")
    # gumbel straight-through bag of words
    with torch.no_grad():
        P, _ = E.ptr_mix_loss(logits.detach(), g.detach(), pi.detach(), labels, row_c, row_vid, csr_off, csr_ent, csr_id, csr_w, c_max, 0.1)
    P = P.detach().requires_grad_(True)
    emb = rnd(V, 20, seed=8)
    noise = -torch.empty(R, c_max).exponential_(generator=torch.Generator().manual_seed(9)).log().to(DEV)
    compare(lambda P, emb: O.gumbel_bow(P, row_c, emb, 0.5, noise=noise), lambda P, emb: E.gumbel_bow(P, row_c, emb, 0.5, noise=noise),
            [P, emb], grad_rtol=2e-3, grad_atol=1e-4, name="gumbel")
    # device-generated noise is Gumbel(0,1): mean ≈ 0.5772, var ≈ 1.645
    rng = O.make_rng(DEV)
    n = torch.empty(200000, device=DEV)
    from svpc_amd import _lib
    _lib.call("gumbel_noise", n.data_ptr(), n.numel(), 1, rng.seed.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert abs(n.mean().item() - 0.5772) < 0.02 and abs(n.var().item() - 1.6449) < 0.05


# ------------------------------------------------------------------------------------------------ bf16 MFMA GEMM
@pytest.fixture
def bf16_mode():
    O.set_precision("bf16")
    yield
    O.set_precision("fp32")


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 5, 3), (64, 64, 32), (100, 33, 42), (130, 257, 300), (192, 768, 768),
                                   (2048, 768, 3072), (768, 768, 19200), (768, 3072, 4096), (33, 951, 128)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_bf16_all_layouts_exact_on_bf16_rounded_operands(bf16_mode, M, N, K, a_kc, b_kc):
    """Operands that are exactly representable in bf16 make the product independent of the rounding step: the kernel must
    then agree with an fp32 matmul up to accumulation order.  Exercises both LDS images (ds_read_b128 rows and the
    ds_read_b64_tr_b16 transposed read) for both operands, split-K, ragged edges and unaligned leading dimensions."""
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    A = torch.randn(M, K, generator=g).bfloat16().float().to(DEV)
    B = torch.randn(N, K, generator=g).bfloat16().float().to(DEV)
    ref = A.double() @ B.double().t()
    Am = A if a_kc else A.t().contiguous()          # [K][M] when the reduction index is strided
    Bm = B if b_kc else B.t().contiguous()
    C = torch.empty(M, N, device=DEV)
    O._gemm(Am, Am.stride(0), a_kc, Bm, Bm.stride(0), b_kc, C, M, N, K)
    err = (C.double() - ref).abs().max().item()
    assert err <= 2e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item()), err


def test_linear_bf16_forward_backward_close_to_fp32(bf16_mode):
    x, w, b = rnd(300, 768, seed=1), rnd(768, 768, seed=2, scale=1 / math.sqrt(768)), rnd(768, seed=3)
    compare(lambda x, w, b: O.linear(x, w, b, act=E.ACT_GELU), lambda x, w, b: E.linear(x, w, b, act=E.ACT_GELU), [x, w, b],
            rtol=1e-2, atol=1e-2, grad_rtol=2e-2, grad_atol=2e-2, name="linear bf16")


# ------------------------------------------------------------------------------------------------ MFMA attention
def _bf16_attn_compare(seq, D, H, causal, with_mask, drop, cross=False, seed=0):
    Rq, Rk = seq.n_q_rows, seq.n_k_rows
    km = None
    if with_mask:
        km = (torch.rand(Rk, generator=torch.Generator().manual_seed(seed)) > 0.3).float().to(DEV)
        for o in seq.h_k_off:
            km[o] = 1.0
    tol = dict(rtol=2e-2, atol=2e-2, grad_rtol=3e-2, grad_atol=3e-2)
    if cross:
        qt, kvt = rnd(Rq, D, seed=seed + 1), rnd(Rk, 2 * D, seed=seed + 2)
        cols = (0, 0, D)
        compare(lambda qt, kvt: O.attention(qt, kvt, cols, D, H, seq, km, causal, drop),
                lambda qt, kvt: E.attention(qt, kvt, cols, D, H, seq, km, causal, drop), [qt, kvt], name="mfma attn cross", **tol)
    else:
        qkv = rnd(Rq, 3 * D, seed=seed + 3)
        cols = (0, D, 2 * D)
        compare(lambda qkv: O.attention(qkv, qkv, cols, D, H, seq, km, causal, drop),
                lambda qkv: E.attention(qkv, qkv, cols, D, H, seq, km, causal, drop), [qkv], name="mfma attn self", **tol)


def test_mfma_attention_encoder_decoder_cross_ragged(bf16_mode):
    _bf16_attn_compare(SeqInfo.uniform(6, 100, 100, DEV), 768, 12, False, True, None)
    _bf16_attn_compare(SeqInfo.uniform(7, 22, 22, DEV), 768, 12, True, True, None, seed=3)
    _bf16_attn_compare(SeqInfo.uniform(9, 22, 3, DEV), 768, 12, False, False, None, cross=True, seed=4)
    _bf16_attn_compare(SeqInfo.uniform(3, 128, 128, DEV), 128, 4, False, True, None, seed=5)     # dh = 32, full tiles
    lens, off = [3, 1, 12, 7], [0, 3, 4, 16]
    _bf16_attn_compare(SeqInfo(off, lens, off, lens, DEV), 128, 2, False, False, None, seed=6)
    _bf16_attn_compare(SeqInfo.uniform(2, 33, 65, DEV), 64, 1, False, True, None, cross=True, seed=7)


def test_mfma_attention_dropout(bf16_mode):
    rng = O.make_rng(DEV, seed=3)
    _bf16_attn_compare(SeqInfo.uniform(3, 40, 40, DEV), 128, 2, True, True, (0.1, rng, 4), seed=8)


@pytest.mark.parametrize("M,N,K,a_kc,b_kc,dts", [
    (256, 768, 768, 1, 1, ("bf16", "f32", "bf16")), (19200, 768, 768, 1, 1, ("bf16", "f32", "bf16")),
    (256, 768, 2304, 1, 0, ("bf16", "f32", "bf16")), (768, 2304, 19200, 0, 0, ("bf16", "bf16", "f32")),
    (128, 128, 4096, 0, 0, ("bf16", "bf16", "f32"))])
def test_gemm_mx_bf16_storage(bf16_mode, M, N, K, a_kc, b_kc, dts):
    """bf16 activation-stream variants: bf16 operands are consumed as they are (no rounding step), fp32 ones rounded to bf16."""
    T = {"bf16": torch.bfloat16, "f32": torch.float32}
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16().float().to(DEV)
    B = torch.randn(N, K, generator=g).bfloat16().float().to(DEV)
    ref = A.double() @ B.double().t()
    Am = (A if a_kc else A.t().contiguous()).to(T[dts[0]])
    Bm = (B if b_kc else B.t().contiguous()).to(T[dts[1]])
    C = torch.empty(M, N, device=DEV, dtype=T[dts[2]])
    O._gemm(Am, Am.stride(0), a_kc, Bm, Bm.stride(0), b_kc, C, M, N, K)
    tol = (1e-2 if dts[2] == "bf16" else 2e-6 * math.sqrt(K)) * max(1.0, ref.abs().max().item())
    assert (C.double() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("M,N,K,a_kc,b_kc,c_bf16", [
    (128, 128, 32, 1, 1, False), (256, 768, 768, 1, 1, True), (19200, 768, 768, 1, 1, True), (256, 2304, 768, 1, 0, True),
    (768, 2304, 19200, 0, 0, False), (128, 256, 4096, 0, 0, False), (256, 128, 1024, 0, 1, False), (384, 3072, 768, 1, 1, False),
    # ragged shapes: clamped M / N edges, K tail of the wgrad layout served from zeros
    (200, 96, 64, 1, 1, True), (19210, 768, 768, 1, 1, True), (100, 768, 2304, 1, 0, True), (1, 128, 32, 1, 1, True),
    (768, 256, 1000, 0, 0, False), (136, 72, 50, 0, 0, False), (768, 768, 4230, 0, 0, False), (8, 8, 7, 0, 0, False),
    (250, 130, 96, 1, 1, False), (72, 200, 160, 0, 1, False)])
def test_gemm_glds_direct_to_lds(bf16_mode, M, N, K, a_kc, b_kc, c_bf16):
    """Direct-to-LDS bf16×bf16 GEMM (global_load_lds ring, source-side chunk swizzle, row and transposed fragments):
    exact up to accumulation order for fp32 output, bf16 rounding for bf16 output; all four layouts, split-K, ragged edges."""
    assert O._lib.load().svpc_gemm_glds_supported(a_kc, b_kc, K if a_kc else M, K if b_kc else N, M, N, K) == 1
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    B = torch.randn(N, K, generator=g).bfloat16().to(DEV)
    ref = A.double() @ B.double().t()
    Am = A if a_kc else A.t().contiguous()
    Bm = B if b_kc else B.t().contiguous()
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16 if c_bf16 else torch.float32)
    O._gemm(Am, Am.stride(0), a_kc, Bm, Bm.stride(0), b_kc, C, M, N, K)
    tol = (1e-2 if c_bf16 else 2e-6 * math.sqrt(K)) * max(1.0, ref.abs().max().item())
    assert (C.double() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("M,N,K,a_kc,b_kc,acc,bias_act", [
    (192, 768, 768, 1, 1, 0, True), (200, 951, 768, 1, 1, 0, True), (4224, 768, 768, 1, 0, 0, False), (768, 768, 192, 0, 0, 1, False),
    (16, 3072, 768, 1, 1, 0, True), (3072, 768, 32, 0, 0, 1, False), (100, 60, 64, 1, 1, 0, False), (64, 64, 4224, 0, 0, 0, False),
    (768, 768, 4224, 0, 0, 1, False), (4224, 2304, 768, 1, 1, 0, True), (12, 20, 96, 0, 1, 0, False), (36, 8, 160, 1, 0, 0, False),
    (2304, 768, 4224, 0, 0, 0, False), (132, 136, 2048, 0, 0, 0, False),
    # wave-split-K form (≤ 256 tiles of 64², ≥ 4 k-tiles): uneven k-tile counts per wave, edges, every layout, long K
    (192, 768, 160, 1, 1, 0, True), (16, 768, 3072, 1, 0, 0, False), (576, 1536, 768, 1, 1, 0, True), (700, 60, 224, 0, 1, 1, False),
    (64, 64, 128, 0, 0, 0, False), (192, 2304, 768, 1, 1, 1, True),
    # k tail (K % 32 != 0, both operands k-contiguous: the 300-wide word vectors): chunks past K are zero-sourced
    (4224, 768, 300, 1, 1, 0, True), (70, 50, 300, 1, 1, 1, False), (192, 768, 44, 1, 1, 0, True), (2112, 300, 36, 1, 1, 0, False)])
def test_gemm_l32_fp32_direct_to_lds(bf16_mode, M, N, K, a_kc, b_kc, acc, bias_act):
    """fp32-operand direct-to-LDS GEMM (deep ring, bf16 rounding at fragment build): all four layouts, clamped M/N edges,
    128² / 64² tiles, split-K, accumulate and bias+activation epilogues — against fp64 on the bf16-rounded operands."""
    assert O._lib.load().svpc_gemm_l32_supported(a_kc, b_kc, K if a_kc else M, K if b_kc else N, M, N, K) == 1
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = torch.randn(N, K, generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV) if bias_act else None
    C0 = torch.randn(M, N, generator=g).to(DEV)
    ref = A.bfloat16().double() @ B.bfloat16().double().t()
    if bias_act:
        ref = torch.relu(ref + bias.double())
    if acc:
        ref = ref + C0.double()
    Am = A if a_kc else A.t().contiguous()
    Bm = B if b_kc else B.t().contiguous()
    C = C0.clone()
    ws = O._ws(C.device)
    O._lib.call("gemm_l32", A_ := Am.data_ptr(), Am.stride(0), a_kc, Bm.data_ptr(), Bm.stride(0), b_kc, C.data_ptr(), C.stride(0), None,
                M, N, K, None if bias is None else bias.data_ptr(), O.ACT_RELU if bias_act else O.ACT_NONE, 0.0, 0, None, acc,
                ws.data_ptr(), ws.numel() * 4, O._stream())
    tol = 2e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item())
    assert (C.double() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("M,N,K,b_kc,gact,has_r", [(192, 768, 768, 0, O.ACT_GELU, False), (192, 768, 768, 0, O.ACT_RELU, True),
                                                    (4224, 768, 768, 0, O.ACT_GELU, True), (200, 96, 64, 1, O.ACT_SIGMOID, False)])
def test_gemm_l32_rg_activation_backward_in_the_dgrad_epilogue(bf16_mode, M, N, K, b_kc, gact, has_r):
    """svpc_gemm_l32_rg: C = (A·B) ⊙ act'(G) + R in fp32 storage (the dgrad of the projection that consumes an activated tensor of the
    step-wise encoder) against fp64 on the bf16-rounded operands"""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(DEV)
    G = torch.randn(M, N, generator=g).to(DEV)
    if gact == O.ACT_SIGMOID:
        G = torch.sigmoid(G)
    R = torch.randn(M, N, generator=g).to(DEV) if has_r else None
    Bm = B.t().contiguous() if b_kc else B
    C = torch.empty(M, N, device=DEV)
    ws = O._ws(C.device)
    O._lib.call("gemm_l32_rg", A.data_ptr(), K, 1, Bm.data_ptr(), Bm.stride(0), b_kc, C.data_ptr(), N, R.data_ptr() if has_r else None,
                G.data_ptr(), gact, M, N, K, 0, ws.data_ptr(), ws.numel() * 4, O._stream())
    ref = A.bfloat16().double() @ B.bfloat16().double()
    gd = G.double()
    if gact == O.ACT_GELU:
        ref = ref * (0.5 * (1.0 + torch.erf(gd / math.sqrt(2.0))) + gd * torch.exp(-0.5 * gd * gd) / math.sqrt(2.0 * math.pi))
    elif gact == O.ACT_RELU:
        ref = ref * (gd > 0).double()
    else:
        ref = ref * gd * (1.0 - gd)
    if has_r:
        ref = ref + R.double()
    assert (C.double() - ref).abs().max().item() <= 2e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item())


def test_gemm_l32_unsupported_shapes_fall_back(bf16_mode):
    """a k tail with a k-strided operand, K < 32, K % 4 != 0, or a k-strided operand whose row count is not a multiple of 4 stay on
    the register-staged kernel; a k tail with both operands k-contiguous runs on the direct-to-LDS kernel."""
    lib = O._lib.load()
    assert lib.svpc_gemm_l32_supported(1, 1, 300, 300, 64, 64, 300) == 1
    assert lib.svpc_gemm_l32_supported(1, 0, 300, 64, 64, 64, 300) == 0
    assert lib.svpc_gemm_l32_supported(1, 1, 28, 28, 64, 64, 28) == 0
    assert lib.svpc_gemm_l32_supported(1, 1, 304, 304, 64, 64, 302) == 0
    assert lib.svpc_gemm_l32_supported(0, 1, 951, 768, 951, 768, 4224) == 0
    assert lib.svpc_gemm_l32_supported(1, 0, 960, 768, 64, 768, 951) == 0
    g = torch.Generator().manual_seed(0)
    A = torch.randn(70, 300, generator=g).to(DEV)
    B = torch.randn(50, 300, generator=g).to(DEV)
    C = torch.empty(70, 50, device=DEV)
    O._gemm(A, 300, 1, B, 300, 1, C, 70, 50, 300)
    ref = A.bfloat16().double() @ B.bfloat16().double().t()
    assert (C.double() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def test_lstm_sequence_matches_stepwise_statement():
    """Fused recurrence node (one direction, ragged lengths 3/1/2 of N=3 videos) against the per-step torch statement:
    outputs, gradient of the input projections and of the recurrent weight."""
    torch.manual_seed(0)
    N, D, lens = 3, 32, [3, 1, 2]
    S, T = max(lens), sum(lens)
    off = [0, 3, 4]
    rows_t = [torch.tensor([off[b] + min(t, lens[b] - 1) for b in range(N)], dtype=torch.int32, device=DEV) for t in range(S)]
    act_t = [torch.tensor([1.0 if t < lens[b] else 0.0 for b in range(N)], device=DEV) for t in range(S)]
    pick = torch.tensor([s_ * N + b for b in range(N) for s_ in range(lens[b])], dtype=torch.int32, device=DEV)
    gx = torch.randn(T, 4 * D, device=DEV, requires_grad=True)
    w = (0.3 * torch.randn(4 * D, D, device=DEV)).requires_grad_(True)
    wt = torch.randn(T, D, device=DEV)
    res = []
    for mod in (O, E):
        gx.grad = w.grad = None
        out = mod.lstm_sequence(gx, w, rows_t, act_t, pick)
        (out * wt).sum().backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), gx.grad.clone(), w.grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5), (a - b).abs().max()


@pytest.mark.parametrize("H,dh,k_lens", [(12, 64, [1, 22, 7, 3]), (4, 32, [100, 33, 64, 1]), (2, 16, [5, 5])])
def test_single_query_attention_for_decoding(H, dh, k_lens):
    """q_len == 1 per sequence under no_grad takes the wave-per-(sequence, head) kernel; against the torch statement, with a key mask"""
    D = H * dh
    n = len(k_lens)
    k_off = [sum(k_lens[:i]) for i in range(n)]
    seq = SeqInfo(list(range(n)), [1] * n, k_off, k_lens, DEV)
    Rk = sum(k_lens)
    q, kv = rnd(n, D, seed=1, grad=False), rnd(Rk, 2 * D, seed=2, grad=False)
    km = (torch.rand(Rk, generator=torch.Generator().manual_seed(3)) > 0.3).float().to(DEV)
    for o in k_off:
        km[o] = 1.0
    for mask in (None, km):
        with torch.no_grad():
            got = O.attention(q, kv, (0, 0, D), D, H, seq, key_mask=mask, causal=False)
            ref = E.attention(q, kv, (0, 0, D), D, H, seq, key_mask=mask, causal=False)
        assert torch.allclose(got, ref, rtol=2e-5, atol=2e-6), (got - ref).abs().max()


@pytest.mark.parametrize("M,N,K,a_kc,b_kc", [(19200, 768, 768, 1, 1), (19200, 768, 2304, 1, 0), (18000, 760, 768, 1, 1), (15608, 1000, 96, 1, 0),
                                              (13000, 1024, 64, 1, 1)])
def test_glds_pingpong_tiles(bf16_mode, M, N, K, a_kc, b_kc):
    """stream-sized bf16 GEMMs take the 256×256 ping-pong kernel (≥200 such tiles): forward (NT) and dgrad (NN) layouts, ragged M and N
    edges, K shallower than the prefetch ring; plain, bias + GELU + pre-activation copy (16-byte row stores), and the element-wise
    epilogue (accumulate) — against fp64 of the same bf16 operands"""
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    A = torch.randn((M, K) if a_kc else (K, M), generator=g).to(bf).to(DEV)
    B = (0.05 * torch.randn((N, K) if b_kc else (K, N), generator=g)).to(bf).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    ref = (A.double() if a_kc else A.double().t()) @ (B.double().t() if b_kc else B.double())
    scale = max(1.0, ref.abs().max().item())
    C = torch.empty(M, N, device=DEV, dtype=bf)
    O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K)
    assert (C.double() - ref).abs().max().item() <= 6e-3 * scale
    Z = torch.empty(M, N, device=DEV, dtype=bf)
    O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K, bias=bias, act=O.ACT_GELU, Z=Z)
    zr = ref + bias.double()
    zscale = max(scale, zr.abs().max().item())                     # bf16 storage: half an ulp of the largest stored value
    assert (Z.double() - zr).abs().max().item() <= 6e-3 * zscale
    assert (C.double() - torch.nn.functional.gelu(zr)).abs().max().item() <= 6e-3 * zscale
    C0 = torch.randn(M, N, generator=g).to(bf).to(DEV)
    C = C0.clone()
    O._gemm(A, A.stride(0), a_kc, B, B.stride(0), b_kc, C, M, N, K, accumulate=True)
    assert (C.double() - (C0.double() + ref)).abs().max().item() <= 1.2e-2 * max(scale, C0.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(19200, 768, 768), (4224, 768, 2304), (18000, 760, 768)])
def test_glds_addend_epilogue(bf16_mode, M, N, K):
    """C = A·B + R in the 16-byte row-store epilogues (ping-pong tiles through the LDS image at M = 19,200 / 18,000, direct row
    pieces of the 128² tiles at M = 4,224): the dgrad that absorbs a residual-path gradient — against fp64"""
    g = torch.Generator().manual_seed(M + N)
    bf = torch.bfloat16
    A = torch.randn(M, K, generator=g).to(bf).to(DEV)
    B = (0.05 * torch.randn(K, N, generator=g)).to(bf).to(DEV)
    R = torch.randn(M, N, generator=g).to(bf).to(DEV)
    C = torch.empty(M, N, device=DEV, dtype=bf)
    O._gemm(A, K, 1, B, N, 0, C, M, N, K, R=R)
    ref = A.double() @ B.double() + R.double()
    assert (C.double() - ref).abs().max().item() <= 6e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K,act", [(19200, 768, 768, 2), (4224, 768, 768, 2), (18000, 760, 768, 2), (4224, 768, 768, 1),
                                        (640, 256, 512, 3)])
def test_glds_activation_backward_epilogue(bf16_mode, M, N, K, act):
    """C = (A·B) ⊙ act'(G) in the row-store epilogues (svpc_gemm_glds_rg): the dgrad of the projection that follows an activation writes
    the pre-activation gradient itself.  Against fp64 with the exact derivative (GELU: Φ(z) + z·φ(z) from the pre-activation; ReLU /
    sigmoid: from the output), on every tile form: ping-pong + LDS image, 128² row pieces, ragged, element-wise."""
    g = torch.Generator().manual_seed(M + N + act)
    bf = torch.bfloat16
    A = torch.randn(M, K, generator=g).to(bf).to(DEV)
    B = (0.05 * torch.randn(K, N, generator=g)).to(bf).to(DEV)
    aux = (1.5 * torch.randn(M, N, generator=g))
    if act == 1:
        aux = aux.clamp_min(0.0)
    elif act == 3:
        aux = torch.sigmoid(aux)
    aux = aux.to(bf).to(DEV)
    C = torch.empty(M, N, device=DEV, dtype=bf)
    assert O._gemm(A, K, 1, B, N, 0, C, M, N, K, G=(aux, act)) is True
    a = aux.double()
    if act == 2:
        d = 0.5 * (1 + torch.erf(a / 2 ** 0.5)) + a * torch.exp(-0.5 * a * a) / (2 * torch.pi) ** 0.5
    elif act == 1:
        d = (a > 0).double()
    else:
        d = a * (1 - a)
    ref = (A.double() @ B.double()) * d
    # the product is rounded to bf16 before the factor is applied (as the separate pass did) and once after: 2 × 2^-9
    assert (C.double() - ref).abs().max().item() <= 9e-3 * max(1.0, ref.abs().max().item())
    # with an addend on top: C = (A·B) ⊙ act'(G) + R
    R = torch.randn(M, N, generator=g).to(bf).to(DEV)
    C2 = torch.empty_like(C)
    assert O._gemm(A, K, 1, B, N, 0, C2, M, N, K, R=R, G=(aux, act)) is True
    ref2 = ref + R.double()
    assert (C2.double() - ref2).abs().max().item() <= 9e-3 * max(1.0, ref2.abs().max().item())


def test_fused_activation_backward_through_two_linears(bf16_mode):
    """linear(act, fuse_act_bwd=True) → linear: the second projection's dgrad applies the activation backward (no act_bwd launch);
    gradients equal the unfused path within bf16 rounding, and a second consumer of the activated tensor fails loudly."""
    torch.manual_seed(1)
    rows, D, F = 768, 256, 384
    x0 = (0.5 * torch.randn(rows, D, device=DEV)).bfloat16()
    w1 = (0.1 * torch.randn(F, D, device=DEV)).requires_grad_(True)
    b1 = (0.1 * torch.randn(F, device=DEV)).requires_grad_(True)
    w2 = (0.1 * torch.randn(D, F, device=DEV)).requires_grad_(True)
    b2 = torch.zeros(D, device=DEV, requires_grad=True)
    wt = torch.randn(rows, D, device=DEV)
    from svpc_amd import _lib
    res = []
    keep = O.FUSE_ACT_BWD
    calls = []
    orig_call = _lib.call
    def spy(name, *a):
        calls.append(name)
        return orig_call(name, *a)
    try:
        for use in (True, False):
            O.FUSE_ACT_BWD = use
            x = x0.clone().requires_grad_(True)
            for t in (w1, b1, w2, b2):
                t.grad = None
            del calls[:]
            _lib.call = spy
            try:
                h = O.linear(x, w1, b1, act=2, fuse_act_bwd=True)
                y = O.linear(h, w2, b2)
                (y.float() * wt).sum().backward()
                O.join_side()
            finally:
                _lib.call = orig_call
            torch.cuda.synchronize()
            assert ("act_bwd_t" in calls) == (not use), calls
            res.append([t.grad.float().clone() for t in (x, w1, b1, w2, b2)])
        for a, b in zip(*res):
            assert torch.allclose(a, b, rtol=2e-2, atol=2e-2 * b.abs().max().item())
        O.FUSE_ACT_BWD = True
        x = x0.clone().requires_grad_(True)
        h = O.linear(x, w1, b1, act=2, fuse_act_bwd=True)
        y = O.linear(h, w2, b2)
        with pytest.raises(Exception, match="second consumer"):
            ((y.float() * wt).sum() + h.float().sum()).backward()
    finally:
        O.FUSE_ACT_BWD = keep
        O._RES_SINK.clear()
        try:
            O.join_side()
        except Exception:
            pass


def test_residual_gradient_hand_over(bf16_mode):
    """layernorm(sub(h) + h, sink=True): the LayerNorm parks its residual-path gradient and the projection consuming h adds it in its
    dgrad epilogue — same gradients as autograd's separate add (within bf16 rounding); a parked gradient nobody absorbs fails loudly"""
    torch.manual_seed(0)
    rows, D = 512, 128
    h0 = (0.5 * torch.randn(rows, D, device=DEV)).bfloat16()
    w = (0.1 * torch.randn(D, D, device=DEV)).requires_grad_(True)
    b = torch.zeros(D, device=DEV, requires_grad=True)
    gamma, beta = torch.ones(D, device=DEV, requires_grad=True), torch.zeros(D, device=DEV, requires_grad=True)
    wt = torch.randn(rows, D, device=DEV)
    grads = []
    keep = O.USE_RES_SINK
    try:
        for use in (True, False):
            O.USE_RES_SINK = use
            h = h0.clone().requires_grad_(True)
            for t in (w, b, gamma, beta):
                t.grad = None
            before = O.SINK_STATS[1]
            y = O.layernorm(O.linear(h, w, b), gamma, beta, 1e-12, residual=h, sink=True)
            (y.float() * wt).sum().backward()
            torch.cuda.synchronize()
            assert (O.SINK_STATS[1] - before) == (1 if use else 0)
            grads.append((h.grad.float().clone(), w.grad.clone()))
        assert torch.allclose(grads[0][0], grads[1][0], rtol=2e-2, atol=2e-2 * grads[1][0].abs().max().item())
        assert torch.allclose(grads[0][1], grads[1][1], rtol=2e-2, atol=2e-2 * grads[1][1].abs().max().item())
        # no projection consumes h: the parked gradient is reported at the end of backward, not dropped
        O.USE_RES_SINK = True
        h = h0.clone().requires_grad_(True)
        y = O.layernorm(h * 2.0, gamma, beta, 1e-12, residual=h, sink=True)
        with pytest.raises(Exception, match="hand-over"):
            (y.float() * wt).sum().backward()
    finally:
        O.USE_RES_SINK = keep
        O._RES_SINK.clear()


def test_grouped_weight_gradients(bf16_mode):
    """svpc_gemm_group_wgrad: several independent dW += dzᵀ·x (+ db += Σ dz) problems of different shapes in one launch, against
    fp64 on the bf16-rounded operands (weights) and an fp32 column sum (bias); accumulation into non-zero targets."""
    import ctypes
    shapes = [(192, 768, 768, True), (576, 1536, 768, True), (192, 64, 300, False), (4224, 768, 768, True), (32, 8, 4, True),
              (192, 3072, 768, True), (96, 100, 60, False),
              # ragged row counts (T = Σ S_b steps is arbitrary): the partial last k-tile is fetched from a block of zeros
              (156, 768, 768, True), (7, 64, 300, True), (468, 1536, 768, False), (3432, 768, 768, True), (33, 2304, 768, True)]
    g = torch.Generator().manual_seed(5)
    keep, probs = [], (O._WgradProblem * len(shapes))()
    for i, (rows, n_out, n_in, with_b) in enumerate(shapes):
        dz = torch.randn(rows, n_out, generator=g).to(DEV)
        x = torch.randn(rows, n_in, generator=g).to(DEV)
        dw0 = torch.randn(n_out, n_in, generator=g).to(DEV)
        db0 = torch.randn(n_out, generator=g).to(DEV) if with_b else None
        dw, db = dw0.clone(), (db0.clone() if with_b else None)
        keep.append((dz, x, dw0, db0, dw, db))
        probs[i] = O._WgradProblem(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if with_b else None, n_out, n_in, rows,
                                   dz.stride(0), x.stride(0), dw.stride(0))
    O._lib.call("gemm_group_wgrad", ctypes.addressof(probs), len(shapes), O._stream())
    torch.cuda.synchronize()
    for (rows, n_out, n_in, with_b), (dz, x, dw0, db0, dw, db) in zip(shapes, keep):
        ref = dw0.double() + dz.bfloat16().double().t() @ x.bfloat16().double()
        tol = 2e-6 * math.sqrt(rows) * max(1.0, ref.abs().max().item())
        assert (dw.double() - ref).abs().max().item() <= tol, (rows, n_out, n_in)
        if with_b:
            rb = db0.double() + dz.double().sum(0)
            assert (db.double() - rb).abs().max().item() <= 1e-5 * max(1.0, rb.abs().max().item()), (rows, n_out)


@pytest.mark.parametrize("N,D,fused", [(3, 32, True), (3, 32, False), (35, 64, True), (16, 768, True)])
def test_bilstm_lockstep_matches_two_direction_nodes(bf16_mode, N, D, fused):
    """Fused two-direction recurrence against the per-direction torch statement: outputs and gradients of both input projections and
    both recurrent weights (bf16 MFMA operands → 2e-2 tolerance), ragged lengths.  fused: recurrent projection + cell in one launch
    per time step (svpc_lstm_pair_step_fwd; 35 videos = two row tiles, 768 = the headline width) — else grouped GEMM + cell launches."""
    torch.manual_seed(0)
    lens = [3, 1, 2] if N == 3 else [1 + (7 * b + 3) % 4 for b in range(N)]
    S, T = max(lens), sum(lens)
    off = [sum(lens[:b]) for b in range(N)]
    keep_fused = O.LSTM_FUSED_STEP
    O.LSTM_FUSED_STEP = fused
    mk = lambda f: [torch.tensor([f(b, t) for b in range(N)], dtype=torch.int32, device=DEV) for t in range(S)]
    rows_f = mk(lambda b, t: off[b] + min(t, lens[b] - 1))
    rows_b = mk(lambda b, t: off[b] + max(lens[b] - 1 - t, 0))
    act_t = [torch.tensor([1.0 if t < lens[b] else 0.0 for b in range(N)], device=DEV) for t in range(S)]
    pick_f = torch.tensor([s_ * N + b for b in range(N) for s_ in range(lens[b])], dtype=torch.int32, device=DEV)
    pick_b = torch.tensor([(lens[b] - 1 - s_) * N + b for b in range(N) for s_ in range(lens[b])], dtype=torch.int32, device=DEV)
    leaves = [torch.randn(T, 4 * D, device=DEV, requires_grad=True), torch.randn(T, 4 * D, device=DEV, requires_grad=True),
              ((0.3 if D <= 64 else 0.04) * torch.randn(4 * D, D, device=DEV)).requires_grad_(True),
              ((0.3 if D <= 64 else 0.04) * torch.randn(4 * D, D, device=DEV)).requires_grad_(True)]
    wt_f, wt_b = torch.randn(T, D, device=DEV), torch.randn(T, D, device=DEV)
    res = []
    for mod in (O, E):
        for l in leaves:
            l.grad = None
        of, ob = mod.bilstm_sequences(leaves[0], leaves[1], leaves[2], leaves[3], rows_f, rows_b, act_t, pick_f, pick_b)
        ((of * wt_f).sum() + (ob * wt_b).sum()).backward()
        torch.cuda.synchronize()
        res.append([of.detach().clone(), ob.detach().clone()] + [l.grad.clone() for l in leaves])
    # summed=True (what the model uses: the two directions picked and added in one launch, one scatter / gather launch each way back)
    summed = []
    for mod in (O, E):
        for l in leaves:
            l.grad = None
        o_ = mod.bilstm_sequences(leaves[0], leaves[1], leaves[2], leaves[3], rows_f, rows_b, act_t, pick_f, pick_b, summed=True)
        (o_ * wt_f).sum().backward()
        torch.cuda.synchronize()
        summed.append([o_.detach().clone()] + [l.grad.clone() for l in leaves])
    O.LSTM_FUSED_STEP = keep_fused
    for a, b in list(zip(res[0], res[1])) + list(zip(summed[0], summed[1])):
        assert (a - b).abs().max().item() <= 2e-2 * max(1.0, b.abs().max().item()), (a - b).abs().max()
    assert (summed[0][0] - (res[0][0] + res[0][1])).abs().max().item() == 0.0


def test_grouped_weight_gradients_bf16(bf16_mode):
    """svpc_gemm_group_wgrad_bf16: bf16-stream problems (ragged row counts, both tile edges) in one launch, whole k-loop per tile,
    accumulating into non-zero fp32 targets — against fp64."""
    import ctypes
    _grouped_wgrad_bf16_case([(19200, 768, 768), (4224, 2304, 768), (1000, 768, 3072), (100, 136, 72), (4230, 768, 768), (7, 8, 8)])


def test_grouped_weight_gradients_bf16_pingpong_tiles(bf16_mode):
    """every problem at least 256 wide both ways → the 256×256 ping-pong tiles (deepest problems dealt first); ragged row counts
    (zero-sourced k tail), widths that are not multiples of 256, one problem shallower than the 3-tile prefetch"""
    _grouped_wgrad_bf16_case([(4230, 768, 768), (19200, 768, 768), (1000, 264, 776), (4224, 2304, 768), (40, 256, 256), (1537, 768, 3072)])


def test_grouped_weight_gradients_bf16_balanced_tail(bf16_mode):
    """more 256×256 tiles than CUs with deep reductions: the tiles dealt after the first round are cut into k-parts (slabs in the
    workspace) and added in part order by the fix-up launch; ragged rows (zero-sourced k tail inside the last part) and tile edges"""
    _grouped_wgrad_bf16_case([(10270, 2296, 3064), (10272, 2304, 3072), (10272, 2304, 3072), (4224, 768, 768)])


def _grouped_wgrad_bf16_case(shapes):
    import ctypes
    g = torch.Generator().manual_seed(9)
    keep, probs = [], (O._WgradProblem * len(shapes))()
    for i, (rows, n_out, n_in) in enumerate(shapes):
        dz = torch.randn(rows, n_out, generator=g).bfloat16().to(DEV)
        x = torch.randn(rows, n_in, generator=g).bfloat16().to(DEV)
        dw0 = torch.randn(n_out, n_in, generator=g).to(DEV)
        dw = dw0.clone()
        keep.append((dz, x, dw0, dw))
        probs[i] = O._WgradProblem(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), None, n_out, n_in, rows, dz.stride(0), x.stride(0), dw.stride(0))
    ws = O._ws(torch.device(DEV))
    O._lib.call("gemm_group_wgrad_bf16_ws", ctypes.addressof(probs), len(shapes), ws.data_ptr(), ws.numel() * 4, O._stream())
    torch.cuda.synchronize()
    for (rows, n_out, n_in), (dz, x, dw0, dw) in zip(shapes, keep):
        ref = dw0.double() + dz.double().t() @ x.double()
        tol = 2e-6 * math.sqrt(rows) * max(1.0, ref.abs().max().item())
        assert (dw.double() - ref).abs().max().item() <= tol, (rows, n_out, n_in)


@pytest.mark.parametrize("M,N,K,act,with_z", [(256, 256, 64, 0, False), (300, 264, 128, 2, True), (1000, 520, 192, 1, False),
                                              (513, 8, 64, 2, True), (19200, 768, 768, 2, True), (18000, 2304, 768, 0, False),
                                              (4224, 768, 3072, 1, False)])
def test_gemm_p8_forward_form(M, N, K, act, with_z):
    """svpc_gemm_p8 (256×256×64 tiles, 8 phases per pair of k-tiles, 16x16x32 MFMA, both operands direct-to-LDS) through its C-ABI
    entry: one to 48 k-tiles (prologue-only, odd and even tile counts → both LDS buffers end the loop), ragged M and N edges (clamped
    rows, guarded 16-byte stores), bias, ReLU / GELU (the kernel's own erf approximation: |error| ≤ 3e-7, below the bf16 rounding of
    the output) and the exact pre-activation copy — against fp64 of the same bf16 operands."""
    from svpc_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    A = torch.randn(M, K, generator=g).to(bf).to(DEV)
    B = (0.05 * torch.randn(N, K, generator=g)).to(bf).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    C = torch.full((M + 2, N), 7.0, device=DEV, dtype=bf)          # two guard rows: nothing may be written past row M-1
    Z = torch.full((M + 2, N), 7.0, device=DEV, dtype=bf) if with_z else None
    _lib.call("gemm_p8", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, Z.data_ptr() if with_z else None, M, N, K, bias.data_ptr(), act,
              torch.cuda.current_stream().cuda_stream)
    zr = A.double() @ B.double().t() + bias.double()
    ref = torch.relu(zr) if act == 1 else torch.nn.functional.gelu(zr) if act == 2 else zr
    scale = max(1.0, zr.abs().max().item())
    assert (C[:M].double() - ref).abs().max().item() <= 6e-3 * scale
    assert (C[M:] == 7.0).all()
    if with_z:
        assert (Z[:M].double() - zr).abs().max().item() <= 6e-3 * scale
        assert (Z[M:] == 7.0).all()


def test_gemm_p8_refuses_what_it_cannot_run():
    from svpc_amd import _lib
    bf = torch.bfloat16
    A = torch.zeros(256, 96, device=DEV, dtype=bf); B = torch.zeros(256, 96, device=DEV, dtype=bf); C = torch.zeros(256, 256, device=DEV, dtype=bf)
    with pytest.raises(_lib.SvpcKernelError):      # K % 64 != 0
        _lib.call("gemm_p8", A.data_ptr(), 96, B.data_ptr(), 96, C.data_ptr(), 256, None, 256, 256, 96, None, 0, torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("with_resim,with_sim", [(True, True), (False, True)])
def test_loss_tail_matches_separate_terms(with_resim, with_sim):
    """ops.loss_tail (all loss terms + their sum in one launch, backward in one) against the per-term torch statement: ragged
    entity widths, rows without any detected action (excluded from the ASL, model.py:1112-1114), probabilities at the BCE clamp."""
    g = torch.Generator().manual_seed(5)
    R, Ce, Ca, n_cap = 37, 7, 45, 531
    widths = [int(v) for v in torch.randint(1, Ce + 1, (R,), generator=g)]
    mk = lambda *sh: torch.rand(*sh, generator=g).clamp(1e-4, 1 - 1e-4)
    e_p, a_p, r_e, r_a = mk(R, Ce), mk(R, Ca), mk(R, Ce), mk(R, Ca)
    e_p[0, 0] = 0.0; e_p[1, 0] = 1.0                       # log clamped at -100 (nn.BCELoss)
    align = (torch.rand(R, Ce, generator=g) < 0.3).float()
    act = (torch.rand(R, Ca, generator=g) < 0.05).float()
    act[3] = 0.0; act[11] = 0.0                            # rows without a detected action
    cap = torch.rand(n_cap, generator=g) * 5
    outs = {}
    for name, mod, dev in (("hip", O, DEV), ("ref", E, "cpu")):
        t = [x.clone().to(dev).requires_grad_(True) for x in (cap, e_p, a_p, r_e, r_a)]
        tot = mod.loss_tail(t[0], t[1], t[2], t[3] if with_resim else None, t[4] if with_resim else None, align.to(dev), act.to(dev),
                            Idx(widths), 0.5 if with_resim else 0.0)
        (tot * 1.7).backward()
        outs[name] = (tot.detach().cpu(), [x.grad.cpu() if x.grad is not None else None for x in t])
    assert torch.allclose(outs["hip"][0], outs["ref"][0], rtol=2e-6)
    for a, b in zip(outs["hip"][1], outs["ref"][1]):
        assert (a is None) == (b is None)
        if a is not None:
            ok = torch.isfinite(b)        # torch's own BCE statement differentiates log(0)·0 to NaN at the clamp; the kernel gives 0 there
            assert torch.isfinite(a).all() and torch.allclose(a[ok], b[ok], rtol=2e-5, atol=1e-6)
            assert (a[~ok] == 0).all()


def test_gather_cast_multi_token_staging():
    """dst[i] = cast(src[idx[i]]) for several (source, index, type) segments in one launch: bit-exact against index_select + to()."""
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 5000, (23424,), generator=g).to(DEV)                       # int64, as the loader hands them
    masks = (torch.rand(23424, generator=g) > 0.3).float().to(DEV)
    labels = torch.randint(-1, 5000, (23424,), generator=g).to(torch.int32).to(DEV)
    vrows = torch.randint(0, 23424, (19200,), generator=g).to(torch.int32).to(DEV)
    trows = torch.randint(0, 23424, (4224,), generator=g).to(torch.int32).to(DEV)
    ingr = torch.randint(0, 900, (16, 100), generator=g).to(DEV)
    empty = torch.zeros(0, dtype=torch.int64, device=DEV)
    items = [(ids, vrows, torch.int32), (masks, vrows, torch.float32), (ids, trows, torch.int32), (masks, trows, torch.float32),
             (labels, trows, torch.int32), (ingr, None, torch.int32), (empty, None, torch.int32), (masks, None, torch.int32),
             (ids, trows, torch.float32)]                                               # 9 items: two launches
    outs = O.gather_cast_multi(items)
    assert len(outs) == len(items)
    for (src, idx, dt), o in zip(items, outs):
        ref = (src.reshape(-1)[idx.long()] if idx is not None else src.reshape(-1)).to(dt)
        assert o.dtype == dt and o.shape == ref.shape and torch.equal(o, ref)


def test_split_cols_gathers_block_gradients_in_place():
    """ops.split_cols: column blocks of a wide projection output whose gradients meet in ONE buffer.  A consumer that honours
    ``_svpc_grad_into`` (ops.attention) writes its block in place; any other consumer's gradient is copied in; an unused block is zero —
    the wide tensor's gradient equals what torch.split would give."""
    torch.manual_seed(4)
    n_seq, lt, n_mem, H, D = 6, 5, 3, 4, 64
    wide0 = torch.randn(n_seq * n_mem, 3 * 2 * D, device=DEV)
    q0 = torch.randn(n_seq * lt, D, device=DEV)
    seq = O.SeqInfo.uniform(n_seq, lt, n_mem, torch.device(DEV))
    wt = torch.randn(n_seq * lt, D, device=DEV)
    res = []
    for mod in ("split_cols", "torch"):
        wide = wide0.clone().requires_grad_(True)
        q = q0.clone().requires_grad_(True)
        blocks = O.split_cols(wide * 1.0, 3) if mod == "split_cols" else tuple((wide * 1.0).split(2 * D, dim=1))
        if mod == "split_cols":
            assert all(getattr(b, "_svpc_grad_into", None) is not None for b in blocks)
        a0 = O.attention(q, blocks[0], (0, 0, D), D, H, seq)                  # writes its K|V gradient into the shared buffer
        extra = (blocks[2] * 0.5).sum()                                       # a plain torch consumer: gradient copied in; block 1 unused
        ((a0 * wt).sum() + extra).backward()
        torch.cuda.synchronize()
        res.append((wide.grad.clone(), q.grad.clone()))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-6)
    assert float(res[0][0][:, 2 * D:4 * D].abs().max()) == 0.0              # the unused block
