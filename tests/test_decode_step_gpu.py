"""The fused decoding-step attention block (svpc_attn_q1_ln_fwd) against a float64 torch statement of the reference's decoder-layer
math for one new position per sentence (reference: src/rtransformer/model.py:620-663 under src/translator.py:88-112)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, K, V, x, g, b, eps, H):
    """q (T, D); K, V (T, n, D); LayerNorm(x + softmax(q·K/sqrt(dh))·V) in float64"""
    T, D = q.shape
    dh = D // H
    qh = q.double().view(T, H, 1, dh)
    Kh = K.double().view(T, -1, H, dh).transpose(1, 2)
    Vh = V.double().view(T, -1, H, dh).transpose(1, 2)
    p = torch.softmax(qh @ Kh.transpose(-1, -2) / math.sqrt(dh), -1)
    y = (p @ Vh).reshape(T, D) + x.double()
    u = y.mean(-1, keepdim=True)
    s = ((y - u) ** 2).mean(-1, keepdim=True)
    return (y - u) / torch.sqrt(s + eps) * g.double() + b.double()


@pytest.mark.parametrize("pos", [0, 1, 3, 7, 8, 15, 16, 21, 23, 31])
def test_self_attention_step_appends_and_matches(pos):
    from svpc_amd import ops
    torch.manual_seed(pos)
    dev, T, D, H, lt = "cuda:0", 37, 768, 12, 32
    qkv = torch.randn(T, 3 * D, device=dev)
    cache = torch.zeros(T * lt, 2 * D, device=dev)
    cache.view(T, lt, 2 * D)[:, :pos] = torch.randn(T, pos, 2 * D, device=dev)
    before = cache.clone()
    x = torch.randn(T, D, device=dev); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    out = ops.attn_q1_ln(qkv, cache, lt, pos + 1, x, g, b, 1e-12, H, new_kv=qkv[:, D:])
    assert out is not None
    cv = cache.view(T, lt, 2 * D)
    assert torch.equal(cv[:, pos], qkv[:, D:])                       # the append, bit for bit
    assert torch.equal(cv[:, :pos], before.view(T, lt, 2 * D)[:, :pos]) and torch.equal(cv[:, pos + 1:], before.view(T, lt, 2 * D)[:, pos + 1:])
    ref = _ref(qkv[:, :D], cv[:, :pos + 1, :D], cv[:, :pos + 1, D:], x, g, b, 1e-12, H)
    assert (out.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # … and the unfused kernels (cache copy, attn_q1, layernorm) agree to rounding
    seq = ops.SeqInfo(list(range(T)), [1] * T, [s * lt for s in range(T)], [pos + 1] * T, dev)
    sa = ops.attention(qkv, cache, (0, 0, D), D, H, seq, key_mask=None, causal=False)
    un = ops.layernorm(sa, g, b, 1e-12, residual=x)
    assert (out - un).abs().max().item() <= 2e-5 * un.abs().max().item()


@pytest.mark.parametrize("n_mem", [1, 2, 3])
def test_cross_attention_step_matches(n_mem):
    from svpc_amd import ops
    torch.manual_seed(n_mem)
    dev, T, D, H, L = "cuda:0", 50, 768, 12, 6
    wide = torch.randn(T * n_mem, L * 2 * D, device=dev)
    kv = wide[:, 2 * D:4 * D]                                        # a layer's K | V columns of the stacked memory projection
    qc = torch.randn(T, D, device=dev); x = torch.randn(T, D, device=dev); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    before = wide.clone()
    out = ops.attn_q1_ln(qc, kv, n_mem, n_mem, x, g, b, 1e-12, H)
    assert out is not None and torch.equal(wide, before)
    ref = _ref(qc, kv[:, :D].reshape(T, n_mem, D), kv[:, D:].reshape(T, n_mem, D), x, g, b, 1e-12, H)
    assert (out.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_shapes_outside_the_kernel_are_declined():
    from svpc_amd import ops
    dev = "cuda:0"
    q = torch.randn(4, 96, device=dev); kv = torch.randn(8, 192, device=dev); x = torch.randn(4, 96, device=dev)
    assert ops.attn_q1_ln(q, kv, 2, 2, x, torch.ones(96, device=dev), torch.zeros(96, device=dev), 1e-12, 3) is None      # heads of 32
    q = torch.randn(4, 768, device=dev); kv = torch.randn(4 * 40, 1536, device=dev); x = torch.randn(4, 768, device=dev)
    assert ops.attn_q1_ln(q, kv, 40, 40, x, torch.ones(768, device=dev), torch.zeros(768, device=dev), 1e-12, 12) is None  # 40 key rows


def test_pointer_attention_with_the_generation_gate_in_one_launch():
    """svpc_ptr_attn_pgen_fwd (decoding iteration: pi and p_gen = sigmoid([dec ; att]·w + b)) against the separate kernels"""
    from svpc_amd import ops
    from svpc_amd.ops_common import ACT_SIGMOID, Idx
    torch.manual_seed(11)
    dev, T, E, D = "cuda:0", 45, 10, 768
    dec = torch.randn(T, D, device=dev); bank = torch.randn(T, E, D, device=dev); proj = torch.randn(T, E, D, device=dev) * 0.05
    w = torch.randn(1, 2 * D, device=dev) * 0.05; b = torch.randn(1, device=dev)
    ne = Idx([1 + (i * 7) % E for i in range(T)])
    with torch.no_grad():
        pi, g = ops.ptr_attn_pgen(dec, proj, bank, ne, w, b)
        pi0, att0 = ops.ptr_attn(dec, proj, bank, ne, 1)
        g0 = ops.linear(torch.cat([dec, att0], 1), w, b, act=ACT_SIGMOID)
    assert torch.equal(pi, pi0)
    ref = torch.sigmoid(torch.cat([dec, att0], 1).double() @ w.double().t() + b.double())
    assert (g.double() - ref).abs().max().item() <= 2e-6 and (g0.double() - ref).abs().max().item() <= 2e-6
