"""Host-side checks that need no GPU: parameter initialisation and the per-batch span cache.

reference: src/rtransformer/model.py:875-885 (init_bert_weights: N(0, initializer_range) for every nn.Linear / nn.Embedding weight,
LayerNorm γ=1 / β=0, zero Linear biases; nn.LSTM keeps torch's default U(−1/√H, 1/√H)), :872 (applied to every sub-module)."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

import emul_ops
from svpc_amd import model as M
from svpc_amd import synthetic as syn


def _check_init(model, cfg):
    sigma = cfg.initializer_range
    n_lin = n_emb = n_ln = 0
    for name, mod in model.named_modules():
        if isinstance(mod, (nn.Linear, nn.Embedding)):
            w = mod.weight.detach().float().cpu()
            n = w.numel()
            # the sample std of n normal draws has relative spread 1/sqrt(2n): 3 % for every tensor of the model proper, wider
            # only for the few tiny ones (W3: 3×D, W4: 1×W, pgen: 1×2D)
            tol = max(0.03, 4.0 / math.sqrt(2.0 * n))
            assert abs(float(w.std()) - sigma) <= tol * sigma, (name, float(w.std()))
            assert abs(float(w.mean())) <= 5.0 * sigma / math.sqrt(n) + 1e-9, (name, float(w.mean()))
            if isinstance(mod, nn.Linear):
                n_lin += 1
                if mod.bias is not None:
                    assert float(mod.bias.detach().abs().max()) == 0.0, name
            else:
                n_emb += 1
        elif isinstance(mod, M.BertLayerNorm):
            assert torch.equal(mod.weight.detach().cpu(), torch.ones(mod.weight.shape)), name
            assert float(mod.bias.detach().abs().max()) == 0.0, name
            n_ln += 1
    assert float(model.decoder_classifier.bias.detach().abs().max()) == 0.0      # nn.Parameter(zeros), model.py:726
    # nn.LSTM is neither Linear nor Embedding: torch's default init stays (U(−k, k), k = 1/sqrt(hidden))
    k = 1.0 / math.sqrt(cfg.hidden_size)
    for name, p in model.recipe_encoder.named_parameters():
        v = p.detach().float().cpu()
        assert float(v.abs().max()) <= k * (1 + 1e-6), name
        assert abs(float(v.std()) - k / math.sqrt(3.0)) <= 0.05 * k, (name, float(v.std()))
    return n_lin, n_emb, n_ln


@pytest.mark.parametrize("mt", ["vivt", "v"])
def test_init_bert_weights_statistics(mt):
    cfg = syn.make_config(model_type=mt, hidden_size=128, num_hidden_layers=2, num_attention_heads=4)
    torch.manual_seed(3)
    model = M.StateAwareRecursiveTransformer(cfg)
    n_lin, n_emb, n_ln = _check_init(model, cfg)
    assert n_lin > 60 and n_emb >= 5 and n_ln > 20
    # set_pretrained_embedding replaces the table AFTER init (train.py:691-707): the given values survive untouched
    tbl = torch.randn(cfg.vocab_size, cfg.word_vec_size)
    model.text_embeddings.set_pretrained_embedding(tbl.clone(), freeze=False)
    assert torch.equal(model.text_embeddings.word_embeddings.weight.detach(), tbl)


@pytest.mark.gpu
def test_init_bert_weights_on_device():
    """same statistics when the module is constructed and moved to the MI355X at the headline width"""
    cfg = syn.make_config(model_type="vivt", hidden_size=768, num_hidden_layers=1, num_attention_heads=12)
    torch.manual_seed(5)
    model = M.StateAwareRecursiveTransformer(cfg).to("cuda:0")
    _check_init(model, cfg)


def test_span_cache_is_not_shared_between_same_shape_batches(monkeypatch):
    """Two batches of identical shapes but different ingredient boundaries through ONE model: the second must be read from its
    own [SEP] mask even when the allocator hands its tensor the first one's address (a fresh ``.to(device)`` per step)."""
    monkeypatch.setattr(M, "ops", emul_ops)
    cfg = syn.make_config(model_type="vivt", hidden_size=32, num_hidden_layers=1, num_attention_heads=4, video_feature_size=64,
                          vocab_size=60, word_vec_size=20, action_vocab_size=12, max_v_len=8, max_t_len=6, max_i_len=24)
    torch.manual_seed(0)
    model = M.StateAwareRecursiveTransformer(cfg)
    g = torch.Generator().manual_seed(1)
    for m in (model.ingredient_embeddings, model.text_embeddings):
        m.set_pretrained_embedding(0.4 * torch.randn(cfg.vocab_size, cfg.word_vec_size, generator=g), freeze=False)
    for m in (model.reasoner, model.recipe_reasoner):
        m.set_pretrained_embedding(0.4 * torch.randn(cfg.action_vocab_size, cfg.word_vec_size, generator=g), freeze=False)
    model.eval()

    def run(seed, m):
        b = syn.make_batch(cfg, n_videos=2, max_steps=3, n_ingr=4, seed=seed)
        noise = [torch.zeros(3, cfg.max_t_len, cfg.vocab_size) for _ in range(2)]
        m.gumbel_noise = noise
        with torch.no_grad():
            loss = m(*syn.forward_args(b))[0]
        ptr = b["ingr_sep_masks"].data_ptr()
        return float(loss), b["ingr_sep_masks"].clone(), ptr

    seeds = [11, 12, 13, 14, 15, 16]
    masks = []
    got = []
    for s in seeds:
        l, mk, _ = run(s, model)
        got.append(l)
        masks.append(mk)
    assert any(not torch.equal(masks[0], mk) for mk in masks[1:]), "the synthetic batches must differ in their [SEP] positions"
    import copy
    for s, l in zip(seeds, got):
        fresh = copy.deepcopy(model)
        fresh._span_cache = {}
        fresh._plans, fresh._ptr_plans = {}, {}
        ref, _, _ = run(s, fresh)
        assert abs(l - ref) <= 1e-6 * abs(ref), (s, l, ref)
    # a resident batch (same tensor object, unmodified) is served from the cache; an in-place edit invalidates it
    b = syn.make_batch(cfg, n_videos=2, max_steps=3, n_ingr=4, seed=21)
    sp1 = model._spans_for(b["ingr_sep_masks"])
    assert model._spans_for(b["ingr_sep_masks"]) is sp1
    other = syn.make_batch(cfg, n_videos=2, max_steps=3, n_ingr=4, seed=22)["ingr_sep_masks"]
    assert not torch.equal(other, b["ingr_sep_masks"])
    b["ingr_sep_masks"].copy_(other)
    sp2 = model._spans_for(b["ingr_sep_masks"])
    assert sp2 is not sp1 and sp2[1].host != sp1[1].host


def test_arena_layout_keeps_packed_projections_contiguous():
    """The gradient arena / weight store lay out (a) every attention block as [Wq Wk Wv | bq bk bv] and (b) the cross-attention
    key / value projections of the decoder stack across the layers as [Wk0 Wv0 Wk1 Wv1 … | bk0 bv0 …]; the packed views alias
    exactly the member parameters (reference: one nn.Linear each, model.py:159-172, consumed with the same memory rows by
    every decoder layer, :643-651)."""
    from svpc_amd.optim import GradArena, WeightStore
    cfg = syn.make_config(model_type="vivt", hidden_size=32, num_hidden_layers=3, num_attention_heads=4, video_feature_size=64,
                          vocab_size=60, word_vec_size=20, action_vocab_size=12, max_v_len=8, max_t_len=6, max_i_len=24)
    torch.manual_seed(1)
    model = M.StateAwareRecursiveTransformer(cfg)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    before = {n: p.detach().clone() for n, p in named}
    store = WeightStore(named)
    arena = GradArena(named)
    for n, p in named:                                     # re-pointing keeps every value
        assert torch.equal(p.detach(), before[n]), n
    D, L = cfg.hidden_size, cfg.num_hidden_layers
    for n, p in named:                                     # a recognisable gradient per tensor
        p.grad.fill_(float(arena.names.index(n) + 1))
    dec = model.decoder
    a = dec.layer[0].dec_enc_attention.key.weight
    wg, bg = a._svpc_stack
    w, b, w16 = a._svpc_stack_w
    assert tuple(wg.shape) == (2 * L * D, D) and tuple(bg.shape) == (2 * L * D,) and w.shape == wg.shape and w16.dtype == torch.bfloat16
    for l, layer in enumerate(dec.layer):
        att = layer.dec_enc_attention
        for j, lin in enumerate((att.key, att.value)):
            r0 = (2 * l + j) * D
            assert torch.equal(w[r0:r0 + D], lin.weight.detach()) and torch.equal(b[r0:r0 + D], lin.bias.detach())
            assert wg[r0:r0 + D].data_ptr() == lin.weight.grad.data_ptr() and bg[r0:r0 + D].data_ptr() == lin.bias.grad.data_ptr()
            assert torch.equal(wg[r0:r0 + D], lin.weight.grad)
        # the per-layer K|V view is still there; Q|K|V of this block is no longer contiguous and must not be offered
        pk = att.query.weight._svpc_packed
        assert "kv" in pk and "q" in pk and "qkv" not in pk
        assert pk["kv"][0].data_ptr() == att.key.weight.grad.data_ptr() and tuple(pk["kv"][0].shape) == (2 * D, D)
        # the self-attention block keeps the full packing
        sp = layer.self_attention.query.weight._svpc_packed
        assert set(sp) == {"qkv", "kv", "q"} and sp["qkv"][0].data_ptr() == layer.self_attention.query.weight.grad.data_ptr()
        assert torch.equal(sp["qkv"][0][D:2 * D], layer.self_attention.key.weight.grad)
    st = dec.stacked_memory_kv()
    assert st is not None and st[0].data_ptr() == w.data_ptr() and st[2].data_ptr() == wg.data_ptr()
    # the fallback concatenation of a block without a contiguous layout still produces the right operand
    wq, bq, _, _, _ = dec.layer[1].dec_enc_attention.packed("qkv")
    assert torch.equal(wq[:D], dec.layer[1].dec_enc_attention.query.weight.detach())
    assert store.numel == arena.numel
