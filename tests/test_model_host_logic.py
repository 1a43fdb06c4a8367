"""Host-side orchestration of the batched model (index plans, ragged handling, return contract, state_dict
names) checked on CPU: the HIP primitives are swapped for their pure-torch statements (tests/emul_ops.py) and
the result is compared with the reference's golden outputs and with the oracle.  No GPU needed."""
import os

import numpy as np
import pytest
import torch

import emul_ops
from oracle import svpc_oracle as orc
from oracle.cases import case_config_and_batch
from svpc_amd import model as M
from svpc_amd import synthetic as syn
from helpers import build_model


@pytest.fixture(autouse=True)
def _emulated_ops(monkeypatch):
    monkeypatch.setattr(M, "ops", emul_ops)


def build(case, mt, golden_dir):
    return build_model(case, mt, golden_dir)


@pytest.mark.parametrize("mt", ["v", "vi", "viv", "vivt"])
def test_state_dict_names_match_reference(golden_dir, mt):
    z, cfg, batch, model = build("tiny", mt, golden_dir)
    ref = {k[len("param/"):]: z[k].shape for k in z.files if k.startswith("param/")}
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert set(ref) == set(mine)
    for k in ref:
        assert tuple(ref[k]) == mine[k], k
    # sinusoid buffers equal the reference's
    for k in ref:
        if k.endswith(".pe"):
            np.testing.assert_allclose(model.state_dict()[k].numpy(), z["param/" + k], atol=1e-6)


@pytest.mark.parametrize("case,mt", [("tiny", "v"), ("tiny", "vi"), ("tiny", "viv"), ("tiny", "vivt"), ("tiny_ls0", "v"), ("tiny_ls0", "vivt")])
def test_batched_forward_backward_matches_reference(golden_dir, case, mt):
    z, cfg, batch, model = build(case, mt, golden_dir)
    loss, probs, ents, acts = model(*syn.forward_args(batch))
    assert abs(loss.item() - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    for b, p in enumerate(probs):
        np.testing.assert_allclose(p.detach().numpy(), z["probs/%d" % b], rtol=1e-4, atol=1e-7)
    for b, e in enumerate(ents):
        np.testing.assert_allclose(e.detach().numpy(), z["ent/%d" % b], rtol=1e-4, atol=1e-7)
    for b, a in enumerate(acts):
        np.testing.assert_allclose(a.detach().numpy(), z["act/%d" % b], rtol=1e-4, atol=1e-7)
    loss.backward()
    n = 0
    for name, p in model.named_parameters():
        k = "grad/" + name
        if k not in z.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        ref = z[k]
        scale = max(1e-6, float(np.abs(ref).max()))
        assert p.grad is not None, name
        assert float(np.abs(p.grad.numpy() - ref).max()) <= 3e-4 * scale + 1e-6, name
        n += 1
    assert n > 20


@pytest.mark.parametrize("mt", ["vivt", "v"])
def test_packed_text_rows_match_reference(golden_dir, mt):
    """``model.pack_text_rows``: the sentence side over the valid tokens only (svpc_amd.model.TextPack) — the loss and the gradients are
    the reference's (config-1 shape: sentences of 7..22 of 22 tokens); the returned probabilities are the reference's at the valid
    positions and zero at the pad positions."""
    from svpc_amd import keep_host_copy
    z, cfg, batch, model = build("c1", mt, golden_dir)
    args = syn.forward_args(batch)
    for m in args[2]:
        keep_host_copy(m, m)
    model.pack_text_rows = True
    loss, probs, ents, acts = model(*args)
    assert model._pack_cache, "the packed path was not taken"
    pk = next(iter(model._pack_cache.values()))
    assert pk.R < sum(p.shape[0] * p.shape[1] for p in probs), "the fixture has no pad tokens"
    assert abs(loss.item() - float(z["loss"])) <= 5e-5 * abs(float(z["loss"]))
    Lv, Lt = cfg.max_v_len, cfg.max_t_len
    for b, p in enumerate(probs):
        ref = z["probs_slice/%d" % b]
        for s_ in range(p.shape[0]):
            n = int(args[2][s_][b, Lv:Lv + Lt].sum())
            np.testing.assert_allclose(p[s_, :n, ::37].detach().numpy(), ref[s_, :n], rtol=2e-4, atol=1e-7)
            if n < Lt:
                assert float(p[s_, n:].detach().abs().max()) == 0.0         # (pad positions: zeros, not the reference's values)
    loss.backward()
    n_checked = 0
    for name, p in model.named_parameters():
        k = "gradnorm/" + name
        if k not in z.files:
            continue
        ref = float(z[k])
        assert abs(float(p.grad.double().norm()) - ref) <= 1e-3 * ref + 1e-6, name
        g = p.grad.reshape(-1)
        sl = g[:: max(1, g.numel() // 64)][:64].numpy()
        rs = z["gradslice/" + name]
        assert float(np.abs(sl - rs).max()) <= 1e-3 * max(1e-6, float(np.abs(rs).max())) + 1e-6, name
        n_checked += 1
    assert n_checked > 20


def test_c1_vivt_loss(golden_dir):
    z, cfg, batch, model = build("c1", "vivt", golden_dir)
    with torch.no_grad():
        loss, probs, ents, acts = model(*syn.forward_args(batch))
    assert abs(loss.item() - float(z["loss"])) <= 5e-5 * abs(float(z["loss"]))
    for b, p in enumerate(probs):
        np.testing.assert_allclose(p.numpy()[:, :, ::37], z["probs_slice/%d" % b], rtol=2e-4, atol=1e-7)


def test_predict_contract(golden_dir):
    z, cfg, batch, model = build("tiny", "vivt", golden_dir)
    with torch.no_grad():
        mem, ents, acts = model(*syn.forward_args(batch), predict=True)
    assert len(mem) == 2 and set(mem[0]) == {"entity_probs", "action_probs", "entity_vectors", "re_pred_entity_probs",
                                             "re_pred_action_probs", "re_entity_vectors"}
    assert mem[0]["entity_vectors"][1].shape == (3, 3, cfg.hidden_size)
    assert mem[1]["entity_vectors"][1].shape == (2, 2, cfg.hidden_size)


@pytest.mark.parametrize("mt", ["v", "vivt"])
def test_reference_shaped_submodules(golden_dir, mt):
    """The call surface src/translator.py uses: forward_step / step encoder / reasoner / decoder / pointer."""
    z, cfg, batch, model = build("tiny", mt, golden_dir)
    P = {k: v for k, v in model.state_dict().items()}
    with torch.no_grad():
        enc = model.forward_step(batch["input_ids_list"][0], batch["video_features_list"][0], batch["input_masks_list"][0])
        np.testing.assert_allclose(enc.numpy(), z["mid/encoder/0"], rtol=1e-4, atol=2e-6)
        ing = model.ingredient_embeddings(batch["ingr_input_ids"], batch["ingr_sep_masks"])
        np.testing.assert_allclose(ing.numpy(), z["mid/ingredient_embeddings/0"], rtol=1e-4, atol=2e-6)
        x = model.step_positional_encoding(enc[:, 0, :].unsqueeze(0))
        g = model.step_wise_encoder(x, torch.ones(1, x.shape[1]))[-1]
        ref_g = orc.encoder(P, "step_wise_encoder", (enc[:, 0] + orc.sinusoid_table(50, cfg.hidden_size)[:2]).unsqueeze(0),
                            torch.ones(1, 2), cfg)
        np.testing.assert_allclose(g.numpy(), ref_g.numpy(), rtol=1e-4, atol=2e-6)
        t = model.text_embeddings(batch["input_ids_list"][0][:, cfg.max_v_len:])
        np.testing.assert_allclose(t.numpy(), orc.text_embed(P, batch["input_ids_list"][0][:, cfg.max_v_len:], cfg).numpy(),
                                   rtol=1e-4, atol=2e-6)
        if mt == "vivt":
            outs = model.reasoner(g, ing[0, :3])
            ref = orc.simulator(P, "reasoner", g[0], ing[0, :3])
            for a, b_ in zip(outs, ref):
                np.testing.assert_allclose(a.numpy(), b_.numpy(), rtol=1e-4, atol=2e-6)
            mem = torch.randn(2, 3, cfg.hidden_size)
            d = model.decoder(t, torch.ones(2, cfg.max_t_len), mem, torch.ones(2, 3), diagonal_mask=True)[-1]
            ref_d = orc.decoder(P, t, torch.ones(2, cfg.max_t_len), mem, torch.ones(2, 3), cfg)
            np.testing.assert_allclose(d.numpy(), ref_d.numpy(), rtol=1e-4, atol=2e-6)
            bank = torch.randn(2, 3, cfg.hidden_size)
            pg = model.pointer_generator_network(d, bank, batch["ingr_id_dict"][0], 1)
            ref_pg = orc.pointer_generator(P, d, bank, batch["ingr_id_dict"][0], 1, cfg)
            np.testing.assert_allclose(pg.numpy(), ref_pg.numpy(), rtol=1e-4, atol=1e-7)
        lg = model.decoder_classifier(torch.ones(2, 3, cfg.hidden_size))
        np.testing.assert_allclose(lg.numpy(), orc.lm_head(P, torch.ones(2, 3, cfg.hidden_size), cfg).numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("incremental", [True, False])
@pytest.mark.parametrize("mt", ["v", "vi", "viv", "vivt"])
def test_translator_greedy_ids_bit_exact_vs_reference(golden_dir, mt, incremental):
    """Batched greedy decoding (host logic over the emulated ops) reproduces the reference Translator's id matrices
    (tests/golden decode/*), KV-cached and re-run-everything forms alike."""
    from svpc_amd import translator as TR
    z, cfg, batch, model = build("tiny", mt, golden_dir)
    import svpc_amd.translator as trmod
    trmod.ops = emul_ops
    try:
        tr = TR.Translator(type("O", (), {"cuda": False})(), {"model_cfg": cfg, "model": model.state_dict()}, model=model,
                           incremental=incremental)
        dec, oov = tr.translate_batch(syn.translate_inputs(batch))
    finally:
        from svpc_amd import ops as real_ops
        trmod.ops = real_ops
    for b, d in enumerate(dec):
        np.testing.assert_array_equal(d.numpy(), z["decode/%d" % b])
