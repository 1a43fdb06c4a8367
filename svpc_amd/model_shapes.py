"""Name → shape table of the trainable parameters of ``StateAwareRecursiveTransformer``.

This is the checkpoint-compatibility contract with the reference (its ``state_dict`` keys and shapes;
reference: src/rtransformer/model.py:826-873 constructor, SURVEY.md §8(b) "Parameters / state").
``model_type`` matters only for the verb tables: after ``set_pretrained_embedding`` the reference stores
them as a bare Parameter ``<sim>.action_embeddings`` (model.py:773-775), otherwise as
``<sim>.action_embeddings.weight`` (src/train.py:699-707 decides which simulators get the call).
"""
from __future__ import annotations

from collections import OrderedDict


def _attn(out, pre, D):
    for n in ("query", "key", "value"):
        out[pre + "." + n + ".weight"] = (D, D)
        out[pre + "." + n + ".bias"] = (D,)


def _ln(out, pre, D):
    out[pre + ".weight"] = (D,)
    out[pre + ".bias"] = (D,)


def _lin(out, pre, o, i, bias=True):
    out[pre + ".weight"] = (o, i)
    if bias:
        out[pre + ".bias"] = (o,)


def _fc_stack(out, pre, I, O):
    _ln(out, pre + ".0", I)
    _lin(out, pre + ".2", O, I)
    _ln(out, pre + ".4", O)


def _encoder(out, pre, L, D, Dff):
    for i in range(L):
        p = "%s.layer.%d" % (pre, i)
        _attn(out, p + ".attention.self", D)
        _lin(out, p + ".attention.output.dense", D, D)
        _ln(out, p + ".attention.output.LayerNorm", D)
        _lin(out, p + ".hidden_intermediate.dense", Dff, D)
        _lin(out, p + ".memory_intermediate.dense", Dff, D)   # dead: never called (model.py:571, 574-591)
        _lin(out, p + ".output.dense", D, Dff)
        _ln(out, p + ".output.LayerNorm", D)


def _simulator(out, pre, D, A, W, bare_table):
    if bare_table:
        out[pre + ".action_embeddings"] = (A, W)
    _lin(out, pre + ".action_selector.0", D, D)
    _lin(out, pre + ".action_selector.3", A, D)
    if not bare_table:
        out[pre + ".action_embeddings.weight"] = (A, W)
    _lin(out, pre + ".W1.0", D, D)
    _lin(out, pre + ".W2", D, D + A)
    _lin(out, pre + ".W3", 3, D)
    _lin(out, pre + ".W4", 1, W)


def parameter_shapes(cfg, model_type=None):
    mode = cfg.model_mode
    D, Dff, F = cfg.hidden_size, cfg.intermediate_size, cfg.video_feature_size
    V, W, A, L = cfg.vocab_size, cfg.word_vec_size, cfg.action_vocab_size, cfg.num_hidden_layers
    out = OrderedDict()
    out["ingredient_embeddings.word_embeddings.weight"] = (V, W)
    _fc_stack(out, "ingredient_embeddings.word_fc", W, D)
    _fc_stack(out, "video_embeddings.video_embeddings", F, D)
    out["text_embeddings.word_embeddings.weight"] = (V, W)
    _fc_stack(out, "text_embeddings.word_fc", W, D)
    out["token_type_embeddings.weight"] = (4, D)
    _encoder(out, "encoder", L, D, Dff)
    _encoder(out, "step_wise_encoder", L, D, Dff)
    _simulator(out, "reasoner", D, A, W, bare_table=mode in ("full", "reason_copy"))
    _lin(out, "Wac.0", D, W)
    _lin(out, "Went.0", D, D)
    for i in range(L):
        p = "decoder.layer.%d" % i
        _attn(out, p + ".self_attention", D)
        _ln(out, p + ".norm1", D)
        _attn(out, p + ".dec_enc_attention", D)
        _ln(out, p + ".norm2", D)
        _lin(out, p + ".output.dense", D, Dff)
        _ln(out, p + ".output.LayerNorm", D)
    out["decoder_classifier.bias"] = (V,)
    _lin(out, "decoder_classifier.transform.dense", D, D)
    _ln(out, "decoder_classifier.transform.LayerNorm", D)
    _lin(out, "decoder_classifier.decoder", V, D, bias=False)
    _lin(out, "Wing", Dff, Dff)
    _lin(out, "pgen_linear.0", 1, 2 * Dff)
    for sfx in ("", "_reverse"):
        out["recipe_encoder.weight_ih_l0" + sfx] = (4 * D, W)
        out["recipe_encoder.weight_hh_l0" + sfx] = (4 * D, D)
        out["recipe_encoder.bias_ih_l0" + sfx] = (4 * D,)
        out["recipe_encoder.bias_hh_l0" + sfx] = (4 * D,)
    _simulator(out, "recipe_reasoner", D, A, W, bare_table=mode == "full")
    return out
