"""StateAwareRecursiveTransformer — MI355X-native drop-in for the reference module of the same name.

Boundary: same constructor (one EasyDict-like config), same ``forward`` signature and
return contract (reference: src/rtransformer/model.py:1027-1189), same sub-module call surface used by the
greedy decoder (src/translator.py:57-104) and the same ``state_dict`` names/shapes (model.py:826-873), so a
reference checkpoint loads unchanged.

Architecture (not a port): the reference loops S×forward_step and then N× per-video python code issuing
~10^4 tiny eager ops.  Here the whole batch is flattened into *row arrays* in HBM —
  clip rows   (T·Lv, ·)   T = Σ_b S_b valid clips, video-major
  text rows   (T·Lt, ·)
  step rows   (T, ·)
  entity rows (Σ_b E_b, ·) and padded banks (T, Emax, D)
— and every stage runs once over all rows through the HIP primitives of ``svpc_amd.ops`` (LayerNorm family,
MFMA GEMM with fused epilogues, segmented attention, simulator recurrence, pointer/loss, Gumbel, LSTM cell).
Ragged videos are handled by index maps built once per batch shape (``BatchPlan``), never by padding work.
All math runs in HIP kernels; torch is used for allocation, views, cat/stack/index copies and autograd wiring.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .ops_common import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, BulkUpload, FIdx, Idx

PAD_ROW = 0  # nn.Embedding(padding_idx=0) in the reference (model.py:492, 519)


def _sinusoid(max_len, d):
    """reference: model.py:87-92."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len).float().unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _i32(v, device):
    return torch.tensor(v, dtype=torch.int32, device=device)


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's names (their own forward()s are reference-shaped wrappers)
# ------------------------------------------------------------------------------------------------
class BertLayerNorm(nn.Module):
    """reference: model.py:143-156 (TF-style, eps inside the sqrt)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        y = ops.layernorm(x.reshape(-1, x.shape[-1]), self.weight, self.bias, self.variance_epsilon)
        return y.view(x.shape)


class PositionEncoding(nn.Module):
    """reference: model.py:67-104."""

    def __init__(self, n_filters=128, max_len=500):
        super().__init__()
        self.register_buffer("pe", _sinusoid(max_len, n_filters))

    def forward(self, x):
        L, D = x.shape[-2], x.shape[-1]
        flat = x.reshape(-1, D)
        R = flat.shape[0]
        y = ops.span_mean(flat, Idx(range(R)), Idx([1] * R), add=self.pe, add_idx=Idx([r % L for r in range(R)]))
        return y.view(x.shape)


class _FcStack(nn.Module):
    """LN → Dropout → Linear → ReLU → LN, indexed 0..4 like the reference's nn.Sequential
    (model.py:493-499, 520-526, 548-554) so parameter names are ``<name>.{0,2,4}.*``."""

    def __init__(self, d_in, d_out, eps):
        super().__init__()
        self.add_module("0", BertLayerNorm(d_in, eps))
        self.add_module("2", nn.Linear(d_in, d_out))
        self.add_module("4", BertLayerNorm(d_out, eps))

    def __getitem__(self, i):
        return getattr(self, str(i))

    def run(self, x, eps, src_rows=None, pad_row=-1, drop=None, add1=None, add1_mod=0, add2=None, add2_idx=None, out_bf16=False,
            final_bf16=False):
        """out_bf16: the whole stack runs on a bf16 stream from the first LayerNorm on; final_bf16: only the last LayerNorm writes
        bf16 (its consumer is a bf16 stream — saves the separate cast)."""
        xn = ops.layernorm(x, self[0].weight, self[0].bias, eps, src_rows=src_rows, pad_row=pad_row, post_drop=drop,
                           out_bf16=out_bf16)
        h = ops.linear(xn, self[2].weight, self[2].bias, act=ACT_RELU)
        return ops.layernorm(h, self[4].weight, self[4].bias, eps, add1=add1, add1_mod=add1_mod,
                             add2=add2, add2_idx=add2_idx, out_bf16=final_bf16)


class _Seq1(nn.Module):
    """nn.Sequential(Linear, act) containers whose Linear is child ``0`` (Wac, Went, W1, pgen_linear)."""

    def __init__(self, d_in, d_out, act):
        super().__init__()
        self.add_module("0", nn.Linear(d_in, d_out))
        self.act = act

    def __getitem__(self, i):
        return getattr(self, str(i))

    def forward(self, x):
        y = ops.linear(x.reshape(-1, x.shape[-1]), self[0].weight, self[0].bias, act=self.act)
        return y.view(*x.shape[:-1], y.shape[-1])


class BertSelfAttention(nn.Module):
    """reference: model.py:159-220 (parameters only; the core runs in ops.attention)."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        D = config.hidden_size
        self.num_attention_heads = config.num_attention_heads
        self.query, self.key, self.value = nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D)

    def packed(self, which="qkv"):
        """Packed projection → (weight, bias, weight-grad view, bias-grad view, bf16 weight shadow).

        Once the parameters live in a ``WeightStore`` (built by the fused optimizer, or ``WeightStore.for_model`` for
        inference) weight / bias / shadow are views of its contiguous [Wq Wk Wv | bq bk bv] block; the gradient views come
        from the optimizer's arena.  Before that the weights are concatenated and autograd splits the gradient."""
        qw = self.query.weight
        pg = getattr(qw, "_svpc_packed", None)
        wg, bg = pg[which] if pg is not None and which in pg else (None, None)
        pw = getattr(qw, "_svpc_packed_w", None)
        if pw is not None and which in pw and (wg is not None or not torch.is_grad_enabled()):
            for p in qw._svpc_packed_w_members[:3]:
                ops._shadow(p)
            w, b, w16 = pw[which]
            return w, b, wg, bg, w16
        mods = {"q": self.query, "k": self.key, "v": self.value}
        w = torch.cat([mods[c].weight for c in which], 0)
        b = torch.cat([mods[c].bias for c in which], 0)
        return w, b, wg, bg, None


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class _Ctx:
    """Per-forward execution context: hyper-parameters + dropout sites."""

    def __init__(self, config, training, rng):
        self.H = config.num_attention_heads
        self.eps = config.layer_norm_eps
        self.training = training
        self.rng = rng
        self.p_h = config.hidden_dropout_prob
        self.p_a = config.attention_probs_dropout_prob

    def drop(self, p):
        if not self.training or p <= 0.0:
            return None
        return (p, self.rng, self.rng.site())


class BertLayerNoMemoryUntied(nn.Module):
    """reference: model.py:565-591.  ``memory_intermediate`` is dead there (never called) and here."""

    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.hidden_intermediate = BertIntermediate(config)
        self.memory_intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def run(self, h, seq, key_mask, cx):
        D = h.shape[1]
        w, b, wg, bg, w16 = self.attention.self.packed()
        qkv = ops.linear(h, w, b, wgrad=wg, bgrad=bg, w16=w16)
        ctx = ops.attention(qkv, qkv, (0, D, 2 * D), D, cx.H, seq, key_mask=key_mask, causal=False, drop=cx.drop(cx.p_a))
        so = self.attention.output
        ao = ops.linear(ctx, so.dense.weight, so.dense.bias)
        # sink=True: h / x1 are consumed by exactly one projection besides the residual path; its dgrad absorbs the residual gradient
        x1 = ops.layernorm(ao, so.LayerNorm.weight, so.LayerNorm.bias, cx.eps, residual=h, pre_drop=cx.drop(cx.p_h), sink=True)
        # (fuse_act_bwd: `it` feeds exactly one projection, whose dgrad applies the GELU backward in its epilogue)
        it = ops.linear(x1, self.hidden_intermediate.dense.weight, self.hidden_intermediate.dense.bias, act=ACT_GELU, fuse_act_bwd=True)
        o = ops.linear(it, self.output.dense.weight, self.output.dense.bias)
        return ops.layernorm(o, self.output.LayerNorm.weight, self.output.LayerNorm.bias, cx.eps,
                             residual=x1, pre_drop=cx.drop(cx.p_h), sink=True)


    def run_rows(self, h, sel_rows, seq_sel, key_mask, cx):
        """Same layer, evaluated only for the query rows ``sel_rows`` (keys/values still come from every row of ``h``).
        The training forward consumes nothing but the [CLS] row of the last clip-encoder layer (model.py:1062-1064), so the
        other 99 % of that layer's query/output/FFN rows — which the reference computes and discards — are never formed;
        the selected rows are bit-for-bit what ``run`` produces for them (in eval mode)."""
        D = h.shape[1]
        att = self.attention.self
        # the few selected rows leave the bf16 / split stream here (no-op in fp32 storage); the key / value projection — the stream's only
        # other consumer — takes the alias, so the rows' gradient joins the stream gradient in place (ops.take_rows_f32_alias)
        hq, h = ops.take_rows_f32_alias(h, sel_rows)
        q = ops.linear(hq, att.query.weight, att.query.bias)
        wkv, bkv, wg, bg, w16 = att.packed("kv")
        kv = ops.linear(h, wkv, bkv, wgrad=wg, bgrad=bg, w16=w16)
        # (fp32 queries against the stream's keys / values in place: bf16 rows, or split rows in the bf16x3 mode — ops.attention)
        ctx = ops.attention(q, kv, (0, 0, D), D, cx.H, seq_sel, key_mask=key_mask, causal=False, drop=cx.drop(cx.p_a)).float()
        so = self.attention.output
        ao = ops.linear(ctx, so.dense.weight, so.dense.bias)
        x1 = ops.layernorm(ao, so.LayerNorm.weight, so.LayerNorm.bias, cx.eps, residual=hq, pre_drop=cx.drop(cx.p_h))
        it = ops.linear(x1, self.hidden_intermediate.dense.weight, self.hidden_intermediate.dense.bias, act=ACT_GELU)
        o = ops.linear(it, self.output.dense.weight, self.output.dense.bias)
        return ops.layernorm(o, self.output.LayerNorm.weight, self.output.LayerNorm.bias, cx.eps,
                             residual=x1, pre_drop=cx.drop(cx.p_h))


class BertEncoderNoMemoryUntied(nn.Module):
    """reference: model.py:594-617."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayerNoMemoryUntied(config) for _ in range(config.num_hidden_layers)])

    def run(self, h, seq, key_mask, cx, last_rows=None, last_seq=None):
        """All layers; with ``last_rows`` the last layer is evaluated for those query rows only (returns (len(rows), D))."""
        n = len(self.layer)
        for i, layer in enumerate(self.layer):
            if last_rows is not None and i == n - 1:
                h = layer.run_rows(h, last_rows, last_seq, key_mask, cx)
            else:
                h = layer.run(h, seq, key_mask, cx)
        return h

    def forward(self, hidden_states, attention_mask, diagonal_mask=False, output_all_encoded_layers=True):
        B, L, D = hidden_states.shape
        cx = _Ctx(self.config, self.training, ops.default_rng(hidden_states.device))
        seq = ops.SeqInfo.uniform(B, L, L, hidden_states.device)
        km = attention_mask.reshape(-1).float().contiguous()
        h = self.run(hidden_states.reshape(B * L, D).contiguous(), seq, km, cx)
        return [h.view(B, L, D)]


class BertDecoderLayerNoMemoryUntied(nn.Module):
    """reference: model.py:620-663."""

    def __init__(self, config):
        super().__init__()
        self.self_attention = BertSelfAttention(config)
        self.norm1 = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dec_enc_attention = BertSelfAttention(config)
        self.norm2 = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.output = BertOutput(config)

    def run(self, x, text_mask, mem, seq_self, seq_cross, mem_mask, cx, kvc=None):
        """``kvc``: this layer's K|V rows of the memory when the stack has projected them for all layers at once."""
        D = x.shape[1]
        w, b, wg, bg, w16 = self.self_attention.packed()
        qkv = ops.linear(x, w, b, wgrad=wg, bgrad=bg, w16=w16)
        sa = ops.attention(qkv, qkv, (0, D, 2 * D), D, cx.H, seq_self, key_mask=text_mask, causal=True,
                           drop=cx.drop(cx.p_a))
        x1 = ops.layernorm(sa, self.norm1.weight, self.norm1.bias, cx.eps, residual=x, sink=True)
        ca_m = self.dec_enc_attention
        qc = ops.linear(x1, ca_m.query.weight, ca_m.query.bias)
        if kvc is None:
            wkv, bkv, wg, bg, w16 = ca_m.packed("kv")
            kvc = ops.linear(mem, wkv, bkv, wgrad=wg, bgrad=bg, w16=w16)
        if kvc.dtype != qc.dtype:          # (memory rows handed over in another storage type than the sentence stream)
            kvc = kvc.to(qc.dtype)
        n_seq = seq_cross.n
        uniform = n_seq > 0 and x1.shape[0] == n_seq * seq_cross.max_q
        ragged = n_seq > 0 and not uniform and x1.shape[0] == seq_cross.n_q_rows and getattr(seq_cross, "packed_rows", False)
        lt = seq_cross.max_q
        if ((uniform or ragged) and kvc.shape[0] == n_seq * seq_cross.max_k and x1.is_cuda
                and (ops.lo_off(qc) is None) == (ops.lo_off(kvc) is None) and (ops.lo_off(qc) is None) == (ops.lo_off(x1) is None)
                and qc.dtype == x1.dtype and ops.cross_attn_ln_usable(D, cx.H, lt, seq_cross.max_k, mem_mask)):
            # sentences (uniform, or ragged: the valid tokens only) over ≤ 3 memory rows each: attention + residual + LayerNorm in one
            # launch, forward and backward (svpc_amd/csrc/cross_attn.hip)
            rows = (seq_cross.table[0], seq_cross.table[1]) if ragged else None
            x2 = ops.cross_attn_ln(qc, x1, kvc, self.norm2.weight, self.norm2.bias, cx.eps, cx.H, lt, seq_cross.max_k,
                                   drop=cx.drop(cx.p_a), sink=True, rows=rows)
        else:
            ca = ops.attention(qc, kvc, (0, 0, D), D, cx.H, seq_cross, key_mask=mem_mask, causal=False, drop=cx.drop(cx.p_a))
            x2 = ops.layernorm(ca, self.norm2.weight, self.norm2.bias, cx.eps, residual=x1, sink=True)
        o = ops.linear(x2, self.output.dense.weight, self.output.dense.bias)
        return ops.layernorm(o, self.output.LayerNorm.weight, self.output.LayerNorm.bias, cx.eps,
                             residual=x2, pre_drop=cx.drop(cx.p_h), sink=True)


    def memory_kv(self, mem):
        """Cross-attention K|V of the memory rows: constant over the decoding iterations (the reference recomputes it Lt times)."""
        wkv, bkv, _, _, _ = self.dec_enc_attention.packed("kv")
        return ops.linear(mem, wkv, bkv)

    def step(self, x, pos, lt, cache, mem_kv, seq_self, seq_cross, cx):
        """The layer for ONE new token per sentence (position ``pos``): its K|V row is appended to ``cache`` ((T·lt, 2D), sentence-
        major), the query attends to the pos+1 cached keys (the causal mask of the reference, model.py:630-640, lets position
        ``pos`` see exactly those), then cross-attention to the precomputed memory K|V and the output block."""
        T, D = x.shape
        w, b, _, _, _ = self.self_attention.packed()
        qkv = ops.linear(x, w, b)
        # cache append + one-query attention + residual + LayerNorm in one launch where the shape allows (fp32 rows, heads of 64)
        x1 = ops.attn_q1_ln(qkv, cache, lt, pos + 1, x, self.norm1.weight, self.norm1.bias, cx.eps, cx.H, new_kv=qkv[:, D:])
        if x1 is None:
            cache.view(T, lt, 2 * D)[:, pos].copy_(qkv[:, D:])
            sa = ops.attention(qkv, cache, (0, 0, D), D, cx.H, seq_self, key_mask=None, causal=False)
            x1 = ops.layernorm(sa, self.norm1.weight, self.norm1.bias, cx.eps, residual=x)
        ca_m = self.dec_enc_attention
        qc = ops.linear(x1, ca_m.query.weight, ca_m.query.bias)
        n_mem = mem_kv.shape[0] // T
        x2 = ops.attn_q1_ln(qc, mem_kv, n_mem, n_mem, x1, self.norm2.weight, self.norm2.bias, cx.eps, cx.H)
        if x2 is None:
            ca = ops.attention(qc, mem_kv, (0, 0, D), D, cx.H, seq_cross, key_mask=None, causal=False)
            x2 = ops.layernorm(ca, self.norm2.weight, self.norm2.bias, cx.eps, residual=x1)
        o = ops.linear(x2, self.output.dense.weight, self.output.dense.bias)
        return ops.layernorm(o, self.output.LayerNorm.weight, self.output.LayerNorm.bias, cx.eps, residual=x2)


class BertDecoderNoMemoryUntied(nn.Module):
    """reference: model.py:666-694."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertDecoderLayerNoMemoryUntied(config) for _ in range(config.num_hidden_layers)])

    def stacked_memory_kv(self):
        """[Wk0 Wv0 Wk1 Wv1 …] (L·2D, D), its bias, the gradient views and the bf16 shadow — available once the parameters live in
        a ``WeightStore`` / ``GradArena``, whose layout keeps the cross-attention key / value projections of the stack together."""
        a = self.layer[0].dec_enc_attention.key.weight
        sw = getattr(a, "_svpc_stack_w", None)
        sg = getattr(a, "_svpc_stack", None)
        if sw is None or (sg is None and torch.is_grad_enabled()):
            return None
        if sw[0].shape[0] != 2 * len(self.layer) * a.shape[0]:
            return None
        for p in a._svpc_stack_w_members:
            ops._shadow(p)
        wg, bg = sg if sg is not None else (None, None)
        return sw[0], sw[1], wg, bg, sw[2]

    def streams_bf16(self, rows, width):
        """whether ``run`` keeps the sentence activations of (rows, width) in bf16"""
        return ops.bf16_stream_ok(rows, width, self.config.intermediate_size)

    def run(self, x, text_mask, mem, seq_self, seq_cross, mem_mask, cx, keep_stream=False):
        """keep_stream: return the rows as the activation stream holds them (bf16 / split) — the caller feeds a projection that reads the
        stream and takes the fp32 copy itself where it needs one."""
        # interior-only row counts in bf16 precision: the sentence activations (and their gradients) stream through HBM as bf16
        stream_bf16 = self.streams_bf16(x.shape[0], x.shape[1])
        x3 = ops.is_x3()
        ops.require_split_tag(x, "decoder.run")
        ops.require_split_tag(mem, "decoder.run (memory)")
        if stream_bf16:
            if x.dtype != torch.bfloat16:      # (the caller may have had the embedding LayerNorm write bf16 / split rows already)
                x = ops.to_split(x) if x3 else x.to(torch.bfloat16)
            # the memory rows join the stream once, not per layer: each layer's K|V projection then reads bf16 and writes bf16
            # (one cast instead of six casts forward and six backward; its weight gradients join the grouped bf16 launch)
            if mem.dtype == torch.float32 and ops.bf16_stream_ok(mem.shape[0], mem.shape[1], 2 * mem.shape[1]):
                mem = ops.to_split(mem) if x3 else mem.to(torch.bfloat16)
        # Every layer projects the SAME memory rows to its keys and values (reference model.py:643-651): one (R, D) x (D, L·2D)
        # projection for the stack instead of L, each layer reads its 2D columns in place; backward likewise gathers the L
        # key / value gradients in one buffer and runs one dgrad (contraction L·2D) and one wgrad.
        kvs = [None] * len(self.layer)
        st = self.stacked_memory_kv() if mem.dtype == x.dtype else None
        if st is not None:
            w, b, wg, bg, w16 = st
            kvs = ops.split_cols(ops.linear(mem, w, b, wgrad=wg, bgrad=bg, w16=w16), len(self.layer))
        for layer, kvc in zip(self.layer, kvs):
            x = layer.run(x, text_mask, mem, seq_self, seq_cross, mem_mask, cx, kvc=kvc)
        return ops.to_f32(x) if (stream_bf16 and not keep_stream) else x

    def forward(self, dec_hidden_states, dec_mask, enc_outputs, enc_mask, diagonal_mask=True,
                output_all_encoded_layers=False):
        B, Lt, D = dec_hidden_states.shape
        M = enc_outputs.shape[1]
        dev = dec_hidden_states.device
        cx = _Ctx(self.config, self.training, ops.default_rng(dev))
        x = self.run(dec_hidden_states.reshape(B * Lt, D).contiguous(), dec_mask.reshape(-1).float().contiguous(),
                     enc_outputs.reshape(B * M, D).contiguous(), ops.SeqInfo.uniform(B, Lt, Lt, dev),
                     ops.SeqInfo.uniform(B, Lt, M, dev), enc_mask.reshape(-1).float().contiguous(), cx)
        return [x.view(B, Lt, D)]


class BertEmbeddingsVideoUntied(nn.Module):
    """reference: model.py:540-562."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.video_embeddings = _FcStack(config.video_feature_size, config.hidden_size, config.layer_norm_eps)
        self.position_embeddings_video = PositionEncoding(config.hidden_size, config.max_position_embeddings)

    def forward(self, video_features):
        B, Lv, F = video_features.shape
        cx = _Ctx(self.config, self.training, ops.default_rng(video_features.device))
        y = self.video_embeddings.run(video_features.reshape(B * Lv, F).contiguous(), cx.eps, drop=cx.drop(cx.p_h),
                                      add1=self.position_embeddings_video.pe[:Lv].contiguous(), add1_mod=Lv)
        return y.view(B, Lv, -1)


class BertEmbeddingsTextUntied(nn.Module):
    """reference: model.py:484-513."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.word_embeddings = nn.Embedding(config.vocab_size, config.word_vec_size, padding_idx=PAD_ROW)
        self.word_fc = _FcStack(config.word_vec_size, config.hidden_size, config.layer_norm_eps)
        self.position_embeddings_text = PositionEncoding(config.hidden_size, config.max_position_embeddings)

    def set_pretrained_embedding(self, pretrained_embedding, freeze=True):
        assert pretrained_embedding.shape == self.word_embeddings.weight.shape
        self.word_embeddings = nn.Embedding.from_pretrained(pretrained_embedding, freeze=freeze,
                                                            padding_idx=self.word_embeddings.padding_idx)

    def run(self, ids_flat, lt, cx, out_bf16=False, pos_idx=None):
        """``pos_idx``: the position of every row inside its sentence (packed rows: svpc_amd.model.TextPack); default row % lt"""
        pe = self.position_embeddings_text.pe[:lt].contiguous()
        if pos_idx is not None:
            return self.word_fc.run(self.word_embeddings.weight, cx.eps, src_rows=ids_flat, pad_row=PAD_ROW, drop=cx.drop(cx.p_h),
                                    add2=pe, add2_idx=pos_idx, final_bf16=out_bf16)
        return self.word_fc.run(self.word_embeddings.weight, cx.eps, src_rows=ids_flat, pad_row=PAD_ROW,
                                drop=cx.drop(cx.p_h), add1=pe, add1_mod=lt, final_bf16=out_bf16)

    def run_at(self, ids, pos, cx):
        """One token per sentence, all at sentence position ``pos`` (incremental decoding)."""
        return self.word_fc.run(self.word_embeddings.weight, cx.eps, src_rows=ids, pad_row=PAD_ROW, drop=cx.drop(cx.p_h),
                                add1=self.position_embeddings_text.pe[pos:pos + 1].contiguous(), add1_mod=1)

    def forward(self, text_input_ids):
        B, Lt = text_input_ids.shape
        cx = _Ctx(self.config, self.training, ops.default_rng(text_input_ids.device))
        return self.run(text_input_ids.reshape(-1).to(torch.int32), Lt, cx).view(B, Lt, -1)


class IngredientPositionEncoding(nn.Module):
    def __init__(self, n_filters=128, max_len=500):
        super().__init__()
        self.register_buffer("pe", _sinusoid(max_len, n_filters))


class BertEmbeddingsIngredientsUntied(nn.Module):
    """reference: model.py:515-537 + :106-140 (mean of the word vectors between [SEP]s + PE at the
    ingredient index).  ``run`` returns the compact (Σ_b E_b, D) entity rows; ``forward`` the padded
    (B, E_max, D) tensor of the reference (padding rows = bare PE)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.word_embeddings = nn.Embedding(config.vocab_size, config.word_vec_size, padding_idx=PAD_ROW)
        self.word_fc = _FcStack(config.word_vec_size, config.lstm_hidden_size, config.layer_norm_eps)
        self.position_embeddings_ingr = IngredientPositionEncoding(config.lstm_hidden_size,
                                                                   config.max_position_embeddings)

    def set_pretrained_embedding(self, pretrained_embedding, freeze=True):
        self.word_embeddings = nn.Embedding.from_pretrained(pretrained_embedding, freeze=freeze,
                                                            padding_idx=self.word_embeddings.padding_idx)

    @staticmethod
    def spans(sep_cpu):
        """host-side span table from the [SEP] mask: (starts, lens, ingredient index, per-video counts)."""
        B, Li = sep_cpu.shape
        starts, lens, eidx, counts = [], [], [], []
        for b in range(B):
            seps = (sep_cpu[b] == 1).nonzero().view(-1).tolist()
            prev = 0
            for e, s in enumerate(seps):
                starts.append(b * Li + prev)
                lens.append(s - prev)
                eidx.append(e)
                prev = s + 1
            counts.append(len(seps))
        return Idx(starts), Idx(lens), Idx(eidx), counts

    def run(self, ids_flat, spans, cx):
        starts, lens, eidx, _ = spans
        tok = self.word_fc.run(self.word_embeddings.weight, cx.eps, src_rows=ids_flat, pad_row=PAD_ROW,
                               drop=cx.drop(cx.p_h))
        return ops.span_mean(tok, starts, lens, add=self.position_embeddings_ingr.pe, add_idx=eidx)

    def forward(self, ingr_input_ids, ingr_sep_masks):
        dev = ingr_input_ids.device
        cx = _Ctx(self.config, self.training, ops.default_rng(dev))
        spans = self.spans(ingr_sep_masks.cpu())
        rows = self.run(ingr_input_ids.reshape(-1).to(torch.int32), spans, cx)
        counts = spans[3]
        e_max = max(counts)
        out = self.position_embeddings_ingr.pe[:e_max].unsqueeze(0).repeat(len(counts), 1, 1)
        off = 0
        pieces = []
        for b, n in enumerate(counts):
            pieces.append(torch.cat([rows[off:off + n], out[b, n:]], 0))
            off += n
        return torch.stack(pieces)


class EntitiyReasoningNetwork(nn.Module):
    """Visual simulator / textual re-simulator.  reference: model.py:742-823, Eqs. (1)-(7).

    Everything that depends only on the step vector (action selector MLP, verb mixture, W1, W2, W3, W4) is
    hoisted out of the recurrence and runs as batched GEMMs over all T step rows; only the entity-state
    recurrence (sigmoid(E·q), gated write) is sequential and runs in one kernel, one workgroup per video."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        D, A, W = config.lstm_hidden_size, config.action_vocab_size, config.word_vec_size
        self.action_selector = nn.Module()
        self.action_selector.add_module("0", nn.Linear(D, D))
        self.action_selector.add_module("3", nn.Linear(D, A))
        self.action_embeddings = nn.Embedding(A, W)
        self.W1 = _Seq1(D, D, ACT_RELU)
        self.W2 = nn.Linear(D + A, D)
        self.W3 = nn.Linear(D, 3)
        self.W4 = nn.Linear(W, 1)

    def set_pretrained_embedding(self, pretrained_embedding, freeze):
        # the reference keeps the bare weight Parameter (model.py:773-775): state_dict key loses ".weight"
        emb = nn.Embedding.from_pretrained(pretrained_embedding, freeze=freeze)
        self.action_embeddings = emb.weight

    def verb_table(self):
        ae = self.action_embeddings
        if isinstance(ae, nn.Embedding):
            raise TypeError("unsupported operand type(s) for @: 'Tensor' and 'Embedding' — call "
                            "set_pretrained_embedding first (reference behaviour, model.py:798)")
        return ae

    def run(self, g, ents, plan_sim, cx):
        """g (T, D) step rows; ents (ΣE, D) → e (T,Emax), a (T,A), ebar (T,D), Eall (T,Emax,D), fbar (T,W)."""
        sel = self.action_selector
        hid = ops.linear(g, getattr(sel, "0").weight, getattr(sel, "0").bias, act=ACT_RELU, drop=cx.drop(0.4))
        a = ops.linear(hid, getattr(sel, "3").weight, getattr(sel, "3").bias, act=ACT_SIGMOID)
        fbar = ops.linear(ops.row_normalize(a), self.verb_table(), None, trans_w=True)
        hat_h = ops.linear(g, self.W1[0].weight, self.W1[0].bias, act=ACT_RELU)
        q = ops.linear(torch.cat([hat_h, a], 1), self.W2.weight, self.W2.bias)
        heads = ops.sim_heads(hat_h, fbar, self.W3.weight, self.W3.bias, self.W4.weight, self.W4.bias)
        if heads is not None:          # the two tiny heads (D → 3 with its softmax, W → 1) in one launch, forward and backward
            c, w4f = heads
        else:
            c = ops.softmax_rows(ops.linear(hat_h, self.W3.weight, self.W3.bias))
            w4f = ops.linear(fbar, self.W4.weight, self.W4.bias).reshape(-1)
        e, ebar, eall = ops.sim_recur(q, c, w4f, ents, *plan_sim)
        return e, a, ebar, eall, fbar

    def forward(self, video_vectors, entity_vectors):
        g = video_vectors.squeeze(0).contiguous()
        S, E = g.shape[0], entity_vectors.shape[0]
        cx = _Ctx(self.config, self.training, ops.default_rng(g.device))
        e, a, ebar, eall, fbar = self.run(g, entity_vectors.contiguous(),
                                          (Idx([0]), Idx([S]), Idx([0]), Idx([E]), E), cx)
        return e, a, ebar, eall, fbar


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    """reference: model.py:712-739."""

    def __init__(self, config, bert_model_embedding_weights=None):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))

    def run(self, x, eps):
        t = ops.linear(x, self.transform.dense.weight, self.transform.dense.bias, act=ACT_GELU)
        t = ops.layernorm(t, self.transform.LayerNorm.weight, self.transform.LayerNorm.bias, eps)
        return ops.linear(t, self.decoder.weight, self.bias)

    def forward(self, hidden_states):
        shp = hidden_states.shape
        y = self.run(hidden_states.reshape(-1, shp[-1]).contiguous(), self.transform.LayerNorm.variance_epsilon)
        return y.view(*shp[:-1], -1)


class _LSTMParams(nn.LSTM):
    """nn.LSTM used as a parameter container (reference init and names, model.py:865)."""

    def flatten_parameters(self):
        return None


# ------------------------------------------------------------------------------------------------
# batch plan: every index map the flattened execution needs, built once per batch shape
# ------------------------------------------------------------------------------------------------
class BatchPlan:
    """Index maps of one batch structure (step counts, ingredient counts): built once per structure on the host, every device
    array of it sent in ONE copy (ops_common.BulkUpload)."""

    def __init__(self, step_nums, ent_nums, n_steps_padded, N, Lv, Lt, L, n_mem, device):
        self.key = (tuple(step_nums), tuple(ent_nums), n_steps_padded, N, Lv, Lt, L, n_mem)
        T = sum(step_nums)
        self.T, self.N = T, N
        up = BulkUpload(device)
        clip_b, clip_s = [], []
        for b in range(N):
            for s in range(step_nums[b]):
                clip_b.append(b)
                clip_s.append(s)
        src = np.asarray([s * N + b for b, s in zip(clip_b, clip_s)], dtype=np.int64)
        up.add(((src * L)[:, None] + np.arange(Lv)[None, :]).reshape(-1), sink=lambda t: setattr(self, "video_rows", t))
        up.add(((src * L + Lv)[:, None] + np.arange(Lt)[None, :]).reshape(-1), sink=lambda t: setattr(self, "text_rows", t))
        self.cls_rows = Idx([c * Lv for c in range(T)])
        self.ones_T = Idx([1] * T)
        self.step_idx = Idx(clip_s)
        self.step_vid = Idx(clip_b)
        self.text_starts = Idx([j * Lt for j in range(T)])
        self.text_lens = Idx([Lt] * T)
        off = [0]
        for n in step_nums:
            off.append(off[-1] + n)
        eoff = [0]
        for n in ent_nums:
            eoff.append(eoff[-1] + n)
        self.step_off, self.step_len = Idx(off[:-1]), Idx(step_nums)
        self.ent_off, self.ent_len = Idx(eoff[:-1]), Idx(ent_nums)
        self.e_max = max(ent_nums) if ent_nums else 0
        self.step_ne = Idx([ent_nums[b] for b in clip_b])
        self.seq_enc = ops.SeqInfo.uniform(T, Lv, Lv, None)
        up.add([c * Lv for c in range(T)], sink=lambda t: setattr(self, "cls_rows_dev", t))
        self.seq_enc_cls = ops.SeqInfo(list(range(T)), [1] * T, [c * Lv for c in range(T)], [Lv] * T, None)
        self.arange_T = Idx(range(T))
        self.seq_step = ops.SeqInfo(off[:-1], step_nums, off[:-1], step_nums, None)
        self.seq_dec_self = ops.SeqInfo.uniform(T, Lt, Lt, None)
        self.seq_dec_cross = ops.SeqInfo.uniform(T, Lt, n_mem, None)
        self.row_vid = Idx([b for b in clip_b for _ in range(Lt)])
        self.h_step_off, self.h_step_len = off[:-1], list(step_nums)
        self.h_ent_off, self.h_ent_len = eoff[:-1], list(ent_nums)
        pick_f, pick_b = [], []
        for b in range(N):
            for s_ in range(step_nums[b]):
                pick_f.append(s_ * N + b)
                pick_b.append((step_nums[b] - 1 - s_) * N + b)
        self.lstm_pick = {}
        up.add(pick_f, sink=lambda t: self.lstm_pick.__setitem__("", t))
        up.add(pick_b, sink=lambda t: self.lstm_pick.__setitem__("_reverse", t))
        up.add(clip_b, sink=lambda t: setattr(self, "step_vid_dev", t))
        # reverse-direction LSTM: at time t video b consumes its step S_b-1-t
        S = max(step_nums)
        self.lstm_fwd_rows, self.lstm_bwd_rows, self.lstm_active = [None] * S, [None] * S, [None] * S
        for t in range(S):
            act = [1.0 if t < step_nums[b] else 0.0 for b in range(N)]
            up.add([off[b] + min(t, step_nums[b] - 1) for b in range(N)], sink=lambda x, t=t: self.lstm_fwd_rows.__setitem__(t, x))
            up.add([off[b] + max(step_nums[b] - 1 - t, 0) for b in range(N)], sink=lambda x, t=t: self.lstm_bwd_rows.__setitem__(t, x))
            up.add_f(act, sink=lambda x, t=t: self.lstm_active.__setitem__(t, x))
        for v in (self.cls_rows, self.ones_T, self.step_idx, self.step_vid, self.text_starts, self.text_lens, self.step_off, self.step_len,
                  self.ent_off, self.ent_len, self.step_ne, self.arange_T, self.row_vid):
            up.add_idx(v)
        for sq in (self.seq_enc, self.seq_enc_cls, self.seq_step, self.seq_dec_self, self.seq_dec_cross):
            up.add_seq(sq)
        up.flush()
        self.sim = (self.step_off, self.step_len, self.ent_off, self.ent_len, self.e_max)
        self._seq_inc = {}

    def seq_dec_incremental(self, pos, lt, device):
        """Segmentation of incremental decoding at position ``pos``: one query row per sentence, keys = the pos+1 cached rows of
        that sentence in a (T·lt)-row sentence-major cache."""
        key = (pos, lt)
        sq = self._seq_inc.get(key)
        if sq is None:
            T = self.T
            sq = self._seq_inc[key] = ops.SeqInfo(list(range(T)), [1] * T, [s * lt for s in range(T)], [pos + 1] * T, device)
        return sq


class TextPack:
    """Row maps for running the sentence side on the VALID tokens only (``model.pack_text_rows``).  A sentence of n ≤ Lt tokens is
    padded to Lt by the loader (recursive_caption_dataset.py:528-576); nothing a pad row computes reaches the loss (its label is -1,
    pad keys are masked in the causal self-attention, ``reconstruct`` weights it 0 — model.py:630-640, :1017-1025), so the decoder
    stack runs over Σ n rows instead of T·Lt: sentence j owns rows [off_j, off_j + n_j).  Built per (batch plan, sentence lengths) on the
    host from the loader's copy of the masks; every table in one upload."""

    def __init__(self, plan, lens, N, Lv, Lt, L, n_mem, device):
        self.lens = [int(n) for n in lens]
        T = plan.T
        ln = np.asarray(self.lens, dtype=np.int64)
        off_a = np.concatenate([[0], np.cumsum(ln)[:-1]]) if T else np.zeros(0, np.int64)
        self.R = int(ln.sum())
        clip_b = np.asarray(plan.step_vid.host, dtype=np.int64)
        clip_s = np.asarray(plan.step_idx.host, dtype=np.int64)
        base = (clip_s * N + clip_b) * L + Lv
        ar = np.arange(self.R, dtype=np.int64)
        pos = ar - np.repeat(off_a, ln)                       # position of each packed row inside its sentence
        rows = np.repeat(base, ln) + pos                      # rows of the flattened (S·N·L) id / label arrays
        full = np.repeat(np.arange(T, dtype=np.int64) * Lt, ln) + pos       # its row in the padded (T·Lt) layout
        inv = np.full(T * Lt, -1, dtype=np.int64)
        inv[full] = ar                                        # … and back: the packed row of a padded row, or -1
        off = off_a.tolist()
        self.seq_self = ops.SeqInfo(off, self.lens, off, self.lens, None)
        self.seq_cross = ops.SeqInfo(off, self.lens, [j * n_mem for j in range(T)], [n_mem] * T, None)
        self.seq_cross.packed_rows = True        # (consecutive sentences, consecutive memory blocks: what the fused cross-attention assumes)
        up = BulkUpload(device)
        up.add(rows, sink=lambda t: setattr(self, "src_rows", t))
        up.add(pos, sink=lambda t: setattr(self, "pos", t))
        up.add(full, sink=lambda t: setattr(self, "full_rows", t))
        up.add(inv, sink=lambda t: setattr(self, "inv_rows", t))
        self.row_vid = Idx(np.repeat(clip_b, ln).tolist())      # the video of each packed row
        self.off_idx, self.len_idx = Idx(off), Idx(self.lens)
        for ix in (self.row_vid, self.off_idx, self.len_idx):
            up.add_idx(ix)
        up.add_seq(self.seq_self)
        up.add_seq(self.seq_cross)
        up.flush()
        self.full_rows64 = self.full_rows.long()


class StateAwareRecursiveTransformer(nn.Module):
    """reference: src/rtransformer/model.py:826-1189."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        D = config.hidden_size
        self.ingredient_embeddings = BertEmbeddingsIngredientsUntied(config)
        self.video_embeddings = BertEmbeddingsVideoUntied(config)
        self.text_embeddings = BertEmbeddingsTextUntied(config)
        self.token_type_embeddings = nn.Embedding(4, D)
        self.encoder = BertEncoderNoMemoryUntied(config)
        self.step_wise_encoder = BertEncoderNoMemoryUntied(config)
        self.step_positional_encoding = PositionEncoding(n_filters=D, max_len=50)
        self.reasoner = EntitiyReasoningNetwork(config)
        self.Wac = _Seq1(config.word_vec_size, D, ACT_RELU)
        self.Went = _Seq1(config.lstm_hidden_size, D, ACT_RELU)
        self.decoder = BertDecoderNoMemoryUntied(config)
        if config.share_wd_cls_weight:
            # the reference dereferences a non-existent attribute here (model.py:854)
            raise AttributeError("'StateAwareRecursiveTransformer' object has no attribute 'embeddings'")
        self.decoder_classifier = BertLMPredictionHead(config, None)
        self.Wing = nn.Linear(config.intermediate_size, config.intermediate_size)
        self.pgen_linear = _Seq1(config.intermediate_size * 2, 1, ACT_SIGMOID)
        self.eps = 1e-12
        self.recipe_encoder = _LSTMParams(config.word_vec_size, D, batch_first=True, bidirectional=True)
        self.recipe_reasoner = EntitiyReasoningNetwork(config)
        # (> 0: LabelSmoothingLoss; else nn.CrossEntropyLoss(ignore_index=-1) applied to the probabilities, a mean per video —
        # reference model.py:869-870; both live in ops.ptr_mix_loss)
        self.label_smoothing = config.label_smoothing if "label_smoothing" in config else 0.0
        if not self.label_smoothing > 0:
            self.label_smoothing = 0.0
        self.apply(self.init_bert_weights)
        self._plans = {}
        self._ptr_plans = {}
        self._span_cache = {}
        self._pack_cache = {}
        self._rng = None
        self.gumbel_noise = None  # test hook: list of (S_b, Lt, V+X_b) tensors, one per video

    def init_bert_weights(self, module):
        """reference: model.py:875-885."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    # ---------------------------------------------------------------- helpers
    def rng(self, device):
        if self._rng is None or self._rng.device != torch.device(device):
            self._rng = ops.make_rng(device)
        return self._rng

    def _cx(self, device):
        return _Ctx(self.config, self.training, self.rng(device))

    def _n_mem(self):
        return {"full": 3, "reason_copy": 3, "copy": 2}.get(self.config.model_mode, 1)

    def plan_for(self, batch_step_num, ent_nums, n_steps_padded, N, L, device):
        cfg = self.config
        key = (tuple(batch_step_num), tuple(ent_nums), n_steps_padded, N, cfg.max_v_len, cfg.max_t_len, L,
               self._n_mem(), str(device))
        p = self._plans.get(key)
        if p is None:
            if len(self._plans) > 64:
                self._plans.clear()
            p = BatchPlan(list(batch_step_num), list(ent_nums), n_steps_padded, N, cfg.max_v_len, cfg.max_t_len, L,
                          self._n_mem(), device)
            self._plans[key] = p
        return p

    def _text_pack(self, plan, input_masks_list, N, L, device):
        """the valid-token row maps of this batch, or None: packing is opt-in (``model.pack_text_rows = True``: the caller does not read
        the pad positions of the returned probabilities — they come back as zeros), needs gradients to be wanted and the loader's host
        copy of every step's mask (``svpc_amd.keep_host_copy``; nothing is read back from the device for it), prefix-shaped masks."""
        if not getattr(self, "pack_text_rows", False) or not torch.is_grad_enabled():
            return None
        hosts = [getattr(m, "_svpc_host", None) for m in input_masks_list]
        if any(h is None or tuple(h.shape) != tuple(m.shape) for h, m in zip(hosts, input_masks_list)):
            return None
        cfg = self.config
        Lv, Lt = cfg.max_v_len, cfg.max_t_len
        key = (plan.key, tuple(id(h) for h in hosts), tuple(int(h._version) for h in hosts), str(device))
        hit = self._pack_cache.get(key)
        if hit is not None:
            return hit
        lens = []
        for b, s_ in zip(plan.step_vid.host, plan.step_idx.host):
            row = hosts[s_][b, Lv:Lv + Lt]
            n = int(row.sum())
            if n < 1 or not bool((row[:n] != 0).all()):
                return None                  # not a prefix mask (or an empty sentence): the padded layout
            lens.append(n)
        if len(self._pack_cache) >= 8:
            self._pack_cache.clear()
        pk = TextPack(plan, lens, N, Lv, Lt, L, self._n_mem(), device)
        pk.hosts = hosts                     # (keeps the ids of the key alive)
        self._pack_cache[key] = pk
        return pk

    def _spans_for(self, ingr_sep_masks):
        """[SEP]-span table of the batch.  The mask lives on the device; reading it is the one host sync of a step, so the result
        is cached for a RESIDENT batch (bench, graph replay): the entry holds a reference to the very tensor object it was read
        from — its storage therefore cannot be freed and handed to another batch — and is valid only for that object at the same
        version counter.  A fresh tensor per step (a real loader's ``.to(device)``) is a different object and is always re-read."""
        key = id(ingr_sep_masks)
        hit = self._span_cache.get(key)
        if hit is not None and hit[0] is ingr_sep_masks and hit[1] == ingr_sep_masks._version:
            return hit[2]
        if len(self._span_cache) >= 4:
            self._span_cache.clear()
        # A loader that built the mask on the host (the reference's collate does: recursive_caption_dataset.py:528-576) can leave that
        # copy on the device tensor (``svpc_amd.keep_host_copy``): the step then reads nothing back from the device.  Otherwise the
        # `.cpu()` below is a synchronous copy on the current stream — the host cannot enqueue this step before the GPU has finished
        # the previous one, which serialises an eager loop over freshly structured batches (1.8 ms of every 16 ms step, measured).
        host = getattr(ingr_sep_masks, "_svpc_host", None)
        if host is None or tuple(host.shape) != tuple(ingr_sep_masks.shape):
            host = ingr_sep_masks.cpu()
        spans = self.ingredient_embeddings.spans(host)
        if ingr_sep_masks.is_cuda:
            # the three span tables in one asynchronous copy: created lazily (Idx.dev) each is a pageable, SYNCHRONOUS host→device copy —
            # the host then waits for the GPU to drain before it can enqueue the step (3 × 0.75 ms per freshly structured batch, measured)
            up = BulkUpload(ingr_sep_masks.device)
            for ix in spans[:3]:
                up.add_idx(ix)
            up.flush()
        self._span_cache[key] = (ingr_sep_masks, ingr_sep_masks._version, spans)
        return spans

    @staticmethod
    def _stacked(tensors):
        """(S, N, ...) view over a list of per-step tensors; zero-copy when they already are consecutive
        slices of one buffer (the bench/DP loader allocates them that way), else one stack copy."""
        t0 = tensors[0]
        n = t0.numel() * t0.element_size()
        base = t0.data_ptr()
        ok = t0.is_contiguous() and all(t.is_contiguous() and t.data_ptr() == base + i * n and t.shape == t0.shape
                                        for i, t in enumerate(tensors))
        if ok and t0._base is not None and t0._base.is_contiguous():
            root = t0._base
            start = (base - root.data_ptr()) // t0.element_size()
            flat = root.reshape(-1)
            if start + len(tensors) * t0.numel() <= flat.numel():
                return flat[start:start + len(tensors) * t0.numel()].view(len(tensors), *t0.shape)
        return torch.stack(tensors)

    # ---------------------------------------------------------------- reference-shaped sub-calls
    def forward_step(self, input_ids, video_features, input_masks):
        """single step forward (reference: model.py:887-894): (B, L) ids, (B, L, F) feats → (B, Lv, D)."""
        cfg = self.config
        Lv = cfg.max_v_len
        B, L, F = video_features.shape
        dev = video_features.device
        cx = self._cx(dev)
        rows = (torch.arange(B, device=dev).unsqueeze(1) * L + torch.arange(Lv, device=dev)).reshape(-1).to(torch.int32)
        h = self._encode_clips(video_features.reshape(B * L, F), rows,
                               ops.take_rows(input_ids.reshape(-1).to(torch.int32), rows),
                               ops.take_rows(input_masks.reshape(-1).float(), rows),
                               ops.SeqInfo.uniform(B, Lv, Lv, dev), cx)
        return h.view(B, Lv, -1)

    def _encode_clips(self, feats_flat, video_rows, ids_v, key_mask_v, seq, cx, cls_only=None):
        Lv = self.config.max_v_len
        ve = self.video_embeddings
        # training forward at interior-only shapes: the clip-encoder activation stream lives in HBM as bf16
        # (bf16x3 mode: the split stream also carries the reference-shaped forward_step — the greedy decoder's encoder side)
        n_rows = video_rows.numel() if video_rows is not None else feats_flat.shape[0]       # (None: the rows are already compact)
        stream_bf16 = (cls_only is not None or ops.is_x3()) and ops.bf16_stream_ok(
            n_rows, self.config.hidden_size, self.config.video_feature_size, self.config.intermediate_size)
        h = ve.video_embeddings.run(feats_flat, cx.eps, src_rows=video_rows, drop=cx.drop(cx.p_h),
                                    add1=ve.position_embeddings_video.pe[:Lv].contiguous(), add1_mod=Lv,
                                    add2=self.token_type_embeddings.weight, add2_idx=ids_v, out_bf16=stream_bf16)
        if cls_only is not None:      # (cls_rows, one-query segmentation): last layer only for the [CLS] rows
            return self.encoder.run(h, seq, key_mask_v, cx, last_rows=cls_only[0], last_seq=cls_only[1])
        return ops.to_f32(self.encoder.run(h, seq, key_mask_v, cx))

    def _lm_probs(self, dec, bank, plan_like, cx, labels=None, proj=None, pack=None, dec_stream=None):
        """Head + pointer-generator (+ caption loss rows).  plan_like carries step_ne, row_vid, csr, row_c, c_max.  ``pack``: ``dec`` holds
        the valid tokens only (TextPack): sentence j owns rows [off_j, off_j + n_j)."""
        cfg = self.config
        lt = plan_like["lt"]
        logits = self.decoder_classifier.run(dec_stream if dec_stream is not None else dec, cx.eps)
        R = dec.shape[0]
        if labels is None:
            labels = torch.full((R,), -1, dtype=torch.int32, device=dec.device)
        if bank is None:
            return ops.ptr_mix_loss(logits, None, None, labels, plan_like["row_c"], plan_like["row_vid"],
                                    plan_like["csr_off"], plan_like["csr_ent"], plan_like["csr_id"], plan_like["csr_w"],
                                    plan_like["c_max"], self.label_smoothing)
        T, e_max, D = bank.shape
        if proj is None:
            proj = self.bank_projection(bank)
        fused = ops.ptr_attn_pgen(dec, proj, bank, plan_like["step_ne"], self.pgen_linear[0].weight, self.pgen_linear[0].bias) \
            if (lt == 1 and not torch.is_grad_enabled()) else None
        if fused is None:               # training / full-sentence form of the same fusion (differentiable)
            rows = (pack.seq_self.table[0], pack.seq_self.table[1]) if pack is not None else None
            fused = ops.ptr_attn_gate(dec, proj, bank, plan_like["step_ne"], lt, self.pgen_linear[0].weight, self.pgen_linear[0].bias,
                                      rows=rows)
        if fused is not None:           # pointer attention and generation gate in one launch
            pi, g = fused
        elif pack is not None:          # (shapes the fused kernels do not take: the unfused form over the padded layout, rows gathered back)
            dpad = ops.scatter_rows(dec, pack.full_rows64, T * lt)
            pi, att = ops.ptr_attn(dpad, proj, bank, plan_like["step_ne"], lt)
            g = ops.linear(torch.cat([dpad, att], 1), self.pgen_linear[0].weight, self.pgen_linear[0].bias, act=ACT_SIGMOID)
            pi, g = torch.index_select(pi, 0, pack.full_rows64), torch.index_select(g, 0, pack.full_rows64)
        else:
            pi, att = ops.ptr_attn(dec, proj, bank, plan_like["step_ne"], lt)
            g = ops.linear(torch.cat([dec, att], 1), self.pgen_linear[0].weight, self.pgen_linear[0].bias, act=ACT_SIGMOID)
        return ops.ptr_mix_loss(logits, g, pi, labels, plan_like["row_c"], plan_like["row_vid"], plan_like["csr_off"],
                                plan_like["csr_ent"], plan_like["csr_id"], plan_like["csr_w"], plan_like["c_max"],
                                self.label_smoothing)

    def bank_projection(self, bank):
        """Wing(E) for the pointer scores (model.py:899) — constant over decoding iterations."""
        T, e_max, D = bank.shape
        return ops.linear(bank.reshape(T * e_max, D), self.Wing.weight, self.Wing.bias).view(T, e_max, D)

    def _ptr_plan(self, ingr_dicts, c_list, lt, step_ne, row_vid, device=None):
        """CSR of (ingredient → word ids, weight 1/len) per video + per-row class counts; cached by content."""
        key = (tuple(tuple((int(e), tuple(int(i) for i in lst)) for e, lst in d.items()) for d in ingr_dicts),
               tuple(c_list), lt, id(step_ne), id(row_vid))
        pl = self._ptr_plans.get(key)
        if pl is not None:
            return pl
        off, ent, ids, w = [0], [], [], []
        for d in ingr_dicts:
            for e, lst in d.items():
                for i in lst:
                    ent.append(int(e)); ids.append(int(i)); w.append(1.0 / len(lst))
            off.append(len(ent))
        pl = dict(lt=lt, step_ne=step_ne, row_vid=row_vid, n_vid=len(c_list), csr_off=Idx(off), csr_ent=Idx(ent),
                  csr_id=Idx(ids), csr_w=FIdx(w), row_c=Idx([c_list[b] for b in row_vid.host]), c_max=max(c_list))
        if device is not None:            # the five tables in one copy (see BatchPlan)
            up = BulkUpload(device)
            for k_ in ("csr_off", "csr_ent", "csr_id", "csr_w", "row_c"):
                up.add_idx(pl[k_])
            up.flush()
        if len(self._ptr_plans) > 64:
            self._ptr_plans.clear()
        self._ptr_plans[key] = pl
        return pl

    def pointer_generator_network(self, decoder_outputs, ingr_vectors, ingr_dict, extra_zero):
        """reference: model.py:896-923.  (S, Lt, D), (S, E, D) → probabilities (S, Lt, V+X)."""
        S, Lt, D = decoder_outputs.shape
        E = ingr_vectors.shape[1]
        cx = self._cx(decoder_outputs.device)
        C = self.config.vocab_size + extra_zero
        pl = self._ptr_plan([ingr_dict], [C], Lt, Idx([E] * S), Idx([0] * (S * Lt)))
        P, _ = self._lm_probs(decoder_outputs.reshape(S * Lt, D).contiguous(), ingr_vectors.contiguous(), pl, cx)
        return P.view(S, Lt, C)

    # ---------------------------------------------------------------- the batched forward
    def forward(self, input_ids_list, video_features_list, input_masks_list, token_type_ids_list, input_labels_list,
                ingr_input_ids, ingr_masks, ingr_sep_masks, batch_step_num, ingr_id_dict, extra_zeros, alignments,
                actions, return_memory=False, predict=False):
        cfg = self.config
        mode = cfg.model_mode
        dev = video_features_list[0].device
        N, L, F = video_features_list[0].shape
        S_pad = len(input_ids_list)
        Lv, Lt, D, V = cfg.max_v_len, cfg.max_t_len, cfg.hidden_size, cfg.vocab_size
        cx = self._cx(dev)
        cx.rng.begin_step(defer=True)          # (the seed advances inside the token-staging launch below, the first of the step)

        spans = self._spans_for(ingr_sep_masks)
        ent_nums = spans[3]
        plan = self.plan_for(batch_step_num, ent_nums, S_pad, N, L, dev)
        T = plan.T

        feats = self._stacked(video_features_list).reshape(S_pad * N * L, F)
        # (zero-copy when the per-step tensors are consecutive slices of one buffer: the input pipeline and bench.py hand them so)
        # token staging in one launch: the clip rows' ids / masks, the sentence rows' ids / masks / labels, the ingredient ids
        ids_src, masks_src, labels_src = (self._stacked(l).reshape(-1) for l in (input_ids_list, input_masks_list, input_labels_list))
        dg = getattr(self, "decoder_graphs", None)
        pack = self._text_pack(plan, input_masks_list, N, L, dev) if dg is None else None
        staged = ops.gather_cast_multi([
            (ids_src, plan.video_rows, torch.int32), (masks_src, plan.video_rows, torch.float32),
            (ids_src, plan.text_rows, torch.int32), (masks_src, plan.text_rows, torch.float32),
            (labels_src, plan.text_rows, torch.int32), (ingr_input_ids, None, torch.int32)] +
            ([(ids_src, pack.src_rows, torch.int32), (labels_src, pack.src_rows, torch.int32)] if pack is not None else []), rng=cx.rng)
        ids_v, mask_v, text_ids, text_mask, labels, ingr_ids = staged[:6]

        # (1) entity initial states, compact (ΣE, D)
        ents = self.ingredient_embeddings.run(ingr_ids, spans, cx)

        # (2) clip encoder over all valid clips at once (reference loops S × forward_step, :1038-1042)
        self.split_boundary = None
        cg = getattr(self, "clip_graphs", None)
        if cg is not None and cg.usable(feats):
            # batches whose structure changes every step: the clip encoder depends on it only through T, so its forward and backward
            # replay hipGraphs captured per clip count (svpc_amd/clip_graphs.py).  Sets ``split_boundary``: the caller's backward
            # must be ``graph.backward_all``.
            cls = cg.run(feats, plan.video_rows, ids_v, mask_v, T, cx)
        else:
            cls = self._encode_clips(feats, plan.video_rows, ids_v, mask_v, plan.seq_enc, cx,
                                     cls_only=(plan.cls_rows_dev, plan.seq_enc_cls))              # (T, D): [CLS] rows only
            # Optional two-phase backward for data parallelism (svpc_amd/graph.py): the [CLS] rows are the ONLY tensor through which
            # the loss reaches the clip encoder, so cutting the autograd graph here lets the caller run the text-side backward,
            # start exchanging those gradients (74 % of the bytes), then run ``split_boundary[0].backward(split_boundary[1].grad)``.
            if getattr(self, "split_backward", False) and torch.is_grad_enabled() and cls.requires_grad:
                cut = cls.detach().requires_grad_(True)
                self.split_boundary = (cls, cut)
                cls = cut

        # (3) [CLS] rows + step PE → step-wise encoder over ragged per-video step sequences (:1062-1065)
        x = ops.span_mean(cls, plan.arange_T, plan.ones_T, add=self.step_positional_encoding.pe, add_idx=plan.step_idx)
        g = self.step_wise_encoder.run(x, plan.seq_step, None, cx)

        # (4) visual simulator, decoder memory
        sim_out = None
        if mode in ("full", "reason_copy"):
            e_p, a_p, ebar, eall, fbar = self.reasoner.run(g, ents, plan.sim, cx)
            sim_out = (e_p, a_p, ebar, eall, fbar)
            went = ops.linear(ebar, self.Went[0].weight, self.Went[0].bias, act=ACT_RELU)
            wac = ops.linear(fbar, self.Wac[0].weight, self.Wac[0].bias, act=ACT_RELU)
            mem = torch.stack([g, went, wac], 1).reshape(T * 3, D)
            bank = eall
        elif mode == "copy":
            mean_ing = ops.span_mean(ents, plan.ent_off, plan.ent_len)                       # (N, D)
            mem = torch.stack([g, ops.take_rows(mean_ing, plan.step_vid_dev)], 1).reshape(T * 2, D)
            bank = self._padded_bank(ents, plan)
        else:
            mem = g
            bank = None

        # (5) decoder over all T sentences at once (reference: per video, :1086/:925-1015)
        dec_s = None
        if pack is not None:
            # valid tokens only (TextPack): the embedding stack, the decoder layers, the head, the pointer mixture, the caption loss and
            # the Gumbel bag of words over Σ n_j rows, ragged segments; only the returned probabilities go back to the padded layout
            xt = self.text_embeddings.run(staged[6], Lt, cx, out_bf16=self.decoder.streams_bf16(pack.R, D), pos_idx=pack.pos)
            dec_s, dec = ops.stream_and_f32(self.decoder.run(xt, None, mem, pack.seq_self, pack.seq_cross, None, cx, keep_stream=True))
            labels = staged[7]
        else:
            xt = self.text_embeddings.run(text_ids, Lt, cx, out_bf16=self.decoder.streams_bf16(T * Lt, D))
            if dg is not None and dg.usable(xt, mem):     # (structure enters the decoder only through T: svpc_amd/clip_graphs.py)
                dec = dg.run(xt, text_mask, mem, T, cx)
            else:       # (the head reads the fp32 rows here, as it does behind a replayed decoder graph: the two paths stay bit-identical)
                dec = self.decoder.run(xt, text_mask, mem, plan.seq_dec_self, plan.seq_dec_cross, None, cx)

        # (6) head + pointer-generator + label-smoothed KL
        c_list = [V + (extra_zeros[b] if mode != "video" else 0) for b in range(N)]
        c_max = max(c_list)
        if mode == "video":
            labels = ops.clamp_labels(labels, V, cfg.unk_id)          # labels ≥ V → UNK (model.py:1013)
        pl = self._ptr_plan(ingr_id_dict if mode != "video" else [{}] * N, c_list, Lt, plan.step_ne,
                            plan.row_vid if pack is None else pack.row_vid, device=dev)
        row_c = pl["row_c"]
        # (the head's first projection reads the decoder's rows as the stream holds them; the pointer takes the fp32 copy)
        P, cap_rows = self._lm_probs(dec, bank, pl, cx, labels=labels, pack=pack, dec_stream=dec_s)
        # (7) simulator losses, the textual re-simulator, and the sum of all terms (one launch: ops.loss_tail)
        ent_list, act_list, mem_list = [], [], []
        if sim_out is None:
            total = ops.sum_all(cap_rows)
        else:
            align = self._pad_cat(alignments, plan.e_max)
            act_t = torch.cat(list(actions), 0)
            r_e = r_a = None
            if mode == "full":
                noise = None
                if self.gumbel_noise is not None:
                    noise = self._pad_cat([n.reshape(-1, n.shape[-1]) for n in self.gumbel_noise], c_max)
                    if pack is not None:
                        noise = torch.index_select(noise, 0, pack.full_rows64)
                bow = ops.gumbel_bow(P, row_c, self.text_embeddings.word_embeddings.weight, cfg.temperature,
                                     noise=noise, rng=cx.rng, site=cx.rng.site())
                if pack is not None:        # (every packed row is a valid token: the mask weights of the padded layout are all ones here)
                    pooled = ops.span_mean(bow, pack.off_idx, pack.len_idx)
                else:
                    pooled = ops.span_mean(bow, plan.text_starts, plan.text_lens, weights=text_mask)
                seq_vec = self._bilstm(pooled, plan)
                r_e, r_a, _, r_all, _ = self.recipe_reasoner.run(seq_vec, ents, plan.sim, cx)
            # caption + entity BCE + action ASL (rows with a detected action) + lambda·(the same two for the re-simulation)
            total = ops.loss_tail(cap_rows, e_p, a_p, r_e, r_a, align, act_t, plan.step_ne, cfg.lambda_ if mode == "full" else 0.0)

        # (8) per-video views for the reference's return contract
        if pack is not None:                # (pad positions: zeros — the caller asked for the packed run and does not read them)
            P = ops.scatter_rows(P, pack.full_rows64, T * Lt, inv=pack.inv_rows)
        prediction_scores_list = []
        for b in range(N):
            o, n = plan.h_step_off[b], plan.h_step_len[b]
            prediction_scores_list.append(P[o * Lt:(o + n) * Lt, :c_list[b]].reshape(n, Lt, c_list[b]))
            if sim_out is not None:
                E_b = plan.h_ent_len[b]
                ent_list.append(e_p[o:o + n, :E_b])
                act_list.append(a_p[o:o + n])
                if predict:
                    md = {"entity_probs": ent_list[-1], "action_probs": act_list[-1],
                          "entity_vectors": [ents[plan.h_ent_off[b]:plan.h_ent_off[b] + E_b], eall[o:o + n, :E_b]]}
                    if mode == "full":
                        md.update({"re_pred_entity_probs": r_e[o:o + n, :E_b], "re_pred_action_probs": r_a[o:o + n],
                                   "re_entity_vectors": r_all[o:o + n, :E_b]})
                    mem_list.append(md)
        if predict:
            return mem_list, ent_list, act_list
        return total, prediction_scores_list, ent_list, act_list

    # ---------------------------------------------------------------- pieces
    @staticmethod
    def _pad_cat(tensors, width):
        out = []
        for t in tensors:
            if t.shape[1] < width:
                t = torch.cat([t, t.new_zeros(t.shape[0], width - t.shape[1])], 1)
            out.append(t)
        return torch.cat(out, 0).contiguous()

    @staticmethod
    def _padded_bank(ents, plan):
        """(T, Emax, D) bank of static ingredient rows for MODEL_TYPE=vi (model.py:988)."""
        D = ents.shape[1]
        per_vid = []
        for b in range(plan.N):
            e = ents[plan.h_ent_off[b]:plan.h_ent_off[b] + plan.h_ent_len[b]]
            if plan.h_ent_len[b] < plan.e_max:
                e = torch.cat([e, e.new_zeros(plan.e_max - plan.h_ent_len[b], D)], 0)
            per_vid.append(e.unsqueeze(0).expand(plan.h_step_len[b], -1, -1))
        return torch.cat(per_vid, 0).contiguous()

    def _bilstm(self, x, plan):
        """Bidirectional LSTM over each video's step sequence, directions summed (model.py:1022-1024).
        Input projections for all steps are one GEMM per direction; the two recurrences advance in lockstep inside one fused
        autograd node (ops.bilstm_sequences: one grouped GEMM + one cell launch per time step for both directions)."""
        rnn = self.recipe_encoder
        gx, whh = [], []
        pairs = [(getattr(rnn, "bias_ih_l0" + sfx), getattr(rnn, "bias_hh_l0" + sfx)) for sfx in ("", "_reverse")]
        direct = [ops.direct_grads(bi, bh) for bi, bh in pairs]
        if all(d is not None for d in direct) and x.is_cuda:
            with torch.no_grad():       # (the sums need no autograd node: both biases of a direction receive their gradient in place)
                sums = torch._foreach_add([p[0] for p in pairs], [p[1] for p in pairs])          # one launch for both directions
        else:
            sums = [bi + bh for bi, bh in pairs]
        for z, sfx in enumerate(("", "_reverse")):
            w_ih, w_hh = getattr(rnn, "weight_ih_l0" + sfx), getattr(rnn, "weight_hh_l0" + sfx)
            # (both biases receive the column sum of the gate gradients in place: no separate column-sum launches, no autograd adds)
            gx.append(ops.linear(x, w_ih, sums[z], bgrad=direct[z]))          # (T, 4D)
            whh.append(w_hh)
        return ops.bilstm_sequences(gx[0], gx[1], whh[0], whh[1], plan.lstm_fwd_rows, plan.lstm_bwd_rows, plan.lstm_active,
                                    plan.lstm_pick[""], plan.lstm_pick["_reverse"], summed=True)

    def reconstruct(self, prediction_scores, text_mask, ga_ingr_vectors):
        """reference-shaped wrapper (model.py:1017-1025) for one video."""
        S, Lt, C = prediction_scores.shape
        dev = prediction_scores.device
        cx = self._cx(dev)
        E = ga_ingr_vectors.shape[0]
        plan = BatchPlan([S], [E], S, 1, self.config.max_v_len, Lt, self.config.max_v_len + Lt, self._n_mem(), dev)
        noise = None if self.gumbel_noise is None else self.gumbel_noise[0].reshape(S * Lt, C)
        bow = ops.gumbel_bow(prediction_scores.reshape(S * Lt, C).contiguous(), Idx([C] * (S * Lt)),
                             self.text_embeddings.word_embeddings.weight, self.config.temperature,
                             noise=noise, rng=cx.rng, site=cx.rng.site())
        pooled = ops.span_mean(bow, plan.text_starts, plan.text_lens, weights=text_mask.reshape(-1).float())
        return self.recipe_reasoner.run(self._bilstm(pooled, plan), ga_ingr_vectors.contiguous(), plan.sim, cx)


RecursiveTransformer = StateAwareRecursiveTransformer  # MART-era alias used by BASELINE.json
