"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A functional, per-video restatement (torch CPU fp32, no nn.Module, parameters passed as a flat
dict keyed by the reference's ``state_dict`` names) of the recurrent-transformer hot path of
awkrail/svpc.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file; the product package ``svpc_amd`` never does.

Parity status: PINNED.  The reference holds no golden vectors of its own (it has no tests), so this
restatement is pinned against outputs of the reference itself, produced in the build container by
``oracle/make_golden.py`` (which imports /root/reference with three in-process shims) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every fixture.

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

UNK_ID = 6


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps):
    """src/rtransformer/model.py:152-156 — biased variance, eps inside the square root."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return w * ((x - mu) / torch.sqrt(var + eps)) + b


def gelu_erf(x):
    """src/rtransformer/model.py:58-64 — exact erf GELU."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def sinusoid_table(max_len, d):
    """src/rtransformer/model.py:87-92 (also :109-114): sin on even columns, cos on odd."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len).float().unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def linear(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _drop(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0) else x


def mha(P, pre, q_in, kv_in, mask, cfg, training=False):
    """src/rtransformer/model.py:181-220.  mask: (B, Lq or 1, Lk) with 1 = attend."""
    H = cfg.num_attention_heads
    B, Lq, D = q_in.shape
    Lk = kv_in.shape[1]
    dh = D // H
    q = linear(P, pre + ".query", q_in).view(B, Lq, H, dh).permute(0, 2, 1, 3)
    k = linear(P, pre + ".key", kv_in).view(B, Lk, H, dh).permute(0, 2, 1, 3)
    v = linear(P, pre + ".value", kv_in).view(B, Lk, H, dh).permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2) / math.sqrt(dh) + (1.0 - mask.unsqueeze(1)) * -10000.0
    p = _drop(torch.softmax(s, dim=-1), cfg.attention_probs_dropout_prob, training)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, Lq, D)


def encoder_layer(P, pre, h, key_mask, cfg, training=False):
    """src/rtransformer/model.py:574-591 with :229-233, :262-265, :285-289 (diagonal_mask=False)."""
    eps = cfg.layer_norm_eps
    ctx = mha(P, pre + ".attention.self", h, h, key_mask.unsqueeze(1), cfg, training)
    a = _drop(linear(P, pre + ".attention.output.dense", ctx), cfg.hidden_dropout_prob, training)
    a = layer_norm(a + h, P[pre + ".attention.output.LayerNorm.weight"],
                   P[pre + ".attention.output.LayerNorm.bias"], eps)
    i = gelu_erf(linear(P, pre + ".hidden_intermediate.dense", a))
    o = _drop(linear(P, pre + ".output.dense", i), cfg.hidden_dropout_prob, training)
    return layer_norm(o + a, P[pre + ".output.LayerNorm.weight"], P[pre + ".output.LayerNorm.bias"], eps)


def encoder(P, pre, h, key_mask, cfg, training=False):
    """src/rtransformer/model.py:599-617, last layer only."""
    for i in range(cfg.num_hidden_layers):
        h = encoder_layer(P, "%s.layer.%d" % (pre, i), h, key_mask, cfg, training)
    return h


def decoder_layer(P, pre, x, dec_mask, mem, mem_mask, cfg, training=False):
    """src/rtransformer/model.py:630-663: causal∧pad self-attention, cross-attention, dense+add+LN."""
    eps = cfg.layer_norm_eps
    Lt = x.shape[1]
    self_mask = dec_mask.unsqueeze(1) * torch.tril(torch.ones(Lt, Lt))
    a = mha(P, pre + ".self_attention", x, x, self_mask, cfg, training)
    a = layer_norm(a + x, P[pre + ".norm1.weight"], P[pre + ".norm1.bias"], eps)
    c = mha(P, pre + ".dec_enc_attention", a, mem, mem_mask.unsqueeze(1), cfg, training)
    c = layer_norm(a + c, P[pre + ".norm2.weight"], P[pre + ".norm2.bias"], eps)
    o = _drop(linear(P, pre + ".output.dense", c), cfg.hidden_dropout_prob, training)
    return layer_norm(o + c, P[pre + ".output.LayerNorm.weight"], P[pre + ".output.LayerNorm.bias"], eps)


def decoder(P, x, dec_mask, mem, mem_mask, cfg, training=False):
    """src/rtransformer/model.py:672-694."""
    for i in range(cfg.num_hidden_layers):
        x = decoder_layer(P, "decoder.layer.%d" % i, x, dec_mask, mem, mem_mask, cfg, training)
    return x


def _fc_ln_relu_ln(P, pre, x, cfg, training=False):
    """The LN→Dropout→Linear→ReLU→LN stack of model.py:493-499 / :520-526 / :548-554."""
    eps = cfg.layer_norm_eps
    x = layer_norm(x, P[pre + ".0.weight"], P[pre + ".0.bias"], eps)
    x = _drop(x, cfg.hidden_dropout_prob, training)
    x = torch.relu(linear(P, pre + ".2", x))
    return layer_norm(x, P[pre + ".4.weight"], P[pre + ".4.bias"], eps)


def video_embed(P, feats, cfg, training=False):
    """src/rtransformer/model.py:558-562."""
    x = _fc_ln_relu_ln(P, "video_embeddings.video_embeddings", feats, cfg, training)
    return x + sinusoid_table(cfg.max_position_embeddings, cfg.hidden_size)[: x.shape[-2]]


def _word_table(P, pre):
    return P[pre + ".word_embeddings.weight"]


def text_embed(P, ids, cfg, training=False):
    """src/rtransformer/model.py:509-513."""
    x = _fc_ln_relu_ln(P, "text_embeddings.word_fc", _word_table(P, "text_embeddings")[ids], cfg, training)
    return x + sinusoid_table(cfg.max_position_embeddings, cfg.hidden_size)[: x.shape[-2]]


def ingredient_embed(P, ingr_ids, ingr_sep, cfg, training=False):
    """src/rtransformer/model.py:534-537 + :116-140: per-ingredient mean of the word vectors between
    consecutive [SEP]s (SEP excluded), plus the sinusoid at the ingredient index; rows past a video's
    ingredient count hold the bare positional term."""
    x = _fc_ln_relu_ln(P, "ingredient_embeddings.word_fc", _word_table(P, "ingredient_embeddings")[ingr_ids],
                       cfg, training)
    B = ingr_ids.shape[0]
    e_max = int((ingr_sep == 1).sum(1).max())
    rows = []
    for b in range(B):
        seps = (ingr_sep[b] == 1).nonzero().view(-1).tolist()
        start, vecs = 0, []
        for s_idx in seps:
            vecs.append(x[b, start:s_idx].mean(0))
            start = s_idx + 1
        vecs += [torch.zeros(x.shape[-1])] * (e_max - len(vecs))
        rows.append(torch.stack(vecs))
    out = torch.stack(rows)
    return out + sinusoid_table(cfg.max_position_embeddings, cfg.lstm_hidden_size)[:e_max].unsqueeze(0)


def forward_step(P, input_ids, feats, masks, cfg, training=False):
    """src/rtransformer/model.py:887-894.  Token-type rows are indexed by the *word ids* of the video half."""
    Lv = cfg.max_v_len
    h = video_embed(P, feats[:, :Lv], cfg, training) + P["token_type_embeddings.weight"][input_ids[:, :Lv]]
    return encoder(P, "encoder", h, masks[:, :Lv], cfg, training)


def _verb_table(P, pre):
    # after set_pretrained_embedding the table is a bare Parameter (model.py:773-775)
    return P[pre + ".action_embeddings"] if (pre + ".action_embeddings") in P else P[pre + ".action_embeddings.weight"]


def simulator(P, pre, step_vecs, ent, training=False):
    """src/rtransformer/model.py:777-823, Eqs. (1)-(7).  step_vecs (S, D), ent (E, D)."""
    S = step_vecs.shape[0]
    verbs = _verb_table(P, pre)
    prev = torch.zeros(ent.shape[0])
    ent_probs, ac_probs, bar_es, all_ents, bar_fs = [], [], [], [], []
    for t in range(S):
        v = step_vecs[t]
        hid = torch.relu(linear(P, pre + ".action_selector.0", v))
        hid = _drop(hid, 0.4, training)
        a = torch.sigmoid(linear(P, pre + ".action_selector.3", hid))
        bar_f = (a / a.sum()).unsqueeze(0) @ verbs                      # Eq. (1)
        hat_h = torch.relu(linear(P, pre + ".W1.0", v))                 # Eq. (2)
        e = torch.sigmoid(ent @ linear(P, pre + ".W2", torch.cat([hat_h, a])))
        c = torch.softmax(linear(P, pre + ".W3", hat_h), dim=-1)        # Eq. (3)
        attn = c[0] * e + c[1] * prev
        bar_e = (attn / attn.sum()).unsqueeze(0) @ ent                  # Eqs. (4), (5)
        k = torch.relu(linear(P, pre + ".W4", bar_f) @ bar_e)           # Eq. (6)
        ent = attn.unsqueeze(1) * k + (1 - attn).unsqueeze(1) * ent     # Eq. (7)
        prev = e
        ent_probs.append(e); ac_probs.append(a); bar_es.append(bar_e[0]); all_ents.append(ent); bar_fs.append(bar_f[0])
    return (torch.stack(ent_probs), torch.stack(ac_probs), torch.stack(bar_es),
            torch.stack(all_ents), torch.stack(bar_fs))


def lm_head(P, x, cfg):
    """src/rtransformer/model.py:704-709, :735-739."""
    h = gelu_erf(linear(P, "decoder_classifier.transform.dense", x))
    h = layer_norm(h, P["decoder_classifier.transform.LayerNorm.weight"],
                   P["decoder_classifier.transform.LayerNorm.bias"], cfg.layer_norm_eps)
    return F.linear(h, P["decoder_classifier.decoder.weight"]) + P["decoder_classifier.bias"]


def pointer_generator(P, dec, bank, ingr_dict, n_oov, cfg):
    """src/rtransformer/model.py:896-923.  dec (S, Lt, D); bank (S, E, D) → probabilities (S, Lt, V+X)."""
    proj = linear(P, "Wing", bank)                                       # (S, E, D)
    score = torch.einsum("sed,std->est", proj, dec)                      # (E, S, Lt)
    pi = torch.softmax(score, dim=0)
    att = torch.einsum("est,sed->std", pi, bank)
    p_gen = torch.sigmoid(linear(P, "pgen_linear.0", torch.cat([dec, att], dim=2)))   # (S, Lt, 1)
    dist = p_gen * torch.softmax(lm_head(P, dec, cfg), dim=-1)
    copy = pi * (1 - p_gen).squeeze(2)
    if n_oov > 0:
        dist = torch.cat([dist, torch.zeros(dist.shape[0], dist.shape[1], n_oov)], dim=-1)
    add = torch.zeros_like(dist)
    for e, ids in ingr_dict.items():
        for i in ids:
            add[:, :, i] = add[:, :, i] + copy[e] / len(ids)
    return dist + add


def label_smoothing_kl(probs, target, smoothing):
    """src/rtransformer/model.py:37-55 — KL(q‖p) summed; the *last* class column gets zero smoothing
    mass (one_hot[ignore_index=-1] = 0) unless it is the target; rows with target −1 are dropped."""
    keep = target != -1
    target = target[keep]
    logp = torch.log(probs[keep] + 1e-12)
    C = logp.shape[1]
    q = torch.full((C,), smoothing / (C - 1))
    q[-1] = 0
    q = q.repeat(target.shape[0], 1)
    q.scatter_(1, target.unsqueeze(1), 1.0 - smoothing)
    return F.kl_div(logp, q, reduction="sum")


def caption_criterion(cfg):
    """src/rtransformer/model.py:869-870 — LabelSmoothingLoss when config.label_smoothing > 0, else nn.CrossEntropyLoss(ignore_index=-1);
    either is applied to the PROBABILITIES (:960, :982, :1002, :1014), once per video: the cross-entropy is a mean over the video's
    non-ignored rows of log_softmax(P)[y] — a softmax of probabilities, reproduced as the reference computes it."""
    ls = cfg.label_smoothing if "label_smoothing" in cfg else 0.0
    if ls > 0:
        return lambda probs, target: label_smoothing_kl(probs, target, ls)
    return lambda probs, target: F.cross_entropy(probs, target, ignore_index=-1)


def asymmetric_loss(p, y, gamma_neg=4.0, gamma_pos=1.0, clip=0.05, eps=1e-8):
    """libs/ASL/src/loss_functions/losses.py:15-50 (probabilities in, focal weights differentiated)."""
    p_neg = (1 - p + clip).clamp(max=1)
    loss = y * torch.log(p.clamp(min=eps)) + (1 - y) * torch.log(p_neg.clamp(min=eps))
    pt = p * y + p_neg * (1 - y)
    w = torch.pow(1 - pt, gamma_pos * y + gamma_neg * (1 - y))
    return -(loss * w).sum()


def bilstm_sum(P, x, hidden):
    """nn.LSTM(W→D, bidirectional, batch 1) at model.py:865,1022-1024; directions summed. x (S, W)."""
    def run(seq, sfx):
        w_ih, w_hh = P["recipe_encoder.weight_ih_l0" + sfx], P["recipe_encoder.weight_hh_l0" + sfx]
        b = P["recipe_encoder.bias_ih_l0" + sfx] + P["recipe_encoder.bias_hh_l0" + sfx]
        h = torch.zeros(hidden); c = torch.zeros(hidden); outs = []
        for t in range(seq.shape[0]):
            g = w_ih @ seq[t] + w_hh @ h + b
            i, f, gg, o = g.chunk(4)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        return torch.stack(outs)
    fwd = run(x, "")
    bwd = run(x.flip(0), "_reverse").flip(0)
    return fwd + bwd


def gumbel_hard(logits, noise, tau):
    """F.gumbel_softmax(hard=True) (torch 2.10) with the noise handed in: straight-through one-hot."""
    y = torch.softmax((logits + noise) / tau, dim=-1)
    idx = y.max(-1, keepdim=True)[1]
    hard = torch.zeros_like(y).scatter_(-1, idx, 1.0)
    return hard - y.detach() + y


def reconstruct(P, probs, text_mask, ingr, noise, cfg, training=False):
    """src/rtransformer/model.py:1017-1025."""
    V = cfg.vocab_size
    onehot = gumbel_hard(torch.log(probs + 1e-12), noise, cfg.temperature)[:, :, :V]
    bow = onehot @ _word_table(P, "text_embeddings")
    pooled = torch.stack([bow[s][text_mask[s] == 1].mean(0) for s in range(bow.shape[0])])
    seq = bilstm_sum(P, pooled, cfg.hidden_size)
    return simulator(P, "recipe_reasoner", seq, ingr, training)


# --------------------------------------------------------------------------------------------
# the forward pass
# --------------------------------------------------------------------------------------------
def forward(P, cfg, input_ids_list, video_features_list, input_masks_list, token_type_ids_list,
            input_labels_list, ingr_input_ids, ingr_masks, ingr_sep_masks, batch_step_num, ingr_id_dict,
            extra_zeros, alignments, actions, gumbel_noise=None, training=False, return_parts=False):
    """src/rtransformer/model.py:1027-1189.  Returns (total, probs_list, ent_list, act_list[, parts])."""
    mode = cfg.model_mode
    Lv = cfg.max_v_len
    ingr_all = ingredient_embed(P, ingr_input_ids, ingr_sep_masks, cfg, training)
    enc = [forward_step(P, input_ids_list[s], video_features_list[s], input_masks_list[s], cfg, training)
           for s in range(len(input_ids_list))]
    pe50 = sinusoid_table(50, cfg.hidden_size)
    cap = ent_l = act_l = re_ent_l = re_act_l = 0.0
    probs_list, ent_list, act_list, extras = [], [], [], []
    for b, S_b in enumerate(batch_step_num):
        cls = torch.stack([enc[s][b, 0] for s in range(S_b)])                        # :1062-1064
        g = encoder(P, "step_wise_encoder", (cls + pe50[:S_b]).unsqueeze(0), torch.ones(1, S_b), cfg, training)[0]
        n_e = int((ingr_sep_masks[b] == 1).sum())
        ingr = ingr_all[b, :n_e]
        ids = torch.stack([input_ids_list[s][b, Lv:] for s in range(S_b)])
        tmask = torch.stack([input_masks_list[s][b, Lv:] for s in range(S_b)])
        labels = torch.stack([input_labels_list[s][b, Lv:] for s in range(S_b)])
        if mode in ("full", "reason_copy"):
            e_p, a_p, bar_e, all_e, bar_f = simulator(P, "reasoner", g, ingr, training)
            mem = torch.stack([g, torch.relu(linear(P, "Went.0", bar_e)), torch.relu(linear(P, "Wac.0", bar_f))], 1)
            bank = all_e
        elif mode == "copy":
            mem = torch.stack([g, ingr.mean(0).unsqueeze(0).expand(S_b, -1)], 1)      # :986-991
            bank = ingr.unsqueeze(0).expand(S_b, -1, -1)
        else:
            mem = g.unsqueeze(1)                                                      # :1006
        dec = decoder(P, text_embed(P, ids, cfg, training), tmask, mem, torch.ones(mem.shape[:2]), cfg, training)
        if mode == "video":
            probs = torch.softmax(lm_head(P, dec, cfg), dim=-1)
            labels = labels.clone()
            labels[labels >= cfg.vocab_size] = cfg.unk_id                             # :1013
        else:
            probs = pointer_generator(P, dec, bank, ingr_id_dict[b], extra_zeros[b], cfg)
        cap = cap + caption_criterion(cfg)(probs.reshape(-1, probs.shape[-1]), labels.reshape(-1))
        probs_list.append(probs)
        if mode in ("full", "reason_copy"):
            any_act = (actions[b] == 1).any(dim=1)
            ent_l = ent_l + F.binary_cross_entropy(e_p, alignments[b], reduction="sum")
            act_l = act_l + asymmetric_loss(a_p[any_act], actions[b][any_act])
            ent_list.append(e_p); act_list.append(a_p)
            extra = dict(step_vecs=g, dec=dec, bar_e=bar_e, all_e=all_e, bar_f=bar_f)
            if mode == "full":
                noise = gumbel_noise[b] if gumbel_noise is not None else \
                    -torch.empty_like(probs).exponential_().log()
                r_e, r_a, _, r_all, _ = reconstruct(P, probs, tmask, ingr, noise, cfg, training)
                re_ent_l = re_ent_l + F.binary_cross_entropy(r_e, alignments[b], reduction="sum")
                re_act_l = re_act_l + asymmetric_loss(r_a[any_act], actions[b][any_act])
                extra.update(re_ent=r_e, re_act=r_a, re_all=r_all)
            extras.append(extra)
        else:
            extras.append(dict(step_vecs=g, dec=dec))
    total = cap + ent_l + act_l + cfg.lambda_ * (re_ent_l + re_act_l)                 # :1188
    if return_parts:
        parts = dict(caption=cap, entity=ent_l, action=act_l, re_entity=re_ent_l, re_action=re_act_l,
                     enc=enc, ingr=ingr_all, extras=extras)
        return total, probs_list, ent_list, act_list, parts
    return total, probs_list, ent_list, act_list


# --------------------------------------------------------------------------------------------
# greedy decode
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def greedy_decode(P, cfg, input_ids_list, video_features_list, input_masks_list, ingr_input_ids,
                  ingr_sep_masks, batch_step_num, ingr_id_dict, oov_word_dict, bos=4, unk=UNK_ID):
    """src/translator.py:45-192 — per video, max_t_len full decoder re-runs, argmax at position i with
    the UNK column suppressed; the emitted stream keeps extended (≥V) ids, the model side sees UNK."""
    mode, Lv, Lt = cfg.model_mode, cfg.max_v_len, cfg.max_t_len
    ingr_input_ids = torch.as_tensor(ingr_input_ids)
    ingr_sep_masks = torch.as_tensor(ingr_sep_masks)
    pe50 = sinusoid_table(50, cfg.hidden_size)
    out = []
    for b, S_b in enumerate(batch_step_num):
        ids = torch.stack([input_ids_list[s][b] for s in range(S_b)]).clone()
        masks = torch.stack([input_masks_list[s][b] for s in range(S_b)]).clone()
        feats = torch.stack([video_features_list[s][b] for s in range(S_b)])
        ids[:, Lv:] = 0; masks[:, Lv:] = 0                                             # :205-228
        ingr = ingredient_embed(P, ingr_input_ids[b:b + 1], ingr_sep_masks[b:b + 1], cfg)[0]
        enc = forward_step(P, ids, feats, masks, cfg)
        g = encoder(P, "step_wise_encoder", (enc[:, 0] + pe50[:S_b]).unsqueeze(0), torch.ones(1, S_b), cfg)[0]
        n_oov = len(oov_word_dict[b])
        if mode in ("full", "reason_copy"):
            _, _, bar_e, all_e, bar_f = simulator(P, "reasoner", g, ingr)
            mem = torch.stack([g, torch.relu(linear(P, "Went.0", bar_e)), torch.relu(linear(P, "Wac.0", bar_f))], 1)
            bank = all_e
        elif mode == "copy":
            mem = torch.stack([g, ingr.mean(0).unsqueeze(0).expand(S_b, -1)], 1)
            bank = ingr.unsqueeze(0).expand(S_b, -1, -1)
        else:
            mem = g.unsqueeze(1)
        text = ids[:, Lv:].clone(); ext = text.clone(); tmask = masks[:, Lv:].clone()
        nxt = torch.full((S_b,), bos, dtype=torch.long); nxt_ext = nxt.clone()
        for i in range(Lt):
            text[:, i] = nxt; ext[:, i] = nxt_ext; tmask[:, i] = 1
            dec = decoder(P, text_embed(P, text, cfg), tmask, mem, torch.ones(mem.shape[:2]), cfg)
            if mode == "video":
                scores = lm_head(P, dec, cfg)
            else:
                scores = pointer_generator(P, dec, bank, ingr_id_dict[b], n_oov, cfg)
            scores[:, :, unk] = -1e10
            nxt_ext = scores[:, i].max(1)[1]
            nxt = nxt_ext.clone()
            if mode != "video":
                nxt[nxt_ext >= scores.shape[-1] - n_oov] = unk
        out.append(text if mode == "video" else ext)
    return out


# --------------------------------------------------------------------------------------------
# optimizer step used by the cpu_baseline leg (BertAdam, no bias correction)
# --------------------------------------------------------------------------------------------
def bert_adam_step(params, grads, state, lr, step_frac_lr=1.0, b1=0.9, b2=0.999, eps=1e-6, wd=None):
    """src/rtransformer/optimization.py:284-331: per-tensor clip to 1.0, m/v update, decoupled decay."""
    for n, p in params.items():
        g = grads.get(n)
        if g is None:
            continue
        g = g * (1.0 / (g.norm() + 1e-6)).clamp(max=1.0)   # clip_grad_norm_(p, 1.0), :306-307
        m, v = state.setdefault(n, (torch.zeros_like(p), torch.zeros_like(p)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        upd = m / (v.sqrt() + eps)
        if wd and wd.get(n, 0.0) > 0:
            upd = upd + wd[n] * p
        p.sub_(lr * step_frac_lr * upd)


def warmup_linear(progress, warmup):
    """src/rtransformer/optimization.py:162-171 (WarmupLinearSchedule.get_lr_)."""
    if progress < warmup:
        return progress / warmup
    return max((progress - 1.0) / (warmup - 1.0), 0.0)


def global_clip_coef(grads, max_norm=1.0):
    """nn.utils.clip_grad_norm_(model.parameters(), max_norm) as src/train.py:141-142 calls it → (total norm, scale factor)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def ema_update(shadow, params, step, decay):
    """src/rtransformer/optimization.py:196-203: shadow ← (1−d)·θ + d·shadow with d = min(decay, (1+step)/(10+step))."""
    d = min(decay, (1.0 + step) / (10.0 + step))
    for n, p in params.items():
        shadow[n] = (1.0 - d) * p + d * shadow[n]


def train_tail_step(params, grads, state, shadow, step, lr, warmup, t_total, grad_clip=1.0, ema_decay=-1.0, wd=None):
    """One training-step tail in the reference's order (src/train.py:140-147): global clip → BertAdam.step → EMA."""
    _, coef = global_clip_coef(grads, grad_clip) if grad_clip and grad_clip > 0 else (None, 1.0)
    sched = warmup_linear(step / t_total, warmup) if t_total > 0 else 1.0
    bert_adam_step(params, {n: g * coef for n, g in grads.items()}, state, lr, step_frac_lr=sched, wd=wd)
    if ema_decay >= 0:
        ema_update(shadow, params, step, ema_decay)
