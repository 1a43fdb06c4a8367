"""Pure-torch statement of every primitive in ``svpc_amd.ops`` — TEST INFRASTRUCTURE ONLY.

Two uses: (1) ``-m "not gpu"`` tests swap it in for ``svpc_amd.ops`` to check the host-side
orchestration of the batched model against the oracle on CPU; (2) ``-m gpu`` tests use it (on cuda
tensors) as the plain-PyTorch fp32 reference each HIP kernel is compared with, forward and backward
(autograd differentiates these definitions).  The product never imports this file.

Signatures are identical to svpc_amd/ops.py.  Dropout: ``drop=(p, rng, site)``; here the mask comes from
``rng.mask(site, n, p, device)`` so that GPU tests can hand in the very mask the kernels generate.
"""
import math

import torch
import torch.nn.functional as F

from svpc_amd.ops_common import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, FIdx, Idx, SeqInfo,  # noqa: F401
                                 as_idx)


def _h(v):
    """host list of an index argument (Idx / FIdx / list / tensor)."""
    if isinstance(v, (Idx, FIdx)):
        return v.host
    if isinstance(v, torch.Tensor):
        return v.tolist()
    return list(v)


def require_split_tag(t, who):
    """(the product refuses an untagged bf16 tensor in bf16x3 mode; the emulation has no split domain)"""
    return None


def _apply_drop(x, drop):
    if drop is None or drop[0] <= 0.0:
        return x
    p, rng, site = drop
    m = rng.mask(site, x.numel(), p, x.device).view(x.shape)
    return x * m * (1.0 / (1.0 - p))


def _act(z, act):
    if act == ACT_RELU:
        return torch.relu(z)
    if act == ACT_GELU:
        return z * 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0)))
    if act == ACT_SIGMOID:
        return torch.sigmoid(z)
    return z


def _shadow(w):
    return None


def direct_grads(*params):
    return None


def linear(x, w, b=None, act=ACT_NONE, trans_w=False, drop=None, wgrad=None, bgrad=None, w16=None, fuse_act_bwd=False):
    z = x @ (w if trans_w else w.t())
    if b is not None:
        z = z + b
    return _apply_drop(_act(z, act), drop)


def layernorm(x, gamma, beta, eps, residual=None, src_rows=None, pad_row=-1, pre_drop=None, post_drop=None,
              add1=None, add1_mod=0, add2=None, add2_idx=None, out_bf16=False, sink=False):
    h = x if src_rows is None else x[src_rows.long()]
    h = _apply_drop(h, pre_drop)
    if residual is not None:
        h = h + residual
    mu = h.mean(-1, keepdim=True)
    var = ((h - mu) ** 2).mean(-1, keepdim=True)
    y = gamma * ((h - mu) / torch.sqrt(var + eps)) + beta
    y = _apply_drop(y, post_drop)
    R = y.shape[0]
    if add1 is not None:
        y = y + add1[torch.arange(R, device=y.device) % add1_mod]
    if add2 is not None:
        y = y + add2[add2_idx.long()]
    return y


class _ZeroPadRowGrad(torch.autograd.Function):
    """nn.Embedding(padding_idx=k): row k of the table never receives a gradient."""

    @staticmethod
    def forward(ctx, table, row):
        ctx.row = row
        return table.view_as(table)

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        g[ctx.row] = 0
        return g, None


def embedding_table(table, pad_row):
    return _ZeroPadRowGrad.apply(table, pad_row) if pad_row >= 0 else table


def attention(qt, kvt, cols, D, n_heads, seq, key_mask=None, causal=False, drop=None):
    """qt (Rq, ≥D) holds the queries in columns [cols[0], cols[0]+D); kvt (Rk, ·) holds keys at cols[1] and
    values at cols[2] (packed projection outputs are consumed in place; qt may be kvt)."""
    q = qt[:, cols[0]:cols[0] + D]
    k = kvt[:, cols[1]:cols[1] + D]
    v = kvt[:, cols[2]:cols[2] + D]
    dh = D // n_heads
    out = []
    for i in range(seq.n):
        qo, ql, ko, kl = seq.h_q_off[i], seq.h_q_len[i], seq.h_k_off[i], seq.h_k_len[i]
        qi = q[qo:qo + ql].reshape(ql, n_heads, dh).permute(1, 0, 2)
        ki = k[ko:ko + kl].reshape(kl, n_heads, dh).permute(1, 0, 2)
        vi = v[ko:ko + kl].reshape(kl, n_heads, dh).permute(1, 0, 2)
        m = torch.ones(ql, kl, device=q.device)
        if key_mask is not None:
            m = m * key_mask[ko:ko + kl].unsqueeze(0)
        if causal:
            m = m * torch.tril(torch.ones(ql, kl, device=q.device))
        s = qi @ ki.transpose(-1, -2) / math.sqrt(dh) + (1.0 - m) * -10000.0
        p = torch.softmax(s, dim=-1)
        if drop is not None and drop[0] > 0:
            pd, rng, site = drop
            # the kernels' own draw for element (row (seq*H + head)*max_q + i, key j)
            full = rng.attn_mask(site, seq.n * n_heads * seq.max_q, seq.max_k, pd, q.device)
            full = full.view(seq.n, n_heads, seq.max_q, seq.max_k)[i, :, :ql, :kl]
            p = p * full * (1.0 / (1.0 - pd))
        out.append((p @ vi).permute(1, 0, 2).reshape(ql, D))
    return torch.cat(out, 0)


def attn_q1_ln(q, kv, k_stride, n_keys, residual, gamma, beta, eps, n_heads, new_kv=None):
    return None          # (the callers then take the unfused path: copy / attention / layernorm, emulated above)


def span_mean(x, starts, lens, weights=None, add=None, add_idx=None):
    outs = []
    starts, lens = _h(starts), _h(lens)
    for g in range(len(starts)):
        s, l = int(starts[g]), int(lens[g])
        rows = x[s:s + l]
        if weights is None:
            outs.append(rows.mean(0))
        else:
            w = weights[s:s + l].unsqueeze(1)
            outs.append((rows * w).sum(0) / w.sum())
    out = torch.stack(outs)
    if add is not None:
        out = out + add[torch.as_tensor(_h(add_idx), device=out.device).long()]
    return out


def row_normalize(a):
    return a / a.sum(-1, keepdim=True)


def softmax_rows(x):
    return torch.softmax(x, dim=-1)


def sim_recur(q, c, w4f, E0, step_off, step_len, ent_off, ent_len, e_max):
    """Batched recurrent part of the simulator (Eqs. (2)-(7) after the per-step projections)."""
    T, D = q.shape
    step_off, step_len, ent_off, ent_len = _h(step_off), _h(step_len), _h(ent_off), _h(ent_len)
    ebar_out, eall_out = [], []
    e_rows = []
    for b in range(len(step_off)):
        E = E0[ent_off[b]:ent_off[b] + ent_len[b]]
        prev = torch.zeros(ent_len[b], device=q.device)
        for t in range(step_len[b]):
            j = step_off[b] + t
            e = torch.sigmoid(E @ q[j])
            alpha = c[j, 0] * e + c[j, 1] * prev
            ebar = (alpha / alpha.sum()) @ E
            k = torch.relu(w4f[j] * ebar)
            E = alpha.unsqueeze(1) * k.unsqueeze(0) + (1 - alpha).unsqueeze(1) * E
            prev = e
            e_rows.append(F.pad(e, (0, e_max - ent_len[b])))
            ebar_out.append(ebar)
            eall_out.append(F.pad(E, (0, 0, 0, e_max - ent_len[b])))
    return torch.stack(e_rows), torch.stack(ebar_out), torch.stack(eall_out)


def ptr_attn(dec, proj, bank, step_ne, lt):
    """dec (T*lt, D); proj, bank (T, Emax, D); step_ne: python list of entity counts per step."""
    T, e_max, D = bank.shape
    d3 = dec.view(T, lt, D)
    score = torch.einsum("jed,jtd->jte", proj, d3)
    valid = torch.arange(e_max, device=dec.device).unsqueeze(0) < torch.as_tensor(_h(step_ne), device=dec.device).unsqueeze(1)
    score = score.masked_fill(~valid.unsqueeze(1), float("-inf"))
    pi = torch.softmax(score, dim=-1)
    att = torch.einsum("jte,jed->jtd", pi, bank)
    return pi.reshape(T * lt, e_max), att.reshape(T * lt, D)


def sim_heads(hh, fb, W3, b3, W4, b4):
    """c = softmax(hh·W3ᵀ + b3), w = fb·W4ᵀ + b4 (model.py:801, :804-805)"""
    return torch.softmax(hh @ W3.t() + b3, -1), (fb @ W4.t() + b4).reshape(-1)


def stream_and_f32(t):
    return t, t.float()


def scatter_rows(src, idx, n_rows, inv=None):
    out = torch.zeros(n_rows, src.shape[1], dtype=src.dtype, device=src.device)
    return out.index_copy(0, idx, src)


def take_rows_f32_alias(t, idx):
    return take_rows_f32(t, idx), t


def cross_attn_ln_usable(*a, **k):
    return False          # (the fused decoder cross-attention is a GPU kernel pair: the emulated model takes the unfused path)


def ptr_attn_gate(dec, proj, bank, step_ne, lt, w, b, rows=None):
    """pointer attention + generation gate (model.py:899-908): (pi, sigmoid([dec ; att]·wᵀ + b))"""
    if rows is not None:                     # ragged sentences: through the padded layout
        off, ln = rows[0].tolist(), rows[1].tolist()
        idx = torch.tensor([j * lt + t for j, n in enumerate(ln) for t in range(n)], dtype=torch.long, device=dec.device)
        dpad = torch.zeros(len(ln) * lt, dec.shape[1], dtype=dec.dtype, device=dec.device).index_copy(0, idx, dec)
        pi, g = ptr_attn_gate(dpad, proj, bank, step_ne, lt, w, b)
        return pi[idx], g[idx]
    pi, att = ptr_attn(dec, proj, bank, step_ne, lt)
    return pi, torch.sigmoid(torch.cat([dec, att], 1) @ w.t() + b)


def ptr_attn_pgen(dec, proj, bank, step_ne, w, b):
    return None          # (decoding-iteration fusion: GPU only; the callers fall back to ptr_attn + linear)


def ptr_mix_loss(logits, g, pi, labels, row_c, row_vid, csr_off, csr_ent, csr_id, csr_w, c_max, smoothing,
                 dp_ext_hook=None):
    """Rows r: P = g*softmax(logits) over V columns, zeros up to C_r, copy mass (1-g)*pi[e]*w scattered to
    csr_id; label-smoothed KL per row (0 for label −1).  g/pi None → plain softmax (MODEL_TYPE=v).
    Returns (P (R, c_max), loss_rows (R,))."""
    R, V = logits.shape
    row_c, row_vid, csr_off, csr_ent, csr_id, csr_w = map(_h, (row_c, row_vid, csr_off, csr_ent, csr_id, csr_w))
    sm = torch.softmax(logits, dim=-1)
    P = torch.zeros(R, c_max, device=logits.device)
    if g is None:
        P = torch.cat([sm, P[:, V:]], 1)
    else:
        P = torch.cat([g * sm, P[:, V:]], 1)
        add = torch.zeros_like(P)
        for b in range(len(csr_off) - 1):
            rows = (torch.as_tensor(row_vid, device=logits.device) == b).nonzero().view(-1)
            for n in range(csr_off[b], csr_off[b + 1]):
                col = torch.zeros(c_max, device=logits.device)
                col[csr_id[n]] = 1.0
                contrib = ((1 - g[rows, 0]) * pi[rows, csr_ent[n]] * csr_w[n]).unsqueeze(1) * col.unsqueeze(0)
                add = add.index_add(0, rows, contrib)
        P = P + add
    losses = []
    if not smoothing > 0:
        # the reference's label_smoothing == 0 branch: nn.CrossEntropyLoss(ignore_index=-1) applied to the probabilities, a MEAN per video
        n_valid = {}
        for r in range(R):
            if int(labels[r]) != -1:
                n_valid[int(row_vid[r])] = n_valid.get(int(row_vid[r]), 0) + 1
    for r in range(R):
        y = int(labels[r])
        if y == -1:
            losses.append(P[r].sum() * 0.0)
            continue
        C = int(row_c[r])
        if not smoothing > 0:
            losses.append((torch.logsumexp(P[r, :C], 0) - P[r, y]) / n_valid[int(row_vid[r])])
            continue
        qv = torch.full((C,), smoothing / (C - 1), device=logits.device)
        qv[C - 1] = 0
        qv[y] = 1.0 - smoothing
        logp = torch.log(P[r, :C] + 1e-12)
        losses.append(F.kl_div(logp, qv, reduction="sum"))
    return P, torch.stack(losses)


def gumbel_bow(P, row_c, emb, tau, noise=None, rng=None, site=0):
    """Straight-through Gumbel-softmax over each row's C_r columns of log(P+1e-12), sliced to V, times emb."""
    R, c_max = P.shape
    V = emb.shape[0]
    assert noise is not None, "the emulation needs explicit noise"
    cols = torch.arange(c_max, device=P.device).unsqueeze(0)
    valid = cols < torch.as_tensor(_h(row_c), device=P.device).unsqueeze(1)
    logits = (torch.log(P + 1e-12) + noise) / tau
    logits = logits.masked_fill(~valid, float("-inf"))
    y = torch.softmax(logits, dim=-1)
    idx = y.max(-1, keepdim=True)[1]
    hard = torch.zeros_like(y).scatter_(-1, idx, 1.0)
    st = hard - y.detach() + y
    return st[:, :V] @ emb


def lstm_cell(gx, gh, c_prev, h_prev, active):
    """gates = gx + gh (i, f, g, o); rows with active == 0 pass (h, c) through."""
    gates = gx + gh
    i, f, g, o = gates.chunk(4, dim=1)
    c = torch.sigmoid(f) * c_prev + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    a = active.unsqueeze(1)
    return a * h + (1 - a) * h_prev, a * c + (1 - a) * c_prev


def lstm_sequence(gx_all, w_hh, rows_t, active_t, pick):
    """One LSTM direction over every video's step sequence (time-major state, inactive steps pass through)."""
    N, D = rows_t[0].numel(), w_hh.shape[1]
    h = gx_all.new_zeros(N, D)
    c = gx_all.new_zeros(N, D)
    hs = []
    for rows, act in zip(rows_t, active_t):
        h, c = lstm_cell(gx_all[rows.long()], h @ w_hh.t(), c, h, act)
        hs.append(h)
    return torch.stack(hs, 0).reshape(-1, D)[pick.long()]


def bilstm_sequences(gx_f, gx_b, w_f, w_b, rows_f, rows_b, active_t, pick_f, pick_b, summed=False):
    of, ob = lstm_sequence(gx_f, w_f, rows_f, active_t, pick_f), lstm_sequence(gx_b, w_b, rows_b, active_t, pick_b)
    return of + ob if summed else (of, ob)


def branch_stream(device):
    return None


def bce_rows(p, y, widths):
    """Per-row sum of binary cross-entropy over the first widths[r] columns (nn.BCELoss clamps log at -100)."""
    cols = torch.arange(p.shape[1], device=p.device).unsqueeze(0)
    valid = (cols < torch.as_tensor(_h(widths), device=p.device).unsqueeze(1)).float()
    pp = torch.where(valid > 0, p, torch.full_like(p, 0.5))
    l = torch.nn.functional.binary_cross_entropy(pp, y, reduction="none")      # (torch's own backward: (p - y) / max(p(1-p), 1e-12))
    return (l * valid).sum(1)


def asl_rows(p, y, row_active, gamma_neg=4.0, gamma_pos=1.0, clip=0.05, eps=1e-8):
    p_neg = (1 - p + clip).clamp(max=1)
    loss = y * torch.log(p.clamp(min=eps)) + (1 - y) * torch.log(p_neg.clamp(min=eps))
    pt = p * y + p_neg * (1 - y)
    w = torch.pow(1 - pt, gamma_pos * y + gamma_neg * (1 - y))
    return -(loss * w).sum(1) * row_active


def sum_all(x):
    return x.sum()


def loss_tail(cap_rows, e_p, a_p, r_e, r_a, align, act, widths, lam, gamma_neg=4.0, gamma_pos=1.0, clip=0.05, eps=1e-8):
    on = row_any_eq1(act)
    total = cap_rows.sum()
    if e_p is not None:
        total = total + bce_rows(e_p, align, widths).sum()
    if a_p is not None:
        total = total + asl_rows(a_p, act, on, gamma_neg, gamma_pos, clip, eps).sum()
    re = 0.0
    if r_e is not None:
        re = re + bce_rows(r_e, align, widths).sum()
    if r_a is not None:
        re = re + asl_rows(r_a, act, on, gamma_neg, gamma_pos, clip, eps).sum()
    return total + lam * re


def take_rows(x, idx):
    return x[idx.long()]


def split_cols(wide, n):
    w = wide.shape[1] // n
    return tuple(wide[:, i * w:(i + 1) * w] for i in range(n))


def add(a, b):
    return a + b


def gather_cast_multi(items, rng=None):
    return [(src.reshape(-1)[idx.long()] if idx is not None else src.reshape(-1)).to(dt) for src, idx, dt in items]


def clamp_labels(labels, vocab, unk):
    return torch.where(labels >= vocab, torch.full_like(labels, unk), labels)


def row_any_eq1(x):
    return (x == 1).any(dim=1).float()


def make_rng(device):
    return EmulRng(device=device)


_default = None


def default_rng(device):
    global _default
    if _default is None:
        _default = EmulRng()
    return _default


class EmulRng:
    """CPU stand-in for svpc_amd.ops.Rng (eval-mode tests never draw from it)."""

    def __init__(self, seed=0, device="cpu"):
        self.seed = seed
        self._site = 0
        self.device = torch.device(device)

    def begin_step(self, defer=False):
        self._site = 0

    def site(self):
        self._site += 1
        return self._site

    def mask(self, site, n, p, device):
        g = torch.Generator().manual_seed(self.seed * 7919 + site)
        return (torch.rand(n, generator=g) >= p).float().to(device)

    def attn_mask(self, site, n_rows, max_k, p, device):
        return self.mask(site, n_rows * max_k, p, device).view(n_rows, max_k)


def greedy_pick(scores, row_c, row_x, lt, pos, unk, append=None):
    row_c, row_x = _h(row_c), _h(row_x)
    n = scores.shape[0] // lt
    ext, mod = [], []
    for j in range(n):
        r = j * lt + pos
        sc = scores[r, :row_c[r]].clone()
        sc[unk] = -1e10
        i = int(sc.max(0)[1])
        ext.append(i)
        mod.append(unk if i >= row_c[r] - row_x[r] else i)
    t = lambda v: torch.tensor(v, dtype=torch.int32, device=scores.device)
    if append is not None:
        append[0][:, append[2]] = t(mod)
        append[1][:, append[2]] = t(ext)
    return t(ext), t(mod)


def bf16_stream_ok(rows, *dims):
    return False


# bf16x3-mode surface of svpc_amd.ops (split tensors do not exist in the torch statement: everything is fp32)
def is_x3():
    return False


def lo_off(t):
    return None


def to_f32(t):
    return t.float()


def take_rows_f32(t, idx):
    return torch.index_select(t, 0, idx.long() if idx.dtype != torch.int64 else idx).float()


def to_split(t):
    return t.float()
