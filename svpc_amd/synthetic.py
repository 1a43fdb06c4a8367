"""Synthetic YouCook2-shaped batches for the recurrent-transformer hot path.

Layouts follow what the reference's data layer hands to ``model.forward``
(reference: src/rtransformer/recursive_caption_dataset.py:289-330 clip_sentence_to_feature,
:389-416 _load_indexed_video_feature, :528-576 caption_collate; src/train.py:91-112 marshalling).
Nothing here reads the reference or any test code; it only produces tensors of the same shape,
dtype and token conventions (PAD=0 CLS=1 SEP=2 VID=3 BOS=4 EOS=5 UNK=6, IGNORE=-1).
"""
from __future__ import annotations

import numpy as np
import torch

PAD, CLS, SEP, VID, BOS, EOS, UNK = 0, 1, 2, 3, 4, 5, 6
IGNORE = -1
N_SPECIAL = 7


class ModelConfig(dict):
    """dict with attribute access; supports ``"key" in cfg`` like the reference's EasyDict
    (reference: src/train.py:657-686 builds it, src/rtransformer/model.py:870 uses ``in``)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        if d:
            self.update(d)
        self.update(kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


MODEL_MODES = {"vivt": "full", "viv": "reason_copy", "vi": "copy", "v": "video"}


def make_config(model_type="vivt", hidden_size=768, num_hidden_layers=6, num_attention_heads=12,
                video_feature_size=3072, vocab_size=951, word_vec_size=300, action_vocab_size=384,
                max_v_len=100, max_t_len=22, max_i_len=100, temperature=0.5, lambda_=0.5,
                hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, label_smoothing=0.1,
                **extra):
    """The field list of src/train.py:657-686 with scripts/train.sh:19-21 defaults."""
    cfg = ModelConfig(
        xl_grad=False,
        hidden_size=hidden_size,
        intermediate_size=hidden_size,
        vocab_size=vocab_size,
        word_vec_size=word_vec_size,
        action_vocab_size=action_vocab_size,
        lstm_hidden_size=hidden_size,
        video_feature_size=video_feature_size,
        max_position_embeddings=max_v_len + max_t_len,
        max_v_len=max_v_len,
        max_t_len=max_t_len,
        max_i_len=max_i_len,
        use_asl="asl",
        model_mode=MODEL_MODES.get(model_type, model_type),
        temperature=temperature,
        lambda_=lambda_,
        type_vocab_size=2,
        unk_id=UNK,
        layer_norm_eps=1e-12,
        hidden_dropout_prob=hidden_dropout_prob,
        num_hidden_layers=num_hidden_layers,
        num_attention_heads=num_attention_heads,
        attention_probs_dropout_prob=attention_probs_dropout_prob,
        n_memory_cells=1,
        memory_dropout_prob=0.1,
        initializer_range=0.02,
        label_smoothing=label_smoothing,
        share_wd_cls_weight=False,
    )
    cfg.update(extra)
    return cfg


def _as_list(v, n):
    if isinstance(v, (list, tuple)):
        assert len(v) == n
        return list(v)
    return [int(v)] * n


def make_batch(cfg, n_videos, max_steps, step_nums=None, n_ingr=10, n_oov=0, seed=2019,
               full_clips=True, p_align=0.15, p_action=None, device="cpu", feature_bank=None):
    """Build the 13 positional arguments of ``StateAwareRecursiveTransformer.forward``.

    Returns a dict with keys named after the forward signature (model.py:1027-1030) plus
    ``oov_word_dict`` (a list of dicts, only their length is used by the hot path) and
    ``ingr_input_ids_list`` etc. in the list form ``Translator.translate_batch`` takes.
    """
    rng = np.random.RandomState(seed)
    N, S = n_videos, max_steps
    Lv, Lt, Li = cfg.max_v_len, cfg.max_t_len, cfg.max_i_len
    F, V, A = cfg.video_feature_size, cfg.vocab_size, cfg.action_vocab_size
    L = Lv + Lt
    step_nums = _as_list(step_nums if step_nums is not None else S, N)
    assert max(step_nums) == S, "the collate pads to the longest video, so one must have max_steps"
    n_ingr = _as_list(n_ingr, N)
    n_oov = _as_list(n_oov, N)
    if p_action is None:
        p_action = 0.01 if A >= 100 else 0.15

    # ---- ingredients (dataset.py: ingredient ids + [SEP] after each ingredient, padded to Li) ----
    ingr_ids = np.zeros((N, Li), np.int64)
    ingr_sep = np.zeros((N, Li), np.int64)
    ingr_mask = np.zeros((N, Li), np.int64)
    ingr_id_dict, oov_word_dict = [], []
    for b in range(N):
        pos, d, oov = 0, {}, {}
        oov_slots = set(rng.choice(n_ingr[b], size=min(n_oov[b], n_ingr[b]), replace=False).tolist())
        for e in range(n_ingr[b]):
            n_w = int(rng.randint(1, 3))
            words = rng.randint(N_SPECIAL, V, size=n_w).tolist()
            ext = list(words)
            if e in oov_slots:  # first word of this ingredient is out-of-vocabulary
                words[0] = UNK
                ext[0] = V + len(oov)
                oov["oov_%d_%d" % (b, len(oov))] = ext[0]
            assert pos + n_w + 1 <= Li, "max_i_len too small for the requested ingredients"
            ingr_ids[b, pos:pos + n_w] = words
            ingr_ids[b, pos + n_w] = SEP
            ingr_sep[b, pos + n_w] = 1
            ingr_mask[b, pos:pos + n_w + 1] = 1
            pos += n_w + 1
            d[e] = ext
        ingr_id_dict.append(d)
        oov_word_dict.append(oov)
    extra_zeros = [len(o) for o in oov_word_dict]

    # ---- per-step clip + sentence tensors ----
    ids = np.zeros((S, N, L), np.int64)
    masks = np.zeros((S, N, L), np.float32)
    labels = np.full((S, N, L), IGNORE, np.int64)
    tt = np.zeros((S, N, L), np.int64)
    tt[:, :, Lv:] = 1
    # feature_bank: a pre-drawn (≥ S, N, L, F) array of frame features reused for every batch (rows 1..Lv-2 of every clip are U[0,1) draws;
    # full_clips only) — drawing 77 M uniforms per batch is what makes a hundred differently STRUCTURED batches slow to build
    bank = feature_bank is not None and full_clips
    feats = feature_bank[:S] if bank else np.zeros((S, N, L, F), np.float32)
    for s in range(S):
        for b in range(N):
            valid = Lv - 2 if full_clips else int(rng.randint(max(1, (Lv - 2) // 2), Lv - 1))
            ids[s, b, 0] = CLS
            ids[s, b, 1:1 + valid] = VID
            ids[s, b, 1 + valid] = SEP
            masks[s, b, :valid + 2] = 1
            if not bank:
                feats[s, b, 1:1 + valid] = rng.rand(valid, F).astype(np.float32)
            n_w = int(rng.randint(min(5, Lt - 2), min(20, Lt - 2) + 1)) if Lt > 4 else Lt - 2
            words = rng.randint(N_SPECIAL, V, size=n_w)
            ext = words.copy()
            if extra_zeros[b] > 0 and n_w > 1 and rng.rand() < 0.5:  # a copied OOV word in the target
                j = int(rng.randint(0, n_w))
                words[j] = UNK
                ext[j] = V + int(rng.randint(0, extra_zeros[b]))
            text = np.concatenate([[BOS], words, [EOS]])
            ext_text = np.concatenate([[BOS], ext, [EOS]])
            ids[s, b, Lv:Lv + len(text)] = text
            masks[s, b, Lv:Lv + len(text)] = 1
            if s < step_nums[b]:  # padded steps keep IGNORE everywhere (caption_collate :559-560)
                labels[s, b, Lv:Lv + len(text) - 1] = ext_text[1:]

    alignments = [torch.from_numpy((rng.rand(step_nums[b], n_ingr[b]) < p_align).astype(np.float32))
                  for b in range(N)]
    actions = []
    for b in range(N):
        a = (rng.rand(step_nums[b], A) < p_action).astype(np.float32)
        a[0, int(rng.randint(0, A))] = 1.0  # at least one row with a detected action
        actions.append(torch.from_numpy(a))

    t = lambda x: torch.from_numpy(x).to(device)
    batch = dict(
        input_ids_list=[t(ids[s]) for s in range(S)],
        video_features_list=[t(feats[s]) for s in range(S)],
        input_masks_list=[t(masks[s]) for s in range(S)],
        token_type_ids_list=[t(tt[s]) for s in range(S)],
        input_labels_list=[t(labels[s]) for s in range(S)],
        ingr_input_ids=t(ingr_ids),
        ingr_masks=t(ingr_mask),
        ingr_sep_masks=t(ingr_sep),
        batch_step_num=list(step_nums),
        ingr_id_dict=ingr_id_dict,
        extra_zeros=extra_zeros,
        alignments=[a.to(device) for a in alignments],
        actions=[a.to(device) for a in actions],
        oov_word_dict=oov_word_dict,
    )
    return batch


FORWARD_ARG_ORDER = ("input_ids_list", "video_features_list", "input_masks_list", "token_type_ids_list",
                     "input_labels_list", "ingr_input_ids", "ingr_masks", "ingr_sep_masks", "batch_step_num",
                     "ingr_id_dict", "extra_zeros", "alignments", "actions")


def forward_args(batch):
    """The 13 positional arguments in the order of model.py:1027-1030."""
    return [batch[k] for k in FORWARD_ARG_ORDER]


def translate_inputs(batch):
    """The 12-element list ``Translator.translate_batch`` unpacks (translator.py:196-198).
    Tensors are cloned because prepare_video_only_inputs overwrites ids/masks in place (:205-228)."""
    return [[x.clone() for x in batch["input_ids_list"]],
            batch["video_features_list"],
            [x.clone() for x in batch["input_masks_list"]],
            batch["token_type_ids_list"],
            *(batch["_ingr_host_lists"] if "_ingr_host_lists" in batch else      # (the reference's collate hands these three as host lists,
              (batch["ingr_input_ids"].tolist(), batch["ingr_masks"].tolist(),  #  translator.py:181-183; a caller holding them on the device
               batch["ingr_sep_masks"].tolist())),                              #  pays three synchronous read-backs per call here)
            batch["ingr_id_dict"],
            batch["oov_word_dict"],
            batch["alignments"],
            batch["actions"],
            batch["batch_step_num"]]


def draw_parameters(named_params, seed=7):
    """Deterministic, test-sensitive parameter values drawn in iteration order from a CPU generator:
    LayerNorm gains 1+0.1n, biases 0.05n, matrices n/sqrt(fan_in)."""
    import zlib
    g = torch.Generator(device="cpu")
    out = {}
    for name, p in named_params:
        shape = tuple(p.shape)
        g.manual_seed(seed * 1000003 + zlib.crc32(name.encode()))  # per-name stream: order-independent
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        if len(shape) == 1 and name.endswith("weight"):
            v = 1.0 + 0.1 * n
        elif len(shape) == 1:
            v = 0.05 * n
        elif "embeddings" in name and "fc" not in name and "video" not in name and "token_type" not in name:
            v = 0.4 * n  # GloVe-like tables (build_vocab.py:79 draws N(0, 0.4^2) for missing words)
        else:
            v = n / (shape[-1] ** 0.5)
        out[name] = v
    return out
