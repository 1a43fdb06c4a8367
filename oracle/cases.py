"""Fixture case definitions shared by oracle/make_golden.py and tests/ (test infrastructure)."""

CASES = {
    # name: (config kwargs, batch kwargs)
    "tiny": (dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=4, video_feature_size=64,
                  vocab_size=50, word_vec_size=20, action_vocab_size=10, max_v_len=8, max_t_len=6, max_i_len=12),
             dict(n_videos=2, max_steps=3, step_nums=[3, 2], n_ingr=[3, 2], n_oov=[1, 0], seed=11, full_clips=False)),
    "c1": (dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, video_feature_size=3072,
                vocab_size=951, word_vec_size=300, action_vocab_size=384, max_v_len=32, max_t_len=22, max_i_len=100),
           dict(n_videos=2, max_steps=4, step_nums=[4, 3], n_ingr=[10, 7], n_oov=[2, 0], seed=2019, full_clips=False)),
    # the reference's label_smoothing == 0 branch (model.py:869-870: nn.CrossEntropyLoss(ignore_index=-1) on the probabilities);
    # generated for MODEL_TYPE v and vivt only
    "tiny_ls0": (dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=4, video_feature_size=64,
                      vocab_size=50, word_vec_size=20, action_vocab_size=10, max_v_len=8, max_t_len=6, max_i_len=12, label_smoothing=0.0),
                 dict(n_videos=2, max_steps=3, step_nums=[3, 2], n_ingr=[3, 2], n_oov=[1, 0], seed=11, full_clips=False)),
}
MODES = ("v", "vi", "viv", "vivt")
CASE_MODES = {"c1": ("v", "vivt"), "tiny_ls0": ("v", "vivt")}          # cases generated for a subset of the modes


def case_config_and_batch(case, model_type, device="cpu"):
    """Rebuild the exact config + synthetic batch a golden fixture was generated from."""
    from svpc_amd import synthetic as syn
    cfg_kw, batch_kw = CASES[case]
    cfg = syn.make_config(model_type=model_type, **cfg_kw)
    return cfg, syn.make_batch(cfg, device=device, **batch_kw)


# ---- optimizer fixture (oracle/make_golden_optim.py ↔ tests/test_optimizer_golden.py) -------------------------------------------
OPTIM_CASE = dict(steps=4, lr=1e-3, warmup=0.1, t_total=20, grad_clip=1.0, ema_decay=0.9999, weight_decay=0.01)
OPTIM_GRAD_SCALE = (1.0, 1.7, 0.6, 2e-4)    # last step: global norm < 1 → neither the global nor the per-tensor clip engages


def optim_step_gradient(base, name, t):
    """Gradient of parameter ``name`` at step ``t`` of the optimizer fixture: the tiny vivt reference gradient (``base``, a numpy
    array from tests/golden/tiny_vivt.npz) rescaled and perturbed by a name/step-seeded pattern — data both sides rebuild."""
    import zlib
    import numpy as np
    rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * t) & 0x7FFFFFFF)
    amp = float(np.abs(base).max()) if base.size else 0.0
    g = OPTIM_GRAD_SCALE[t] * (base.astype(np.float64) + 0.25 * amp * rs.standard_normal(base.shape))
    return g.astype(np.float32)
