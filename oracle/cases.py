"""Fixture case definitions shared by oracle/make_golden.py and tests/ (test infrastructure)."""

CASES = {
    # name: (config kwargs, batch kwargs)
    "tiny": (dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=4, video_feature_size=64,
                  vocab_size=50, word_vec_size=20, action_vocab_size=10, max_v_len=8, max_t_len=6, max_i_len=12),
             dict(n_videos=2, max_steps=3, step_nums=[3, 2], n_ingr=[3, 2], n_oov=[1, 0], seed=11, full_clips=False)),
    "c1": (dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, video_feature_size=3072,
                vocab_size=951, word_vec_size=300, action_vocab_size=384, max_v_len=32, max_t_len=22, max_i_len=100),
           dict(n_videos=2, max_steps=4, step_nums=[4, 3], n_ingr=[10, 7], n_oov=[2, 0], seed=2019, full_clips=False)),
}
MODES = ("v", "vi", "viv", "vivt")


def case_config_and_batch(case, model_type, device="cpu"):
    """Rebuild the exact config + synthetic batch a golden fixture was generated from."""
    from svpc_amd import synthetic as syn
    cfg_kw, batch_kw = CASES[case]
    cfg = syn.make_config(model_type=model_type, **cfg_kw)
    return cfg, syn.make_batch(cfg, device=device, **batch_kw)
