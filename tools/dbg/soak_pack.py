"""60 eager training steps with a different sentence-length pattern every step through the packed sentence side (model.pack_text_rows):
loss finite, allocator flat, the pack cache bounded.  python tools/dbg/soak_pack.py   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from svpc_amd import StateAwareRecursiveTransformer, make_config, make_batch, keep_host_copy, ops, synthetic as syn
from svpc_amd.graph import backward_all, ops_stream
from svpc_amd.optim import FusedBertAdam
DEV = torch.device("cuda", 0)
cfg = make_config(model_type="vivt", hidden_size=256, num_hidden_layers=2, num_attention_heads=4)
torch.manual_seed(1)
model = StateAwareRecursiveTransformer(cfg)
g = torch.Generator().manual_seed(2)
glove = 0.4 * torch.randn(cfg.vocab_size, cfg.word_vec_size, generator=g)
verb = 0.4 * torch.randn(cfg.action_vocab_size, cfg.word_vec_size, generator=g)
model.ingredient_embeddings.set_pretrained_embedding(glove.clone(), freeze=False)
model.text_embeddings.set_pretrained_embedding(glove.clone(), freeze=False)
model.reasoner.set_pretrained_embedding(verb.clone(), freeze=False)
model.recipe_reasoner.set_pretrained_embedding(verb.clone(), freeze=False)
model = model.to(DEV).train()
model.pack_text_rows = True
ops.set_precision("bf16x3")
opt = FusedBertAdam(list(model.named_parameters()), lr=1e-4, warmup=0.1, t_total=10000, grad_clip=1.0)
peak0 = None
with torch.cuda.stream(ops_stream()):
    for step in range(60):
        b = make_batch(cfg, n_videos=4, max_steps=6, n_ingr=5, n_oov=1, seed=100 + step, full_clips=False)
        bd = {}
        for k, v in b.items():
            if isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
                bd[k] = [t.to(DEV) for t in v]
            elif isinstance(v, torch.Tensor):
                bd[k] = v.to(DEV)
            else:
                bd[k] = v
        for m, h in zip(bd["input_masks_list"], b["input_masks_list"]):
            keep_host_copy(m, h)
        keep_host_copy(bd["ingr_sep_masks"], b["ingr_sep_masks"])
        opt.zero_grad()
        loss = model(*syn.forward_args(bd))[0]
        backward_all(model, loss)
        opt.step()
        if step == 9:
            torch.cuda.synchronize(); peak0 = torch.cuda.memory_allocated()
        assert len(model._pack_cache) <= 8
    torch.cuda.synchronize()
    lv = float(loss)
    assert lv == lv and abs(lv) < 1e9, lv
    now = torch.cuda.memory_allocated()
    pk = next(iter(model._pack_cache.values()))
    print("final loss %.3f; packed rows of the last batch %d of %d; allocated %.1f MB at step 10, %.1f MB at step 60; pack cache %d entries"
          % (lv, pk.R, len(pk.lens) * cfg.max_t_len, peak0 / 2**20, now / 2**20, len(model._pack_cache)))
    assert now <= peak0 * 1.2 + (64 << 20)
