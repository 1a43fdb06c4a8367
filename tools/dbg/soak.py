"""soak: 600 ragged training steps with the per-clip-count graphs (24 structures, LRU of 8 entries: constant eviction / recapture) followed
by 40 greedy decodes on the trained weights with ONE Translator; every loss finite, memory flat, decode ids reproducible"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from svpc_amd import ops, synthetic as syn, clip_graphs, make_batch
from svpc_amd.optim import FusedBertAdam
from svpc_amd.graph import backward_all
from svpc_amd.translator import Translator
args = bench.parse_args([])
dev = torch.device("cuda:0")
ops.set_precision("bf16x3")
cfg, model = bench.build(args, dev, model_type="vivt")
model.train()
rb, clips = bench.ragged_batches(cfg, args, dev, 24)
rargs = [syn.forward_args(b) for b in rb]
opt = FusedBertAdam(list(model.named_parameters()), lr=1e-4, warmup=0.1, t_total=100000, weight_decay=0.01, grad_clip=1.0)
st = torch.cuda.Stream()
def step(k):
    model._plans.clear(); model._ptr_plans.clear(); model._span_cache.clear()
    opt.zero_grad(); loss = model(*rargs[k % len(rargs)])[0]; backward_all(model, loss); opt.step(); return loss
with torch.cuda.stream(st):
    for k in range(3): step(k)
    cg, dg = clip_graphs.enable(model)
    cg.max_entries = dg.max_entries = 8
    torch.cuda.synchronize(); m0 = torch.cuda.memory_reserved()
    t0 = time.time(); losses = []
    for k in range(600):
        l = step(k)
        if k % 50 == 49:
            losses.append(float(l)); print("step %d loss %.2f reserved %.1f GB captures %d hits %d" % (k + 1, losses[-1], torch.cuda.memory_reserved() / 2**30, cg.stats["captures"], cg.stats["hits"]), flush=True)
    torch.cuda.synchronize()
    print("600 ragged steps in %.1f s; distinct clip counts %d" % (time.time() - t0, len(set(clips))))
    assert all(l == l and abs(l) < 1e8 for l in losses)
    clip_graphs.enable(model, False)
    model.eval()
    b = make_batch(cfg, n_videos=64, max_steps=args.clips, n_ingr=10, n_oov=0, seed=2019, full_clips=True)
    b["_ingr_host_lists"] = (b["ingr_input_ids"].tolist(), b["ingr_masks"].tolist(), b["ingr_sep_masks"].tolist())
    for k_, v in list(b.items()):
        if isinstance(v, list) and v and isinstance(v[0], torch.Tensor): b[k_] = [t.to(dev) for t in v]
        elif isinstance(v, torch.Tensor): b[k_] = v.to(dev)
    tr = Translator(type("O", (), {"cuda": True})(), {"model_cfg": cfg, "model": model.state_dict()}, model=model, graph=True)
    first = None
    for i in range(40):
        out, _ = tr.translate_batch(syn.translate_inputs(b))
        cat = torch.cat([o.reshape(-1) for o in out])
        if first is None: first = cat.clone()
        assert torch.equal(cat, first), i
    torch.cuda.synchronize()
    print("40 decodes reproducible; reserved %.1f GB" % (torch.cuda.memory_reserved() / 2**30))
print("SOAK OK")
