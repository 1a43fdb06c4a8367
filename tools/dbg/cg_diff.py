"""which parameter gradients differ between the eager step and the step with the clip / decoder graphs (tests/test_clip_graphs_gpu.py's
bucketed case)?  python tools/dbg/cg_diff.py   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import build_model
from svpc_amd import synthetic as syn, keep_host_copy, ops, clip_graphs
from svpc_amd.graph import backward_all, ops_stream
from svpc_amd.optim import FusedBertAdam
DEV = "cuda:0"
z, cfg, _, model = build_model("c1", "vivt", os.path.join(ROOT, "tests", "golden"), DEV)
structs = [dict(S=[5, 3, 7], E=[3, 1, 10], X=[0, 0, 1], seed=61), dict(S=[4, 6, 3], E=[2, 25, 8], X=[1, 2, 0], seed=62)]
batches = []
for st in structs:
    b_cpu = syn.make_batch(cfg, n_videos=len(st["S"]), max_steps=max(st["S"]), step_nums=st["S"], n_ingr=st["E"], n_oov=st["X"], seed=st["seed"], full_clips=False)
    gn = torch.Generator().manual_seed(100 + st["seed"])
    noise = [-torch.empty(s_, cfg.max_t_len, cfg.vocab_size + x).exponential_(generator=gn).log().to(DEV) for s_, x in zip(st["S"], st["X"])]
    b = {kk: ([t.to(DEV) if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list) else (v.to(DEV) if isinstance(v, torch.Tensor) else v)) for kk, v in b_cpu.items()}
    keep_host_copy(b["ingr_sep_masks"], b_cpu["ingr_sep_masks"])
    batches.append((syn.forward_args(b), noise))
ops.set_precision("bf16x3")
opt = FusedBertAdam(list(model.named_parameters()), lr=0.0, warmup=0.1, t_total=1000, weight_decay=0.0, grad_clip=1.0)
def run(k):
    args, noise = batches[k % 2]
    model._plans.clear(); model._ptr_plans.clear(); model._span_cache.clear()
    model.gumbel_noise = noise
    opt.zero_grad()
    tot = model(*args)[0]
    backward_all(model, tot)
    ops.join_side(); torch.cuda.synchronize()
    arena = opt.ensure_built()
    return float(tot.detach()), {n: p.grad.detach().clone() for n, p in zip(arena.names, arena.params)}
with torch.cuda.stream(ops_stream()):
    run(0)
    eager = [run(k) for k in range(2)]
    clip_graphs.enable(model)
    graphed = [run(k) for k in range(4)]
out = os.environ.get("CG_DIFF_SAVE")
if out:
    torch.save({"eager": [(l, {n: t.cpu() for n, t in d.items()}) for l, d in eager], "graphed": [(l, {n: t.cpu() for n, t in d.items()}) for l, d in graphed]}, out)
ref = os.environ.get("CG_DIFF_REF")
if ref:
    R = torch.load(ref)
    for tag, mine in (("eager", eager), ("graphed", graphed)):
        for k, (l, d) in enumerate(mine):
            rl, rd = R[tag][k]
            worst = sorted(((float((d[n].cpu() - rd[n]).abs().max()), float(rd[n].abs().max()), n) for n in d), reverse=True)[:3]
            print("vs ref", tag, k, "loss", l, rl, " ".join("%s %.2e/%.2e" % (n[-40:], a, b) for a, b, n in worst))
for k in range(4):
    e, g = eager[k % 2], graphed[k]
    print("k", k, "loss", e[0], g[0])
    worst = sorted(((float((e[1][n] - g[1][n]).abs().max()), float(e[1][n].abs().max()), n) for n in e[1]), reverse=True)[:6]
    for d, m, n in worst:
        print("   %-60s diff %.3e  max %.3e" % (n, d, m))
