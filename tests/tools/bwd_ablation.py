"""Backward ablation of the bf16x3 mode (test infrastructure — run on the MI355X: `python tests/tools/bwd_ablation.py`).

The headline arithmetic computes every FORWARD contraction as a three-term split-bf16 product and runs the bf16 mode's BACKWARD.  Over a
20-step trajectory its parameters drift 8× further from the oracle's than the fp32 mode's do (profiles/trajectory_parity.json: drift /
travel 0.155 vs 0.019).  Which part of the backward is that?  One family at a time is switched to exact arithmetic through the switches of
svpc_amd.ops (all off in the product):

  stream storage   ops.BF16_STREAM = False — activations AND gradients of the clip encoder / decoder in fp32 storage instead of bf16 planes
                   (their backward contractions still round the operands to bf16 when the MFMA fragments are built)
  text-side GEMMs  ops.BWD_EXACT = True — every dgrad / wgrad on fp32 storage on the exact f32 MFMA (ungrouped)
  attention        ops.ATTN_BWD_EXACT = True — attention on fp32 storage takes the exact fp32 backward kernel

Reference: the fp32 mode on the same device (it follows the CPU oracle to 2e-6 over the first steps; tests/test_trajectory_gpu.py).  Per row:
cosine / relative norm error of the whole step-0 gradient against the fp32 mode's, worst loss deviation over 20 steps of {forward, backward,
global clip, BertAdam}, parameter drift / travel after them, ms per eager step.  Two weight seeds, the config-1 shape and the headline shape.
→ gpurun_out/bwd_ablation.json (committed as profiles/r05_bwd_ablation.json).  Reference for the step: src/train.py:125-147,
src/rtransformer/optimization.py:284-331."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from svpc_amd import ops, synthetic as syn  # noqa: E402
from svpc_amd.graph import backward_all, ops_stream  # noqa: E402
from svpc_amd.optim import FusedBertAdam  # noqa: E402

DEV = "cuda:0"
STEPS, LR, WARMUP, T_TOTAL, WD = 20, 2e-4, 0.1, 100, 0.01

CONFIGS = [
    ("fp32", dict(precision="fp32")),
    ("bf16x3 (the product)", dict(precision="bf16x3")),
    ("bf16x3 + exact text-side GEMM backward", dict(precision="bf16x3", exact=True)),
    ("bf16x3 + exact fp32-storage attention backward", dict(precision="bf16x3", attn=True)),
    ("bf16x3, fp32 stream storage", dict(precision="bf16x3", stream=False)),
    ("bf16x3, fp32 stream storage + exact GEMM backward", dict(precision="bf16x3", stream=False, exact=True)),
    ("bf16x3, fp32 stream storage + exact GEMM and attention backward", dict(precision="bf16x3", stream=False, exact=True, attn=True)),
    ("bf16 (one-term forward too)", dict(precision="bf16")),
]


def build(shape, seed):
    import bench
    if shape == "c1":
        from oracle.cases import CASES
        cfg_kw, batch_kw = CASES["c1"]
        cfg = syn.make_config(model_type="vivt", **cfg_kw)
        batch = syn.make_batch(cfg, device="cpu", **batch_kw)
        from svpc_amd import StateAwareRecursiveTransformer
        model = StateAwareRecursiveTransformer(cfg)
        V, W, A = cfg.vocab_size, cfg.word_vec_size, cfg.action_vocab_size
        for m_, n_ in ((model.ingredient_embeddings, V), (model.text_embeddings, V), (model.reasoner, A), (model.recipe_reasoner, A)):
            m_.set_pretrained_embedding(torch.zeros(n_, W), freeze=False)
        n_vid, steps_v = batch_kw["n_videos"], batch_kw["step_nums"]
    else:
        args = bench.parse_args([])
        cfg, model = bench.build(args, "cpu", model_type="vivt")
        batch = syn.make_batch(cfg, n_videos=16, max_steps=12, n_ingr=10, n_oov=0, seed=2019, full_clips=True)
        n_vid, steps_v = 16, [12] * 16
    drawn = syn.draw_parameters(list(model.named_parameters()), seed=seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(drawn[n])
    model.eval()
    g = torch.Generator().manual_seed(99 + seed)
    noise = [-torch.empty(s_, cfg.max_t_len, cfg.vocab_size + x_).exponential_(generator=g).log()
             for s_, x_ in zip(steps_v, batch["extra_zeros"])]
    return cfg, model, batch, noise


def to_dev(batch):
    out = {}
    for k, v in batch.items():
        if isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            out[k] = [t.to(DEV) for t in v]
        elif isinstance(v, torch.Tensor):
            out[k] = v.to(DEV)
        else:
            out[k] = v
    return out


def run(cfg_model, conf):
    import copy
    cfg, model_cpu, batch, noise = cfg_model
    ops.set_precision(conf["precision"])
    ops.BF16_STREAM = conf.get("stream", True)
    ops.BWD_EXACT = conf.get("exact", False)
    ops.ATTN_BWD_EXACT = conf.get("attn", False)
    try:
        model = copy.deepcopy(model_cpu).to(DEV)
        model.eval()
        model.gumbel_noise = [n.to(DEV) for n in noise]
        fargs = syn.forward_args(to_dev(batch))
        opt = FusedBertAdam(list(model.named_parameters()), lr=LR, warmup=WARMUP, t_total=T_TOTAL, weight_decay=WD, grad_clip=1.0)
        losses, g0 = [], None
        with torch.cuda.stream(ops_stream()):
            t_steps = []
            for k in range(STEPS):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                opt.zero_grad()
                loss = model(*fargs)[0]
                backward_all(model, loss)
                ops.join_side()
                if k == 0:
                    torch.cuda.synchronize()
                    g0 = {n: p.grad.detach().double().cpu().reshape(-1) for n, p in model.named_parameters() if p.grad is not None}
                opt.step()
                torch.cuda.synchronize()
                t_steps.append(time.perf_counter() - t0)
                losses.append(float(loss))
        params = {n: p.detach().double().cpu().reshape(-1) for n, p in model.named_parameters()}
        return dict(losses=losses, g0=g0, params=params, ms=1000.0 * sorted(t_steps[2:])[len(t_steps[2:]) // 2])
    finally:
        ops.set_precision("fp32")
        ops.BF16_STREAM, ops.BWD_EXACT, ops.ATTN_BWD_EXACT = True, False, False


def main():
    shapes = sys.argv[1:] or ["c1", "headline"]
    report = {"steps": STEPS, "lr": LR, "reference": "the fp32 mode on the same device", "rows": []}
    for shape in shapes:
        for seed in (7, 8):
            cm = build(shape, seed)
            p0 = {n: p.detach().double().reshape(-1) for n, p in cm[1].named_parameters()}
            ref = None
            for name, conf in CONFIGS:
                r = run(cm, conf)
                if ref is None:
                    ref = r
                names = [n for n in ref["g0"] if n in r["g0"]]
                ga = torch.cat([r["g0"][n] for n in names]); gb = torch.cat([ref["g0"][n] for n in names])
                cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
                worst = sorted(((float(torch.dot(r["g0"][n], ref["g0"][n]) / (r["g0"][n].norm() * ref["g0"][n].norm() + 1e-300)), n) for n in names
                                if float(ref["g0"][n].norm()) > 0))[:3]
                rel = [abs(a - b) / abs(b) for a, b in zip(r["losses"], ref["losses"])]
                num = sum(float((r["params"][n] - ref["params"][n]).pow(2).sum()) for n in ref["params"])
                den = sum(float((ref["params"][n] - p0[n]).pow(2).sum()) for n in ref["params"])
                row = dict(shape=shape, seed=seed, config=name, grad_cosine_step0=cos, grad_norm_rel_step0=abs(float(ga.norm() / gb.norm()) - 1.0),
                           worst_tensor_cosines_step0=[(round(c, 6), n) for c, n in worst], loss_rel_worst=max(rel), loss_rel_first7=max(rel[:7]),
                           loss_rel_last=rel[-1], drift_over_travel=(num / max(den, 1e-300)) ** 0.5, ms_per_eager_step=r["ms"],
                           loss_first=r["losses"][0], loss_last=r["losses"][-1])
                report["rows"].append(row)
                print("%-9s seed %d  %-66s cos %.6f  |g| %.2e  loss worst %.2e  drift/travel %.3f  %.1f ms" %
                      (shape, seed, name, cos, row["grad_norm_rel_step0"], row["loss_rel_worst"], row["drift_over_travel"], r["ms"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bwd_ablation.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
