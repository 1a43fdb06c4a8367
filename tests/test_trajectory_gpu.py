"""Parity of a training TRAJECTORY, not of one forward / backward: k steps of {forward, backward, global clip, BertAdam}
(src/train.py:125-147, src/rtransformer/optimization.py:284-331) on the config-1 shape (D=128, L=2, N=2, S=4, Lv=32, F=3072, V=951),
dropout off (eval mode), the reference's recorded Gumbel noise injected — the captured hipGraph step of the product in fp32 and in
the headline arithmetic (bf16x3: three-term forward products, bf16 BACKWARD) against the CPU oracle stepping with
``oracle.train_tail_step`` (pinned to the reference's own BertAdam by tests/golden/optim.npz).  What it shows: what the bf16 backward's
gradient error (≤ 0.3 % in norm, cosine ≥ 0.9999 per step) does to the loss after 20 optimizer steps.  The curves are written to
gpurun_out/trajectory_parity.json (committed as profiles/trajectory_parity.json)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import build_model  # noqa: E402
from oracle import svpc_oracle as orc  # noqa: E402
from svpc_amd import ops, synthetic as syn  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STEPS, LR, WARMUP, T_TOTAL, WD = 20, 2e-4, 0.1, 100, 0.01
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")          # src/train.py:339

# Stated bounds on |loss_gpu(k) - loss_oracle(k)| / loss_oracle(k) (≈ 2× measured on the MI355X, profiles/trajectory_parity.json).  The
# problem is a fast descent (loss 1257 → 217 in 20 steps of BertAdam, whose update m/√v turns a gradient element's sign noise into a
# full-size step), so any rounding difference grows: even fp32-on-GPU against fp32-on-CPU — arithmetic of the same precision in another
# summation order — reaches 9e-3 by step 16 after staying below 1e-6 for eight steps.  The headline mode (three-term forward, bf16
# BACKWARD: gradient norms within 0.3 %, cosines ≥ 0.9999 per step) starts at the forward's 3e-7 and sits ≈ 7× above the fp32 run
# from step 8 on; both descend to within 4 % of the oracle's final loss.  `early` bounds steps 0-6, `all` every step.
# Round 5: bounds at 1.3× what THIS build measures (profiles/trajectory_parity.json: fp32 worst 9.5e-3 / drift 0.018;
# bf16x3 worst 1.8e-2 / final 5.4e-3 / drift 0.075) — a backward that got 1.3× worse now fails.  What the bound cannot do is stay put across
# kernel changes: the trajectory is chaotic (Adam's m/√v turns rounding into full-size steps), so a mere RE-DRAW of the rounding pattern
# moves it by 2× — round 4's kernels gave worst 6.3e-2 / drift 0.155 with the same arithmetic; tests/tools/bwd_ablation.py
# (profiles/r05_bwd_ablation.json) shows the same on two seeds and two shapes: no single family of the bf16 backward carries the drift
# (exact text-side GEMMs, exact attention, fp32 stream storage each leave it where it was; only ALL of them together — a 4× slower
# step — bring the config-1 run to the fp32 run's 0.001–0.017, and at the headline shape even that nearly exact backward, step-0 gradient
# cosine 1.000000, drifts 0.17–0.24 from the fp32 mode in 20 steps).  Re-measure and re-state these numbers when the arithmetic changes.
BOUND = {"fp32": dict(early=2e-6, all=1.25e-2, final=7.7e-3, drift=2.4e-2), "bf16x3": dict(early=6.6e-4, all=2.3e-2, final=7.1e-3, drift=9.8e-2)}

_ORACLE = {}


def _oracle_curve(golden_dir):
    if "curve" in _ORACLE:
        return _ORACLE["curve"]
    z, cfg, batch, model = build_model("c1", "vivt", golden_dir, "cpu")
    noise = model.gumbel_noise
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters()]
    wd = {n: (0.0 if any(t in n for t in NO_DECAY) else WD) for n in names}
    state, losses = {}, []
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for k in range(STEPS):
        for n in names:
            P[n].requires_grad_(True)
            P[n].grad = None
        total = orc.forward(P, cfg, *syn.forward_args(batch), gumbel_noise=noise)[0]
        total.backward()
        grads = {n: P[n].grad for n in names if P[n].grad is not None}
        losses.append(float(total))
        with torch.no_grad():
            for n in names:
                P[n].requires_grad_(False)
            orc.train_tail_step({n: P[n] for n in grads}, grads, state, None, k, LR, WARMUP, T_TOTAL, grad_clip=1.0, wd=wd)
    _ORACLE["curve"] = (losses, {n: P[n].detach().clone() for n in names})
    return _ORACLE["curve"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_twenty_captured_steps_follow_the_oracle(golden_dir, precision):
    from svpc_amd.graph import GraphedTrainStep
    from svpc_amd.optim import FusedBertAdam
    ref_losses, ref_params = _oracle_curve(golden_dir)
    assert ref_losses[-1] < 0.9 * ref_losses[0] and all(b < a * 1.02 for a, b in zip(ref_losses, ref_losses[1:])), ref_losses          # the trajectory is a real descent, not a standstill
    ops.set_precision(precision)
    try:
        z, cfg, batch, model = build_model("c1", "vivt", golden_dir, DEV)       # eval mode: no dropout; recorded Gumbel noise
        opt = FusedBertAdam(list(model.named_parameters()), lr=LR, warmup=WARMUP, t_total=T_TOTAL, weight_decay=WD, grad_clip=1.0)
        fargs = syn.forward_args(batch)
        losses = []
        from svpc_amd.graph import ops_stream
        st = ops_stream()                            # ONE stream for the eager step and the capture (autograd binds a leaf's
        st.wait_stream(torch.cuda.current_stream())  # AccumulateGrad node to the stream of the backward that created it)
        with torch.cuda.stream(st):
            opt.zero_grad()
            loss = model(*fargs)[0]
            loss.backward()
            opt.step()                               # step 0 eagerly: builds the gradient arena the capture needs
            losses.append(float(loss))
            step = GraphedTrainStep(model, opt, fargs, warmup=0)
            for _ in range(STEPS - 1):
                losses.append(float(step().item()))  # the loss of the parameters BEFORE this step's update
        torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, ref_losses)]
    # parameter distance after the 20 steps, relative to how far the oracle's parameters moved
    named = dict(model.named_parameters())
    z0, _, _, m0 = build_model("c1", "vivt", golden_dir, "cpu")
    p0 = dict(m0.named_parameters())
    num = sum(float((named[n].detach().cpu().double() - ref_params[n].double()).pow(2).sum()) for n in ref_params if n in named)
    den = sum(float((ref_params[n].double() - p0[n].detach().double()).pow(2).sum()) for n in ref_params if n in named)
    rec_path = os.path.join(ROOT, "gpurun_out", "trajectory_parity.json")
    os.makedirs(os.path.dirname(rec_path), exist_ok=True)
    try:
        with open(rec_path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        rec = {}
    rec[precision] = dict(loss_gpu=losses, loss_oracle=ref_losses, loss_rel=rel, worst_rel=max(rel), first_rel=rel[0], last_rel=rel[-1],
                          param_drift_over_travel=(num / max(den, 1e-300)) ** 0.5, steps=STEPS, lr=LR, shape="config 1 (c1), vivt")
    with open(rec_path, "w") as f:
        from helpers import product_sources_sha16
        rec["_sources_sha16"] = product_sources_sha16()
        json.dump(rec, f, indent=1)
    print("%s: loss %.4f -> %.4f (oracle %.4f -> %.4f), worst rel %.2e, last rel %.2e, parameter drift / travel %.2e" %
          (precision, losses[0], losses[-1], ref_losses[0], ref_losses[-1], max(rel), rel[-1], rec[precision]["param_drift_over_travel"]))
    assert np.all(np.isfinite(losses))
    b = BOUND[precision]
    assert max(rel[:7]) <= b["early"], (precision, rel)
    assert max(rel) <= b["all"], (precision, rel)
    assert rel[-1] <= b["final"], (precision, rel)
    assert rec[precision]["param_drift_over_travel"] <= b["drift"], (precision, rec[precision]["param_drift_over_travel"])
    assert losses[-1] < 0.25 * losses[0]                    # the product's own run descends as the oracle's does
