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
# Round 5.  The FREE-RUNNING curve is chaotic: Adam's m/√v turns rounding into full-size steps, so a mere RE-DRAW of
# the rounding pattern moves it by 2× — three builds of this round with the SAME arithmetic (only kernels fused differently) measured
# worst loss deviation 1.8e-2 / 5.6e-2 / 6.3e-2 and drift / travel 0.075 / 0.133 / 0.155 in bf16x3; tests/tools/bwd_ablation.py
# (profiles/r05_bwd_ablation.json) shows on two seeds and two shapes that no single family of the bf16 backward carries the drift and that
# at the headline shape even a nearly exact backward (step-0 gradient cosine 1.000000) drifts 0.17–0.24 from the fp32 mode in 20 steps.
# A bound at 1.3× one build's draw therefore fails on the next benign change (tried: it did, within three commits).  So the curve keeps
# bounds that hold over the observed re-draws, and the REGRESSION gate of the backward is the teacher-forced test below
# (test_backward_along_the_oracle_trajectory): the product's gradient at the ORACLE's parameters of every step against the oracle's
# gradient there — no amplification, errors average over millions of elements, bounds at 1.3× measured.
BOUND = {"fp32": dict(early=2e-6, all=2e-2, final=1.5e-2, drift=4e-2), "bf16x3": dict(early=1e-3, all=1.3e-1, final=8e-2, drift=3.2e-1)}
# 1 − cosine and relative norm error of the WHOLE gradient (all tensors, flattened) at each of the 20 oracle states; measured on the MI355X
# (profiles/trajectory_parity.json "teacher_forced"): fp32 5.2e-12 / 8.3e-7, bf16x3 1.33e-5 / 6.8e-4 → bounds at 1.3× (fp32: a floor an
# fp32 summation re-order stays under)
TF_BOUND = {"fp32": dict(one_minus_cos=1e-10, norm=2e-6), "bf16x3": dict(one_minus_cos=1.75e-5, norm=8.9e-4)}

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
    _ORACLE["states"] = []            # (parameters, gradients) of every step: the teacher-forced test replays the product at these points
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for k in range(STEPS):
        for n in names:
            P[n].requires_grad_(True)
            P[n].grad = None
        total = orc.forward(P, cfg, *syn.forward_args(batch), gumbel_noise=noise)[0]
        total.backward()
        grads = {n: P[n].grad for n in names if P[n].grad is not None}
        _ORACLE["states"].append(({n: P[n].detach().clone() for n in names}, {n: g.detach().clone() for n, g in grads.items()}))
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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_backward_along_the_oracle_trajectory(golden_dir, precision):
    """Teacher-forced: at the ORACLE's parameters of each of its 20 steps the product's whole gradient (every tensor, flattened) against the
    oracle's gradient at that point — the backward's accuracy along a real descent, without the chaotic amplification of a free run."""
    _oracle_curve(golden_dir)
    states = _ORACLE["states"]
    ops.set_precision(precision)
    worst_c = worst_n = 0.0
    per_step = []
    try:
        z, cfg, batch, model = build_model("c1", "vivt", golden_dir, DEV)
        fargs = syn.forward_args(batch)
        named = dict(model.named_parameters())
        for k, (params, grads) in enumerate(states):
            with torch.no_grad():
                for n, p in named.items():
                    p.copy_(params[n].to(DEV))
                    p.grad = None
            loss = model(*fargs)[0]
            loss.backward()
            ops.join_side()
            torch.cuda.synchronize()
            names = [n for n in grads if named[n].grad is not None]
            assert len(names) == len(grads)
            ga = torch.cat([named[n].grad.detach().double().cpu().reshape(-1) for n in names])
            gb = torch.cat([grads[n].double().reshape(-1) for n in names])
            omc = 1.0 - float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
            nr = abs(float(ga.norm() / gb.norm()) - 1.0)
            per_step.append((omc, nr))
            worst_c, worst_n = max(worst_c, omc), max(worst_n, nr)
    finally:
        ops.set_precision("fp32")
    rec_path = os.path.join(ROOT, "gpurun_out", "trajectory_parity.json")
    try:
        with open(rec_path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        rec = {}
    rec.setdefault("teacher_forced", {})[precision] = dict(one_minus_cos_worst=worst_c, norm_rel_worst=worst_n, per_step=per_step, steps=len(states))
    with open(rec_path, "w") as f:
        json.dump(rec, f, indent=1)
    print("%s teacher-forced: worst 1 - cos %.3e, worst |g| rel %.3e" % (precision, worst_c, worst_n))
    b = TF_BOUND[precision]
    assert worst_c <= b["one_minus_cos"], (precision, worst_c, per_step)
    assert worst_n <= b["norm"], (precision, worst_n, per_step)
