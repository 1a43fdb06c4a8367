"""Parity of THE THING THAT IS TIMED: the headline configuration of bench.py (MODEL_TYPE=vivt, N=16 videos × S=12 clips × Lv=100
frames × F=3072, Lt=22, D=768, H=12, L=6, V=951, A=384, E=10 — BASELINE.json configs 2-4) against the CPU oracle
(oracle/svpc_oracle.py, pinned to the reference by tests/golden), on the same seeded inputs, in ALL THREE arithmetic modes:

* ``fp32``   (f32-MFMA parity mode)  — north_star's bar: loss ≤ 1e-4 relative.
* ``bf16x3`` (≤1e-4-parity THROUGHPUT mode, what bench.py's ``value`` is measured in) — every forward contraction a three-term
  split-bf16 product, the clip-encoder stream stored as two bf16 planes; held to the SAME loss bar (≤ 1e-4) as fp32; its backward is
  the bf16 mode's, so its gradients are held to the bf16 entry's bounds.
* ``bf16``   (fastest mode, reported beside it with its measured loss error) — bf16 GEMM/attention operands and bf16 activation
  streams; the tolerance stated below is ≤ 2× what that arithmetic measures at this size.
The Gumbel hard arg-max of the re-simulation (model.py:1018) is a discontinuity of the loss: the test counts the positions where
the GPU's probabilities and the oracle's pick a different word under the same injected noise and reports them (``gumbel_flips``):
a flip moves one bag-of-words row, which is what the residual gradient differences of the re-simulator's tensors in fp32 mode are.

Dropout is off (eval mode) and the Gumbel noise is injected, so both sides are deterministic functions of the same inputs; the
gradients are taken twice on the GPU: through autograd's own accumulation (first backward) and through the optimizer's gradient
arena (direct in-place writes, grouped weight gradients, residual-gradient hand-over) — the path the captured training step runs.
Also runs the bf16 mode once for MODEL_TYPE=vi and viv (configs 2 and 3).
reference: src/rtransformer/model.py:1027-1189 (forward), loss sum :1188."""
import json
import os
import sys
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import svpc_oracle as orc  # noqa: E402
from svpc_amd import synthetic as syn  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# gradient tensors compared (a dozen, spread over every component of the path)
GRAD_NAMES = [
    "video_embeddings.video_embeddings.2.weight", "encoder.layer.0.attention.self.query.weight",
    "encoder.layer.2.attention.output.dense.weight", "encoder.layer.4.hidden_intermediate.dense.weight",
    "encoder.layer.5.attention.self.value.weight", "step_wise_encoder.layer.3.output.dense.weight",
    "decoder.layer.0.self_attention.key.weight", "decoder.layer.5.dec_enc_attention.query.weight",
    "decoder.layer.3.output.dense.weight", "decoder_classifier.decoder.weight", "text_embeddings.word_fc.2.weight",
    "text_embeddings.word_embeddings.weight", "token_type_embeddings.weight", "encoder.layer.1.output.LayerNorm.weight",
    "reasoner.W2.weight", "reasoner.action_selector.3.weight", "recipe_reasoner.W2.weight", "recipe_encoder.weight_hh_l0",
    "Wing.weight", "pgen_linear.0.weight", "Went.0.weight",
]

# stated tolerances: (loss rel, probability abs, gradient-norm rel, gradient cosine ≥)
# bf16: measured 3e-4 … 2.4e-3 (loss), 0.987 … 0.992 (argmax agreement), ≤ 5.2e-2 (gradient norms), ≥ 0.9925 (cosines) on the four
# headline cases (profiles/r02_c_headline_parity.json); the single worst probability entry moves by up to 0.24 where the pointer's
# softmax over ≤ 10 entities is nearly tied ("What the bf16 mode costs in accuracy"), hence a mean criterion beside the max.
# round 3: the bf16 entry tightened to ≤ 2× the measured values of profiles/r02_f_headline_parity.json (loss 7.7e-4 worst case, worst
# probability entry 0.19, mean 1.5e-5, gradient norms 5.8 % — 7.6 % on `Wing.weight` (viv) once the softmax used exp2: the pointer's
# near-ties make that tensor the noisiest —, cosine 0.9930); bf16x3: the fp32 entry's loss bar, measured probability
# errors ≈1e-5, the bf16 entry's gradient bounds (its backward is the bf16 backward)
# round 4: (i) bf16x3 gradient bounds at 2× the round-3 measurements (norms ≤ 2.7e-3, cosines ≥ 0.99994, worst probability 3.1e-4 over
# the eight cases of profiles/headline_parity.json) instead of the bf16 entry's: a regression that costs its backward a factor of
# ten now fails.  (ii) bf16: ONE named exception for the tensor the pointer's near-ties make noisy (`Wing.weight`, 7.6 %) instead of a
# bound widened for every tensor; the loss bar back at 3e-3: the loss deviation of this mode is the net of ~10^7 independent 2^-9
# roundings and is re-drawn by any change of the rounding pattern — the same inputs gave 5.7e-6 with the one-workgroup-per-pair
# attention forward and 2.1e-3 with the pipelined one, whose own error against an fp64 reference is SMALLER (rms 3.4e-4 vs 3.8e-4,
# no bias: tools/dbg/attn_bf16_error.py, asserted in tests/test_x3_gpu.py); round 2 had measured 3e-4 … 2.4e-3 over the four cases.
TOL = {"fp32": dict(loss=1e-4, prob=5e-5, prob_mean=1e-6, gnorm=2e-3, cos=0.99999, argmax=0.9999),
       "bf16x3": dict(loss=1e-4, prob=8e-4, prob_mean=2e-6, gnorm=6e-3, cos=0.9998, argmax=0.999),
       "bf16": dict(loss=3e-3, prob=0.4, prob_mean=5e-5, gnorm=8e-2, cos=0.990, argmax=0.98)}
# per-tensor exceptions: (precision, parameter) → overrides
# (`Wing.weight` in bf16 mode: 7.6 % in round 3, 13.2 % with round 4's attention forward — the pointer's softmax over ≤ 10 entities sits on
# near-ties that a 2^-9 change of its inputs flips; every other tensor stays inside the 8 % of the table)
TOL_TENSOR = {("bf16", "Wing.weight"): dict(gnorm=2e-1)}

_REPORT = {}
# the wider bf16 bars (loss 3e-3 instead of 1.5e-3, the `Wing.weight` exception) are justified only by "the pipelined
# attention forward re-draws the rounding pattern, it is not less accurate".  That claim is now CHECKED on the device before the wider
# bars apply: both forward kernels on the same bf16 inputs against an fp64 reference — the pipelined kernel's rms error and bias must be
# no worse than the one-workgroup-per-pair kernel's.  If the check fails the previous bars (1.5e-3, no per-tensor exception) are used.
_PREV_BF16 = dict(loss=1.5e-3)
_AB = {}


def _attn_ab():
    """→ dict(rms_pipe, rms_old, bias_pipe, bias_old, floor, ok) — clip-encoder attention forward, bf16, 100 × 100 × 64, fp64 reference"""
    if _AB:
        return _AB
    import math
    from svpc_amd import _lib, ops
    from svpc_amd.ops_common import SeqInfo
    H, dh, B, L = 12, 64, 48, 100
    D = H * dh
    g = torch.Generator().manual_seed(70)
    qkv = (1.5 * torch.randn(B * L, 3 * D, generator=g)).to(DEV).to(torch.bfloat16).contiguous()
    xv = qkv.double()
    q, k, v = (xv[:, i * D:(i + 1) * D].view(B, L, H, dh).permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1) @ v).permute(0, 2, 1, 3).reshape(B * L, D)
    seq = SeqInfo.uniform(B, L, L, DEV)
    prev = ops.get_precision()
    ops.set_precision("bf16")
    st = {}
    try:
        for on, tag in ((1, "pipe"), (0, "old")):
            was = _lib.load().svpc_attn_pipe_enable(on)
            try:
                out = ops.attention(qkv, qkv, (0, D, 2 * D), D, H, seq, key_mask=None, causal=False, drop=None)
            finally:
                _lib.load().svpc_attn_pipe_enable(was)
            e = out.double() - ref
            st["rms_" + tag], st["bias_" + tag] = float(e.pow(2).mean().sqrt()), float(e.mean())
    finally:
        ops.set_precision(prev)
    st["floor"] = float((ref.to(torch.bfloat16).double() - ref).pow(2).mean().sqrt())
    st["ok"] = bool(st["rms_pipe"] <= 1.02 * st["rms_old"] and abs(st["bias_pipe"]) <= max(abs(st["bias_old"]), 0.02 * st["floor"]))
    _AB.update(st)
    return _AB


def _build(mt, init):
    import bench
    args = bench.parse_args([])
    cfg, model = bench.build(args, "cpu", model_type=mt)
    if init == "drawn":      # test-sensitive weights (n/sqrt(fan_in), gains 1±0.1): as the config-1 fixtures use
        drawn = syn.draw_parameters(list(model.named_parameters()), seed=7)
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.copy_(drawn[n])
    model.eval()
    batch = syn.make_batch(cfg, n_videos=16, max_steps=12, n_ingr=10, n_oov=0, seed=2019, full_clips=True)
    g = torch.Generator().manual_seed(99)
    noise = [-torch.empty(12, cfg.max_t_len, cfg.vocab_size).exponential_(generator=g).log() for _ in range(16)]
    return cfg, model, batch, noise


def _oracle(cfg, model, batch, noise):
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters()]
    for n in names:
        P[n].requires_grad_(True)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.time()
    loss, probs, ents, acts = orc.forward(P, cfg, *syn.forward_args(batch), gumbel_noise=noise if cfg.model_mode == "full" else None)
    loss.backward()
    dt = time.time() - t0
    grads = {n: P[n].grad for n in names if P[n].grad is not None}
    return dict(loss=float(loss), probs=[p.detach() for p in probs], ents=[e.detach() for e in ents], grads=grads, seconds=dt)


_CACHE = {}


def _case(mt, init):
    key = (mt, init)
    if key not in _CACHE:
        while len(_CACHE) >= 2:  # at most two cases resident (each holds ≈0.8 GB of parameters + gradients)
            _CACHE.pop(next(iter(_CACHE)))
        cfg, model, batch, noise = _build(mt, init)
        ref = _oracle(cfg, model, batch, noise)
        _CACHE[key] = (cfg, model, batch, noise, ref)
    return _CACHE[key]


def _to_dev(batch):
    out = {}
    for k, v in batch.items():
        if isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            out[k] = [t.to(DEV) for t in v]
        elif isinstance(v, torch.Tensor):
            out[k] = v.to(DEV)
        else:
            out[k] = v
    return out


def _gumbel_flips(probs, ref_probs, noise, tau):
    """positions whose hard Gumbel pick (argmax of log(P + 1e-12) + G, model.py:1018) differs between the GPU's and the oracle's P"""
    flips = total = 0
    for p, r, g in zip(probs, ref_probs, noise):
        p, g = p.detach().cpu().float(), g.cpu().float()
        a = (torch.log(p + 1e-12) + g[..., :p.shape[-1]]).argmax(-1)
        b = (torch.log(r + 1e-12) + g[..., :r.shape[-1]]).argmax(-1)
        flips += int((a != b).sum())
        total += a.numel()
    return flips, total


def _compare(tag, precision, loss, probs, grads, ref, names, noise=None):
    tol = dict(TOL[precision])
    tol_tensor = dict(TOL_TENSOR)
    if precision == "bf16":
        ab = _attn_ab()
        _REPORT["_attention_ab_vs_fp64"] = dict(ab)
        if not ab["ok"]:         # the pipelined forward is NOT "the same error, re-drawn": the previous bars apply
            tol.update(_PREV_BF16)
            tol_tensor = {}
    rep = {"loss": loss, "ref_loss": ref["loss"], "loss_rel": abs(loss - ref["loss"]) / abs(ref["loss"])}
    if noise is not None:
        rep["gumbel_flips"], rep["gumbel_positions"] = _gumbel_flips(probs, ref["probs"], noise, None)
    perr = max(float((p.detach().cpu() - r).abs().max()) for p, r in zip(probs, ref["probs"]))
    rep["prob_abs_max"] = perr
    rep["prob_abs_mean"] = float(np.mean([float((p.detach().cpu() - r).abs().mean()) for p, r in zip(probs, ref["probs"])]))
    agree = np.mean([float((p.detach().cpu().argmax(-1) == r.argmax(-1)).float().mean()) for p, r in zip(probs, ref["probs"])])
    rep["argmax_agreement"] = float(agree)
    rep["grads"] = {}
    for n in names:
        if n not in ref["grads"]:
            continue
        g, r = grads[n].detach().double().cpu().reshape(-1), ref["grads"][n].double().reshape(-1)
        rn = float(r.norm())
        rep["grads"][n] = dict(norm_rel=abs(float(g.norm()) - rn) / max(rn, 1e-30), cos=float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-300)),
                               max_abs_over_max=float((g - r).abs().max() / r.abs().max()))
    _REPORT[tag] = rep
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    from helpers import product_sources_sha16
    _REPORT["_sources_sha16"] = product_sources_sha16()       # bench.py quotes the record only for these sources
    with open(os.path.join(out_dir, "headline_parity.json"), "w") as f:
        json.dump(_REPORT, f, indent=1)
    assert rep["loss_rel"] <= tol["loss"], (tag, rep["loss"], rep["ref_loss"], rep["loss_rel"])
    assert perr <= tol["prob"], (tag, perr)
    assert rep["prob_abs_mean"] <= tol["prob_mean"], (tag, rep["prob_abs_mean"])
    assert rep["argmax_agreement"] >= tol["argmax"], (tag, rep["argmax_agreement"])
    assert len(rep["grads"]) >= 12
    for n, d in rep["grads"].items():
        t = dict(tol, **tol_tensor.get((precision, n), {}))
        assert d["norm_rel"] <= t["gnorm"], (tag, n, d)
        assert d["cos"] >= t["cos"], (tag, n, d)


def _run_gpu(mt, init, precision):
    from svpc_amd import ops
    from svpc_amd.optim import FusedBertAdam
    import copy
    cfg, model_cpu, batch, noise, ref = _case(mt, init)
    model = copy.deepcopy(model_cpu).to(DEV)
    model.eval()
    if cfg.model_mode == "full":
        model.gumbel_noise = [n.to(DEV) for n in noise]
    b = _to_dev(batch)
    names = [n for n in GRAD_NAMES if n in ref["grads"]]
    ops.set_precision(precision)
    try:
        # (1) gradients through autograd's own accumulation
        loss, probs, _, _ = model(*syn.forward_args(b))
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        gn = noise if cfg.model_mode == "full" else None
        _compare("%s/%s/%s/autograd" % (mt, init, precision), precision, float(loss), probs, grads, ref, names, noise=gn)
        # (2) the captured step's path: gradients written in place into the optimizer's arena
        opt = FusedBertAdam(list(model.named_parameters()), lr=0.0, grad_clip=-1.0, max_grad_norm=-1.0)
        opt.ensure_built()
        opt.zero_grad()
        loss2, probs2, _, _ = model(*syn.forward_args(b))
        loss2.backward()
        ops.join_side()
        torch.cuda.synchronize()
        grads2 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        _compare("%s/%s/%s/arena" % (mt, init, precision), precision, float(loss2), probs2, grads2, ref, names, noise=gn)
    finally:
        ops.set_precision("fp32")
    return _REPORT


@pytest.mark.timeout(900)
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("init", ["bench", "drawn"])
def test_headline_vivt_step_vs_oracle(init, precision):
    _run_gpu("vivt", init, precision)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mt", ["vi", "viv"])
def test_headline_vi_viv_bf16_vs_oracle(mt):
    """BASELINE.json configs 2 and 3 in their stated dtype, in the ≤1e-4 throughput mode and in the fp32 mode, on one oracle run"""
    _run_gpu(mt, "drawn", "bf16")
    _run_gpu(mt, "drawn", "bf16x3")
    _run_gpu(mt, "drawn", "fp32")


# ------------------------------------------------------------------------------------------------ the fp32 mode's simulator residue
@pytest.mark.timeout(1500)
def test_fp32_mode_simulator_gradients_and_a_float64_oracle():
    """Round 3 attributed the fp32 mode's 1.5e-3 norm residue on `recipe_reasoner.W2.weight` (test-sensitive weights) to the CPU oracle's
    own fp32 conditioning.  The same headline case against the oracle run in FLOAT64 said otherwise (gpurun_out/fp64_oracle_leg.json,
    committed under profiles/), and led to the cause:
      * the visual simulator's loss IS conditioning-limited in the reference itself: nn.BCELoss(sum) clamps log(1 - e) at -100 once a
        sigmoid output rounds to exactly 1.0f, so the fp32 and the float64 evaluation of the reference's formulas differ by 17 % in the
        loss and 16-34 % on `reasoner.*` gradients — parity is defined by the fp32 reference, and the product matches THAT to 5e-4;
      * on `recipe_reasoner.*` the fp32 and float64 oracles agree to 3e-7, so the 2.5e-3 the product showed there was the product's:
        torch's BCE backward is (p - y) / max(p(1-p), 1e-12), not the derivative of the clamped forward the kernels used; a handful of
        re-simulated entity probabilities of 1e-12…1e-13 with label 1 made the difference (tests/tools/recipe_residue_full.py).  With the
        reference's formula in `bce_rows_bwd` / `loss_tail_bwd` the distance is 4e-7 — asserted here at 5e-6."""
    from svpc_amd import ops
    import copy
    cfg, model_cpu, batch, noise, ref32 = _case("vivt", "drawn")
    names = ["recipe_reasoner.W2.weight", "reasoner.W2.weight", "recipe_reasoner.action_selector.3.weight", "reasoner.action_selector.3.weight"]
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        P = {k: (v.detach().clone().double() if v.dtype.is_floating_point else v.detach().clone()) for k, v in model_cpu.state_dict().items()}
        pn = [n for n, _ in model_cpu.named_parameters()]
        for n in pn:
            P[n].requires_grad_(True)
        b64 = {k: ([t.double() if isinstance(t, torch.Tensor) and t.dtype == torch.float32 else t for t in v] if isinstance(v, list)
                   else (v.double() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else v)) for k, v in batch.items()}
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        loss64 = orc.forward(P, cfg, *syn.forward_args(b64), gumbel_noise=[n_.double() for n_ in noise])[0]
        loss64.backward()
        g64 = {n: P[n].grad.detach().clone() for n in pn if P[n].grad is not None}
    finally:
        torch.set_default_dtype(old)
    model = copy.deepcopy(model_cpu).to(DEV)
    model.eval()
    model.gumbel_noise = [n_.to(DEV) for n_ in noise]
    ops.set_precision("fp32")
    loss = model(*syn.forward_args(_to_dev(batch)))[0]
    loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    rep = {"loss_gpu_vs_oracle32": abs(float(loss) - ref32["loss"]) / abs(ref32["loss"]),
           "loss_oracle32_vs_oracle64": abs(ref32["loss"] - float(loss64)) / abs(float(loss64)), "tensors": {}}
    for n in names:
        g, r64, r32 = named[n].grad.detach().double().cpu().reshape(-1), g64[n].reshape(-1), ref32["grads"][n].double().reshape(-1)
        rep["tensors"][n] = dict(gpu_vs_oracle32=float((g - r32).norm() / r32.norm()), gpu_vs_oracle64=float((g - r64).norm() / r64.norm()),
                                 oracle32_vs_oracle64=float((r32 - r64).norm() / r64.norm()))
    # the audit this leg makes possible, over EVERY parameter: where the reference is well conditioned (its fp32 and float64 evaluations
    # agree to 1e-5) the product must agree with it too — a larger distance there is a difference in semantics, not rounding
    audit = []
    for n in sorted(g64):
        if named[n].grad is None or n not in ref32["grads"]:
            continue
        g, r64, r32 = named[n].grad.detach().double().cpu().reshape(-1), g64[n].reshape(-1), ref32["grads"][n].double().reshape(-1)
        if float(r64.norm()) == 0.0:
            continue
        audit.append((n, float((g - r32).norm() / r32.norm()), float((r32 - r64).norm() / r64.norm())))
    well = [a for a in audit if a[2] <= 1e-5]
    rep["audit"] = {"parameters": len(audit), "well_conditioned_in_the_reference": len(well),
                    "worst_well_conditioned": [dict(name=a[0], gpu_vs_oracle32=a[1], oracle32_vs_oracle64=a[2]) for a in sorted(well, key=lambda a: -a[1])[:8]]}
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "fp64_oracle_leg.json"), "w") as f:
        from helpers import product_sources_sha16
        rep["_sources_sha16"] = product_sources_sha16()
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))
    assert rep["loss_gpu_vs_oracle32"] <= 1e-6
    assert len(well) >= 100
    # (Went.0.*: ONE of the 147,456 pre-activations of went = relu(Went·ē) is 3.9e-7 from zero and falls on the other side with the
    # kernel's ē, which is 3.4e-6 from the oracle's — a ReLU flip, i.e. rounding: tests/tools/went_residue.py)
    flip = {"Went.0.bias": 5e-3, "Went.0.weight": 2e-3}
    for a in well:
        assert a[1] <= flip.get(a[0], 5e-5), a
    for n, d in rep["tensors"].items():
        if n.startswith("recipe_reasoner."):
            assert d["oracle32_vs_oracle64"] <= 1e-5, (n, d)          # the reference is well conditioned here …
            assert d["gpu_vs_oracle32"] <= 5e-6 and d["gpu_vs_oracle64"] <= 5e-6, (n, d)      # … and the product agrees with it (4e-7 measured)
        else:
            assert d["gpu_vs_oracle32"] <= 1.5e-3, (n, d)             # conditioning-limited in the reference itself (4.5e-4 measured)
