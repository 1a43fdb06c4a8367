"""The training-step tail {global clip → BertAdam → EMA} against tests/golden/optim.npz, which holds the outputs of the
REFERENCE's own BertAdam / EMA / clip_grad_norm_ sequence (oracle/make_golden_optim.py; src/train.py:140-147,
src/rtransformer/optimization.py:183-216,284-331).  CPU: the oracle's restatement (what bench.py's cpu_baseline leg steps with).
GPU: the three-launch fused kernel (svpc_amd/csrc/optimizer.hip) through FusedBertAdam."""
import os

import numpy as np
import pytest
import torch

from oracle import svpc_oracle as orc
from oracle.cases import OPTIM_CASE, optim_step_gradient

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")      # src/train.py:339


def _load(golden_dir):
    z = np.load(os.path.join(golden_dir, "optim.npz"))
    t = np.load(os.path.join(golden_dir, "tiny_vivt.npz"))
    return z, t, [str(n) for n in z["names"]]


def test_oracle_tail_matches_reference_optimizer(golden_dir):
    z, t, names = _load(golden_dir)
    c = OPTIM_CASE
    P = {n: torch.from_numpy(t["param/" + n].copy()) for n in names}
    shadow = {n: p.clone() for n, p in P.items()}
    state = {}
    wd = {n: (0.0 if any(k in n for k in NO_DECAY) else c["weight_decay"]) for n in names}
    for s in range(c["steps"]):
        grads = {n: torch.from_numpy(optim_step_gradient(t["grad/" + n], n, s)) for n in names}
        total, _ = orc.global_clip_coef(grads, c["grad_clip"])
        assert abs(float(total) - float(z["gnorm/%d" % s])) <= 1e-5 * float(z["gnorm/%d" % s])
        orc.train_tail_step(P, grads, state, shadow, s, c["lr"], c["warmup"], c["t_total"], c["grad_clip"], c["ema_decay"], wd)
        for n in names:
            np.testing.assert_allclose(P[n].numpy(), z["p/%d/%s" % (s, n)], rtol=1e-5, atol=1e-7, err_msg="%s step %d" % (n, s))
    for n in names:
        np.testing.assert_allclose(state[n][0].numpy(), z["m/" + n], rtol=1e-5, atol=1e-9, err_msg=n)
        np.testing.assert_allclose(state[n][1].numpy(), z["v/" + n], rtol=1e-5, atol=1e-12, err_msg=n)
        np.testing.assert_allclose(shadow[n].numpy(), z["ema/" + n], rtol=1e-5, atol=1e-7, err_msg=n)
    assert float(z["gnorm/%d" % (c["steps"] - 1)]) < 1.0 < float(z["gnorm/0"])      # both clip branches are in the fixture


@pytest.mark.gpu
def test_fused_bert_adam_matches_reference_optimizer(golden_dir):
    from helpers import build_model
    from svpc_amd.optim import FusedBertAdam
    z, t, names = _load(golden_dir)
    c = OPTIM_CASE
    _, cfg, batch, model = build_model("tiny", "vivt", golden_dir, "cuda:0")
    named = dict(model.named_parameters())
    before_dead = {str(n): named[str(n)].detach().cpu().clone() for n in z["dead"]}
    opt = FusedBertAdam(list(model.named_parameters()), lr=c["lr"], warmup=c["warmup"], t_total=c["t_total"],
                        weight_decay=c["weight_decay"], grad_clip=c["grad_clip"], ema_decay=c["ema_decay"])
    for s in range(c["steps"]):
        opt.zero_grad()
        for n in names:
            g = torch.from_numpy(optim_step_gradient(t["grad/" + n], n, s)).to("cuda:0")
            if named[n].grad is None:
                named[n].grad = g
            else:
                named[n].grad.copy_(g)          # a view into the gradient arena once the optimizer is built
        opt.step()
        assert abs(float(opt.grad_norm()) - float(z["gnorm/%d" % s])) <= 1e-4 * float(z["gnorm/%d" % s])
        for n in names:
            np.testing.assert_allclose(named[n].detach().cpu().numpy(), z["p/%d/%s" % (s, n)], rtol=2e-5, atol=2e-7,
                                       err_msg="%s step %d" % (n, s))
    assert sorted(opt.arena.names) == sorted(names)
    for (n, p), o in zip(zip(opt.arena.names, opt.arena.params), opt.arena.offsets):
        k = p.numel()
        np.testing.assert_allclose(opt.m[o:o + k].cpu().numpy(), z["m/" + n].reshape(-1), rtol=2e-5, atol=1e-9, err_msg=n)
        np.testing.assert_allclose(opt.v[o:o + k].cpu().numpy(), z["v/" + n].reshape(-1), rtol=2e-5, atol=1e-12, err_msg=n)
        np.testing.assert_allclose(opt.ema[o:o + k].cpu().numpy(), z["ema/" + n].reshape(-1), rtol=2e-5, atol=2e-7, err_msg=n)
        # the bf16 weight shadow the GEMMs read is the rounding of the new fp32 weight
        assert torch.equal(opt.weights.shadow[o:o + k].cpu(), p.detach().reshape(-1).to(torch.bfloat16).cpu()), n
    for n, v in before_dead.items():
        assert torch.equal(named[n].detach().cpu(), v), n
