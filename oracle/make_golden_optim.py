"""Generate tests/golden/optim.npz by running the REAL reference optimizer tail on CPU (TEST INFRASTRUCTURE, build container only).

What is pinned: the training-step tail of the reference, in its own order (src/train.py:140-147):
    nn.utils.clip_grad_norm_(model.parameters(), grad_clip)  →  BertAdam.step()  →  EMA(model, niter)
with the reference's own classes imported from /root/reference (src/rtransformer/optimization.py: BertAdam :219-338 incl. its
per-tensor clip :306-307, warmup-linear schedule :162-171, no bias correction; EMA :183-216) and the reference's weight-decay
grouping by parameter name (src/train.py:338-343), on the reference's own model object (tiny vivt configuration, the parameters of
tests/golden/tiny_vivt.npz).  Gradients are DATA: the tiny vivt golden gradients, rescaled/perturbed per step by
``oracle.cases.optim_step_gradient`` (the last step has a global norm < 1 so that the un-clipped branch is covered too).
Nothing of the reference is copied; only parameter values after each step, the final m / v and the EMA shadow are stored.

Usage:  python oracle/make_golden_optim.py
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle.cases import CASES, OPTIM_CASE, optim_step_gradient  # noqa: E402
from oracle.make_golden import build_reference_model, install_shims  # noqa: E402


def main():
    EasyDict = install_shims()
    from src.rtransformer.model import StateAwareRecursiveTransformer as Model
    from src.rtransformer.optimization import EMA, BertAdam
    gdir = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(gdir, "tiny_vivt.npz"))
    cfg_kw, _ = CASES["tiny"]
    cfg, model = build_reference_model(EasyDict, Model, cfg_kw, "vivt")
    with torch.no_grad():     # the fixture's parameters (build_reference_model draws the same ones; assert rather than assume)
        for n, p in model.named_parameters():
            assert np.array_equal(p.numpy(), z["param/" + n]), n
    c = OPTIM_CASE
    live = [n for n, _ in model.named_parameters() if "grad/" + n in z.files]
    named = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]                                   # src/train.py:339
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": c["weight_decay"]},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = BertAdam(groups, lr=c["lr"], warmup=c["warmup"], t_total=c["t_total"], schedule="warmup_linear")
    ema = EMA(c["ema_decay"])
    for n, p in named:
        if p.requires_grad:
            ema.register(n, p.data)
    out = {"names": np.array(live)}
    for t in range(c["steps"]):
        opt.zero_grad()
        for n, p in named:
            p.grad = torch.from_numpy(optim_step_gradient(z["grad/" + n], n, t)) if n in live else None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")        # the reference uses the deprecated add_(scalar, tensor) overloads
            total = torch.nn.utils.clip_grad_norm_(model.parameters(), c["grad_clip"])              # src/train.py:141-142
            opt.step()                                                                              # :143
            ema(model, t)                                                                           # :146-147 (niter)
        out["gnorm/%d" % t] = np.float64(float(total))
        for n, p in named:
            if n in live:
                out["p/%d/%s" % (t, n)] = p.detach().numpy().copy()
    for n, p in named:
        if n in live:
            st = opt.state[p]
            out["m/" + n] = st["next_m"].numpy().copy()
            out["v/" + n] = st["next_v"].numpy().copy()
            out["ema/" + n] = ema.shadow[n].numpy().copy()
    # parameters the reference never touches (no gradient → BertAdam skips them, :290-291) must stay put
    dead = [n for n, _ in named if n not in live]
    out["dead"] = np.array(dead)
    path = os.path.join(gdir, "optim.npz")
    np.savez_compressed(path, **out)
    print("%s: %d live tensors, %d dead, %d steps, global norms %s, %.1f KB"
          % (os.path.basename(path), len(live), len(dead), c["steps"],
             ["%.3g" % float(out["gnorm/%d" % t]) for t in range(c["steps"])], os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
