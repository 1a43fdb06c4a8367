"""Generate tests/golden/*.npz by running the REAL reference (awkrail/svpc @ /root/reference) on CPU.

TEST INFRASTRUCTURE — runs only in the build container (the reference never travels to the GPU box).
Nothing of the reference is copied: this script imports it in-process with three shims installed
before import and stores *data only* — inputs are regenerated from seeds
by ``svpc_amd.synthetic``; the fixtures hold parameters, Gumbel noise and the reference's outputs.

  shims: (1) a stub ``easydict`` module (model.py:8 imports it; not installed here),
         (2) ``torch.Tensor.cuda`` → identity (the model hard-codes .cuda(), e.g. model.py:50,123,783-790),
         (3) a stub ``nltk`` (recursive_caption_dataset.py:5 imports it; only needed for src.translator).

Usage:  python oracle/make_golden.py [case …]   (writes tests/golden/; no argument: every case of oracle/cases.py)
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"

from svpc_amd import synthetic as syn  # noqa: E402


def install_shims():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            if d:
                self.update(d)
            self.update(kw)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    m = types.ModuleType("easydict")
    m.EasyDict = EasyDict
    sys.modules["easydict"] = m
    torch.Tensor.cuda = lambda self, *a, **k: self
    nl = types.ModuleType("nltk")
    tk = types.ModuleType("nltk.tokenize")
    tk.word_tokenize = lambda s: s.split()
    nl.tokenize = tk
    nl.word_tokenize = tk.word_tokenize
    sys.modules["nltk"] = nl
    sys.modules["nltk.tokenize"] = tk
    sys.path.insert(0, REF)
    return EasyDict


from oracle.cases import CASES, CASE_MODES, MODES  # noqa: E402


def build_reference_model(EasyDict, Model, cfg_kw, model_type, seed=7):
    cfg = syn.make_config(model_type=model_type, **cfg_kw)
    model = Model(EasyDict(dict(cfg)))
    V, W, A = cfg.vocab_size, cfg.word_vec_size, cfg.action_vocab_size
    g = torch.Generator().manual_seed(seed + 1000)
    glove = 0.4 * torch.randn(V, W, generator=g)
    verb = 0.4 * torch.randn(A, W, generator=g)
    # src/train.py:691-707: which tables are replaced depends on the mode
    model.ingredient_embeddings.set_pretrained_embedding(glove.clone(), freeze=False)
    model.text_embeddings.set_pretrained_embedding(glove.clone(), freeze=False)
    if model_type in ("vivt", "viv"):
        model.reasoner.set_pretrained_embedding(verb.clone(), freeze=False)
    if model_type == "vivt":
        model.recipe_reasoner.set_pretrained_embedding(verb.clone(), freeze=False)
    drawn = syn.draw_parameters(list(model.named_parameters()), seed=seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(drawn[n])
    model.eval()
    return cfg, model


def run_case(EasyDict, Model, Translator, case, model_type, full_dump):
    import torch.nn.functional as F
    cfg_kw, batch_kw = CASES[case]
    cfg, model = build_reference_model(EasyDict, Model, cfg_kw, model_type)
    batch = syn.make_batch(cfg, **batch_kw)
    out = {}
    for k, v in model.state_dict().items():
        out["param/" + k] = v.detach().numpy().copy()

    # ---- record the Gumbel noise the reference draws (one call per video, in loop order) ----
    noises = []
    real = F.gumbel_softmax

    def recording_gumbel(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        st = torch.get_rng_state()
        res = real(logits, tau=tau, hard=hard, eps=eps, dim=dim)
        st2 = torch.get_rng_state()
        torch.set_rng_state(st)
        noises.append(-torch.empty_like(logits, memory_format=torch.legacy_contiguous_format).exponential_().log())
        torch.set_rng_state(st2)
        return res

    F.gumbel_softmax = recording_gumbel
    captured = {}
    hooks = []
    if full_dump:
        def mk(name):
            def hook(mod, inp, outp):
                captured.setdefault(name, []).append(outp)
            return hook
        for name in ("ingredient_embeddings", "video_embeddings", "encoder", "step_wise_encoder", "reasoner",
                     "text_embeddings", "decoder", "decoder_classifier", "recipe_encoder", "recipe_reasoner"):
            hooks.append(getattr(model, name).register_forward_hook(mk(name)))
    try:
        torch.manual_seed(1234)
        args = syn.forward_args(batch)
        args[4] = [x.clone() for x in args[4]]
        loss, probs, ents, acts = model(*args)
        loss.backward()
    finally:
        F.gumbel_softmax = real
        for h in hooks:
            h.remove()

    out["loss"] = np.float64(loss.item())
    for b, n in enumerate(noises):
        out["gumbel/%d" % b] = n.numpy()
    for b, p in enumerate(probs):
        p = p.detach().numpy()
        if full_dump:
            out["probs/%d" % b] = p
        else:
            out["probs_sum/%d" % b] = p.sum(-1)
            out["probs_slice/%d" % b] = p[:, :, ::37].copy()
            out["probs_argmax/%d" % b] = p.argmax(-1)
    for b, e in enumerate(ents):
        out["ent/%d" % b] = e.detach().numpy()
    for b, a in enumerate(acts):
        out["act/%d" % b] = a.detach().numpy()
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach()
        if full_dump:
            out["grad/" + n] = g.numpy().copy()
        else:
            out["gradnorm/" + n] = np.float64(g.double().norm().item())
            out["gradslice/" + n] = g.reshape(-1)[:: max(1, g.numel() // 64)][:64].numpy().copy()
    if full_dump:
        def last(x):
            return x[-1] if isinstance(x, (list, tuple)) and not isinstance(x[0], (int, float)) else x
        for name, calls in captured.items():
            for i, o in enumerate(calls):
                if name in ("encoder", "step_wise_encoder", "decoder"):
                    out["mid/%s/%d" % (name, i)] = o[-1].detach().numpy()
                elif name in ("reasoner", "recipe_reasoner"):
                    for j, t in enumerate(o):
                        out["mid/%s/%d/%d" % (name, i, j)] = t.detach().numpy()
                elif name == "recipe_encoder":
                    out["mid/%s/%d" % (name, i)] = o[0].detach().numpy()
                else:
                    out["mid/%s/%d" % (name, i)] = o.detach().numpy()

    # ---- greedy decode through the reference Translator (bit-exact id targets) ----
    model.zero_grad()
    tr = Translator(opt=EasyDict(cuda=False),
                    checkpoint={"model_cfg": model.config, "model": model.state_dict()}, model=model)
    dec, _ = tr.translate_batch(syn.translate_inputs(batch))
    for b, d in enumerate(dec):
        out["decode/%d" % b] = d.numpy().astype(np.int64)
    return out


def main():
    EasyDict = install_shims()
    from src.rtransformer.model import StateAwareRecursiveTransformer as Model
    from src.translator import Translator
    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)
    only = sys.argv[1:]          # optional case names: regenerate those fixtures only
    for case in CASES:
        if only and case not in only:
            continue
        for mt in MODES:
            if mt not in CASE_MODES.get(case, MODES):
                continue
            out = run_case(EasyDict, Model, Translator, case, mt, full_dump=case.startswith("tiny"))
            if case == "c1":  # parameters are re-drawn from the seed by the tests (too large to commit)
                out = {k: v for k, v in out.items() if not k.startswith("param/")}
            path = os.path.join(gdir, "%s_%s.npz" % (case, mt))
            np.savez_compressed(path, **out)
            print("%-14s loss=%.6f  keys=%d  %.1f KB" % (os.path.basename(path), float(out["loss"]), len(out),
                                                          os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
