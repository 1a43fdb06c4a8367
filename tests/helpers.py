"""Shared test helpers: build the product model from a golden fixture."""
import os

import numpy as np
import torch

from oracle.cases import case_config_and_batch
from svpc_amd import model as M
from svpc_amd import synthetic as syn
from svpc_amd.model_shapes import parameter_shapes


def build_model(case, mt, golden_dir, device="cpu"):
    z = np.load(os.path.join(golden_dir, "%s_%s.npz" % (case, mt)))
    cfg, batch = case_config_and_batch(case, mt, device=device)
    model = M.StateAwareRecursiveTransformer(cfg)
    V, W, A = cfg.vocab_size, cfg.word_vec_size, cfg.action_vocab_size
    model.ingredient_embeddings.set_pretrained_embedding(torch.zeros(V, W), freeze=False)
    model.text_embeddings.set_pretrained_embedding(torch.zeros(V, W), freeze=False)
    if mt in ("vivt", "viv"):
        model.reasoner.set_pretrained_embedding(torch.zeros(A, W), freeze=False)
    if mt == "vivt":
        model.recipe_reasoner.set_pretrained_embedding(torch.zeros(A, W), freeze=False)
    if case.startswith("tiny"):
        sd = {k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    else:
        sd = syn.draw_parameters([(n, torch.empty(s)) for n, s in parameter_shapes(cfg, mt).items()], seed=7)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(".pe") for k in missing), missing
    model.to(device)
    model.eval()
    noise = [torch.from_numpy(z[k]).to(device) for k in sorted((k for k in z.files if k.startswith("gumbel/")),
                                                              key=lambda s: int(s.split("/")[1]))]
    model.gumbel_noise = noise or None
    return z, cfg, batch, model


def product_sources_sha16():
    """hash of everything a parity record depends on: the kernel sources and the host-side package (bench.py recomputes it and refuses
    to quote a record made from other sources)"""
    import glob
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "svpc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "svpc_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "svpc_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(root, "svpc_amd", "*.py")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
