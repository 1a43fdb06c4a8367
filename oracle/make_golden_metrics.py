"""TEST INFRASTRUCTURE — generates tests/golden/metrics.npz by RUNNING the reference's own logging arithmetic
(src/train.py:32-49 cal_performance / calculate_f1, :51-68 compute_total_f1) in this container on seeded inputs.

In-process stubs installed before the import (nothing of the reference is modified or copied): ``easydict``, ``nltk``,
``tensorboardX`` (not installed here; imported at module top), ``torch.Tensor.cuda`` → identity.
Run:  PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_metrics.py
"""
import os
import sys
import types

import numpy as np
import torch

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
nltk = types.ModuleType("nltk"); nltk.tokenize = types.ModuleType("nltk.tokenize"); nltk.tokenize.word_tokenize = lambda s: s.split()
sys.modules.setdefault("nltk", nltk); sys.modules.setdefault("nltk.tokenize", nltk.tokenize)
ed = types.ModuleType("easydict")
class EasyDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
ed.EasyDict = EasyDict
sys.modules.setdefault("easydict", ed)
tb = types.ModuleType("tensorboardX"); tb.SummaryWriter = object
sys.modules.setdefault("tensorboardX", tb)
if not hasattr(np, "int"):
    np.int = int
torch.Tensor.cuda = lambda self, *a, **k: self

import src.train as T   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "metrics.npz")


def main():
    g = torch.Generator().manual_seed(11)
    out = {}
    n_vid = 5
    out["n_vid"] = np.array(n_vid)
    tot = np.zeros(8)
    for b in range(n_vid):
        S, Lt, C = int(torch.randint(1, 5, (1,), generator=g)), 6, 20 + b
        pred = torch.rand(S, Lt, C, generator=g)
        pred[0, 0, 3] = pred[0, 0, 7] = 2.0                    # a tie: arg-max must take the first
        gold = torch.randint(0, C, (S, Lt), generator=g)
        gold[torch.rand(S, Lt, generator=g) < 0.3] = -1
        gold[0, 0] = 3
        E, A = 4 + b, 9
        eprob, aprob = torch.rand(S, E, generator=g), torch.rand(S, A, generator=g)
        egold, agold = (torch.rand(S, E, generator=g) < 0.3).float(), (torch.rand(S, A, generator=g) < 0.1).float()
        n_correct = T.cal_performance(pred, gold)
        n_word = int(gold.ne(-1).sum())
        ec, er, ep = T.calculate_f1([eprob], [egold])
        ac, ar, ap = T.calculate_f1([aprob], [agold])
        tot += np.array([n_word, n_correct, ec, er, ep, ac, ar, ap], dtype=np.float64)
        for k, v in (("pred", pred), ("gold", gold), ("eprob", eprob), ("egold", egold), ("aprob", aprob), ("agold", agold)):
            out["v%d/%s" % (b, k)] = v.numpy()
        out["v%d/counts" % b] = np.array([n_word, n_correct, ec, er, ep, ac, ar, ap], dtype=np.float64)
    out["total"] = tot
    r = T.compute_total_f1(tot[2], tot[3], tot[4]); out["entity_f1"] = np.array([r["recall"], r["precision"], r["f1"]])
    r = T.compute_total_f1(tot[5], tot[6], tot[7]); out["action_f1"] = np.array([r["recall"], r["precision"], r["f1"]])
    r = T.compute_total_f1(0, 0, 0); out["zero_f1"] = np.array([r["recall"], r["precision"], r["f1"]])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, tot)


if __name__ == "__main__":
    main()
