"""TEST INFRASTRUCTURE — generates tests/golden/input_pipeline.npz by RUNNING the reference's own feature-windowing functions
(src/rtransformer/recursive_caption_dataset.py:380-416) in this container on seeded synthetic videos.

Shims installed in-process before the import (no reference file is modified, nothing of it is copied): stub ``nltk`` and
``easydict`` modules (not installed here; the module imports them at the top) and ``np.int = int`` (removed in numpy ≥ 1.24; the
reference calls ``.astype(np.int)`` at :404).  Run:  PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_input.py
"""
import os
import sys
import types

import numpy as np

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
if not hasattr(np, "int"):
    np.int = int

from src.rtransformer.recursive_caption_dataset import RecursiveCaptionDataset as DS   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "input_pipeline.npz")


def main():
    rng = np.random.RandomState(7)
    stub = types.SimpleNamespace(max_v_len=12, max_t_len=5, CLS_TOKEN="[CLS]", VID_TOKEN="[VID]", SEP_TOKEN="[SEP]", PAD_TOKEN="[PAD]",
                                 _convert_to_feat_index_st_ed=DS._convert_to_feat_index_st_ed)
    tok = {"[CLS]": 1, "[VID]": 3, "[SEP]": 2, "[PAD]": 0}
    out = {}
    cases = []
    # (feat_len, frm2sec, [st, ed]) — short windows, windows longer than max_v_len-2 (down-sampled), clamped ends, st == ed
    for k in range(24):
        feat_len = int(rng.randint(6, 60))
        frm2sec = float(rng.choice([0.5, 1.0, 1.7, 2.56]))
        dur = feat_len * frm2sec
        t0 = float(rng.uniform(0, dur * 0.8))
        t1 = float(min(dur * 1.2, t0 + rng.choice([0.0, 0.4, 3.0, 9.0, 25.0, 70.0])))
        cases.append((feat_len, frm2sec, t0, t1))
    cases += [(30, 1.0, 0.0, 29.0), (30, 1.0, 28.5, 40.0), (11, 2.0, 0.0, 100.0), (8, 1.0, 3.0, 3.0)]
    for k, (feat_len, frm2sec, t0, t1) in enumerate(cases):
        raw = rng.rand(feat_len, 6).astype(np.float32)
        st, ed_ = DS._convert_to_feat_index_st_ed(feat_len, [t0, t1], frm2sec)
        feat, tokens, mask = DS._load_indexed_video_feature(stub, raw, [t0, t1], frm2sec)
        out["case%d/args" % k] = np.array([feat_len, frm2sec, t0, t1], dtype=np.float64)
        out["case%d/raw" % k] = raw
        out["case%d/st_ed" % k] = np.array([st, ed_], dtype=np.int64)
        out["case%d/feat" % k] = feat
        out["case%d/ids" % k] = np.array([tok[t] for t in tokens], dtype=np.int64)
        out["case%d/mask" % k] = np.array(mask, dtype=np.float32)
    out["n_cases"] = np.array(len(cases))
    out["max_v_len"] = np.array(12); out["max_t_len"] = np.array(5)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
