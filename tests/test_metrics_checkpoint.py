"""on-device logging counters against numbers produced by the reference's own functions
(tests/golden/metrics.npz ← oracle/make_golden_metrics.py runs src/train.py:32-68), and checkpoint files interchangeable with the
reference's ``.chkpt`` layout (src/train.py:401-405)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from svpc_amd import checkpoint as CK          # noqa: E402
from svpc_amd import metrics as MT             # noqa: E402
from svpc_amd import synthetic as syn          # noqa: E402

GOLD = os.path.join(HERE, "golden", "metrics.npz")


def test_compute_total_f1_matches_reference_values():
    z = np.load(GOLD)
    t = z["total"]
    for key, args in (("entity_f1", t[2:5]), ("action_f1", t[5:8]), ("zero_f1", (0, 0, 0))):
        r = MT.compute_total_f1(*[float(a) for a in args])
        np.testing.assert_allclose([r["recall"], r["precision"], r["f1"]], z[key], rtol=0, atol=0)


@pytest.mark.gpu
def test_device_counters_match_reference_counts():
    z = np.load(GOLD)
    dev = torch.device("cuda")
    m = MT.TrainMetrics(dev)
    for rep in range(2):                      # accumulation over steps
        for b in range(int(z["n_vid"])):
            g = lambda k: torch.from_numpy(z["v%d/%s" % (b, k)]).to(dev)
            m.update(torch.tensor(1.5, device=dev), [g("pred")], [g("gold")], [g("eprob")], [g("egold")], [g("aprob")], [g("agold")])
    r = m.result()
    np.testing.assert_array_equal(np.array(r["counts"]), 2 * z["total"])
    assert r["total_loss"] == pytest.approx(1.5 * 2 * int(z["n_vid"]))
    assert r["accuracy"] == pytest.approx(z["total"][1] / z["total"][0])
    np.testing.assert_allclose([r["entity"][k] for k in ("recall", "precision", "f1")], z["entity_f1"], rtol=1e-12)
    np.testing.assert_allclose([r["action"][k] for k in ("recall", "precision", "f1")], z["action_f1"], rtol=1e-12)
    # per-video counts too (fresh counters each)
    for b in range(int(z["n_vid"])):
        m.reset()
        g = lambda k: torch.from_numpy(z["v%d/%s" % (b, k)]).to(dev)
        m.update(None, [g("pred")], [g("gold")], [g("eprob")], [g("egold")], [g("aprob")], [g("agold")])
        np.testing.assert_array_equal(np.array(m.result()["counts"]), z["v%d/counts" % b])


def test_metrics_need_the_gpu():
    with pytest.raises(Exception):
        MT.TrainMetrics("cpu")


def _tiny_model():
    from svpc_amd.model import StateAwareRecursiveTransformer
    cfg = syn.make_config(model_type="vi", hidden_size=32, num_hidden_layers=1, num_attention_heads=4, video_feature_size=16,
                          vocab_size=30, word_vec_size=10, action_vocab_size=6, max_v_len=8, max_t_len=5, max_i_len=8)
    torch.manual_seed(0)
    return cfg, StateAwareRecursiveTransformer(cfg)


def test_checkpoint_roundtrip_and_reference_layout(tmp_path):
    cfg, model = _tiny_model()
    path = str(tmp_path / "model.chkpt")
    opt = syn.ModelConfig(lr=1e-4, max_v_len=8)
    CK.save_checkpoint(path, model, opt=opt, epoch=3)
    raw = CK.load_checkpoint(path)       # (a bare torch.load needs the reference's easydict: test_saved_checkpoint_opens_without_this_package)
    assert set(raw) == {"model", "model_cfg", "opt", "epoch"} and raw["epoch"] == 3            # src/train.py:401-405
    cfg2, model2 = _tiny_model()
    with torch.no_grad():
        for p in model2.parameters():
            p.add_(1.0)
    ck = CK.load_checkpoint(path, model2)
    assert ck["model_cfg"].hidden_size == 32 and "model_mode" in ck["model_cfg"] and ck["opt"].lr == 1e-4
    for (n1, p1), (n2, p2) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2)


def test_load_checkpoint_written_with_easydict(tmp_path):
    """A file as the reference writes it: configs are pickled ``easydict.EasyDict`` objects (package absent in this image)."""
    assert "easydict" not in sys.modules
    ed = types.ModuleType("easydict")
    exec("class EasyDict(dict):\n"
         "    def __init__(self, d=None, **kw):\n"
         "        super().__init__()\n"
         "        for k, v in dict(d or {}, **kw).items():\n"
         "            setattr(self, k, v)\n"
         "    def __setattr__(self, k, v):\n"
         "        super().__setattr__(k, v)\n"
         "        super().__setitem__(k, v)\n"
         "    __setitem__ = __setattr__\n", ed.__dict__)
    EasyDict = ed.EasyDict
    sys.modules["easydict"] = ed
    try:
        cfg, model = _tiny_model()
        path = str(tmp_path / "ref.chkpt")
        torch.save({"model": model.state_dict(), "model_cfg": EasyDict(dict(cfg)), "opt": EasyDict(lr=3e-4, debug=False), "epoch": 7}, path)
    finally:
        del sys.modules["easydict"]
    cfg2, model2 = _tiny_model()
    ck = CK.load_checkpoint(path, model2)
    assert isinstance(ck["model_cfg"], syn.ModelConfig) and ck["model_cfg"].hidden_size == cfg.hidden_size
    assert ck["opt"].lr == 3e-4 and ck["epoch"] == 7
    for (n1, p1), (n2, p2) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2)


def test_saved_checkpoint_opens_without_this_package(tmp_path):
    """ONE deterministic format: the configs are pickled under the name ``easydict.EasyDict`` whether or not that package is installed
    where the file is written, and no class of svpc_amd is pickled — a fresh interpreter that cannot import this package but has an
    ``easydict`` module (the reference's own dependency, src/rtransformer/model.py:8; stubbed here as the oracle harness stubs it)
    reads it with a bare ``torch.load`` and uses the reference Translator's access pattern (src/translator.py:33-38: attribute access
    on ``checkpoint["model_cfg"]``, ``checkpoint["model"]`` a plain state_dict)."""
    import subprocess
    cfg, model = _tiny_model()
    path = str(tmp_path / "plain.chkpt")
    CK.save_checkpoint(path, model, opt=syn.ModelConfig(lr=1e-4), epoch=1)
    raw = open(path, "rb").read()
    assert b"easydict" in raw and b"svpc_amd" not in raw
    code = ("import sys, types, torch\n"
            "sys.modules['svpc_amd'] = None\n"                       # any 'import svpc_amd…' inside the unpickler would raise
            "m = types.ModuleType('easydict')\n"
            "class EasyDict(dict):\n"
            "    def __init__(self, d=None, **kw):\n"
            "        super().__init__()\n"
            "        for k, v in dict(d or {}, **kw).items(): self[k] = v\n"
            "    def __setitem__(self, k, v):\n"
            "        super().__setitem__(k, v); self.__dict__[k] = v\n"
            "    __setattr__ = __setitem__\n"
            "m.EasyDict = EasyDict; sys.modules['easydict'] = m\n"
            "checkpoint = torch.load(%r, weights_only=False)\n"
            "model_config = checkpoint['model_cfg']\n"
            "assert type(model_config).__name__ == 'EasyDict' and type(checkpoint['opt']).__name__ == 'EasyDict'\n"
            "assert model_config.hidden_size == 32 and model_config.max_t_len > 0 and model_config.max_v_len > 0\n"     # translator.py:34-36
            "assert 'label_smoothing' in model_config and checkpoint['opt'].lr == 1e-4 and checkpoint['epoch'] == 1\n"
            "assert all(isinstance(v, torch.Tensor) for v in checkpoint['model'].values())\n"
            "print('ok')\n" % path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), timeout=240)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    # and this package's own loader reads the same file back (EasyDict → ModelConfig)
    cfg2, model2 = _tiny_model()
    ck = CK.load_checkpoint(path, model2)
    assert isinstance(ck["model_cfg"], syn.ModelConfig) and ck["model_cfg"].hidden_size == 32 and ck["opt"].lr == 1e-4
