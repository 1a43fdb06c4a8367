"""Checkpoint files interchangeable with the reference's ``.chkpt``.

reference: src/train.py:401-405 writes ``{"model": state_dict, "model_cfg": EasyDict, "opt": EasyDict, "epoch": int}`` with
``torch.save``; src/translator.py:33-38 / src/translate.py read ``checkpoint["model_cfg"]`` and ``checkpoint["model"]``.

``save_checkpoint`` writes the same four keys, ALWAYS in one format: the two configs are pickled as ``easydict.EasyDict`` — by
qualified name, without importing that package (it is not needed to write the file) — which is what the reference's loaders expect
(attribute access on ``checkpoint["model_cfg"]``, src/translator.py:33-36) and what the reference itself writes; no class of this
package is pickled, so a reference process (which always has easydict: src/rtransformer/model.py:8 imports it) reads the file with
its bare ``torch.load``.  ``load_checkpoint`` maps that one class name onto ``ModelConfig`` while unpickling (so it opens files
written by the reference, or by this function, on a box without easydict); nothing else is remapped.
"""
from __future__ import annotations

import pickle
import types

import torch

from .synthetic import ModelConfig


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "easydict" and name == "EasyDict":
            return ModelConfig
        return super().find_class(module, name)


_pickle_module = types.ModuleType("svpc_amd._checkpoint_pickle")
_pickle_module.Unpickler = _Unpickler
_pickle_module.load = lambda f, **kw: _Unpickler(f, **kw).load()
_pickle_module.__name__ = "pickle"


class _AsEasyDict(dict):
    """a config on its way into a checkpoint: pickled under the name ``easydict.EasyDict`` (see _Pickler.save_global)"""


class _Pickler(pickle._Pickler):          # the pure-Python pickler: save_global can be overridden (tensors travel by persistent id)
    def save_global(self, obj, name=None):
        if obj is _AsEasyDict:
            self.write(pickle.GLOBAL + b"easydict\nEasyDict\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)


_save_module = types.ModuleType("svpc_amd._checkpoint_pickle_save")
_save_module.Pickler = _Pickler
_save_module.dump = lambda obj, f, protocol=None, **kw: _Pickler(f, protocol).dump(obj)
_save_module.__name__ = "pickle"


def save_checkpoint(path, model, opt=None, epoch=0, state_dict=None):
    """``state_dict`` overrides ``model.state_dict()`` (e.g. the EMA weights, as train.py:401 saves them)."""
    sd = state_dict if state_dict is not None else model.state_dict()
    sd = {k: v.detach().cpu().clone() for k, v in sd.items()}
    ckpt = {"model": sd, "model_cfg": _AsEasyDict(dict(model.config)), "opt": _AsEasyDict(dict(opt)) if opt is not None else None,
            "epoch": int(epoch)}
    torch.save(ckpt, path, pickle_module=_save_module)
    return ckpt


def load_checkpoint(path, model=None, map_location="cpu", strict=True):
    """→ the checkpoint dict; with ``model`` given its weights are loaded (same key set and shapes as the reference's
    ``state_dict`` — svpc_amd/model_shapes.py)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_pickle_module)
    for k in ("model_cfg", "opt"):
        if isinstance(ckpt.get(k), dict) and not isinstance(ckpt[k], ModelConfig):
            ckpt[k] = ModelConfig(ckpt[k])
    if model is not None:
        model.load_state_dict(ckpt["model"], strict=strict)
    return ckpt
