"""Checkpoint files interchangeable with the reference's ``.chkpt``.

reference: src/train.py:401-405 writes ``{"model": state_dict, "model_cfg": EasyDict, "opt": EasyDict, "epoch": int}`` with
``torch.save``; src/translator.py:33-38 / src/translate.py read ``checkpoint["model_cfg"]`` and ``checkpoint["model"]``.

``save_checkpoint`` writes the same four keys.  The two configs are pickled as ``easydict.EasyDict`` when that package is
importable (what the reference's loaders expect: they use attribute access on ``checkpoint["model_cfg"]``), otherwise as plain
``dict``s — never as a class of this package, so the file opens with a bare ``torch.load`` wherever it is read (a reference
process without easydict-typed configs needs one line, ``EasyDict(ckpt["model_cfg"])``).  ``load_checkpoint``
turns either form into ``ModelConfig`` and also opens files written by the reference on a box without easydict: that one class
name is mapped onto ``ModelConfig`` while unpickling; nothing else is remapped.
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


def save_checkpoint(path, model, opt=None, epoch=0, state_dict=None):
    """``state_dict`` overrides ``model.state_dict()`` (e.g. the EMA weights, as train.py:401 saves them)."""
    sd = state_dict if state_dict is not None else model.state_dict()
    sd = {k: v.detach().cpu().clone() for k, v in sd.items()}
    try:
        from easydict import EasyDict as _cfg_type       # the reference's own container, when the environment has it
    except ImportError:
        _cfg_type = dict
    ckpt = {"model": sd, "model_cfg": _cfg_type(dict(model.config)), "opt": _cfg_type(dict(opt)) if opt is not None else None,
            "epoch": int(epoch)}
    torch.save(ckpt, path)
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
