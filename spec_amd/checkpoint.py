"""Checkpoint loading with the reference's conventions.

``load_pretrained_model`` restates the contract the reference relies on from
``pare.utils.train_utils`` (call sites ``spec/tester.py:69-70``: ``overwrite_shape_mismatch=True,
remove_lightning=True``; ``scripts/camcalib_demo.py:81``: ``remove_lightning=True, strict=True``;
``spec/models/hmr.py:128``): strip the Lightning ``model.`` prefix, load (non-)strictly and, on
a shape mismatch, patch or drop the offending keys and reload non-strictly.

``read_checkpoint`` opens Lightning ``.ckpt`` files whose pickles reference classes that are
not installed here (yacs ``CfgNode``, pytorch_lightning containers) by stubbing them.
"""
from __future__ import annotations

import pickle
import types
from collections import OrderedDict

import torch


def strip_lightning_prefix(state_dict, prefix='model.'):
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k[len(prefix):] if k.startswith(prefix) else k] = v
    return out


def load_pretrained_model(model, state_dict, strict=False, overwrite_shape_mismatch=True,
                          remove_lightning=False):
    if remove_lightning:
        state_dict = strip_lightning_prefix(state_dict)
    own = model.state_dict()
    mismatched = [k for k, v in state_dict.items()
                  if k in own and hasattr(v, 'shape') and tuple(own[k].shape) != tuple(v.shape)]
    if not mismatched:
        model.load_state_dict(state_dict, strict=strict)
        return model
    if not overwrite_shape_mismatch:
        raise RuntimeError(f'shape mismatch for {mismatched} and overwrite_shape_mismatch=False')
    patched = OrderedDict(state_dict)
    for k in mismatched:
        src, dst = state_dict[k], own[k]
        if k.endswith('head.fc1.weight') and src.dim() == 2 and src.shape[0] == dst.shape[0]:
            # regressor input grew/shrank by the 7 camera features (2205 <-> 2212 columns)
            if src.shape[1] + 7 == dst.shape[1]:
                patched[k] = torch.cat([src, src[:, -7:]], dim=-1)
                continue
            if src.shape[1] - 7 == dst.shape[1]:
                patched[k] = src[:, :-7]
                continue
        del patched[k]
    model.load_state_dict(patched, strict=False)
    return model


class _AnyStub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)

    def __call__(self, *a, **k):
        return _AnyStub()


class _StubDict(dict):
    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError, ModuleNotFoundError):
            return _StubDict if 'CfgNode' in name or 'Dict' in name else _AnyStub


_tolerant_pickle = types.SimpleNamespace(
    Unpickler=_TolerantUnpickler, load=lambda f, **kw: _TolerantUnpickler(f, **kw).load(),
    __name__='spec_amd_tolerant_pickle')


def read_checkpoint(path, map_location='cpu'):
    """torch.load for trusted reference checkpoints, tolerant of missing third-party classes."""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except Exception:
        return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_tolerant_pickle)
