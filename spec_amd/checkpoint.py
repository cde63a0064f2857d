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
    """Deliberate deviations from the PARE function this restates (documented, not silent):

    * only a LEADING ``model.`` is stripped (upstream uses ``str.replace('model.', '')``, which would also mangle a
      key that merely contains ``model.`` further in);
    * the regressor-input patch applies to any key ending in ``head.fc1.weight`` (upstream: the literal
      ``model.head.fc1.weight``), so it also works after the prefix was stripped or for a bare head module;
    * besides upstream's growth case (2205 -> 2212 columns; like upstream, the seven new camera-feature columns are
      filled with COPIES of the last seven source columns - a placeholder initialisation, not a meaningful one), the
      inverse shrink case (2212 -> 2205: drop the camera-feature columns) is handled too;
    * ``strict=True`` keeps the reference's failure mode: after patching, missing / unexpected keys or a shape
      mismatch that cannot be patched raise ``RuntimeError`` (CamCalib loads strictly, scripts/camcalib_demo.py:81).
    """
    if remove_lightning:
        state_dict = strip_lightning_prefix(state_dict)
    own = model.state_dict()
    # a one-element tensor saved as shape (1,) for a 0-d buffer (num_batches_tracked) is the same value: reshape it
    state_dict = OrderedDict((k, v.reshape(own[k].shape) if k in own and hasattr(v, 'shape') and v.numel() == 1
                              and own[k].numel() == 1 else v) for k, v in state_dict.items())
    mismatched = [k for k, v in state_dict.items()
                  if k in own and hasattr(v, 'shape') and tuple(own[k].shape) != tuple(v.shape)]
    if not mismatched:
        model.load_state_dict(state_dict, strict=strict)
        return model
    if not overwrite_shape_mismatch:
        raise RuntimeError(f'shape mismatch for {mismatched} and overwrite_shape_mismatch=False')
    patched = OrderedDict(state_dict)
    dropped = []
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
        dropped.append(k)
    if strict:
        if dropped:
            raise RuntimeError(f'strict load: shape mismatch that cannot be patched for {dropped}')
        model.load_state_dict(patched, strict=True)        # raises on missing / unexpected keys
        return model
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


_ALLOWED_ROOTS = ('torch', 'collections', 'numpy', 'builtins', '_codecs', 'copyreg')
_BLOCKED_BUILTINS = {'eval', 'exec', 'compile', 'open', '__import__', 'getattr', 'setattr', 'delattr', 'input', 'breakpoint'}


class _TolerantUnpickler(pickle.Unpickler):
    """Fallback unpickler for Lightning checkpoints: only torch / collections / numpy (and harmless builtins) are
    resolved for real; every other global a checkpoint names (yacs CfgNode, pytorch_lightning containers, argparse
    namespaces, arbitrary user classes) becomes an inert stub, so nothing outside that allow-list can execute."""

    def find_class(self, module, name):
        root = module.split('.')[0]
        if root in _ALLOWED_ROOTS and not (root == 'builtins' and name in _BLOCKED_BUILTINS):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError, ModuleNotFoundError):
                pass
        return _StubDict if 'CfgNode' in name or 'Dict' in name else _AnyStub


_tolerant_pickle = types.SimpleNamespace(
    Unpickler=_TolerantUnpickler, load=lambda f, **kw: _TolerantUnpickler(f, **kw).load(),
    __name__='spec_amd_tolerant_pickle')


def read_checkpoint(path, map_location='cpu'):
    """torch.load of a reference checkpoint.  First ``weights_only=True``; ONLY when that fails because the pickle
    names a class outside torch's allow-list (``pickle.UnpicklingError``) it is re-read with the restricted stubbing
    unpickler above.  I/O errors, truncated or corrupt files propagate instead of triggering the fallback."""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError:
        return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_tolerant_pickle)
