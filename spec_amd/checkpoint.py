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


def _allowed_globals():
    """(module, name) pairs the fallback unpickler resolves for real: what tensors, storages and plain containers are
    rebuilt from - the same set ``torch.load(weights_only=True)`` trusts, spelled out - plus the NumPy array
    reconstructors Lightning checkpoints carry.  No package is trusted wholesale: ``torch.utils.collect_env.run``,
    ``torch.hub.*``, ``numpy.load``, ``builtins.type`` / ``vars`` / ``map`` ... are all OUTSIDE this list and stubbed."""
    allowed = {
        ('collections', 'OrderedDict'), ('collections', 'Counter'), ('collections', 'defaultdict'),
        ('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'bytearray'), ('builtins', 'complex'),
        ('builtins', 'dict'), ('builtins', 'list'), ('builtins', 'tuple'), ('builtins', 'int'), ('builtins', 'float'),
        ('builtins', 'str'), ('builtins', 'bool'), ('builtins', 'bytes'), ('builtins', 'slice'), ('builtins', 'range'),
        ('_codecs', 'encode'),
        ('torch', 'Size'), ('torch', 'Tensor'), ('torch', 'device'), ('torch', 'dtype'),
        ('torch.nn.parameter', 'Parameter'),
        ('torch._utils', '_rebuild_tensor'), ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_tensor_v3'),
        ('torch._utils', '_rebuild_parameter'), ('torch._utils', '_rebuild_parameter_with_state'),
        ('torch._utils', '_rebuild_device_tensor_from_cpu_tensor'), ('torch._utils', '_rebuild_device_tensor_from_numpy'),
        ('torch._tensor', '_rebuild_from_type_v2'),
        ('torch.storage', 'UntypedStorage'), ('torch.storage', 'TypedStorage'), ('torch.storage', '_load_from_bytes'),
        ('torch.serialization', '_get_layout'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'),
        ('numpy.core.multiarray', '_reconstruct'), ('numpy.core.multiarray', 'scalar'),
        ('numpy._core.multiarray', '_reconstruct'), ('numpy._core.multiarray', 'scalar'),
        ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer'),
    }
    allowed.discard(('torch.storage', '_load_from_bytes'))   # unpickles a nested stream with the stock pickle: not trusted
    for t in ('Float', 'Double', 'Half', 'BFloat16', 'Long', 'Int', 'Short', 'Char', 'Byte', 'Bool',
              'ComplexFloat', 'ComplexDouble'):
        allowed.add(('torch', t + 'Storage'))
        allowed.add(('torch', t + 'Tensor'))
    for t in ('float32', 'float64', 'float16', 'bfloat16', 'int64', 'int32', 'int16', 'int8', 'uint8', 'bool',
              'complex64', 'complex128', 'float', 'double', 'half', 'long', 'int', 'short'):
        allowed.add(('torch', t))
    for t in ('float32', 'float64', 'float16', 'int64', 'int32', 'int16', 'int8', 'uint8', 'uint16', 'uint32', 'uint64',
              'bool_', 'longlong', 'intc'):
        allowed.add(('numpy', t))
    return allowed


_ALLOWED = _allowed_globals()


class _TolerantUnpickler(pickle.Unpickler):
    """Fallback unpickler for Lightning checkpoints.  Only the explicit (module, name) pairs of ``_allowed_globals`` -
    tensor / storage / array reconstructors, dtypes and plain containers - are resolved for real; every other global a
    checkpoint names (yacs CfgNode, pytorch_lightning containers, argparse namespaces, user classes, and any callable
    of torch / numpy / builtins that is not on the list) becomes an inert stub whose call returns another stub."""

    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
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
    names a class outside torch's allow-list (``pickle.UnpicklingError``) it is re-read with the allow-list
    unpickler above (explicit reconstructor names only; everything else is stubbed).  I/O errors, truncated or corrupt files propagate instead of triggering the fallback."""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError:
        return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_tolerant_pickle)
