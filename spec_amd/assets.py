"""Body-model / mean-parameter assets for the SPEC head.

The reference reads ``data/body_models/smpl`` (licence-gated SMPL pickle),
``data/smpl_mean_params.npz`` and ``data/J_regressor_extra.npy`` at construction time
(``spec/config.py:35-37``, used inside the un-vendored ``pare`` heads).  Those files cannot
ship, so the asset source is explicit:

* ``use_synthetic_assets(seed)``  - seeded SMPL-shaped tensors (tests, benchmark);
* ``load_assets(smpl_path, mean_params_path, j_regressor_extra_path)`` - user supplied files
  (SMPL ``.pkl`` with chumpy/scipy-sparse members or ``.npz``), no chumpy needed;
* otherwise, the reference's default relative paths are tried and a clear error is raised.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Optional

import numpy as np

from . import constants as C

_STATE: Dict[str, Optional[dict]] = {'smpl': None, 'mean': None}

DEFAULT_SMPL_DIR = 'data/body_models/smpl'
DEFAULT_MEAN_PARAMS = 'data/smpl_mean_params.npz'
DEFAULT_J_EXTRA = 'data/J_regressor_extra.npy'


def use_synthetic_assets(seed: int = 1003, mean_params: Optional[dict] = None):
    from . import synth
    _STATE['smpl'] = synth.smpl_model(seed)
    _STATE['mean'] = mean_params or {
        'pose': np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), 24),
        'shape': np.zeros(10, np.float32), 'cam': np.array([0.9, 0., 0.], np.float32)}
    return _STATE['smpl']


class _Stub:
    """Placeholder for classes referenced by SMPL pickles (chumpy.Ch, scipy sparse)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'state': state})


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith('numpy') or module.startswith('scipy') or module in ('builtins', 'collections', '_codecs'):
            return super().find_class(module, name)
        if module.startswith('chumpy'):
            return _Stub
        raise pickle.UnpicklingError(f'refusing to load {module}.{name}')


def _dense(a):
    if isinstance(a, _Stub):
        for key in ('x', 'r', 'state'):
            if key in a.__dict__:
                return np.asarray(a.__dict__[key])
        raise ValueError('cannot densify chumpy object')
    if hasattr(a, 'toarray'):
        return np.asarray(a.toarray())
    return np.asarray(a)


def load_smpl_file(path: str, j_regressor_extra: Optional[np.ndarray] = None) -> dict:
    """SMPL .pkl / .npz -> the tensor dict libspecmi expects (smplx.SMPL buffer layouts)."""
    if os.path.isdir(path):
        for cand in ('SMPL_NEUTRAL.pkl', 'SMPL_NEUTRAL.npz', 'basicModel_neutral_lbs_10_207_0_v1.0.0.pkl'):
            if os.path.exists(os.path.join(path, cand)):
                path = os.path.join(path, cand)
                break
    if path.endswith('.npz'):
        try:
            raw = dict(np.load(path, allow_pickle=False))     # plain arrays only: no pickle execution
        except ValueError as e:
            raise ValueError(f'{path}: object (pickled) members are not loaded from .npz files; convert the model to '
                             f'plain arrays or use the .pkl (read through a restricted unpickler)') from e
    else:
        with open(path, 'rb') as f:
            raw = _RestrictedUnpickler(f, encoding='latin1').load()
    v_template = _dense(raw['v_template']).astype(np.float32)
    nv = v_template.shape[0]
    shapedirs = _dense(raw['shapedirs']).astype(np.float32)[:, :, :C.NUM_BETAS]
    posedirs = _dense(raw['posedirs']).astype(np.float32)            # (V,3,207)
    posedirs = posedirs.reshape(nv * 3, -1).T.copy()                 # smplx: (207, 3V)
    parents = _dense(raw['kintree_table'])[0].astype(np.int64).copy()
    parents[0] = -1
    ids = np.array(C.SMPL_EXTRA_VERTEX_IDS, dtype=np.int32)
    if j_regressor_extra is None:
        if os.path.exists(DEFAULT_J_EXTRA):
            j_regressor_extra = np.load(DEFAULT_J_EXTRA)
        else:
            raise FileNotFoundError(f'J_regressor_extra not given and {DEFAULT_J_EXTRA} missing')
    return {
        'v_template': v_template, 'shapedirs': np.ascontiguousarray(shapedirs), 'posedirs': posedirs,
        'J_regressor': _dense(raw['J_regressor']).astype(np.float32),
        'lbs_weights': _dense(raw['weights']).astype(np.float32),
        'J_regressor_extra': np.asarray(j_regressor_extra, dtype=np.float32),
        'parents': parents.astype(np.int32), 'extra_vertex_ids': ids,
        'joint_map': np.array(C.JOINT_MAP49, dtype=np.int32),
    }


def load_assets(smpl_path: str = DEFAULT_SMPL_DIR, mean_params_path: str = DEFAULT_MEAN_PARAMS,
                j_regressor_extra_path: str = DEFAULT_J_EXTRA):
    jx = np.load(j_regressor_extra_path)
    _STATE['smpl'] = load_smpl_file(smpl_path, jx)
    mp = np.load(mean_params_path)
    _STATE['mean'] = {'pose': mp['pose'].astype(np.float32), 'shape': mp['shape'].astype(np.float32),
                      'cam': mp['cam'].astype(np.float32)}
    return _STATE['smpl']


def _ensure():
    if _STATE['smpl'] is None:
        if os.path.exists(DEFAULT_SMPL_DIR) and os.path.exists(DEFAULT_MEAN_PARAMS):
            load_assets()
        else:
            raise FileNotFoundError(
                'SMPL assets not configured: call spec_amd.assets.load_assets(...) with your licensed '
                f'SMPL files (defaults {DEFAULT_SMPL_DIR}, {DEFAULT_MEAN_PARAMS}, {DEFAULT_J_EXTRA}) or '
                'spec_amd.assets.use_synthetic_assets() for synthetic tensors.')


def smpl_model() -> dict:
    _ensure()
    return _STATE['smpl']


def mean_params() -> dict:
    _ensure()
    return _STATE['mean']
