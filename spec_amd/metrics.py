"""Evaluation metrics on the device (MPJPE / PA-MPJPE / V2V in mm) - the step that follows the
hot path in ``spec/trainer.py:272-316`` and ``spec/utils/compute_error.py:33-86``, computed by
``specmi_eval_mesh`` / ``specmi_eval_joints`` without copying the vertices to the host."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, constants
from .cam_utils import _engine
from .engine import _dev_f32, _ptr


@torch.no_grad()
def eval_single(pred_vertices, gt_vertices, J_regressor, joint_sel=constants.H36M_TO_J14):
    """-> (mpjpe, pampjpe, v2v) device tensors (B,), millimetres.  ``J_regressor`` is (J,V)
    (a leading batch dimension as in the reference's ``J_regressor_batch`` is accepted)."""
    if pred_vertices.device.type != 'cuda':
        raise RuntimeError('spec_amd.metrics needs device tensors (no CPU path)')
    eng = _engine(pred_vertices.device)
    dev = eng.device
    pv, gv = _dev_f32(pred_vertices, dev), _dev_f32(gt_vertices, dev)
    Jr = _dev_f32(J_regressor[0] if J_regressor.dim() == 3 else J_regressor, dev)
    B, V, _ = pv.shape
    sel = torch.as_tensor(list(joint_sel), dtype=torch.int32, device=dev)
    out = torch.empty(3, B, device=dev, dtype=torch.float32)
    _lib.check(eng.h, eng.lib.specmi_eval_mesh(eng.h, _ptr(pv), _ptr(gv), B, V, _ptr(Jr), Jr.shape[0], _ptr(sel),
                                               sel.numel(), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), eng._stream()))
    return out[0], out[1], out[2]


@torch.no_grad()
def eval_j_24(pred_joints, gt_joints):
    """-> (mpjpe, pampjpe) device tensors (B,), millimetres (pelvis = joint 0)."""
    if pred_joints.device.type != 'cuda':
        raise RuntimeError('spec_amd.metrics needs device tensors (no CPU path)')
    eng = _engine(pred_joints.device)
    pj, gj = _dev_f32(pred_joints, eng.device), _dev_f32(gt_joints, eng.device)
    B, J, _ = pj.shape
    out = torch.empty(2, B, device=eng.device, dtype=torch.float32)
    _lib.check(eng.h, eng.lib.specmi_eval_joints(eng.h, _ptr(pj), _ptr(gj), B, J, _ptr(out[0]), _ptr(out[1]), eng._stream()))
    return out[0], out[1]


@torch.no_grad()
def w_mpjpe_24(pred_vertices, gt_vertices, J_regressor24):
    """README metric for SPEC-SYN / SPEC-MTP: eval_j_24 on J_regressor(24xV) @ vertices
    (spec/utils/compute_error.py:184,192,216).  The regression runs in the mesh kernel."""
    return eval_single(pred_vertices, gt_vertices, J_regressor24, joint_sel=range(24))[:2]
