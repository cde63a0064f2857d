"""Evaluation on the device - the step that follows the hot path in the reference:
``spec/trainer.py:272-316`` (validation step) and ``spec/utils/compute_error.py`` (``eval_j_24`` :33-49,
``eval_single`` :52-86, ``compute_error`` :89-223 - the numbers of the README table, ``README.md:155-159``).

Everything numeric runs in libspecmi on the GPU (joint regression, rotations, pelvis alignment, MPJPE, Procrustes,
V2V, the ground-truth SMPL meshes through ``specmi_smpl_native``); only per-image scalars come back to the host.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import _lib, constants
from .cam_utils import _engine
from .engine import Engine, _dev_f32, _ptr


@torch.no_grad()
def eval_single(pred_vertices, gt_vertices, J_regressor, joint_sel=constants.H36M_TO_J14):
    """spec/utils/compute_error.py:52-86 -> (mpjpe, pampjpe, v2v) device tensors (B,), millimetres.  Both joint sets
    are regressed from the vertices with ``J_regressor`` (J,V) (the reference passes the H36M regressor; a leading
    batch dimension as in ``J_regressor_batch`` is accepted), pelvis = regressed joint 0, ``joint_sel`` = H36M_TO_J14."""
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
    """spec/utils/compute_error.py:33-49 -> (mpjpe, pampjpe) device tensors (B,), millimetres (pelvis = joint 0)."""
    if pred_joints.device.type != 'cuda':
        raise RuntimeError('spec_amd.metrics needs device tensors (no CPU path)')
    eng = _engine(pred_joints.device)
    pj, gj = _dev_f32(pred_joints, eng.device), _dev_f32(gt_joints, eng.device)
    B, J, _ = pj.shape
    out = torch.empty(2, B, device=eng.device, dtype=torch.float32)
    _lib.check(eng.h, eng.lib.specmi_eval_joints(eng.h, _ptr(pj), _ptr(gj), B, J, _ptr(out[0]), _ptr(out[1]), eng._stream()))
    return out[0], out[1]


@torch.no_grad()
def regress_joints(vertices, J_regressor):
    """``torch.einsum('bik,ji->bjk', vertices, J_regressor)`` (compute_error.py:184,187) on the device."""
    eng = _engine(vertices.device)
    v = _dev_f32(vertices, eng.device)
    Jr = _dev_f32(J_regressor, eng.device)
    B, V, _ = v.shape
    out = torch.empty(B, Jr.shape[0], 3, device=eng.device, dtype=torch.float32)
    _lib.check(eng.h, eng.lib.specmi_regress_joints(eng.h, _ptr(v), B, V, _ptr(Jr), Jr.shape[0], _ptr(out), eng._stream()))
    return out


@torch.no_grad()
def rotate_points(R, points):
    """``torch.bmm(R, x.transpose(2,1)).transpose(2,1)`` (compute_error.py:164-165,186) on the device."""
    eng = _engine(points.device)
    x = _dev_f32(points, eng.device)
    B, N, _ = x.shape
    Rm = _dev_f32(R, eng.device, (B, 3, 3))
    out = torch.empty_like(x)
    _lib.check(eng.h, eng.lib.specmi_rotate_points(eng.h, _ptr(Rm), _ptr(x), B, N, _ptr(out), eng._stream()))
    return out


@torch.no_grad()
def w_mpjpe_24(pred_vertices, gt_joints_24, J_regressor24):
    """The README metric of SPEC-SYN / SPEC-MTP (compute_error.py:156-160,184,192,216): prediction = 24 joints
    REGRESSED from the predicted vertices with the SMPL ``J_regressor``; ground truth = the 24 posed
    kinematic-chain joints of the body model (``body_model_orig(...).joints[:, :24]``, e.g. ``BodyModel.native``
    of the GT pose / shape) - NOT joints regressed from GT vertices.  -> (w_mpjpe_24, pa_mpjpe_24) in mm."""
    return eval_j_24(regress_joints(pred_vertices, J_regressor24), gt_joints_24)


class BodyModel:
    """The SMPL body model alone on the GPU - what the evaluation code instantiates as ``SMPL`` / ``SMPLorig``
    (compute_error.py:118-130) and the trainer as ``smpl_native`` (spec/trainer.py:79-86).  ``model`` is the tensor
    dict of ``spec_amd.assets`` (loaded from the licensed SMPL file, or synthetic)."""

    def __init__(self, model: Optional[dict] = None, device='cuda'):
        from . import assets
        self.model = model if model is not None else assets.smpl_model()
        dev = torch.device(device)
        self.engine = Engine('smpl', dev if dev.index is not None else torch.device('cuda', torch.cuda.current_device()))
        self.engine.load({}, smpl=self.model)
        self.J_regressor = torch.from_numpy(np.ascontiguousarray(self.model['J_regressor'])).float().to(self.engine.device)

    @torch.no_grad()
    def native(self, pose, betas, vertices=True, joints24=True):
        """pose: (B,72) axis-angle [global_orient | body_pose] or (B,24,3,3) rotation matrices; returns
        (vertices (B,V,3), joints24 (B,24,3)) like ``smplx.SMPL(...)(...)`` ``.vertices`` / ``.joints[:, :24]``."""
        return self.engine.smpl_native(pose, betas, vertices, joints24)


def _load_pkl(path):
    import joblib
    return joblib.load(path)


@torch.no_grad()
def compute_error(results_file, dataset_file=None, data_root='.', body_model: Optional[BodyModel] = None,
                  J_regressor_h36m=None, pred_cam_rotmat=None, log=print, num_chunks=100):
    """``spec/utils/compute_error.py:89-223`` on the GPU.  ``results_file`` = ``evaluation_results_<dataset>.pkl``
    (``{'vertices': ...}``, spec/trainer.py:348-353); the dataset annotations (``imgname``, ``pose`` /
    ``pose_0yaw_inverseyz``, ``shape``, ``cam_rotmat`` | ``pose_cam``), ``data/J_regressor_h36m.npy`` and
    ``data/camcalib/<dataset>_cam_rotmat.pkl`` default to the reference's relative paths (spec/config.py:34-56) under
    ``data_root``.  Prints the reference's log lines and returns the per-sample arrays + means."""
    dataset_name = os.path.basename(results_file).replace('evaluation_results_', '').replace('.pkl', '')
    files = {'spec-mtp': 'data/dataset_folders/spec-mtp/annotations/test.npz',
             'spec-syn': 'data/dataset_folders/spec-syn/annotations/test.npz',
             '3dpw-test-cam': 'data/dataset_extras/3dpw_test_0yaw_inverseyz_w_camcalib.npz'}
    if dataset_file is None:
        dataset_file = os.path.join(data_root, files[dataset_name])
    results = _load_pkl(results_file)
    data = np.load(dataset_file)
    pose_key = 'pose_0yaw_inverseyz' if dataset_name.startswith('3dpw') else 'pose'
    pred_vertices = np.asarray(results['vertices'], dtype=np.float32)
    del results
    n = len(data['imgname'])
    if pred_vertices.shape[0] != n:
        raise ValueError(f'{results_file}: {pred_vertices.shape[0]} predictions for {n} annotations')
    if pred_cam_rotmat is None and dataset_name != 'spec-syn':
        pred_cam_rotmat = _load_pkl(os.path.join(data_root, f'data/camcalib/{dataset_name}_cam_rotmat.pkl'))
    if pred_cam_rotmat is not None and not isinstance(pred_cam_rotmat, torch.Tensor):
        pred_cam_rotmat = torch.as_tensor(np.asarray(pred_cam_rotmat))
    if J_regressor_h36m is None:
        J_regressor_h36m = np.load(os.path.join(data_root, 'data/J_regressor_h36m.npy'))
    body = body_model if body_model is not None else BodyModel()
    dev = body.engine.device
    Jh = torch.as_tensor(np.asarray(J_regressor_h36m), dtype=torch.float32).to(dev)
    J24 = body.J_regressor
    keys = ('wv2v', 'v2v', 'wmpjpe', 'mpjpe', 'pampjpe', 'pampjpe_24', 'wmpjpe_24', 'mpjpe_24')
    acc = {k: np.zeros(n) for k in keys}
    f = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).to(dev)
    for idx in np.array_split(np.arange(n), min(num_chunks, n)):
        if idx.size == 0:
            continue
        gt_pose, gt_betas = f(data[pose_key][idx]), f(data['shape'][idx])
        gt_vertices, gt_joints = body.native(gt_pose, gt_betas)
        if dataset_name == 'spec-syn':
            R = f(data['cam_rotmat'][idx])
            gt_cam_vertices, gt_cam_joints = rotate_points(R, gt_vertices), rotate_points(R, gt_joints)
            pred_R = R
        else:
            gt_cam_vertices, gt_cam_joints = body.native(f(data['pose_cam'][idx]), gt_betas)
            pred_R = pred_cam_rotmat[idx].float().to(dev)
        pred_verts = f(pred_vertices[idx])
        pred_joints = regress_joints(pred_verts, J24)
        pred_vertices_gt_cam = rotate_points(pred_R, pred_verts)
        pred_cam_joints = regress_joints(pred_vertices_gt_cam, J24)
        wmpjpe, pampjpe, wv2v = eval_single(pred_verts, gt_vertices, Jh)
        mpjpe, _, v2v = eval_single(pred_vertices_gt_cam, gt_cam_vertices, Jh)
        wmpjpe_24, pampjpe_24 = eval_j_24(pred_joints, gt_joints)
        mpjpe_24, _ = eval_j_24(pred_cam_joints, gt_cam_joints)
        for k, v in (('wv2v', wv2v), ('v2v', v2v), ('wmpjpe', wmpjpe), ('mpjpe', mpjpe), ('pampjpe', pampjpe),
                     ('pampjpe_24', pampjpe_24), ('wmpjpe_24', wmpjpe_24), ('mpjpe_24', mpjpe_24)):
            acc[k][idx] = v.cpu().numpy()
    m = {k: float(v.mean()) for k, v in acc.items()}
    log(f'***** RESULTS ON {dataset_name.upper()} *****')
    if dataset_name == '3dpw-test-cam':          # standard protocol for 3dpw is 14 joint evaluation
        log(f'W-MPJPE: {m["wmpjpe"]:.3f}')
        log(f'C-MPJPE: {m["wmpjpe"]:.3f}')
        log(f'MPJPE: {m["mpjpe"]:.3f}')
        log(f'PA-MPJPE: {m["pampjpe"]:.3f}')
    else:                                        # 24 SMPL joints for SPEC-SYN and SPEC-MTP
        log(f'W-MPJPE-24: {m["wmpjpe_24"]:.3f}')
        log(f'C-MPJPE-24: {m["wmpjpe_24"]:.3f}')
        log(f'MPJPE-24: {m["mpjpe_24"]:.3f}')
        log(f'PA-MPJPE-24: {m["pampjpe_24"]:.3f}')
    log(f'W-V2V: {m["wv2v"]:.3f}')
    log(f'C-V2V: {m["wv2v"]:.3f}')
    log(f'V2V: {m["v2v"]:.3f}')
    return {'mean': m, 'per_sample': acc, 'dataset': dataset_name}
