"""CamCalib decode + CamCalib->SPEC hand-off on the GPU.

Mirrors the call surface of ``camcalib/cam_utils.py:110-145`` (``convert_preds_to_angles``,
soft-argmax branch) and ``spec/utils/cam_params.py:24-50`` (R, K construction) but runs as one
fused HIP launch (``specmi_camcalib_decode``) on the logits that are already in HBM, instead
of the reference's per-image subprocess + pickle hand-off (``spec/tester.py:86-88``).
"""
from __future__ import annotations

import torch

from . import constants as C
from .engine import Engine

_engines = {}


def _engine(device) -> Engine:
    dev = torch.device('cuda', device.index if device.index is not None else torch.cuda.current_device())
    if dev not in _engines:
        _engines[dev] = Engine('camcalib', dev)   # decode needs no parameters
    return _engines[dev]


def soft_idx_to_angle(soft_idx, min, max):
    return (max - min) * ((soft_idx + 1) / 2) + min


def angle_to_soft_idx(angle, min, max):
    return 2 * ((angle - min) / (max - min)) - 1


@torch.no_grad()
def decode_camera(pred_vfov, pred_pitch, pred_roll, img_h=None, img_w=None):
    """Logits (B,256)x3 [+ full-image sizes] -> dict(vfov, pitch, roll, f_pix, cam_rotmat,
    cam_intrinsics) as device tensors (one kernel launch)."""
    if pred_vfov.device.type != 'cuda':
        raise RuntimeError('decode_camera needs device tensors (no CPU path in spec_amd)')
    return _engine(pred_vfov.device).camcalib_decode(pred_vfov, pred_pitch, pred_roll, img_h, img_w)


@torch.no_grad()
def convert_preds_to_angles(pred_vfov, pred_pitch, pred_roll, loss_type='softargmax_l2',
                            return_type='torch', legacy=False):
    """Reference signature (camcalib/cam_utils.py:121).  Only the soft-argmax, non-legacy branch
    used by the released model (scripts/camcalib_demo.py:74-78,227) is built."""
    if loss_type not in ('softargmax_l2', 'softargmax_biased_l2') or legacy:
        raise NotImplementedError('only the soft-argmax (non-legacy) decode of the released CamCalib model')
    d = decode_camera(pred_vfov, pred_pitch, pred_roll)
    out = (d['vfov'], d['pitch'], d['roll'])
    if return_type == 'np':
        return tuple(t.cpu().numpy() for t in out)
    return out


@torch.no_grad()
def cam_params_from_angles(pitch, roll, f_pix, img_w, img_h, device='cuda'):
    """(pitch, roll, f_pix, img_w, img_h) (B,) -> cam_rotmat (B,3,3), cam_intrinsics (B,3,3) on the
    device - the R / K construction of ``spec/utils/cam_params.py:37-48`` (K[2,2] = 0)."""
    from . import _lib
    from .engine import _dev_f32, _ptr
    dev = torch.device(device)
    eng = _engine(dev if dev.type == 'cuda' else torch.device('cuda'))
    args = [_dev_f32(torch.as_tensor(a, dtype=torch.float32).reshape(-1), eng.device) for a in (pitch, roll, f_pix, img_w, img_h)]
    B = args[0].shape[0]
    R = torch.empty(B, 3, 3, device=eng.device, dtype=torch.float32)
    K = torch.empty(B, 3, 3, device=eng.device, dtype=torch.float32)
    _lib.check(eng.h, eng.lib.specmi_cam_params(eng.h, *[_ptr(a) for a in args], B, _ptr(R), _ptr(K), eng._stream()))
    return R, K


VFOV_RANGE, PITCH_RANGE, ROLL_RANGE = C.VFOV_RANGE, C.PITCH_RANGE, C.ROLL_RANGE
