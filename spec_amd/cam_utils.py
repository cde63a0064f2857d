"""CamCalib decode + CamCalib->SPEC hand-off on the GPU.

Mirrors the whole call surface of ``camcalib/cam_utils.py`` (bin tables :23-63, ``bins2*`` :66-91,
``*2soft_idx`` / ``angle_to_soft_idx`` / ``soft_idx_to_angle`` :94-107, ``get_softargmax`` :110-118,
``convert_preds_to_angles`` :121-145 with the reference's defaults) and ``spec/utils/cam_params.py:24-50``
(R, K construction).  The reductions over the 256 bins (arg-max, soft-arg-max) run on the logits that are
already in HBM (``specmi_camcalib_bins`` / ``specmi_camcalib_decode``); bin tables are host float64 NumPy
arrays and ``bins2*`` return float64 NumPy arrays exactly like the reference.  Host (CPU) logits are moved
to the GPU first - there is no CPU arithmetic path.
"""
from __future__ import annotations

import threading

import numpy as np
import torch

from . import constants as C
from .engine import Engine

_tls = threading.local()


def _engine(device) -> Engine:
    """The parameter-less decode engine of the calling thread on ``device``.  A handle is not thread-safe and is driven from one
    stream at a time (include/specmi.h), so the module-level helpers below keep one per (thread, device) in thread-local storage: worker
    threads of a server loop may call them concurrently, each on its own ``torch.cuda.Stream``, and a thread's handles are
    destroyed with it (tests/test_gpu_threads.py)."""
    dev = torch.device('cuda', device.index if device.index is not None else torch.cuda.current_device())
    engines = getattr(_tls, 'engines', None)
    if engines is None:
        engines = _tls.engines = {}
    if dev not in engines:
        engines[dev] = Engine('camcalib', dev)   # decode needs no parameters
    return engines[dev]


# ---- bin tables (camcalib/cam_utils.py:23-63) -------------------------------------------------------------
def get_bins(minval, maxval, sigma, alpha, beta, kappa):
    """Remember, bin 0 = below value! last bin mean >= maxval  (camcalib/cam_utils.py:23-37; scipy.stats.norm(0,
    sigma).pdf written out: exp(-(x/sigma)^2 / 2) / sqrt(2 pi) / sigma)."""
    x = np.linspace(minval, maxval, 255)
    y = x / sigma
    pdf = np.exp(-y ** 2 / 2.0) / np.sqrt(2 * np.pi) / sigma
    pdf /= (pdf.max())
    pdf *= alpha
    pdf = pdf.max() * beta - pdf
    cumsum = np.cumsum(pdf)
    cumsum = cumsum / cumsum.max() * kappa
    cumsum -= cumsum[pdf.size // 2]
    return cumsum


def _centers(bins):
    c = bins.copy()
    c[:-1] += np.diff(c) / 2
    return np.append(c, bins[-1])


pitch_bins = np.linspace(-0.6, 0.6, 255)
pitch_bins_centers = _centers(pitch_bins)
horizon_bins = np.linspace(-0.5, 1.5, 255)
horizon_bins_centers = _centers(horizon_bins)
roll_bins = get_bins(-np.pi / 6, np.pi / 6, 0.5, 0.04, 1.1, np.pi)
roll_bins_centers = _centers(roll_bins)
vfov_bins = np.linspace(0.2617, 2.1, 255)
vfov_bins_centers = _centers(vfov_bins)
roll_new_bins = np.linspace(-0.6, 0.6, 255)
roll_new_bins_centers = _centers(roll_new_bins)


def _argmax_idx(bins) -> np.ndarray:
    """np.argmax over the last axis, reduced on the GPU (first maximum; index work, bit-exact)."""
    if not isinstance(bins, torch.Tensor):
        bins = torch.as_tensor(np.asarray(bins))
    if not torch.cuda.is_available():
        raise RuntimeError('spec_amd.cam_utils needs an AMD GPU (no CPU path)')
    dev = bins.device if bins.device.type == 'cuda' else torch.device('cuda')
    if bins.dtype == torch.float64:
        # float64 input: order-preserving only if no two distinct doubles collapse to one float; the reference's
        # logits are network outputs (fp32), so refuse instead of silently changing ties
        if not torch.equal(bins.float().double(), bins):
            raise NotImplementedError('bins2*: float64 logits that are not exactly representable in fp32')
    idx, _ = _engine(dev).camcalib_bins(bins.to(dev), argmax=True, soft=False)
    return idx.cpu().numpy().astype(np.int64)


def bins2centers_device(bins, centers) -> torch.Tensor:
    """The ``bins2*`` look-up WITHOUT leaving the device: arg-max bin of (..., nbins) device logits -> float64 centres as a device
    tensor, no host synchronisation (the reference's ``bins2*`` return NumPy arrays - camcalib/cam_utils.py:66-91 - which costs a
    device-to-host copy per call: fine for the demo script, not for a serving loop; ``centers`` = one of the ``*_bins_centers``)."""
    if not isinstance(bins, torch.Tensor) or bins.device.type != 'cuda':
        raise RuntimeError('bins2centers_device needs device logits')
    idx, _ = _engine(bins.device).camcalib_bins(bins, argmax=True, soft=False)
    table = torch.from_numpy(np.ascontiguousarray(centers)).to(bins.device)
    return table[idx.long()]


def bins2horizon(bins):
    return horizon_bins_centers[_argmax_idx(bins)]


def bins2pitch(bins):
    return pitch_bins_centers[_argmax_idx(bins)]


def bins2roll(bins):
    return roll_bins_centers[_argmax_idx(bins)]


def bins2vfov(bins):
    return vfov_bins_centers[_argmax_idx(bins)]


def vfov2soft_idx(angle):
    return angle_to_soft_idx(angle, min=np.min(vfov_bins), max=np.max(vfov_bins))


def pitch2soft_idx(angle):
    return angle_to_soft_idx(angle, min=np.min(pitch_bins), max=np.max(pitch_bins))


def roll2soft_idx(angle):
    return angle_to_soft_idx(angle, min=-0.6, max=0.6)


def soft_idx_to_angle(soft_idx, min, max):
    return (max - min) * ((soft_idx + 1) / 2) + min


def angle_to_soft_idx(angle, min, max):
    return 2 * ((angle - min) / (max - min)) - 1


@torch.no_grad()
def get_softargmax(pred):
    """(N, 256) logits -> (N,) soft-arg-max in [-1, 1] (camcalib/cam_utils.py:110-118), device tensor."""
    dev = pred.device if pred.device.type == 'cuda' else torch.device('cuda')
    _, soft = _engine(dev).camcalib_bins(pred.to(dev), argmax=False, soft=True)
    return soft.reshape(-1)


@torch.no_grad()
def decode_camera(pred_vfov, pred_pitch, pred_roll, img_h=None, img_w=None, angles_out=None):
    """Logits (B,256)x3 [+ full-image sizes] -> dict(vfov, pitch, roll, f_pix, cam_rotmat,
    cam_intrinsics) as device tensors (one kernel launch).  ``angles_out``: optional (vfov, pitch, roll)
    destination tensors with a common stride (columns of a packed record)."""
    if pred_vfov.device.type != 'cuda':
        raise RuntimeError('decode_camera needs device tensors (no CPU path in spec_amd)')
    return _engine(pred_vfov.device).camcalib_decode(pred_vfov, pred_pitch, pred_roll, img_h, img_w, angles_out)


@torch.no_grad()
def convert_preds_to_angles(pred_vfov, pred_pitch, pred_roll, loss_type='kl', return_type='torch', legacy=False):
    """Reference signature, defaults and return types (camcalib/cam_utils.py:121-145): 'kl' / 'ce' -> arg-max bin
    centres (float64 NumPy, converted with torch.from_numpy for return_type='torch'); 'softargmax_l2' /
    'softargmax_biased_l2' -> soft-arg-max angles as fp32 tensors on the logits' device (``legacy``: roll through
    the arg-max table)."""
    if loss_type in ('kl', 'ce'):
        pred_vfov, pred_pitch, pred_roll = bins2vfov(pred_vfov), bins2pitch(pred_pitch), bins2roll(pred_roll)
    elif loss_type in ('softargmax_l2', 'softargmax_biased_l2'):
        if pred_vfov.device.type != 'cuda':
            pred_vfov, pred_pitch, pred_roll = pred_vfov.cuda(), pred_pitch.cuda(), pred_roll.cuda()
        roll_logits = pred_roll
        d = decode_camera(pred_vfov, pred_pitch, pred_roll)
        pred_vfov, pred_pitch, pred_roll = d['vfov'], d['pitch'], d['roll']
        if legacy:
            pred_roll = bins2roll(roll_logits)

    if return_type == 'np' and isinstance(pred_vfov, torch.Tensor):
        return (pred_vfov.cpu().numpy(), pred_pitch.cpu().numpy(),
                pred_roll.cpu().numpy() if isinstance(pred_roll, torch.Tensor) else pred_roll)

    if return_type == 'torch' and isinstance(pred_vfov, np.ndarray):
        return torch.from_numpy(pred_vfov), torch.from_numpy(pred_pitch), torch.from_numpy(pred_roll)

    return pred_vfov, pred_pitch, pred_roll


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
