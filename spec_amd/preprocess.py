"""Pre-processing on the device.

Crop + normalise: the detection loop of ``spec/tester.py:116-128``
(``get_single_image_crop_demo`` per bbox, ``bbox_scale = bbox[2]/200``, ``bbox_center``) as one
HIP launch (``specmi_crop_normalize``) from a uint8 RGB frame that already sits in HBM."""
from __future__ import annotations

import torch

from . import _lib
from .cam_utils import _engine
from .engine import _dev_f32, _ptr


@torch.no_grad()
def crop_detections(frame_rgb_u8, dets, scale: float = 1.0, crop_size: int = 224, return_raw: bool = False, out=None):
    """frame (H,W,3) uint8 device tensor, dets (n,4) [cx, cy, w, h] ->
    dict(inp_images (n,3,S,S) fp32, bbox_scale (n,), bbox_center (n,2)[, raw (n,S,S,3) uint8]).
    ``out``: optional dict of preallocated contiguous tensors (``inp_images``, ``bbox_scale``, ``bbox_center`` - e.g. slices
    of a larger batch buffer) to write into instead of allocating."""
    if not isinstance(frame_rgb_u8, torch.Tensor) or frame_rgb_u8.device.type != 'cuda':
        raise RuntimeError('crop_detections needs a device tensor (no CPU path in spec_amd)')
    if frame_rgb_u8.dtype != torch.uint8 or frame_rgb_u8.dim() != 3 or frame_rgb_u8.shape[2] != 3:
        raise ValueError('frame must be (H,W,3) uint8 RGB')
    eng = _engine(frame_rgb_u8.device)
    dev = eng.device
    frame = frame_rgb_u8.contiguous()
    boxes = _dev_f32(dets, dev)
    if boxes.dim() != 2 or boxes.shape[1] != 4:
        raise ValueError('dets must be (n,4) [cx, cy, w, h]')
    n, (H, W) = boxes.shape[0], frame.shape[:2]
    img, sc, ce = _crop_outputs(out, n, crop_size, dev)
    raw = torch.empty(n, crop_size, crop_size, 3, device=dev, dtype=torch.uint8) if return_raw else None
    if n > 0:
        _lib.check(eng.h, eng.lib.specmi_crop_normalize(eng.h, _ptr(frame), H, W, _ptr(boxes), n, float(scale), crop_size,
                                                        _ptr(img), _ptr(raw), _ptr(sc), _ptr(ce), eng._stream()))
    res = {'inp_images': img, 'bbox_scale': sc, 'bbox_center': ce}
    if return_raw:
        res['raw'] = raw
    return res


def _crop_outputs(out, n, crop_size, dev):
    if out is None:
        return (torch.empty(n, 3, crop_size, crop_size, device=dev, dtype=torch.float32),
                torch.empty(n, device=dev, dtype=torch.float32), torch.empty(n, 2, device=dev, dtype=torch.float32))
    img, sc, ce = out['inp_images'], out['bbox_scale'], out['bbox_center']
    for t_, shp in ((img, (n, 3, crop_size, crop_size)), (sc, (n,)), (ce, (n, 2))):
        if tuple(t_.shape) != shp or t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != dev:
            raise ValueError(f'out tensor must be a contiguous fp32 device tensor of shape {shp}, got {tuple(t_.shape)} {t_.dtype}')
    return img, sc, ce


@torch.no_grad()
def crop_detections_batch(frames_u8, frame_index, dets, scale: float = 1.0, crop_size: int = 224, out=None):
    """The detections of MANY equal-sized frames in one launch (``specmi_crop_normalize_batch``): ``frames_u8`` (F,H,W,3) uint8
    device slab, ``frame_index`` (n,) int32 (which frame each detection belongs to), ``dets`` (n,4) [cx, cy, w, h] -> the same
    dict as ``crop_detections`` for all n crops, bit-identical to cutting them frame by frame."""
    if not isinstance(frames_u8, torch.Tensor) or frames_u8.device.type != 'cuda':
        raise RuntimeError('crop_detections_batch needs a device tensor (no CPU path in spec_amd)')
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3 or not frames_u8.is_contiguous():
        raise ValueError('frames must be a contiguous (F,H,W,3) uint8 RGB slab')
    eng = _engine(frames_u8.device)
    dev = eng.device
    boxes = _dev_f32(dets, dev)
    if boxes.dim() != 2 or boxes.shape[1] != 4:
        raise ValueError('dets must be (n,4) [cx, cy, w, h]')
    n = boxes.shape[0]
    fidx = frame_index if isinstance(frame_index, torch.Tensor) else torch.as_tensor(frame_index)
    F, H, W = frames_u8.shape[:3]
    if fidx.device.type == 'cpu' and fidx.numel():
        # still on the host: check the range here (the kernel clamps what reaches it - a wrong crop instead of an
        # out-of-bounds read - but cannot report it)
        lo, hi = int(fidx.min()), int(fidx.max())
        if lo < 0 or hi >= F:
            raise ValueError(f'frame_index values must lie in [0, {F}), got [{lo}, {hi}]')
    fidx = fidx.to(device=dev, dtype=torch.int32).contiguous()
    if fidx.shape != (n,):
        raise ValueError('frame_index must have one entry per detection')
    img, sc, ce = _crop_outputs(out, n, crop_size, dev)
    if n > 0:
        _lib.check(eng.h, eng.lib.specmi_crop_normalize_batch(eng.h, _ptr(frames_u8), F, H, W, _ptr(fidx), _ptr(boxes), n,
                                                              float(scale), crop_size, _ptr(img), None, _ptr(sc), _ptr(ce),
                                                              eng._stream()))
    return {'inp_images': img, 'bbox_scale': sc, 'bbox_center': ce}


def pare_crop_boxes(centers, scales, res: int = 224):
    """Integer crop boxes of pare's ``crop(img, center, scale, [res, res])`` (SPIN / PARE image_utils): ``ul = transform([1, 1],
    ..., invert=1) - 1``, ``br = transform([res + 1, res + 1], ..., invert=1) - 1`` with ``transform`` = 3x3 float64 matrix
    (``h = 200 * scale``), ``np.linalg.inv``, ``.astype(int)`` (truncation toward zero) - restated with the same NumPy calls.
    -> (n,4) int32 [ul_x, ul_y, br_x, br_y]."""
    import numpy as np
    out = []
    for c, sc in zip(np.asarray(centers, dtype=np.float64).reshape(-1, 2), np.asarray(scales, dtype=np.float64).reshape(-1)):
        h = 200 * sc
        t = np.zeros((3, 3))
        t[0, 0] = float(res) / h
        t[1, 1] = float(res) / h
        t[0, 2] = res * (-float(c[0]) / h + .5)
        t[1, 2] = res * (-float(c[1]) / h + .5)
        t[2, 2] = 1
        ti = np.linalg.inv(t)

        def tr(pt):
            new_pt = np.dot(ti, np.array([pt[0] - 1, pt[1] - 1, 1.]).T)
            return new_pt[:2].astype(int) + 1
        ul = np.array(tr([1, 1])) - 1
        br = np.array(tr([res + 1, res + 1])) - 1
        out.append([ul[0], ul[1], br[0], br[1]])
    return np.asarray(out, dtype=np.int32).reshape(-1, 4)


@torch.no_grad()
def dataset_crops(frame_rgb_u8, centers, scales, crop_size: int = 224):
    """The evaluation dataset's image path on the device (spec/dataset/cam_dataset.py:253-287,367-377): pare ``crop`` (integer
    box copy + cv2.resize bilinear) + clip + ``/ 255`` + ImageNet Normalize.  frame (H,W,3) uint8 device tensor, centers (n,2),
    scales (n,) (bbox height / 200) -> (n,3,S,S) fp32."""
    if not isinstance(frame_rgb_u8, torch.Tensor) or frame_rgb_u8.device.type != 'cuda':
        raise RuntimeError('dataset_crops needs a device tensor (no CPU path in spec_amd)')
    if frame_rgb_u8.dtype != torch.uint8 or frame_rgb_u8.dim() != 3 or frame_rgb_u8.shape[2] != 3:
        raise ValueError('frame must be (H,W,3) uint8 RGB')
    eng = _engine(frame_rgb_u8.device)
    frame = frame_rgb_u8.contiguous()
    boxes = torch.from_numpy(pare_crop_boxes(centers, scales, crop_size)).to(eng.device)
    n, (H, W) = boxes.shape[0], frame.shape[:2]
    out = torch.empty(n, 3, crop_size, crop_size, device=eng.device, dtype=torch.float32)
    _lib.check(eng.h, eng.lib.specmi_crop_resize_normalize(eng.h, _ptr(frame), H, W, _ptr(boxes), n, crop_size, _ptr(out),
                                                           eng._stream()))
    return out


def resize_output_size(w: int, h: int, min_size: int = 600):
    """``torchvision.transforms.Resize(min_size)`` geometry: shorter side -> min_size, longer ->
    ``int(min_size * long / short)``.  Returns (ow, oh)."""
    if w <= h:
        return min_size, int(min_size * h / w)
    return int(min_size * w / h), min_size


@torch.no_grad()
def camcalib_transform(frame_rgb_u8, min_size: int = 600, return_raw: bool = False):
    """The CamCalib demo's ``ImageFolder`` transform (``camcalib/pano_dataset.py:156-162``): Resize(600) of
    the PIL image (Pillow's antialiased bilinear), ToTensor, ImageNet Normalize - one HIP launch
    (``specmi_resize_normalize``), bit-identical to Pillow + torchvision.  frame (H,W,3) uint8 device tensor ->
    (1,3,oh,ow) fp32 [, (oh,ow,3) uint8]."""
    if not isinstance(frame_rgb_u8, torch.Tensor) or frame_rgb_u8.device.type != 'cuda':
        raise RuntimeError('camcalib_transform needs a device tensor (no CPU path in spec_amd)')
    if frame_rgb_u8.dtype != torch.uint8 or frame_rgb_u8.dim() != 3 or frame_rgb_u8.shape[2] != 3:
        raise ValueError('frame must be (H,W,3) uint8 RGB')
    eng = _engine(frame_rgb_u8.device)
    frame = frame_rgb_u8.contiguous()
    H, W = frame.shape[:2]
    ow, oh = resize_output_size(W, H, min_size)
    out = torch.empty(1, 3, oh, ow, device=eng.device, dtype=torch.float32)
    raw = torch.empty(oh, ow, 3, device=eng.device, dtype=torch.uint8) if return_raw else None
    _lib.check(eng.h, eng.lib.specmi_resize_normalize(eng.h, _ptr(frame), H, W, oh, ow, _ptr(out), _ptr(raw), eng._stream()))
    return (out, raw) if return_raw else out


@torch.no_grad()
def camcalib_transform_batch(frames_u8, min_size: int = 600, out=None):
    """``camcalib_transform`` for a slab of F equal-sized frames: (F,H,W,3) uint8 device -> (F,3,oh,ow) fp32, one
    ``specmi_resize_normalize`` launch per frame into ONE batch tensor (CamCalib then runs once on all F frames instead of once
    per frame, ``scripts/camcalib_demo.py:95-102``); each frame's pixels are bit-identical to the single-frame call."""
    if not isinstance(frames_u8, torch.Tensor) or frames_u8.device.type != 'cuda':
        raise RuntimeError('camcalib_transform_batch needs a device tensor (no CPU path in spec_amd)')
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3 or not frames_u8.is_contiguous():
        raise ValueError('frames must be a contiguous (F,H,W,3) uint8 RGB slab')
    eng = _engine(frames_u8.device)
    F, H, W = frames_u8.shape[:3]
    ow, oh = resize_output_size(W, H, min_size)
    if out is None:
        out = torch.empty(F, 3, oh, ow, device=eng.device, dtype=torch.float32)
    elif tuple(out.shape) != (F, 3, oh, ow) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError(f'out must be a contiguous (F,3,{oh},{ow}) fp32 tensor')
    for f in range(F):
        _lib.check(eng.h, eng.lib.specmi_resize_normalize(eng.h, _ptr(frames_u8[f]), H, W, oh, ow, _ptr(out[f]), None, eng._stream()))
    return out
