"""Thin object wrapper over a libspecmi handle: parameter upload and forward calls with torch
device tensors (torch is plumbing here: HBM allocations + the current HIP stream)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from . import _lib

KIND = {'camcalib': _lib.MODEL_CAMCALIB, 'hmr': _lib.MODEL_HMR, 'smpl': _lib.MODEL_SMPL}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev_f32(t, device, shape=None):
    """-> contiguous fp32 tensor on `device` (accepts tensors / arrays / scalars), or None."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t, dtype=np.float32))
    t = t.to(device=device, dtype=torch.float32).contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f'expected shape {tuple(shape)}, got {tuple(t.shape)}')
    return t


class Engine:
    def __init__(self, kind: str, device: torch.device):
        if device.type != 'cuda':
            raise RuntimeError('spec_amd runs on an AMD GPU (torch device "cuda"); there is no CPU path')
        self.lib = _lib.load()
        self.kind = kind
        self.device = torch.device('cuda', device.index if device.index is not None else torch.cuda.current_device())
        h = C.c_void_p()
        rc = self.lib.specmi_create(C.byref(h), self.device.index, KIND[kind])
        if rc != _lib.OK:
            raise _lib.SpecmiError(rc, (self.lib.specmi_last_error(None) or b'?').decode())
        self.h = h
        self.nbins = 256
        self.feat_channels = 2048
        self.num_verts = 0
        self._ld_opts = {}        # current value of the stride options ("output_ld", "angle_ld") on the handle

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.specmi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters ----------------------------------------------------------------------
    def set_option(self, name: str, value):
        if isinstance(value, float):
            _lib.check(self.h, self.lib.specmi_set_option_f32(self.h, name.encode(), value))
        else:
            _lib.check(self.h, self.lib.specmi_set_option_i32(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        """The effective value of an integer option on this handle (what was set, else the library's default)."""
        v = C.c_int(0)
        _lib.check(self.h, self.lib.specmi_get_option_i32(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def experimental(self, on: bool = True):
        """Let this handle accept the experimental option names (include/specmi.h): tuning thresholds, debug pins, opt-ins."""
        self.set_option('experimental', int(bool(on)))
        return self

    PLAN_NAMES = ('throughput', 'latency', 'single')

    def trunk_plan(self, B: int, H: int = 224, W: int = 224, pair: bool = False) -> str:
        """The execution plan a trunk forward of (B, 3, H, W) takes under the current options ('throughput' | 'latency' | 'single')."""
        mode = C.c_int32(0)
        _lib.check(self.h, self.lib.specmi_trunk_plan(self.h, int(B), int(H), int(W), int(bool(pair)), C.byref(mode)))
        return self.PLAN_NAMES[int(mode.value)]

    def sync_status(self) -> int:
        """Synchronises; 0 = every in-launch hand-off of this handle's persistent launches completed (include/specmi.h)."""
        err = C.c_int32(0)
        _lib.check(self.h, self.lib.specmi_sync_status(self.h, C.byref(err)))
        return int(err.value)

    def sync_reset(self):
        _lib.check(self.h, self.lib.specmi_sync_reset(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def debug_poison_sync(self, value: int = 0xDEADBEEF):
        _lib.check(self.h, self.lib.specmi_debug_poison_sync(self.h, C.c_uint32(value)))

    def set_tensor(self, name: str, value):
        if isinstance(value, torch.Tensor):
            value = value.detach().cpu().numpy()
        a = np.asarray(value)
        if a.dtype.kind in 'iu' or a.dtype == np.bool_:
            a = np.ascontiguousarray(a, dtype=np.int32)
            fn = self.lib.specmi_set_tensor_i32
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            fn = self.lib.specmi_set_tensor_f32
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        _lib.check(self.h, fn(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))

    def load(self, tensors: Mapping[str, object], smpl: Optional[Mapping[str, object]] = None, **options):
        """Stage a state_dict-like mapping (+ the SMPL body model under 'smpl.*'), then commit."""
        for k, v in options.items():
            self.set_option(k, v)
        for k, v in tensors.items():
            if k.endswith('num_batches_tracked'):
                continue
            self.set_tensor(k, v)
        if smpl is not None:
            for k, v in smpl.items():
                self.set_tensor('smpl.' + k, v)
            self.num_verts = int(np.asarray(smpl['v_template']).shape[0]) if not isinstance(
                smpl['v_template'], torch.Tensor) else int(smpl['v_template'].shape[0])
        if 'fc_vfov.weight' in tensors:
            self.nbins = int(tensors['fc_vfov.weight'].shape[0])
        else:   # Sequential heads (num_fc_layers > 1): the last Linear of the chain has the bins
            last = [k for k in tensors if k.startswith('fc_vfov.') and k.endswith('.weight')]
            if last:
                self.nbins = int(tensors[sorted(last)[-1]].shape[0])
        self.feat_channels = {18: 512, 34: 512, 32: 480, 48: 720}.get(int(options.get('backbone', 50)), 2048)
        _lib.check(self.h, self.lib.specmi_commit(self.h))

    def _set_ld(self, name: str, value: int):
        if self._ld_opts.get(name, 0) != value:
            self.set_option(name, int(value))
            self._ld_opts[name] = value

    # ---- packed per-image record ------------------------------------------------
    def record_layout(self):
        """[(key, offset, per-image shape)] of the packed record the kernels can write directly:
        85,164 B of SPEC outputs + 12 B of camera angles = 21,294 floats for V = 6890."""
        lay, off = [], 0
        for k, shp in (('smpl_vertices', (self.num_verts, 3)), ('smpl_joints3d', (49, 3)), ('smpl_joints2d', (49, 2)),
                       ('pred_cam_t', (3,)), ('pred_pose', (24, 3, 3)), ('pred_cam', (3,)), ('pred_shape', (10,)),
                       ('pred_pose_6d', (144,)), ('cam_vfov', ()), ('cam_pitch', ()), ('cam_roll', ())):
            n = int(np.prod(shp)) if shp else 1
            lay.append((k, off, shp))
            off += n
        return lay, off

    def record_views(self, record: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Views of a (B, record_floats) record as the output dict (no copy)."""
        lay, total = self.record_layout()
        if record.dim() != 2 or record.shape[1] != total or record.dtype != torch.float32 or record.stride(1) != 1:
            raise ValueError(f'record must be a (B, {total}) fp32 tensor with unit column stride')
        B = record.shape[0]
        return {k: record[:, off:off + (int(np.prod(shp)) if shp else 1)].view(B, *shp) if shp
                else record[:, off] for k, off, shp in lay}

    # ---- forward ---------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _images(self, images):
        if not isinstance(images, torch.Tensor) or images.device.type != 'cuda':
            raise RuntimeError('images must be a device tensor (no CPU path in spec_amd)')
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f'images must be (B,3,H,W), got {tuple(images.shape)}')
        return images.to(device=self.device, dtype=torch.float32).contiguous()

    def trunk(self, images):
        x = self._images(images)
        B, _, H, W = x.shape

        def o(n, k, s, p):
            return (n + 2 * p - k) // s + 1
        fh, fw = H, W
        fh, fw = o(fh, 7, 2, 3), o(fw, 7, 2, 3)
        fh, fw = o(fh, 3, 2, 1), o(fw, 3, 2, 1)
        for _ in range(3):
            fh, fw = o(fh, 3, 2, 1), o(fw, 3, 2, 1)
        feat = torch.empty(B, fh, fw, self.feat_channels, device=self.device, dtype=torch.float32)
        if B == 0:
            return feat
        _lib.check(self.h, self.lib.specmi_trunk_forward(self.h, _ptr(x), B, H, W, _ptr(feat), self._stream()))
        return feat

    def _feat_shape(self, H, W):
        def o(n, k, s_, p):
            return (n + 2 * p - k) // s_ + 1
        fh, fw = o(H, 7, 2, 3), o(W, 7, 2, 3)
        for _ in range(4):
            fh, fw = o(fh, 3, 2, 1), o(fw, 3, 2, 1)
        return fh, fw

    def trunk_pair(self, other: 'Engine', images, other_images):
        """The ResNet trunks of this engine and of ``other`` (same depth, same input shape) in lockstep, every layer of both
        as one grouped launch (``specmi_trunk_forward_pair``) -> (features of self, features of other), NHWC."""
        xa, xb = self._images(images), other._images(other_images)
        if xa.shape != xb.shape:
            raise ValueError(f'grouped trunk launches need equal input shapes, got {tuple(xa.shape)} and {tuple(xb.shape)}')
        B, _, H, W = xa.shape
        fh, fw = self._feat_shape(H, W)
        fa = torch.empty(B, fh, fw, self.feat_channels, device=self.device, dtype=torch.float32)
        fb = torch.empty(B, fh, fw, other.feat_channels, device=self.device, dtype=torch.float32)
        if B == 0:
            return fa, fb
        _lib.check(self.h, self.lib.specmi_trunk_forward_pair(self.h, other.h, _ptr(xa), _ptr(xb), B, H, W, _ptr(fa), _ptr(fb),
                                                              self._stream()))
        return fa, fb

    def camcalib_head(self, feat_nhwc):
        """avg-pool + the three Linear chains of CameraRegressorNetwork.forward from a trunk feature map."""
        f = _dev_f32(feat_nhwc, self.device)
        B, fh, fw, _ = f.shape
        out = torch.empty(3, B, self.nbins, device=self.device, dtype=torch.float32)
        if B == 0:
            return [out[0], out[1], out[2]]
        _lib.check(self.h, self.lib.specmi_camcalib_head_forward(self.h, _ptr(f), B, fh, fw, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                                                 self._stream()))
        return [out[0], out[1], out[2]]

    def camcalib_forward(self, images):
        x = self._images(images)
        B, _, H, W = x.shape
        out = torch.empty(3, B, self.nbins, device=self.device, dtype=torch.float32)
        if B == 0:                      # an empty batch gives empty outputs, as the reference's torch modules do
            return [out[0], out[1], out[2]]
        _lib.check(self.h, self.lib.specmi_camcalib_forward(
            self.h, _ptr(x), B, H, W, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), self._stream()))
        return [out[0], out[1], out[2]]

    def camcalib_head_decode(self, feat_nhwc, img_h=None, img_w=None, angles_out=None):
        """``camcalib_head`` + ``camcalib_decode`` as one call (one launch at small batches: ``specmi_camcalib_head_decode``) ->
        ([logits_vfov, logits_pitch, logits_roll], dict(vfov, pitch, roll, f_pix, cam_rotmat, cam_intrinsics))."""
        f = _dev_f32(feat_nhwc, self.device)
        B, fh, fw, _ = f.shape
        out = torch.empty(3, B, self.nbins, device=self.device, dtype=torch.float32)
        img_h = _dev_f32(img_h, self.device, (B,))
        img_w = _dev_f32(img_w, self.device, (B,))
        mk = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        if angles_out is not None:
            vf, pt, rl = angles_out
            ld = vf.stride(0) if B > 1 else 1
            if any(t.shape != (B,) or t.dtype != torch.float32 or (B > 1 and t.stride(0) != ld) for t in (vf, pt, rl)):
                raise ValueError('angles_out: three (B,) fp32 tensors with one common stride')
            self._set_ld('angle_ld', ld if ld != 1 else 0)
        else:
            vf, pt, rl = mk(B), mk(B), mk(B)
            self._set_ld('angle_ld', 0)
        fp = mk(B) if img_h is not None else None
        R = mk(B, 3, 3)
        K = mk(B, 3, 3) if (img_h is not None and img_w is not None) else None
        if B > 0:
            _lib.check(self.h, self.lib.specmi_camcalib_head_decode(
                self.h, _ptr(f), B, fh, fw, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(img_h), _ptr(img_w), _ptr(vf), _ptr(pt),
                _ptr(rl), _ptr(fp), _ptr(R), _ptr(K), self._stream()))
        return [out[0], out[1], out[2]], {'vfov': vf, 'pitch': pt, 'roll': rl, 'f_pix': fp, 'cam_rotmat': R, 'cam_intrinsics': K}

    def camcalib_decode(self, lv, lp, lr, img_h=None, img_w=None, angles_out=None):
        """``angles_out``: optional (vfov, pitch, roll) tensors of shape (B,) with a common element stride
        (e.g. three columns of the packed record) that the kernel writes directly."""
        lv, lp, lr = (_dev_f32(t, self.device) for t in (lv, lp, lr))
        B, nb = lv.shape
        img_h = _dev_f32(img_h, self.device, (B,))
        img_w = _dev_f32(img_w, self.device, (B,))
        mk = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        if angles_out is not None:
            vf, pt, rl = angles_out
            ld = vf.stride(0) if B > 1 else 1
            if any(t.shape != (B,) or t.dtype != torch.float32 or (B > 1 and t.stride(0) != ld) for t in (vf, pt, rl)):
                raise ValueError('angles_out: three (B,) fp32 tensors with one common stride')
            self._set_ld('angle_ld', ld if ld != 1 else 0)
        else:
            vf, pt, rl = mk(B), mk(B), mk(B)
            self._set_ld('angle_ld', 0)
        f = mk(B) if img_h is not None else None
        R = mk(B, 3, 3)
        K = mk(B, 3, 3) if (img_h is not None and img_w is not None) else None
        _lib.check(self.h, self.lib.specmi_camcalib_decode(
            self.h, _ptr(lv), _ptr(lp), _ptr(lr), B, nb, _ptr(img_h), _ptr(img_w), _ptr(vf), _ptr(pt),
            _ptr(rl), _ptr(f), _ptr(R), _ptr(K), self._stream()))
        return {'vfov': vf, 'pitch': pt, 'roll': rl, 'f_pix': f, 'cam_rotmat': R, 'cam_intrinsics': K}

    def camcalib_bins(self, logits, argmax=True, soft=False):
        """Per-row argmax (int32) and / or normalised soft-argmax of (..., nbins) device logits."""
        x = _dev_f32(logits, self.device)
        nb = x.shape[-1]
        rows = x.numel() // nb
        idx = torch.empty(x.shape[:-1], device=self.device, dtype=torch.int32) if argmax else None
        sf = torch.empty(x.shape[:-1], device=self.device, dtype=torch.float32) if soft else None
        _lib.check(self.h, self.lib.specmi_camcalib_bins(self.h, _ptr(x), rows, nb, _ptr(idx), _ptr(sf), self._stream()))
        return idx, sf

    def _cam_args(self, B, cam_rotmat, cam_intrinsics, bbox_scale, bbox_center, img_w, img_h):
        d = self.device
        return (_dev_f32(cam_rotmat, d, (B, 3, 3)), _dev_f32(cam_intrinsics, d, (B, 3, 3)),
                _dev_f32(bbox_scale, d, (B,)), _dev_f32(bbox_center, d, (B, 2)),
                _dev_f32(img_w, d, (B,)), _dev_f32(img_h, d, (B,)))

    def _hmr_outputs(self, B) -> Dict[str, torch.Tensor]:
        mk = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        return {'smpl_vertices': mk(B, self.num_verts, 3), 'smpl_joints3d': mk(B, 49, 3),
                'smpl_joints2d': mk(B, 49, 2), 'pred_cam_t': mk(B, 3), 'pred_pose': mk(B, 24, 3, 3),
                'pred_cam': mk(B, 3), 'pred_shape': mk(B, 10), 'pred_pose_6d': mk(B, 144)}

    def _out_for(self, B, record):
        """Output dict + the handle's output stride: dense tensors, or views of a (B, record_floats) record the
        kernels write in place."""
        if record is None:
            self._set_ld('output_ld', 0)
            return None
        if record.shape[0] != B or record.device != self.device:
            raise ValueError('record must be a (B, record_floats) tensor on the model device')
        views = self.record_views(record)
        self._set_ld('output_ld', record.stride(0))
        return views

    def hmr_forward(self, images, cam_rotmat=None, cam_intrinsics=None, bbox_scale=None,
                    bbox_center=None, img_w=None, img_h=None, out: Optional[Dict[str, torch.Tensor]] = None,
                    record: Optional[torch.Tensor] = None):
        x = self._images(images)
        B, _, H, W = x.shape
        R, K, sc, ce, iw, ih = self._cam_args(B, cam_rotmat, cam_intrinsics, bbox_scale, bbox_center, img_w, img_h)
        views = self._out_for(B, record)
        out = views if views is not None else (out if out is not None else self._hmr_outputs(B))
        if B == 0:
            return out
        o = _lib.HmrOutputs(**{k: out[k].data_ptr() for k, _ in _lib.HmrOutputs._fields_})
        _lib.check(self.h, self.lib.specmi_hmr_forward(
            self.h, _ptr(x), B, H, W, _ptr(R), _ptr(K), _ptr(sc), _ptr(ce), _ptr(iw), _ptr(ih),
            C.byref(o), self._stream()))
        return out

    def hmr_regress(self, feat_nhwc, cam_rotmat=None, cam_intrinsics=None, bbox_scale=None, bbox_center=None,
                    img_w=None, img_h=None, record: Optional[torch.Tensor] = None):
        """Regressor head + SMPL head from a trunk feature map (hmr.py:94-122), one library call."""
        f = _dev_f32(feat_nhwc, self.device)
        B, fh, fw, _ = f.shape
        R, K, sc, ce, iw, ih = self._cam_args(B, cam_rotmat, cam_intrinsics, bbox_scale, bbox_center, img_w, img_h)
        views = self._out_for(B, record)
        out = views if views is not None else self._hmr_outputs(B)
        if B == 0:                      # empty batch: empty outputs, like hmr_forward / trunk
            return out
        o = _lib.HmrOutputs(**{k: out[k].data_ptr() for k, _ in _lib.HmrOutputs._fields_})
        _lib.check(self.h, self.lib.specmi_hmr_regress(
            self.h, _ptr(f), B, fh, fw, _ptr(R), _ptr(K), _ptr(sc), _ptr(ce), _ptr(iw), _ptr(ih),
            C.byref(o), self._stream()))
        return out

    def hmr_uncertainty(self, B: int):
        """(pred_pose_var (B, 288), pred_shape_var (B, 20)) of a model committed with ``estimate_var``: call right after the head
        forward of the same batch on the same stream (``specmi_hmr_uncertainty``)."""
        pv = torch.empty(B, 288, device=self.device, dtype=torch.float32)
        sv = torch.empty(B, 20, device=self.device, dtype=torch.float32)
        if B > 0:
            _lib.check(self.h, self.lib.specmi_hmr_uncertainty(self.h, int(B), _ptr(pv), _ptr(sv), self._stream()))
        return pv, sv

    def hmr_head(self, feat_nhwc, cam_rotmat=None, cam_intrinsics=None, img_h=None, record=None):
        f = _dev_f32(feat_nhwc, self.device)
        B, fh, fw, _ = f.shape
        R = _dev_f32(cam_rotmat, self.device, (B, 3, 3))
        K = _dev_f32(cam_intrinsics, self.device, (B, 3, 3))
        ih = _dev_f32(img_h, self.device, (B,))
        mk = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        views = self._out_for(B, record)
        out = ({k: views[k] for k in ('pred_pose', 'pred_shape', 'pred_cam', 'pred_pose_6d')} if views is not None else
               {'pred_pose': mk(B, 24, 3, 3), 'pred_shape': mk(B, 10), 'pred_cam': mk(B, 3), 'pred_pose_6d': mk(B, 144)})
        if B == 0:
            return out
        _lib.check(self.h, self.lib.specmi_hmr_head_forward(
            self.h, _ptr(f), B, fh, fw, _ptr(R), _ptr(K), _ptr(ih), _ptr(out['pred_pose']),
            _ptr(out['pred_shape']), _ptr(out['pred_cam']), _ptr(out['pred_pose_6d']), self._stream()))
        return out

    def smpl(self, rotmat, betas, cam, cam_rotmat=None, cam_intrinsics=None, bbox_scale=None,
             bbox_center=None, img_w=None, img_h=None, record=None):
        rot = _dev_f32(rotmat, self.device)
        B = rot.shape[0]
        rot = rot.reshape(B, 24, 3, 3).contiguous()
        be = _dev_f32(betas, self.device, (B, 10))
        cm = _dev_f32(cam, self.device, (B, 3))
        R, K, sc, ce, iw, ih = self._cam_args(B, cam_rotmat, cam_intrinsics, bbox_scale, bbox_center, img_w, img_h)
        mk = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        views = self._out_for(B, record)
        out = ({k: views[k] for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t')}
               if views is not None else
               {'smpl_vertices': mk(B, self.num_verts, 3), 'smpl_joints3d': mk(B, 49, 3),
                'smpl_joints2d': mk(B, 49, 2), 'pred_cam_t': mk(B, 3)})
        if B == 0:
            return out
        _lib.check(self.h, self.lib.specmi_smpl_forward(
            self.h, _ptr(rot), _ptr(be), _ptr(cm), B, _ptr(R), _ptr(K), _ptr(sc), _ptr(ce), _ptr(iw),
            _ptr(ih), _ptr(out['smpl_vertices']), _ptr(out['smpl_joints3d']), _ptr(out['smpl_joints2d']),
            _ptr(out['pred_cam_t']), self._stream()))
        return out

    def smpl_native(self, pose, betas, vertices=True, joints24=True):
        """smplx-style body model call: ``pose`` (B,24,3,3) rotation matrices or (B,72) axis-angle; returns
        (vertices (B,V,3) | None, joints24 (B,24,3) | None)."""
        p = _dev_f32(pose, self.device)
        B = p.shape[0]
        aa = p.dim() == 2
        if aa and p.shape[1] != 72 or not aa and tuple(p.shape[1:]) != (24, 3, 3):
            raise ValueError(f'pose must be (B,72) axis-angle or (B,24,3,3) rotation matrices, got {tuple(p.shape)}')
        be = _dev_f32(betas, self.device, (B, 10))
        mk = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        v = mk(B, self.num_verts, 3) if vertices else None
        j = mk(B, 24, 3) if joints24 else None
        self._set_ld('output_ld', 0)
        _lib.check(self.h, self.lib.specmi_smpl_native(self.h, _ptr(p), int(aa), _ptr(be), B, _ptr(v), _ptr(j),
                                                       self._stream()))
        return v, j

    def conv2d(self, x, w_oihw, scale, shift, stride, pad, residual=None, relu=True, nchw_input=False):
        """Single fused layer (tests).  x NHWC device tensor (NCHW for the 7x7 stem)."""
        x = _dev_f32(x, self.device)
        w = np.ascontiguousarray(np.asarray(w_oihw, dtype=np.float32))
        sc = np.ascontiguousarray(np.asarray(scale, dtype=np.float32))
        sh = np.ascontiguousarray(np.asarray(shift, dtype=np.float32))
        cout, cin, kh, kw = w.shape
        if nchw_input:
            B, _, H, W = x.shape
        else:
            B, H, W, _ = x.shape
        oh, ow = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        out = torch.empty(B, oh, ow, cout, device=self.device, dtype=torch.float32)
        res = _dev_f32(residual, self.device, (B, oh, ow, cout))
        _lib.check(self.h, self.lib.specmi_conv2d(
            self.h, _ptr(x), B, H, W, cin, w.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
            sh.ctypes.data_as(C.c_void_p), cout, kh, kw, stride, pad, _ptr(res), int(relu), _ptr(out),
            self._stream()))
        return out

    def maxpool(self, x):
        x = _dev_f32(x, self.device)
        B, H, W, Cc = x.shape
        out = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc, device=self.device, dtype=torch.float32)
        _lib.check(self.h, self.lib.specmi_maxpool3x3s2(self.h, _ptr(x), B, H, W, Cc, _ptr(out), self._stream()))
        return out

    def avgpool(self, x):
        x = _dev_f32(x, self.device)
        B, H, W, Cc = x.shape
        out = torch.empty(B, Cc, device=self.device, dtype=torch.float32)
        _lib.check(self.h, self.lib.specmi_avgpool(self.h, _ptr(x), B, H * W, Cc, _ptr(out), self._stream()))
        return out

    # ---- profiling -------------------------------------------------------------------------
    def profile(self, on: bool):
        _lib.check(self.h, self.lib.specmi_profile_enable(self.h, int(on)))

    def profile_read(self, max_entries: int = 512):
        arr = (_lib.ProfEntry * max_entries)()
        n = C.c_int(0)
        _lib.check(self.h, self.lib.specmi_profile_read(self.h, arr, max_entries, C.byref(n)))
        return [{'kernel': arr[i].kernel.decode(), 'label': arr[i].label.decode(), 'ms': arr[i].ms,
                 'flops': arr[i].flops, 'bytes': arr[i].bytes, 'launches': arr[i].launches}
                for i in range(min(n.value, max_entries))]
