"""Wire / disk formats of the hand-offs around the hot path, kept
interoperable with the reference's scripts even though the fused pipeline no longer needs the
files:

* CamCalib result pickle ``<out>/camcalib/<image name>.pkl`` = joblib dump of
  ``{'vfov', 'f_pix', 'pitch', 'roll'}`` (writer ``scripts/camcalib_demo.py:135-140,174``,
  reader ``spec/utils/cam_params.py:28-35`` which calls ``.item()`` on vfov/pitch/roll);
* SPEC result pickle ``<out>/spec_results/<image stem>.pkl`` = joblib dump of the forward's
  output dict as NumPy arrays (``spec/tester.py:153-163``);
* evaluation dump ``evaluation_results_<ds>.pkl`` (``spec/trainer.py:118-135,348-353,533-536``): the
  accumulated ``pose / shape / cam / vertices`` arrays (+ ``imgname``, ``dataset_name``, per-joint errors), read
  back by ``compute_error`` (``spec/utils/compute_error.py:95-110``).
"""
from __future__ import annotations

import os
from typing import Dict

import joblib
import numpy as np
import torch


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def camcalib_result_path(output_path: str, img_fname: str) -> str:
    return os.path.join(output_path, 'camcalib', os.path.basename(img_fname) + '.pkl')


def write_camcalib_result(output_path: str, img_fname: str, vfov, pitch, roll, orig_img_h) -> str:
    """One image's CamCalib record in the reference's format; f_pix = h/2/tan(vfov/2)."""
    vfov, pitch, roll = (np.float32(_np(v).reshape(())) for v in (vfov, pitch, roll))
    f_pix = float(orig_img_h) / 2. / np.tan(vfov / 2.)
    path = camcalib_result_path(output_path, img_fname)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    joblib.dump({'vfov': np.asarray(vfov), 'f_pix': np.float64(f_pix), 'pitch': np.asarray(pitch),
                 'roll': np.asarray(roll)}, path)
    return path


def read_cam_params(output_path: str, img_fname: str, orig_shape, device='cuda'):
    """Same call signature and return tuple as ``spec/utils/cam_params.py:24-50``:
    (cam_rotmat (3,3), cam_int (3,3), vfov, pitch, roll, focal_length); R and K are built on
    the device by ``specmi_cam_params``."""
    from .cam_utils import cam_params_from_angles
    rec = joblib.load(camcalib_result_path(output_path, img_fname))
    pitch, roll, vfov = rec['pitch'].item(), rec['roll'].item(), rec['vfov'].item()
    f = rec['f_pix']
    R, K = cam_params_from_angles([pitch], [roll], [float(f)], [orig_shape[1]], [orig_shape[0]], device=device)
    return R[0], K[0], vfov, pitch, roll, f


def write_spec_result(output_path: str, img_fname: str, output: Dict[str, torch.Tensor]) -> str:
    ext = img_fname.split('.')[-1]
    path = os.path.join(output_path, 'spec_results', os.path.basename(img_fname).replace(ext, 'pkl'))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    joblib.dump({k: _np(v) for k, v in output.items()}, path)
    return path


class EvalDump:
    """Accumulates what ``validation_step`` stashes in ``self.evaluation_results`` (spec/trainer.py:118-135,
    337-353) and writes ``evaluation_results_<ds>.pkl`` like ``validation_epoch_end`` (:533-536): keys ``imgname``,
    ``dataset_name``, ``mpjpe``, ``pampjpe``, ``mpjpe_24``, ``pampjpe_24`` (per-joint errors, when given) and - with
    ``TESTING.SAVE_RESULTS`` - ``pose`` (N,24,3,3), ``shape`` (N,10), ``cam`` (N,3), ``vertices`` (N,6890,3), the
    array ``compute_error`` reads back (spec/utils/compute_error.py:108)."""
    KEYS = ('pose', 'shape', 'cam', 'vertices')

    def __init__(self):
        self.data = {k: [] for k in self.KEYS}
        self.meta = {'imgname': [], 'dataset_name': []}
        self.errors = {}

    def add(self, pred: Dict[str, torch.Tensor], imgnames=None, dataset_name=None, **per_joint_errors):
        self.data['pose'].append(_np(pred['pred_pose']))
        self.data['shape'].append(_np(pred['pred_shape']))
        self.data['cam'].append(_np(pred['pred_cam']))
        self.data['vertices'].append(_np(pred['smpl_vertices']))
        n = self.data['cam'][-1].shape[0]
        if imgnames is not None:
            self.meta['imgname'] += list(imgnames)
            self.meta['dataset_name'] += [dataset_name] * n
        for k, v in per_joint_errors.items():
            self.errors.setdefault(k, []).append(_np(v))

    def write(self, log_dir: str, dataset_name: str) -> str:
        path = os.path.join(log_dir, f'evaluation_results_{dataset_name}.pkl')
        os.makedirs(log_dir, exist_ok=True)
        out = {k: np.concatenate(v) for k, v in self.data.items() if v}
        out.update({k: np.concatenate(v) for k, v in self.errors.items() if v})
        if self.meta['imgname']:
            out.update({k: np.array(v) for k, v in self.meta.items()})
        joblib.dump(out, path)
        return path
