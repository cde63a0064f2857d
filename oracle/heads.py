"""Oracle (test infrastructure): HMRHead / SMPLCamHead / SMPLHead of the SPEC path.

Restated from the published PARE heads (un-vendored, requirements.txt:28).  Constructor and
forward keyword contracts are the ones the reference uses at ``spec/models/hmr.py:57-74``
and ``:96-120``; output keys are the ones consumed in-tree (``spec/tester.py:166-167``,
``spec/losses.py:172-177``, ``spec/trainer.py:246-254,350-353``).
"""
import numpy as np
import torch
import torch.nn as nn

from .geometry import (rot6d_to_rotmat, rotmat_to_rot6d, convert_pare_to_full_img_cam,
                       perspective_projection, convert_weak_perspective_to_perspective)
from .smpl import SMPLOracle

# assets are injected (synthetic or user supplied); see oracle.models.set_assets
_ASSETS = {'smpl_model': None, 'mean_params': None}


def set_assets(smpl_model=None, mean_params=None):
    if smpl_model is not None:
        _ASSETS['smpl_model'] = smpl_model
    if mean_params is not None:
        _ASSETS['mean_params'] = mean_params


class HMRHead(nn.Module):
    """Iterative error feedback regressor, 3 iterations, no activation between fc1 and fc2.

    fc1 input = [xf(2048), pose6d(144), shape(10), cam(3)] (+ [rot6d(cam_rotmat)(6), vfov(1)]
    when ``use_cam_feats``) - ``spec/models/hmr.py:57-64,94-98``.

    ``estimate_var`` (constructor flags passed at ``spec/models/hmr.py:59-61``; the two extra outputs are consumed by
    ``spec/losses.py:61-62`` as ``pred['pred_pose_var']`` / ``pred['pred_shape_var']`` against 144- / 10-wide targets): the
    decoders also emit one variance per pose / shape number - doubled ``decpose`` / ``decshape`` whose first half is the mean
    update and whose second half is the variance, or separate ``decpose_var`` / ``decshape_var`` layers
    (``use_separate_var_branch``); the variance of the LAST iteration, through ``F.<uncertainty_activation>`` when one is named,
    is returned concatenated behind the final mean.
    """

    def __init__(self, num_input_features, smpl_mean_params=None, estimate_var=False,
                 use_separate_var_branch=False, uncertainty_activation='', backbone='resnet50',
                 use_cam_feats=False):
        super().__init__()
        npose = 24 * 6
        self.npose = npose
        self.estimate_var = estimate_var
        self.use_separate_var_branch = use_separate_var_branch
        self.uncertainty_activation = uncertainty_activation
        self.use_cam_feats = use_cam_feats
        if use_cam_feats:
            num_input_features += 7
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Linear(num_input_features + npose + 13, 1024)
        self.drop1 = nn.Dropout()
        self.fc2 = nn.Linear(1024, 1024)
        self.drop2 = nn.Dropout()
        if estimate_var and use_separate_var_branch:
            self.decpose = nn.Linear(1024, npose)
            self.decshape = nn.Linear(1024, 10)
            self.deccam = nn.Linear(1024, 3)
            self.decpose_var = nn.Linear(1024, npose)
            self.decshape_var = nn.Linear(1024, 10)
        elif estimate_var:
            self.decpose = nn.Linear(1024, npose * 2)       # double the output sizes to estimate var
            self.decshape = nn.Linear(1024, 10 * 2)
            self.deccam = nn.Linear(1024, 3)
        else:
            self.decpose = nn.Linear(1024, npose)
            self.decshape = nn.Linear(1024, 10)
            self.deccam = nn.Linear(1024, 3)
        mp = smpl_mean_params if smpl_mean_params is not None else _ASSETS['mean_params']
        if mp is None:
            mp = {'pose': np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), 24),
                  'shape': np.zeros(10, np.float32), 'cam': np.array([0.9, 0, 0], np.float32)}
        self.register_buffer('init_pose', torch.from_numpy(np.asarray(mp['pose'], np.float32)).reshape(1, -1))
        self.register_buffer('init_shape', torch.from_numpy(np.asarray(mp['shape'], np.float32)).reshape(1, -1))
        self.register_buffer('init_cam', torch.from_numpy(np.asarray(mp['cam'], np.float32)).reshape(1, -1))

    def forward(self, features, init_pose=None, init_shape=None, init_cam=None,
                cam_rotmat=None, cam_vfov=None, n_iter=3):
        batch_size = features.shape[0]
        if init_pose is None:
            init_pose = self.init_pose.expand(batch_size, -1)
        if init_shape is None:
            init_shape = self.init_shape.expand(batch_size, -1)
        if init_cam is None:
            init_cam = self.init_cam.expand(batch_size, -1)
        xf = self.avgpool(features)
        xf = xf.view(xf.size(0), -1)
        pred_pose, pred_shape, pred_cam = init_pose, init_shape, init_cam
        pred_pose_var = pred_shape_var = None
        for _ in range(n_iter):
            if self.use_cam_feats:
                xc = torch.cat([xf, pred_pose, pred_shape, pred_cam,
                                rotmat_to_rot6d(cam_rotmat), cam_vfov.unsqueeze(-1)], 1)
            else:
                xc = torch.cat([xf, pred_pose, pred_shape, pred_cam], 1)
            xc = self.drop1(self.fc1(xc))
            xc = self.drop2(self.fc2(xc))
            if self.estimate_var:
                pred_pose = self.decpose(xc)[:, :self.npose] + pred_pose
                pred_shape = self.decshape(xc)[:, :10] + pred_shape
                pred_cam = self.deccam(xc) + pred_cam
                if self.use_separate_var_branch:
                    pred_pose_var = self.decpose_var(xc)
                    pred_shape_var = self.decshape_var(xc)
                else:
                    pred_pose_var = self.decpose(xc)[:, self.npose:]
                    pred_shape_var = self.decshape(xc)[:, 10:]
                if self.uncertainty_activation != '':
                    act = getattr(torch.nn.functional, self.uncertainty_activation)     # eval(f'F.{name}') upstream
                    pred_pose_var = act(pred_pose_var)
                    pred_shape_var = act(pred_shape_var)
            else:
                pred_pose = self.decpose(xc) + pred_pose
                pred_shape = self.decshape(xc) + pred_shape
                pred_cam = self.deccam(xc) + pred_cam
        pred_rotmat = rot6d_to_rotmat(pred_pose).view(batch_size, 24, 3, 3)
        output = {'pred_pose': pred_rotmat, 'pred_cam': pred_cam, 'pred_shape': pred_shape,
                  'pred_pose_6d': pred_pose}
        if self.estimate_var:
            output.update({'pred_pose_var': torch.cat([pred_pose, pred_pose_var], dim=1),
                           'pred_shape_var': torch.cat([pred_shape, pred_shape_var], dim=1)})
        return output


class SMPLCamHead(nn.Module):
    """SMPL forward + full-image camera + perspective projection (``spec/models/hmr.py:69``)."""

    def __init__(self, img_res=224):
        super().__init__()
        self.smpl = SMPLOracle(_ASSETS['smpl_model'])
        self.img_res = img_res

    def forward(self, rotmat, shape, cam, cam_rotmat, cam_intrinsics, bbox_scale, bbox_center,
                img_w, img_h, normalize_joints2d=False):
        vertices, joints3d = self.smpl(shape, rotmat)
        output = {'smpl_vertices': vertices, 'smpl_joints3d': joints3d}
        cam_t = convert_pare_to_full_img_cam(
            pare_cam=cam, bbox_height=bbox_scale * 200., bbox_center=bbox_center,
            img_w=img_w, img_h=img_h, focal_length=cam_intrinsics[:, 0, 0], crop_res=self.img_res)
        joints2d = perspective_projection(joints3d, rotation=cam_rotmat, translation=cam_t,
                                          cam_intrinsics=cam_intrinsics)
        if normalize_joints2d:
            joints2d = joints2d / (self.img_res / 2.)
        output['smpl_joints2d'] = joints2d
        output['pred_cam_t'] = cam_t
        return output


class SMPLHead(nn.Module):
    """Non-camera variant (``spec/models/hmr.py:71-74,115-120``): weak perspective, R = I."""

    def __init__(self, focal_length=5000., img_res=224):
        super().__init__()
        self.smpl = SMPLOracle(_ASSETS['smpl_model'])
        self.focal_length = focal_length
        self.img_res = img_res

    def forward(self, rotmat, shape, cam=None, normalize_joints2d=False):
        vertices, joints3d = self.smpl(shape, rotmat)
        output = {'smpl_vertices': vertices, 'smpl_joints3d': joints3d}
        if cam is not None:
            B = joints3d.shape[0]
            cam_t = convert_weak_perspective_to_perspective(cam, self.focal_length, self.img_res)
            K = torch.zeros(B, 3, 3)
            K[:, 0, 0] = self.focal_length
            K[:, 1, 1] = self.focal_length
            K[:, 2, 2] = 1.0
            R = torch.eye(3).unsqueeze(0).expand(B, -1, -1)
            joints2d = perspective_projection(joints3d, R, cam_t, K)
            if normalize_joints2d:
                joints2d = joints2d / (self.img_res / 2.)
            output['smpl_joints2d'] = joints2d
            output['pred_cam_t'] = cam_t
        return output
