"""Oracle (test infrastructure): evaluation metrics that follow the hot path.

In-tree reference code restated here: ``eval_j_24`` (spec/utils/compute_error.py:33-49),
``eval_single`` (:52-86) and the validation-step variants (spec/trainer.py:272-316).
Un-vendored leaf functions restated from the published SPIN/PARE code
(``pare.utils.eval_utils``): ``reconstruction_error`` (Procrustes / similarity alignment with
NumPy SVD) and ``compute_error_verts`` (mean per-vertex L2).  W-MPJPE-24 for SPEC-SYN / SPEC-MTP is
``eval_j_24(J_regressor(24xV) @ pred_vertices, body_model_orig(gt pose, gt shape).joints[:, :24])``
(compute_error.py:156-160,184,192,216) - regressed prediction against kinematic-chain ground truth;
``compute_error`` below restates the whole function (:89-223) on arrays instead of files.
"""
import numpy as np
import torch


def compute_similarity_transform(S1, S2):
    """(sR, t) taking S1 (N,3) closest to S2 (N,3) in the least-squares sense; returns S1_hat."""
    transposed = False
    if S1.shape[0] != 3 and S1.shape[0] != 2:
        S1, S2 = S1.T, S2.T
        transposed = True
    mu1 = S1.mean(axis=1, keepdims=True)
    mu2 = S2.mean(axis=1, keepdims=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = np.sum(X1 ** 2)
    K = X1.dot(X2.T)
    U, s, Vh = np.linalg.svd(K)
    V = Vh.T
    Z = np.eye(U.shape[0])
    Z[-1, -1] *= np.sign(np.linalg.det(U.dot(V.T)))
    R = V.dot(Z.dot(U.T))
    scale = np.trace(R.dot(K)) / var1
    t = mu2 - scale * (R.dot(mu1))
    S1_hat = scale * R.dot(S1) + t
    return S1_hat.T if transposed else S1_hat


def reconstruction_error(S1, S2, reduction='mean'):
    S1_hat = np.stack([compute_similarity_transform(a, b) for a, b in zip(S1, S2)])
    re_per_joint = np.sqrt(((S1_hat - S2) ** 2).sum(axis=-1))
    re = re_per_joint.mean(axis=-1)
    if reduction == 'mean':
        re = re.mean()
    elif reduction == 'sum':
        re = re.sum()
    return re, re_per_joint


def compute_error_verts(pred_verts, target_verts=None, target_theta=None):
    return np.sqrt(np.sum((target_verts - pred_verts) ** 2, axis=2)).mean(axis=1)


H36M_TO_J14 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10]


def eval_j_24(pred_joints, gt_joints):
    """spec/utils/compute_error.py:33-49."""
    pred_joints = pred_joints - pred_joints[:, [0], :].clone()
    gt_joints = gt_joints - gt_joints[:, [0], :].clone()
    pampjpe, _ = reconstruction_error(pred_joints.cpu().numpy(), gt_joints.cpu().numpy(), reduction=None)
    pampjpe = pampjpe * 1000
    mpjpe = torch.sqrt(((pred_joints - gt_joints) ** 2).sum(dim=-1)).mean(dim=-1).cpu().numpy() * 1000
    return mpjpe, pampjpe


def eval_single(pred_vertices, gt_vertices, J_regressor_batch, joint_sel=H36M_TO_J14):
    """spec/utils/compute_error.py:52-86."""
    pred_joints = torch.matmul(J_regressor_batch, pred_vertices)
    pred_pelvis = pred_joints[:, [0], :].clone()
    pred_joints = pred_joints[:, joint_sel, :] - pred_pelvis
    gt_joints = torch.matmul(J_regressor_batch, gt_vertices)
    gt_pelvis = gt_joints[:, [0], :].clone()
    gt_joints = gt_joints[:, joint_sel, :] - gt_pelvis
    v2v = compute_error_verts(pred_verts=(pred_vertices - pred_pelvis).cpu().numpy(),
                              target_verts=(gt_vertices - gt_pelvis).cpu().numpy()) * 1000
    pampjpe, _ = reconstruction_error(pred_joints.cpu().numpy(), gt_joints.cpu().numpy(), reduction=None)
    pampjpe = pampjpe * 1000
    mpjpe = torch.sqrt(((pred_joints - gt_joints) ** 2).sum(dim=-1)).mean(dim=-1).cpu().numpy() * 1000
    return mpjpe, pampjpe, v2v


def compute_error(pred_vertices, data, dataset_name, smpl_oracle, J_regressor_h36m, pred_cam_rotmat=None, num_chunks=100):
    """spec/utils/compute_error.py:89-223 with the file reads replaced by arguments: ``data`` = the annotation
    arrays, ``smpl_oracle`` = oracle.smpl.SMPLOracle (stands for both ``SMPL`` and ``SMPLorig``: the reference only
    uses ``.vertices`` of the first and ``.joints[:, :24]`` / ``.J_regressor`` of the second).  Returns the per-sample
    error arrays keyed like the reference's local variables."""
    n = len(data['imgname'])
    pose_key = 'pose_0yaw_inverseyz' if dataset_name.startswith('3dpw') else 'pose'
    pred_vertices = torch.as_tensor(np.asarray(pred_vertices)).float()
    J_regressor = torch.as_tensor(np.asarray(J_regressor_h36m)).float()
    J_regressor_batch = J_regressor[None, :].expand(1, -1, -1)
    out = {k: np.zeros(n) for k in ('wv2v', 'v2v', 'wmpjpe', 'mpjpe', 'pampjpe', 'pampjpe_24', 'wmpjpe_24', 'mpjpe_24')}
    for idx in np.array_split(np.arange(n), min(num_chunks, n)):
        if idx.size == 0:
            continue
        gt_pose = torch.from_numpy(data[pose_key][idx]).float()
        gt_betas = torch.from_numpy(data['shape'][idx]).float()
        gt_vertices, j45 = smpl_oracle.native_axis_angle(gt_betas, gt_pose)
        gt_joints = j45[:, :24]
        if dataset_name == 'spec-syn':
            gt_cam_rotmat = torch.from_numpy(data['cam_rotmat'][idx]).float()
            gt_cam_vertices = torch.bmm(gt_cam_rotmat, gt_vertices.transpose(2, 1)).transpose(2, 1)
            gt_cam_joints = torch.bmm(gt_cam_rotmat, gt_joints.transpose(2, 1)).transpose(2, 1)
            pred_cam_rotmat_ = gt_cam_rotmat
        else:
            pred_cam_rotmat_ = pred_cam_rotmat[idx].float()
            gt_pose_cam = torch.from_numpy(data['pose_cam'][idx]).float()
            gt_cam_vertices, j45c = smpl_oracle.native_axis_angle(gt_betas, gt_pose_cam)
            gt_cam_joints = j45c[:, :24]
        pred_verts = pred_vertices[idx]
        pred_joints = torch.einsum('bik,ji->bjk', [pred_verts, smpl_oracle.J_regressor])
        pred_vertices_gt_cam = torch.bmm(pred_cam_rotmat_, pred_verts.transpose(2, 1)).transpose(2, 1)
        pred_cam_joints = torch.einsum('bik,ji->bjk', [pred_vertices_gt_cam, smpl_oracle.J_regressor])
        wmpjpe, pampjpe, wv2v = eval_single(pred_verts, gt_vertices, J_regressor_batch)
        mpjpe, _, v2v = eval_single(pred_vertices_gt_cam, gt_cam_vertices, J_regressor_batch)
        wmpjpe_24, pampjpe_24 = eval_j_24(pred_joints, gt_joints)
        mpjpe_24, _ = eval_j_24(pred_cam_joints, gt_cam_joints)
        for k, v in (('wv2v', wv2v), ('v2v', v2v), ('wmpjpe', wmpjpe), ('mpjpe', mpjpe), ('pampjpe', pampjpe),
                     ('pampjpe_24', pampjpe_24), ('wmpjpe_24', wmpjpe_24), ('mpjpe_24', mpjpe_24)):
            out[k][idx] = v
    return out
