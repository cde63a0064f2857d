"""Oracle (test infrastructure): evaluation metrics that follow the hot path.

In-tree reference code restated here: ``eval_j_24`` (spec/utils/compute_error.py:33-49),
``eval_single`` (:52-86) and the validation-step variants (spec/trainer.py:272-316).
Un-vendored leaf functions restated from the published SPIN/PARE code
(``pare.utils.eval_utils``): ``reconstruction_error`` (Procrustes / similarity alignment with
NumPy SVD) and ``compute_error_verts`` (mean per-vertex L2).  W-MPJPE for SPEC-SYN is
``eval_j_24`` on ``J_regressor(24xV) @ vertices`` (compute_error.py:184,192,216).
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
