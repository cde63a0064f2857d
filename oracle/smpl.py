"""Oracle (test infrastructure): SMPL linear blend skinning, restating smplx 0.1.28
(``lbs``, ``batch_rigid_transform``, ``VertexJointSelector``; un-vendored dependency,
reference requirements.txt:7) and the SPIN/PARE 49-joint wrapper (requirements.txt:28).

In-tree evidence that constrains this contract: ``spec/trainer.py:71-86,249-254``
(``pose2rot=False``, ``create_transl=False``, ``joints[:, :24]`` of the native model),
``spec/constants.py:87-105`` (JOINT_MAP indexes up to 53 = 24 + 21 + 9 - 1),
``spec/utils/compute_error.py:184`` (``J_regressor @ vertices``).

Two implementations: ``SMPLOracle`` (torch fp32, same op order as upstream) and
``smpl_forward_f64`` (NumPy float64, used to budget rounding error).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def blend_shapes(betas, shape_disps):
    return torch.einsum('bl,mkl->bmk', [betas, shape_disps])


def vertices2joints(J_regressor, vertices):
    return torch.einsum('bik,ji->bjk', [vertices, J_regressor])


def transform_mat(R, t):
    return torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def batch_rigid_transform(rot_mats, joints, parents):
    joints = torch.unsqueeze(joints, dim=-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] -= joints[:, parents[1:]]
    transforms_mat = transform_mat(rot_mats.reshape(-1, 3, 3),
                                   rel_joints.reshape(-1, 3, 1)).reshape(-1, joints.shape[1], 4, 4)
    chain = [transforms_mat[:, 0]]
    for i in range(1, parents.shape[0]):
        chain.append(torch.matmul(chain[parents[i]], transforms_mat[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = F.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - F.pad(torch.matmul(transforms, joints_homogen),
                                        [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """smplx.lbs.batch_rodrigues (0.1.28): (N,3) axis-angle -> (N,3,3)."""
    batch_size = rot_vecs.shape[0]
    dtype = rot_vecs.dtype
    angle = torch.norm(rot_vecs + epsilon, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((batch_size, 1), dtype=dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((batch_size, 3, 3))
    ident = torch.eye(3, dtype=dtype).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def lbs_rotmat(betas, rot_mats, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """``smplx.lbs.lbs(..., pose2rot=False)``: rot_mats (B,24,3,3) -> verts (B,V,3), joints (B,24,3)."""
    batch_size = max(betas.shape[0], rot_mats.shape[0])
    dtype, device = betas.dtype, betas.device
    v_shaped = v_template + blend_shapes(betas, shapedirs)
    J = vertices2joints(J_regressor, v_shaped)
    ident = torch.eye(3, dtype=dtype, device=device)
    pose_feature = rot_mats[:, 1:].view(batch_size, -1, 3, 3) - ident
    rot_mats = rot_mats.view(batch_size, -1, 3, 3)
    pose_offsets = torch.matmul(pose_feature.view(batch_size, -1), posedirs).view(batch_size, -1, 3)
    v_posed = pose_offsets + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    W = lbs_weights.unsqueeze(dim=0).expand([batch_size, -1, -1])
    num_joints = J_regressor.shape[0]
    T = torch.matmul(W, A.view(batch_size, num_joints, 16)).view(batch_size, -1, 4, 4)
    ones = torch.ones([batch_size, v_posed.shape[1], 1], dtype=dtype, device=device)
    v_posed_homo = torch.cat([v_posed, ones], dim=2)
    v_homo = torch.matmul(T, torch.unsqueeze(v_posed_homo, dim=-1))
    return v_homo[:, :, :3, 0], J_transformed


class SMPLOracle(nn.Module):
    """49-joint SMPL wrapper (PARE ``pare.models.SMPL`` over ``smplx.SMPL``).

    Buffers are named like smplx's so that checkpoints carrying ``smpl.smpl.*`` keys load.
    ``model`` is the dict produced by ``spec_amd.synth.smpl_model`` or the asset loader.
    """

    def __init__(self, model):
        super().__init__()
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        self.register_buffer('v_template', t(model['v_template']).float())
        self.register_buffer('shapedirs', t(model['shapedirs']).float())
        self.register_buffer('posedirs', t(model['posedirs']).float())
        self.register_buffer('J_regressor', t(model['J_regressor']).float())
        self.register_buffer('lbs_weights', t(model['lbs_weights']).float())
        self.register_buffer('J_regressor_extra', t(model['J_regressor_extra']).float())
        parents = t(model['parents']).long().clone()
        parents[0] = -1
        self.register_buffer('parents', parents)
        self.register_buffer('extra_joints_idxs', t(model['extra_vertex_ids']).long())
        self.register_buffer('joint_map', t(model['joint_map']).long())

    def native(self, betas, rot_mats):
        """smplx.SMPL.forward(pose2rot=False): vertices + 45 joints (24 + 21 vertex-picked)."""
        vertices, joints = lbs_rotmat(betas, rot_mats, self.v_template, self.shapedirs,
                                      self.posedirs, self.J_regressor, self.parents,
                                      self.lbs_weights)
        extra = torch.index_select(vertices, 1, self.extra_joints_idxs)
        return vertices, torch.cat([joints, extra], dim=1)

    def native_axis_angle(self, betas, pose_aa):
        """smplx.SMPL.forward(global_orient=pose[:, :3], body_pose=pose[:, 3:], betas) with pose2rot=True."""
        B = pose_aa.shape[0]
        return self.native(betas, batch_rodrigues(pose_aa.reshape(-1, 3)).view(B, 24, 3, 3))

    def forward(self, betas, rot_mats):
        vertices, joints45 = self.native(betas, rot_mats)
        extra_joints = vertices2joints(self.J_regressor_extra, vertices)
        joints = torch.cat([joints45, extra_joints], dim=1)
        return vertices, joints[:, self.joint_map, :]


def smpl_forward_f64(model, betas, rot_mats):
    """NumPy float64 reference of the same forward: returns verts (B,V,3), joints49 (B,49,3)."""
    f = lambda a: np.asarray(a, dtype=np.float64)
    vt, sdirs, pdirs = f(model['v_template']), f(model['shapedirs']), f(model['posedirs'])
    Jr, W, Jx = f(model['J_regressor']), f(model['lbs_weights']), f(model['J_regressor_extra'])
    parents = np.asarray(model['parents']).astype(np.int64)
    betas, R = f(betas), f(rot_mats)
    B = R.shape[0]
    v_shaped = vt[None] + np.einsum('bl,mkl->bmk', betas, sdirs)
    J = np.einsum('bik,ji->bjk', v_shaped, Jr)
    pf = (R[:, 1:] - np.eye(3)).reshape(B, -1)
    v_posed = v_shaped + (pf @ pdirs).reshape(B, -1, 3)
    nj = parents.shape[0]
    T = np.zeros((B, nj, 4, 4))
    for j in range(nj):
        loc = np.zeros((B, 4, 4))
        loc[:, :3, :3] = R[:, j]
        loc[:, :3, 3] = J[:, j] - (J[:, parents[j]] if j > 0 else 0.0)
        loc[:, 3, 3] = 1.0
        T[:, j] = loc if j == 0 else T[:, parents[j]] @ loc
    posed = T[:, :, :3, 3].copy()
    A = T.copy()
    A[:, :, :3, 3] -= np.einsum('bjik,bjk->bji', T[:, :, :3, :3], J)
    Tv = np.einsum('vj,bjik->bvik', W, A)
    verts = np.einsum('bvik,bvk->bvi', Tv[:, :, :3, :3], v_posed) + Tv[:, :, :3, 3]
    extra_v = verts[:, np.asarray(model['extra_vertex_ids']).astype(np.int64)]
    extra_r = np.einsum('bik,ji->bjk', verts, Jx)
    joints = np.concatenate([posed, extra_v, extra_r], axis=1)
    return verts, joints[:, np.asarray(model['joint_map']).astype(np.int64)]
