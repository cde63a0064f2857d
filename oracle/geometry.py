"""Oracle (test infrastructure): geometry leaf functions of the SPEC path, PyTorch-CPU.

Restated from the published PARE implementation (un-vendored, requirements.txt:28); every
function lists the reference call sites that constrain its contract.
"""
import torch
import torch.nn.functional as F


def rot6d_to_rotmat(x):
    """6D rotation -> 3x3 (Zhou et al.), Gram-Schmidt with columns b1,b2,b3.

    Used inside HMRHead (reference ``spec/models/hmr.py:96``) and imported by
    ``spec/losses.py:23``.  ``x`` is (B, 144) or (N, 6); view is (-1, 3, 2): element [i, c]
    is row i of column c.
    """
    x = x.reshape(-1, 3, 2)
    a1 = x[:, :, 0]
    a2 = x[:, :, 1]
    b1 = F.normalize(a1)                                   # eps = 1e-12, dim = 1
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def rotmat_to_rot6d(x):
    """First two columns, flattened row-major: (B,3,3) -> (B,6).  Fed to fc1 when
    ``use_cam_feats`` (reference ``spec/models/hmr.py:94-96``)."""
    rotmat = x.reshape(-1, 3, 3)
    return rotmat[:, :, :2].reshape(x.shape[0], -1)


def batch_euler2matrix(r):
    """Euler (x, y, z) -> quaternion -> matrix = Rx(x) Ry(y) Rz(z).

    Reference call site: ``spec/utils/cam_params.py:37`` with ``[pitch, 0, roll]``.
    """
    return quat_to_rotmat(euler_to_quaternion(r))


def euler_to_quaternion(r):
    x = r[..., 0]
    y = r[..., 1]
    z = r[..., 2]
    z = z / 2.0
    y = y / 2.0
    x = x / 2.0
    cz = torch.cos(z)
    sz = torch.sin(z)
    cy = torch.cos(y)
    sy = torch.sin(y)
    cx = torch.cos(x)
    sx = torch.sin(x)
    quaternion = torch.zeros_like(r.repeat(1, 2))[..., :4].to(r.device)
    quaternion[..., 0] += cx * cy * cz - sx * sy * sz
    quaternion[..., 1] += cx * sy * sz + cy * cz * sx
    quaternion[..., 2] += cx * cz * sy - sx * cy * sz
    quaternion[..., 3] += cx * cy * sz + sx * cz * sy
    return quaternion


def quat_to_rotmat(quat):
    norm_quat = quat
    norm_quat = norm_quat / norm_quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = norm_quat[:, 0], norm_quat[:, 1], norm_quat[:, 2], norm_quat[:, 3]
    B = quat.size(0)
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    rotMat = torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                          2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                          2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(B, 3, 3)
    return rotMat


def convert_pare_to_full_img_cam(pare_cam, bbox_height, bbox_center, img_w, img_h,
                                 focal_length, crop_res=224):
    """Weak-perspective (s,tx,ty) in bbox coords -> translation in the full-image camera.

    Called from SMPLCamHead with ``bbox_height = bbox_scale*200`` and
    ``focal_length = cam_intrinsics[:,0,0]`` (contract: ``spec/models/hmr.py:101-112``;
    consumer of ``pred_cam_t``: ``spec/tester.py:166-167``).  ``res`` is the literal 224.
    """
    s, tx, ty = pare_cam[:, 0], pare_cam[:, 1], pare_cam[:, 2]
    res = 224
    r = bbox_height / res
    tz = 2 * focal_length / (r * res * s)
    cx = 2 * (bbox_center[:, 0] - (img_w / 2.)) / (s * bbox_height)
    cy = 2 * (bbox_center[:, 1] - (img_h / 2.)) / (s * bbox_height)
    return torch.stack([tx + cx, ty + cy, tz], dim=-1)


def perspective_projection(points, rotation, translation, cam_intrinsics):
    """p = K ((R X + t) / z), returns the first two rows.  K[2,2] may be 0
    (``spec/utils/cam_params.py:39-46`` leaves it 0); the third row is discarded."""
    K = cam_intrinsics
    points = torch.einsum('bij,bkj->bki', rotation, points)
    points = points + translation.unsqueeze(1)
    projected = points / points[:, :, -1].unsqueeze(-1)
    projected = torch.einsum('bij,bkj->bki', K, projected.float())
    return projected[:, :, :-1]


def convert_weak_perspective_to_perspective(cam, focal_length=5000., img_res=224):
    """Non-camera SMPLHead variant (``spec/models/hmr.py:71-74``)."""
    return torch.stack([cam[:, 1], cam[:, 2],
                        2 * focal_length / (img_res * cam[:, 0] + 1e-9)], dim=-1)


def softargmax1d(heatmaps, temperature=None, normalize_keypoints=True):
    """Softmax expectation of the bin index; (N, C, D) -> (N, C) in [-1, 1].

    Reference call site ``camcalib/cam_utils.py:114-118`` (shape (N,1,256), reshaped to (N,)).
    """
    dtype, device = heatmaps.dtype, heatmaps.device
    if temperature is None:
        temperature = torch.tensor(1.0, dtype=dtype, device=device)
    n, c, dim = heatmaps.shape
    points = torch.arange(0, dim, device=device, dtype=dtype).reshape(1, 1, dim).expand(n, -1, -1)
    prob = F.softmax(heatmaps.reshape(n, c, -1) * temperature.reshape(1, -1, 1), dim=-1)
    keypoints = (prob.reshape(n, -1, dim) * points).sum(dim=-1)
    if normalize_keypoints:
        keypoints = (keypoints / (dim - 1) * 2 - 1)
    return keypoints, prob.reshape(n, -1, dim)
