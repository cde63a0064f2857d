"""Closed-form known-answer tests that pin the oracle's leaf functions (the reference holds no
tests of its own for this path - SURVEY.md section 4 / 8c)."""
import numpy as np
import torch

from oracle import geometry as G
from oracle.models import decode_angles, cam_params
from oracle.smpl import SMPLOracle, smpl_forward_f64
from tests.util import smpl_model, t

torch.set_grad_enabled(False)


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def test_decode_one_hot_closed_form():
    nb = 256
    for k in (0, 1, 100, 255):
        row = torch.full((1, nb), -80.0)
        row[0, k] = 40.0
        vf, pt, rl = decode_angles(row, row, row)
        s = k / 255.0 * 2 - 1
        assert abs(vf.item() - ((2.1 - 0.2617) * (s + 1) / 2 + 0.2617)) < 2e-6
        assert abs(pt.item() - (1.2 * (s + 1) / 2 - 0.6)) < 2e-6
        assert abs(rl.item() - pt.item()) < 1e-7


def test_decode_uniform_is_centre():
    z = torch.zeros(3, 256)
    vf, pt, rl = decode_angles(z, z, z)
    assert np.allclose(vf.numpy(), (0.2617 + 2.1) / 2, atol=2e-6)
    assert np.allclose(pt.numpy(), 0.0, atol=2e-6) and np.allclose(rl.numpy(), 0.0, atol=2e-6)


def test_euler2matrix_is_rx_ry_rz():
    ang = torch.tensor([[0.3, -0.2, 0.5], [-0.6, 0.0, 0.6], [0.0, 0.0, 0.0]])
    R = G.batch_euler2matrix(ang).numpy()
    for i, (x, y, z) in enumerate(ang.numpy()):
        assert np.allclose(R[i], _rx(x) @ _ry(y) @ _rz(z), atol=1e-6)


def test_cam_params_k22_zero_and_centre():
    R, K = cam_params([0.1], [-0.2], [500.0], torch.tensor([640.0]), torch.tensor([480.0]))
    assert K[0, 2, 2].item() == 0.0 and K[0, 0, 2].item() == 320.0 and K[0, 1, 2].item() == 240.0
    assert K[0, 0, 0].item() == 500.0 and K[0, 1, 1].item() == 500.0
    assert np.allclose(R[0].numpy(), _rx(0.1) @ _rz(-0.2), atol=1e-6)


def test_rot6d_identity_and_orthonormal():
    ident = torch.tensor([[1., 0., 0., 1., 0., 0.]])
    assert np.allclose(G.rot6d_to_rotmat(ident).numpy()[0], np.eye(3), atol=1e-7)
    x = torch.randn(50, 6, generator=torch.Generator().manual_seed(0))
    R = G.rot6d_to_rotmat(x).numpy().astype(np.float64)
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-5)
    assert np.allclose(np.linalg.det(R), 1.0, atol=1e-5)
    # round trip through the 6d representation
    assert np.allclose(G.rot6d_to_rotmat(G.rotmat_to_rot6d(t(R.astype(np.float32)))).numpy(), R, atol=1e-5)


def test_projection_on_axis_hits_principal_point():
    K = torch.tensor([[[500., 0., 320.], [0., 500., 240.], [0., 0., 0.]]])
    p = G.perspective_projection(torch.tensor([[[0., 0., 0.]]]), torch.eye(3)[None], torch.tensor([[0., 0., 4.]]), K)
    assert np.allclose(p.numpy(), [[[320., 240.]]])


def test_full_img_cam_closed_form():
    cam = torch.tensor([[0.8, 0.1, -0.2]])
    ct = G.convert_pare_to_full_img_cam(cam, torch.tensor([224.0]), torch.tensor([[320., 240.]]),
                                        torch.tensor([640.]), torch.tensor([480.]), torch.tensor([500.]))
    assert np.allclose(ct.numpy(), [[0.1, -0.2, 2 * 500 / (224 * 0.8)]], rtol=1e-6)


def test_smpl_zero_pose_zero_shape():
    m = smpl_model()
    so = SMPLOracle(m)
    B = 2
    R = torch.eye(3).expand(B, 24, 3, 3).contiguous()
    verts, joints45 = so.native(torch.zeros(B, 10), R)
    assert np.allclose(verts.numpy(), m['v_template'][None], atol=1e-6)
    J = m['J_regressor'].astype(np.float64) @ m['v_template'].astype(np.float64)
    assert np.allclose(joints45[:, :24].numpy(), J[None], atol=1e-5)
    assert np.allclose(joints45[:, 24:].numpy(), m['v_template'][m['extra_vertex_ids']][None], atol=1e-6)


def test_smpl_global_rotation_rotates_about_root():
    m = smpl_model()
    so = SMPLOracle(m)
    Rg = (_rz(0.7) @ _rx(-0.4)).astype(np.float32)
    R = torch.eye(3).expand(1, 24, 3, 3).clone()
    R[0, 0] = t(Rg)
    verts, _ = so.native(torch.zeros(1, 10), R)
    J0 = (m['J_regressor'].astype(np.float64) @ m['v_template'].astype(np.float64))[0]
    expect = (m['v_template'].astype(np.float64) - J0) @ Rg.astype(np.float64).T + J0
    assert np.allclose(verts[0].numpy(), expect, atol=2e-5)


def test_smpl_fp32_vs_fp64_budget():
    m = smpl_model()
    so = SMPLOracle(m)
    g = torch.Generator().manual_seed(3)
    R = G.rot6d_to_rotmat(torch.randn(4 * 24, 6, generator=g)).view(4, 24, 3, 3)
    betas = torch.randn(4, 10, generator=g)
    v32, j32 = so(betas, R)
    v64, j64 = smpl_forward_f64(m, betas.numpy(), R.numpy())
    assert np.abs(v32.numpy() - v64).max() < 5e-6
    assert np.abs(j32.numpy() - j64).max() < 5e-6
    assert j32.shape == (4, 49, 3)


def test_softargmax_matches_numpy():
    x = torch.randn(5, 1, 256, generator=torch.Generator().manual_seed(1)) * 3
    k, p = G.softargmax1d(x)
    xn = x.numpy().astype(np.float64)
    e = np.exp(xn - xn.max(-1, keepdims=True))
    pr = e / e.sum(-1, keepdims=True)
    exp = (pr * np.arange(256)).sum(-1) / 255 * 2 - 1
    assert np.allclose(k.numpy(), exp, atol=1e-5)
