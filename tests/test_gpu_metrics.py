"""On-device evaluation metrics against the reference's compute_error.py fixture
and the NumPy oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import golden, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'


def _data(seed, B, V=6890):
    gt_v = synth.normal(seed, 'gt_verts', (B, V, 3), std=0.3)
    pr_v = gt_v + synth.normal(seed, 'noise', (B, V, 3), std=0.03) + synth.normal(seed, 'shift', (B, 1, 3), std=0.2)
    return pr_v, gt_v


def test_mesh_metrics_vs_reference_fixture():
    from spec_amd import metrics
    g = golden('metrics.npz')
    B, seed = int(g['batch']), int(g['seed'])
    pr_v, gt_v = _data(seed, B)
    J17 = synth.h36m_regressor(int(g['seed_smpl']))
    mp, pa, vv = metrics.eval_single(t(pr_v).to(DEV), t(gt_v).to(DEV), t(J17).to(DEV)[None].expand(B, -1, -1))
    assert np.allclose(mp.cpu().numpy(), g['mpjpe'], rtol=1e-4)
    assert np.allclose(pa.cpu().numpy(), g['pampjpe'], rtol=1e-4)
    assert np.allclose(vv.cpu().numpy(), g['v2v'], rtol=1e-4)
    J24 = synth.smpl_model(int(g['seed_smpl']))['J_regressor']
    # the fixture's eval_j_24 inputs are einsum('bik,ji->bjk') regressions of both meshes (make_fixtures.py)
    pj, gj = metrics.regress_joints(t(pr_v).to(DEV), t(J24).to(DEV)), metrics.regress_joints(t(gt_v).to(DEV), t(J24).to(DEV))
    ref_pj = torch.einsum('bik,ji->bjk', t(pr_v), t(J24))
    assert np.abs(pj.cpu().numpy() - ref_pj.numpy()).max() < 2e-6
    mp24, pa24 = metrics.eval_j_24(pj, gj)
    assert np.allclose(mp24.cpu().numpy(), g['mpjpe24'], rtol=1e-4)
    assert np.allclose(pa24.cpu().numpy(), g['pampjpe24'], rtol=1e-4)
    # W-MPJPE-24 (README metric): regressed prediction against GIVEN ground-truth joints
    mp24b, _ = metrics.w_mpjpe_24(t(pr_v).to(DEV), gj, t(J24).to(DEV))
    assert np.allclose(mp24b.cpu().numpy(), g['mpjpe24'], rtol=1e-4)


@pytest.mark.parametrize('B,J', [(1, 14), (37, 24), (256, 17)])
def test_joint_metrics_vs_oracle(B, J):
    from oracle import metrics as M
    from spec_amd import metrics
    g = torch.Generator().manual_seed(B + J)
    gt = torch.randn(B, J, 3, generator=g) * 0.4
    # similarity-transformed + noisy prediction, including a few mirrored (det < 0) cases
    A = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    pred = (gt @ A.transpose(1, 2)) * (0.8 + 0.4 * torch.rand(B, 1, 1, generator=g)) \
        + 0.05 * torch.randn(B, J, 3, generator=g) + torch.randn(B, 1, 3, generator=g)
    mp_ref, pa_ref = M.eval_j_24(pred, gt)
    mp, pa = metrics.eval_j_24(pred.to(DEV), gt.to(DEV))
    assert np.allclose(mp.cpu().numpy(), mp_ref, rtol=1e-4, atol=1e-3)
    assert np.allclose(pa.cpu().numpy(), pa_ref, rtol=2e-4, atol=2e-3)


def test_metrics_properties():
    """Identical meshes -> 0; PA-MPJPE invariant to a similarity transform of the prediction."""
    from spec_amd import metrics
    pr_v, gt_v = _data(11, 4, V=1200)
    J = t(synth.h36m_regressor(1003, 1200)).to(DEV)
    same = metrics.eval_single(t(gt_v).to(DEV), t(gt_v).to(DEV), J)
    assert all(float(x.abs().max()) < 1e-3 for x in same)
    a = metrics.eval_single(t(pr_v).to(DEV), t(gt_v).to(DEV), J)[1]
    Rz = torch.tensor([[0.36, -0.8, 0.48], [0.8, 0.0, -0.6], [0.48, 0.6, 0.64]])
    moved = (t(pr_v) @ Rz.T) * 1.7 + torch.tensor([0.3, -2.0, 5.0])
    b = metrics.eval_single(moved.to(DEV), t(gt_v).to(DEV), J)[1]
    assert np.allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4)
