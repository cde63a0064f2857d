"""HMR(estimate_var=True): the uncertainty outputs of pare's HMRHead (constructor flags spec/models/hmr.py:35-38,57-64; the two
extra keys are what spec/losses.py:61-62 reads).  Both parameter layouts (doubled decoders / separate variance layers), every
activation, the composed affine map and the nine-GEMM loop, GEMV and GEMM batch sizes - against the CPU oracle and against a fixture
made through the reference's own hmr.py."""
import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import golden, pinned_plan, rel_err, smpl_model, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'
TOL = 1e-4


def _models(separate, act, ucf=True):
    from oracle import heads
    from oracle.models import HMROracle, load_numpy_state
    from spec_amd import assets
    from spec_amd.modules import HMR
    assets.use_synthetic_assets(1003)
    heads.set_assets(smpl_model=smpl_model())
    sd = synth.hmr_state(1002, ucf, estimate_var=True, use_separate_var_branch=separate)
    ref = load_numpy_state(HMROracle(use_cam=True, use_cam_feats=ucf, estimate_var=True, use_separate_var_branch=separate,
                                     uncertainty_activation=act).eval(), sd)
    m = HMR(use_cam=True, use_cam_feats=ucf, estimate_var=True, use_separate_var_branch=separate, uncertainty_activation=act)
    missing, unexpected = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith('smpl.') for k in missing), (missing, unexpected)
    return m.to(DEV).eval(), ref


def _inputs(B, seed):
    from oracle.models import cam_params
    x = t(synth.images(seed, B))
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(seed, B, 640., 480.)]
    g = torch.Generator().manual_seed(seed)
    R, K = cam_params(0.3 * torch.randn(B, generator=g), 0.2 * torch.randn(B, generator=g), (400 + 200 * torch.rand(B, generator=g)).numpy(), iw, ih)
    return x, R, K, sc, ce, iw, ih


@pytest.mark.parametrize('separate', [False, True])
def test_uncertainty_outputs_vs_oracle(separate):
    from spec_amd.modules import UNCERTAINTY_ACTIVATIONS
    m, ref = _models(separate, 'softplus')
    eng = m.engine(torch.device(DEV))
    assert [k for k in m.state_dict() if k.startswith('head.')] == [k for k in ref.state_dict() if k.startswith('head.')]
    for act in ('softplus', '', 'relu', 'sigmoid', 'tanh', 'elu'):
        # the activation is read at every call: one committed model serves all of them
        m.uncertainty_activation = ref.head.uncertainty_activation = act
        eng.set_option('uncertainty_activation', UNCERTAINTY_ACTIVATIONS[act])
        # GEMV (<= 10 rows) and GEMM (12 rows) head paths, composed map and nine-GEMM loop (all five for the first activation)
        for B, collapse in (((1, 1), (3, 0), (3, 1), (12, 1), (12, 0)) if act == 'softplus' else ((3, 1),)):
            ins = _inputs(B, 700 + B)
            want = ref(*ins)
            eng.set_option('head_collapse', collapse)
            out = m(*[a.to(DEV) for a in ins])
            assert sorted(out.keys()) == sorted(want.keys())
            assert out['pred_pose_var'].shape == (B, 288) and out['pred_shape_var'].shape == (B, 20)
            for k in want:
                assert rel_err(out[k].cpu().numpy(), want[k].numpy()) < TOL, (separate, act, B, collapse, k)
            # the mean halves ARE the regressed pose / shape; the variance halves on their own scale
            assert torch.equal(out['pred_pose_var'][:, :144], out['pred_pose_6d']) and torch.equal(out['pred_shape_var'][:, :10], out['pred_shape'])
            for k, n in (('pred_pose_var', 144), ('pred_shape_var', 10)):
                assert rel_err(out[k][:, n:].cpu().numpy(), want[k][:, n:].numpy()) < TOL, (separate, act, B, collapse, k, 'variance half')
    eng.set_option('head_collapse', 1)


def test_uncertainty_call_order_and_errors():
    from spec_amd._lib import SpecmiError
    m, _ = _models(False, 'softplus')
    eng = m.engine(torch.device(DEV))
    with pytest.raises(SpecmiError):                   # no head forward yet
        eng.hmr_uncertainty(2)
    ins = [a.to(DEV) for a in _inputs(2, 9)]
    out = m(*ins)
    with pytest.raises(SpecmiError):                   # another batch size than the last forward
        eng.hmr_uncertainty(3)
    pv, sv = eng.hmr_uncertainty(2)
    assert torch.equal(pv, out['pred_pose_var']) and torch.equal(sv, out['pred_shape_var'])
    from tests.util import gpu_models
    _, plain = gpu_models(True, True, DEV)
    plain(*ins)
    with pytest.raises(SpecmiError):                   # a model without estimate_var
        plain.engine(torch.device(DEV)).hmr_uncertainty(2)
    for plan in ('single', 'latency', 'throughput'):   # the plans differ in the trunk only: same keys, values within the contract
        with pinned_plan(plan, m):
            o = m(*ins)
        assert rel_err(o['pred_pose_var'].cpu().numpy(), out['pred_pose_var'].cpu().numpy()) < TOL


@pytest.mark.parametrize('tag', ['doubled', 'separate'])
def test_estimate_var_reference_fixture(tag):
    """The fixture the reference's own spec/models/hmr.py produced with estimate_var=True (tests/golden/make_fixtures.py)."""
    g = golden(f'hmr_e2e_var_{tag}.npz')
    m, _ = _models(tag == 'separate', str(g['uncertainty_activation']))
    B = int(g['batch'])
    x = t(synth.images(int(g['seed_images']), B)).to(DEV)
    out = m(x, t(g['cam_rotmat']).to(DEV), t(g['cam_intrinsics']).to(DEV), t(g['bbox_scale']).to(DEV), t(g['bbox_center']).to(DEV),
            t(g['img_w']).to(DEV), t(g['img_h']).to(DEV))
    assert sorted(out.keys()) == sorted(g['out_keys'])
    for k in out:
        assert tuple(out[k].shape) == g[f'out_{k}'].shape, k
        assert rel_err(out[k].cpu().numpy(), g[f'out_{k}']) < TOL, (tag, k)
