"""The oracle's own composition (oracle/models.py) against the golden vectors that were
produced by the REFERENCE's in-tree modules (tests/golden/make_fixtures.py)."""
import numpy as np
import pytest
import torch

from oracle.models import decode_angles, cam_params
from spec_amd import synth
from tests.util import golden, oracle_models, t

torch.set_grad_enabled(False)


def test_decode_matches_reference_cam_utils():
    g = golden('camcalib_decode.npz')
    vf, pt, rl = decode_angles(t(g['logits_vfov']), t(g['logits_pitch']), t(g['logits_roll']))
    assert np.array_equal(vf.numpy(), g['vfov'])
    assert np.array_equal(pt.numpy(), g['pitch'])
    assert np.array_equal(rl.numpy(), g['roll'])
    assert np.allclose(g['vfov_range'], [0.2617, 2.1]) and np.allclose(g['pitch_range'], [-0.6, 0.6])


def test_cam_params_matches_reference():
    g = golden('cam_params.npz')
    meta = g['meta']
    R, K = cam_params(meta[:, 0], meta[:, 1], meta[:, 5], t(meta[:, 4]).float(), t(meta[:, 3]).float())
    assert np.array_equal(R.numpy(), g['R'])
    assert np.array_equal(K.numpy(), g['K'])
    assert np.all(g['K'][:, 2, 2] == 0.0)


def test_camcalib_network_matches_reference_composition():
    g = golden('camcalib_e2e.npz')
    cc, _ = oracle_models()
    imgs = t(synth.images(int(g['seed_images']), int(g['batch'])))
    lg = cc(imgs)
    for i, k in enumerate(('logits_vfov', 'logits_pitch', 'logits_roll')):
        assert np.array_equal(lg[i].numpy(), g[k]), k
    assert list(cc.state_dict().keys()) == list(g['state_keys'])


@pytest.mark.parametrize('tag,use_cam,ucf', [('camfeats', True, True), ('cam', True, False), ('nocam', False, False)])
def test_hmr_matches_reference_composition(tag, use_cam, ucf):
    g = golden(f'hmr_e2e_{tag}.npz')
    _, hm = oracle_models(use_cam=use_cam, use_cam_feats=ucf)
    B = int(g['batch'])
    imgs = t(synth.images(int(g['seed_images']), B))
    if use_cam:
        out = hm(imgs, t(g['cam_rotmat']), t(g['cam_intrinsics']), t(g['bbox_scale']), t(g['bbox_center']),
                 t(g['img_w']), t(g['img_h']))
    else:
        out = hm(imgs)
    assert list(out.keys()) == list(g['out_keys'])
    for k in out:
        assert np.array_equal(out[k].numpy(), g[f'out_{k}']), k
    own = [k for k in hm.state_dict().keys()]
    assert own == list(g['state_keys'])


def test_metrics_match_reference_compute_error():
    """oracle/metrics.py against spec/utils/compute_error.py (eval_single, eval_j_24) run in the
    build container (tests/golden/metrics.npz)."""
    from oracle import metrics as M
    g = golden('metrics.npz')
    Bm, seed, V = int(g['batch']), int(g['seed']), 6890
    J24 = synth.smpl_model(int(g['seed_smpl']))['J_regressor']
    J17 = synth.h36m_regressor(int(g['seed_smpl']))
    gt_v = synth.normal(seed, 'gt_verts', (Bm, V, 3), std=0.3)
    pr_v = gt_v + synth.normal(seed, 'noise', (Bm, V, 3), std=0.03) + synth.normal(seed, 'shift', (Bm, 1, 3), std=0.2)
    mp, pa, vv = M.eval_single(t(pr_v), t(gt_v), t(J17)[None].expand(Bm, -1, -1))
    assert np.array_equal(mp, g['mpjpe']) and np.array_equal(pa, g['pampjpe']) and np.array_equal(vv, g['v2v'])
    pj = torch.einsum('bik,ji->bjk', t(pr_v), t(J24))
    gj = torch.einsum('bik,ji->bjk', t(gt_v), t(J24))
    mp24, pa24 = M.eval_j_24(pj, gj)
    assert np.array_equal(mp24, g['mpjpe24']) and np.array_equal(pa24, g['pampjpe24'])
