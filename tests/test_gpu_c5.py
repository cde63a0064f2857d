"""BASELINE.json config 5 as far as it can be exercised without the licensed data: the whole evaluation flow
(``scripts/spec_eval.py`` -> run_evaluation -> compute_error) over a stand-in ``data/`` tree written in the REAL
container formats, against the CPU oracle's restatement of spec/utils/compute_error.py:89-223 - including the
north-star criterion |delta W-MPJPE| < 0.1 mm."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.util import t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = torch.device('cuda:0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ('wv2v', 'v2v', 'wmpjpe', 'mpjpe', 'pampjpe', 'pampjpe_24', 'wmpjpe_24', 'mpjpe_24')


def _oracle_models(gt):
    from oracle import heads
    from oracle.models import HMROracle, load_numpy_state
    from oracle.smpl import SMPLOracle
    heads.set_assets(smpl_model=gt['smpl_model'])
    ohm = load_numpy_state(HMROracle(use_cam=True, use_cam_feats=True).eval(), gt['hmr_state'])
    return ohm, SMPLOracle(gt['smpl_model'])


def test_smpl_native_axis_angle_vs_oracle():
    from oracle.smpl import SMPLOracle
    from spec_amd import metrics, synth
    model = synth.smpl_model(1003)
    body = metrics.BodyModel(model, device=DEV)
    rng = np.random.default_rng(5)
    pose = (rng.standard_normal((9, 72)) * 0.4).astype(np.float32)
    pose[0] = 0.0                          # |r + 1e-8| guard of batch_rodrigues
    pose[1, :3] = [3.1, 0.0, 0.0]          # near pi
    betas = (rng.standard_normal((9, 10)) * 0.8).astype(np.float32)
    v, j = body.native(t(pose).to(DEV), t(betas).to(DEV))
    ov, oj = SMPLOracle(model).native_axis_angle(t(betas), t(pose))
    assert np.abs(v.cpu().numpy() - ov.numpy()).max() < 5e-6
    assert np.abs(j.cpu().numpy() - oj[:, :24].numpy()).max() < 5e-6
    # rotation-matrix input and the joints-only / vertices-only variants
    from oracle.smpl import batch_rodrigues
    R = batch_rodrigues(t(pose).reshape(-1, 3)).view(9, 24, 3, 3)
    v2, j2 = body.native(R.to(DEV), t(betas).to(DEV))
    assert np.abs(v2.cpu().numpy() - ov.numpy()).max() < 5e-6 and np.abs(j2.cpu().numpy() - oj[:, :24].numpy()).max() < 5e-6
    assert body.native(R.to(DEV), t(betas).to(DEV), vertices=False)[0] is None


@pytest.mark.parametrize('dataset', ['spec-syn', 'spec-mtp'])
def test_c5_standin_flow_vs_oracle(tmp_path, dataset):
    from oracle import metrics as OM
    from spec_amd import assets, evaluation
    d = str(tmp_path)
    gt = evaluation.write_standin_data_tree(d, n_images=6, dataset=dataset)
    hp = evaluation.load_config(os.path.join(d, 'data/spec/checkpoints/spec_config.yaml'))
    lines = []
    res = evaluation.run_evaluation(hp, data_root=d, log=lines.append)[dataset]
    assert any('W-MPJPE-24' in l for l in lines) and any('README' in l for l in lines)
    import joblib
    ev = joblib.load(os.path.join(d, 'logs/eval_standin', f'evaluation_results_{dataset}.pkl'))
    assert ev['vertices'].shape == (6, 6890, 3) and len(ev['imgname']) == 6
    # oracle side: the same crops (the device crop has its own parity tests) through the CPU restatement
    ohm, osmpl = _oracle_models(gt)
    ds = evaluation.EvalDataset(dataset, d)
    b = ds.batch(np.arange(6), DEV, 224, False)
    pred = ohm(b['img'].cpu(), cam_rotmat=b['cam_rotmat'].cpu(), cam_intrinsics=b['cam_int'].cpu(),
               bbox_scale=b['scale'].cpu(), bbox_center=b['center'].cpu(), img_w=b['img_w'].cpu(), img_h=b['img_h'].cpu())
    assert np.abs(ev['vertices'] - pred['smpl_vertices'].numpy()).max() / np.abs(ev['vertices']).max() < 1e-4
    pred_R = None if dataset == 'spec-syn' else joblib.load(os.path.join(d, f'data/camcalib/{dataset}_cam_rotmat.pkl'))
    ref = OM.compute_error(pred['smpl_vertices'], ds.data, dataset, osmpl, np.load(os.path.join(d, 'data/J_regressor_h36m.npy')),
                           pred_cam_rotmat=pred_R)
    for k in KEYS:
        diff = np.abs(res['per_sample'][k] - ref[k]).max()
        assert diff < 0.05, (k, diff, res['per_sample'][k], ref[k])              # millimetres
    assert abs(res['mean']['wmpjpe_24'] - ref['wmpjpe_24'].mean()) < 0.1          # BASELINE.json: W-MPJPE within 0.1 mm
    assets.use_synthetic_assets(1003)


def test_spec_eval_cli_standin(tmp_path):
    import json
    rep_path = os.path.join(str(tmp_path), 'eval.json')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'spec_eval.py'), '--standin', str(tmp_path), '--report', rep_path],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'W-MPJPE-24:' in r.stdout and 'PA-MPJPE-24:' in r.stdout and 'W-V2V:' in r.stdout and 'README 74.9' in r.stdout
    rep = json.load(open(rep_path))      # the machine-readable README delta (config 5): scores, table row, delta, verdict
    ds = rep['datasets']['spec-syn']
    assert rep['standin_tree'] is True and rep['target_abs_delta_wmpjpe_mm'] == 0.1
    assert ds['readme'] == {'wmpjpe': 74.9, 'pampjpe': 54.5, 'wpve': 90.5}
    assert abs(ds['delta_mm']['wmpjpe'] - (ds['mean']['wmpjpe_24'] - 74.9)) < 1e-9 and isinstance(ds['within_target'], bool)


def test_reference_demo_commands_on_standin_tree(tmp_path):
    """The reference's two demo commands with their DEFAULT paths, run from a directory that holds a ``data/`` tree in the
    real formats: ``scripts/camcalib_demo.py --img_folder ... --out_folder ...`` and ``scripts/spec_demo.py --image_folder
    data/sample_images --output_folder ...`` (README.md:100-104) - SPECTester, read_cam_params, result pickles."""
    import joblib
    from spec_amd import evaluation
    d = str(tmp_path)
    evaluation.write_standin_data_tree(d, n_images=2)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'camcalib_demo.py'), '--img_folder', 'data/sample_images',
                        '--out_folder', 'out_cc', '--no_save'], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = joblib.load(os.path.join(d, 'out_cc', 'im0.png.pkl'))
    assert set(rec) == {'vfov', 'f_pix', 'pitch', 'roll'} and 0.2617 <= float(rec['vfov']) <= 2.1
    assert abs(float(rec['f_pix']) - 300 / 2. / np.tan(float(rec['vfov']) / 2.)) < 1e-3
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'spec_demo.py'), '--image_folder', 'data/sample_images',
                        '--output_folder', 'logs/demo', '--exp', 'x'], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = os.path.join(d, 'logs/demo', 'sample_images_x')
    cam = joblib.load(os.path.join(out, 'camcalib', 'im1.png.pkl'))
    assert abs(float(cam['vfov']) - float(joblib.load(os.path.join(d, 'out_cc', 'im1.png.pkl'))['vfov'])) < 1e-6
    res = joblib.load(os.path.join(out, 'spec_results', 'im1.pkl'))
    assert res['smpl_vertices'].shape == (1, 6890, 3) and res['smpl_joints2d'].shape == (1, 49, 2)
    assert set(res) == {'smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'pred_cam', 'pred_shape',
                        'pred_pose_6d'}
    # kl-trained CamCalib variants decode through the arg-max tables
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'camcalib_demo.py'), '--img_folder', 'data/sample_images',
                        '--out_folder', 'out_kl', '--loss', 'kl', '--no_save'], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    from spec_amd import cam_utils as CU
    assert float(joblib.load(os.path.join(d, 'out_kl', 'im0.png.pkl'))['pitch']) in set(CU.pitch_bins_centers.tolist())


@pytest.mark.parametrize('nv,B', [(1000, 3), (257, 9), (6890, 17)])
def test_body_model_other_sizes_vs_oracle(nv, B):
    """The SMPL-only handle (SPECMI_MODEL_SMPL) on body models of other vertex counts (partial 256-vertex workgroups, batch
    sizes that are not a multiple of the 8-image skinning tile): native outputs and the 49-joint / projection head."""
    from oracle import heads
    from oracle.smpl import SMPLOracle, batch_rodrigues
    from spec_amd import metrics, synth
    model = synth.smpl_model(77, nv)
    body = metrics.BodyModel(model, device=DEV)
    rng = np.random.default_rng(nv + B)
    pose = (rng.standard_normal((B, 72)) * 0.3).astype(np.float32)
    betas = (rng.standard_normal((B, 10)) * 0.7).astype(np.float32)
    v, j = body.native(t(pose).to(DEV), t(betas).to(DEV))
    ov, oj = SMPLOracle(model).native_axis_angle(t(betas), t(pose))
    assert v.shape == (B, nv, 3)
    assert np.abs(v.cpu().numpy() - ov.numpy()).max() < 5e-6 and np.abs(j.cpu().numpy() - oj[:, :24].numpy()).max() < 5e-6
    # SMPLHead (weak perspective) through the same handle: specmi_smpl_forward without a trunk
    heads.set_assets(smpl_model=model)
    R = batch_rodrigues(t(pose).reshape(-1, 3)).view(B, 24, 3, 3)
    cam = torch.tensor([[0.9, 0.05, -0.02]]).repeat(B, 1)
    ref = heads.SMPLHead(focal_length=5000., img_res=224)(R, t(betas), cam, normalize_joints2d=True)
    out = body.engine.smpl(R.to(DEV), t(betas).to(DEV), cam.to(DEV))
    for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t'):
        err = float(np.abs(out[k].cpu().numpy() - ref[k].numpy()).max() / np.abs(ref[k].numpy()).max())
        assert err < 1e-5, (k, err)
