"""Host-side logic: checkpoint conventions, state_dict layout, synthetic generator, packing,
asset loading."""
import io
import os
import pickle

import numpy as np
import pytest
import torch

from spec_amd import assets, synth
from spec_amd.checkpoint import load_pretrained_model, strip_lightning_prefix, read_checkpoint
from tests.util import golden, t


@pytest.fixture(scope='module')
def models():
    from spec_amd.modules import HMR, CameraRegressorNetwork
    assets.use_synthetic_assets(1003)
    return CameraRegressorNetwork(), HMR(use_cam=True, use_cam_feats=True)


def test_state_dict_layout_matches_reference(models):
    cc, hm = models
    g = golden('camcalib_e2e.npz')
    assert list(cc.state_dict().keys()) == list(g['state_keys'])      # strict load must be possible
    gk = [k for k in golden('hmr_e2e_camfeats.npz')['state_keys'] if not k.startswith('smpl.')]
    own = [k for k in hm.state_dict().keys() if not k.startswith('smpl.')]
    assert sorted(own) == sorted(gk)
    assert len([k for k in cc.state_dict() if k.startswith('backbone.')]) == 318
    assert tuple(hm.state_dict()['head.fc1.weight'].shape) == (1024, 2212)
    assert 'head.init_pose' in hm.state_dict()                           # scripts/spec_eval.py:57


def test_lightning_prefix_and_strict_load(models):
    cc, _ = models
    sd = {('model.' + k): v.clone() for k, v in cc.state_dict().items()}
    load_pretrained_model(cc, sd, remove_lightning=True, strict=True)
    assert list(strip_lightning_prefix(sd).keys()) == list(cc.state_dict().keys())


def test_shape_mismatch_tolerant_load():
    from spec_amd.modules import HMR
    assets.use_synthetic_assets(1003)
    src = HMR(use_cam=True, use_cam_feats=False)          # fc1 has 2205 columns
    dst = HMR(use_cam=True, use_cam_feats=True)           # fc1 has 2212 columns
    sd = {('model.' + k): v.clone() for k, v in src.state_dict().items()}
    sd['smpl.some_trainer_buffer'] = torch.zeros(3)        # trainer-level keys are ignored
    sd['J_regressor'] = torch.zeros(17, 6890)
    load_pretrained_model(dst, sd, overwrite_shape_mismatch=True, remove_lightning=True)
    w_src, w_dst = src.state_dict()['head.fc1.weight'], dst.state_dict()['head.fc1.weight']
    assert torch.equal(w_dst[:, :2205], w_src) and torch.equal(w_dst[:, 2205:], w_src[:, -7:])
    assert torch.equal(dst.state_dict()['backbone.conv1.weight'], src.state_dict()['backbone.conv1.weight'])
    with pytest.raises(RuntimeError):
        load_pretrained_model(dst, sd, overwrite_shape_mismatch=False, remove_lightning=True)


def test_read_checkpoint_with_unknown_classes(tmp_path):
    class Hyper:  # stands for a yacs CfgNode living in a module that is not importable later
        pass
    Hyper.__module__ = 'yacs_not_installed.config'
    Hyper.__qualname__ = 'CfgNode'
    p = tmp_path / 'lit.ckpt'
    # hand-build a pickle that references the missing class next to a state_dict
    payload = {'state_dict': {'model.fc_vfov.bias': torch.arange(4.)}, 'epoch': 3}
    torch.save(payload, p)
    ck = read_checkpoint(str(p))
    assert torch.equal(ck['state_dict']['model.fc_vfov.bias'], torch.arange(4.))


def test_synth_is_deterministic_and_exact():
    a = synth.normal(5, 'x', (1000,), std=0.3)
    b = synth.normal(5, 'x', (1000,), std=0.3)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert not np.array_equal(a, synth.normal(6, 'x', (1000,), std=0.3))
    assert abs(float(a.std()) - 0.3) < 0.03
    u = synth.uniform01(1, 'u', 10000)
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.02
    # pinned values: the committed golden fixtures assume exactly this generator
    assert a[:3].tolist() == [-0.12882201373577118, 0.0512717105448246, -0.16897457838058472]
    assert u[:3].tolist() == [0.6008992791175842, 0.9093006253242493, 0.32369232177734375]


def test_synth_smpl_shapes():
    m = synth.smpl_model(1003)
    assert m['v_template'].shape == (6890, 3) and m['shapedirs'].shape == (6890, 3, 10)
    assert m['posedirs'].shape == (207, 20670) and m['J_regressor'].shape == (24, 6890)
    assert m['lbs_weights'].shape == (6890, 24) and m['J_regressor_extra'].shape == (9, 6890)
    assert np.allclose(m['J_regressor'].sum(1), 1, atol=1e-5) and np.allclose(m['lbs_weights'].sum(1), 1, atol=1e-5)
    assert m['joint_map'].dtype == np.int32 and m['parents'][0] == -1


def test_pack_unpack_roundtrip():
    from spec_amd.pipeline import pack_outputs, unpack_outputs, PACKED_KEYS
    B, V = 3, 6890
    g = torch.Generator().manual_seed(0)
    out = {}
    for k, shp in PACKED_KEYS:
        shp = (V, 3) if shp is None else shp
        out[k] = torch.randn(B, *shp, generator=g)
    packed = pack_outputs(out)
    assert packed.shape == (B, 20670 + 147 + 98 + 3 + 216 + 3 + 10 + 144 + 3)
    assert packed.shape[1] * 4 == 85164 + 12                      # SURVEY.md 8e record size
    back = unpack_outputs(packed, V)
    for k in out:
        assert torch.equal(back[k], out[k])


def test_smpl_file_loader_without_chumpy(tmp_path):
    import scipy.sparse as sp
    nv = 50
    rng = np.random.default_rng(0)
    raw = {'v_template': rng.normal(size=(nv, 3)), 'shapedirs': rng.normal(size=(nv, 3, 300)),
           'posedirs': rng.normal(size=(nv, 3, 207)), 'J_regressor': sp.csc_matrix(rng.random((24, nv))),
           'weights': rng.random((nv, 24)), 'kintree_table': np.stack([np.array([2**32 - 1] + list(range(23))), np.arange(24)]),
           'f': np.zeros((10, 3), dtype=np.uint32)}
    p = tmp_path / 'SMPL_NEUTRAL.pkl'
    with open(p, 'wb') as f:
        pickle.dump(raw, f)
    m = assets.load_smpl_file(str(tmp_path), j_regressor_extra=rng.random((9, nv)))
    assert m['shapedirs'].shape == (nv, 3, 10) and m['posedirs'].shape == (207, nv * 3)
    assert np.allclose(m['posedirs'][5].reshape(nv, 3), raw['posedirs'][:, :, 5])
    assert m['parents'][0] == -1 and m['J_regressor'].shape == (24, nv) and m['J_regressor'].dtype == np.float32


@pytest.mark.parametrize('backbone,nl,nc', [('resnet34', 1, 1024), ('resnet34', 3, 256), ('resnet50', 2, 512)])
def test_camcalib_variants_state_dict_layout(backbone, nl, nc):
    """camcalib/model.py:25-70 variants (the reference's test_model matrix): parameter names, order and shapes are
    the torchvision / nn.Sequential ones, so a checkpoint trained with the reference loads strictly."""
    from oracle.models import CamCalibOracle
    from spec_amd.modules import CameraRegressorNetwork
    own = CameraRegressorNetwork(backbone=backbone, num_fc_layers=nl, num_fc_channels=nc).state_dict()
    ref = CamCalibOracle(backbone, nl, nc).state_dict()
    assert list(own.keys()) == list(ref.keys())
    assert all(tuple(own[k].shape) == tuple(ref[k].shape) for k in ref)
    sd = synth.camcalib_state(5, backbone=backbone, num_fc_layers=nl, num_fc_channels=nc)
    assert sorted(sd.keys()) == sorted(ref.keys())
    n_backbone = len([k for k in ref if k.startswith('backbone.')])
    assert n_backbone == (318 if backbone == 'resnet50' else 216)   # (53 | 36) conv + bn pairs x 6 tensors


def test_unsupported_variants_raise():
    from spec_amd.modules import CameraRegressorNetwork, HMR
    with pytest.raises(NotImplementedError):
        CameraRegressorNetwork(backbone='hrnet_w32')
    with pytest.raises(NotImplementedError):
        CameraRegressorNetwork(num_fc_layers=4)
    with pytest.raises(NotImplementedError):
        HMR(backbone='mobilenet_v2')            # pare's one trunk outside the torchvision ResNet / HRNet families
    with pytest.raises(NotImplementedError):
        HMR(backbone='hrnet_w18-conv')
    from spec_amd import assets
    assets.use_synthetic_assets(1003)
    with pytest.raises(NotImplementedError):
        HMR(estimate_var=True, uncertainty_activation='exp')     # not a torch.nn.functional the library evaluates
    # HMRHead's uncertainty layouts (pare; flags at spec/models/hmr.py:35-38,59-61): doubled decoders or separate variance layers
    keys = lambda m: {k: tuple(v.shape) for k, v in m.head.state_dict().items() if k.startswith('dec')}
    assert keys(HMR(estimate_var=True)) == {'decpose.weight': (288, 1024), 'decpose.bias': (288,), 'decshape.weight': (20, 1024),
                                            'decshape.bias': (20,), 'deccam.weight': (3, 1024), 'deccam.bias': (3,)}
    sep = keys(HMR(estimate_var=True, use_separate_var_branch=True))
    assert sep['decpose.weight'] == (144, 1024) and sep['decpose_var.weight'] == (144, 1024) and sep['decshape_var.bias'] == (10,)
    m = HMR(estimate_var=True)
    split = m._engine_state({'head.' + k: v for k, v in m.head.state_dict().items()})
    assert split['head.decpose.weight'].shape == (144, 1024) and split['head.decpose_var.weight'].shape == (144, 1024)
    assert torch.equal(split['head.decshape_var.bias'], m.head.decshape.bias[10:])
    # every trunk the build carries, with the regressor input width the reference derives from get_backbone_info
    for bb, feat in (('resnet50', 2048), ('resnet34', 512), ('resnet18', 512), ('resnet101', 2048), ('resnet152', 2048),
                     ('hrnet_w32-conv', 480), ('hrnet_w48-interp', 720)):
        assert HMR(backbone=bb, use_cam_feats=True).head.fc1.weight.shape == (1024, feat + 144 + 13 + 7)


def test_bench_stage_table_formulae():
    """bench.py's per-stage roofline: frac = max(bytes / 8 TB/s, executed flops / 157.3 TF/s) / t, Winograd entries count
    16/36 of their algorithmic flops, stages are grouped over both trunks."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    e = lambda kernel, label, ms, flops, by, n=1: {'kernel': kernel, 'label': label, 'ms': ms, 'flops': flops, 'bytes': by, 'launches': n}
    entries = [
        e('conv_igemm_f32<64x64,2x2>', 'backbone.layer3.1.conv1', 0.25, 26.3e9, 0.3e9),
        e('conv_igemm_f32<64x64,2x2>', 'backbone.layer3.2.conv1', 0.25, 26.3e9, 0.3e9),
        e('conv_wino_f32<32t x64,F(2x2,3x3)>', 'backbone.layer3.1.conv2', 0.25, 59.2e9, 0.1e9),
        e('conv_igemm_f32<128x128,4x2>', 'backbone.layer1.1.conv3', 0.35, 26.3e9, 1.85e9),
        e('maxpool3x3s2_f32', 'backbone.maxpool', 0.23, 0.0, 1.03e9),
        e('conv_igemm_f32<64x64,2x2,splitK>', 'head.ief_collapsed', 0.02, 0.18e9, 4e6),
        e('smpl_skin_lbs', 'smpl', 0.047, 3.3e9, 39.6e6),
        e('smpl_joints_project', 'smpl', 0.025, 0.095e9, 21.8e6),
    ]
    rows = {r['stage']: r for r in bench.stage_table(entries)}
    assert rows['layer3.conv1']['launches'] == 2 and rows['layer3.conv1']['bound'] == 'mfma'
    assert abs(rows['layer3.conv1']['frac'] - (52.6e9 / 157.3e12) / 0.5e-3) < 1e-3
    assert abs(rows['layer3.conv2']['TFLOPs'] - 59.2e9 * 16 / 36 / 0.25e-3 / 1e12) < 0.01          # executed, not algorithmic
    assert rows['layer1.conv3']['bound'] == 'hbm' and abs(rows['layer1.conv3']['frac'] - (1.85e9 / 8e12) / 0.35e-3) < 1e-3
    assert rows['stem.maxpool']['bound'] == 'hbm' and rows['hmr.regressor']['launches'] == 1
    sk = [r for r in rows.values() if r['kernel'] == 'smpl_skin_lbs'][0]       # the skinning contractions run on the matrix cores
    assert sk['bound'] == 'mfma' and abs(sk['frac'] - (3.3e9 / 157.3e12) / 0.047e-3) < 1e-3
    jp = [r for r in rows.values() if r['kernel'] == 'smpl_joints_project'][0]
    assert jp['bound'] in ('hbm', 'valu')                                       # no matrix instructions: never labelled mfma
    assert all(0 < r['frac'] <= 1.0 for r in rows.values())
    roof = bench.roofline_from_profile(entries)
    igemm_ms = 0.25 + 0.25 + 0.35 + 0.02
    assert abs(roof['achieved'] - (26.3e9 * 3 + 0.18e9) / (igemm_ms * 1e-3) / 1e12) < 0.01
    assert roof['second_kernel']['executed_mfma_TFLOPs'] == round(59.2e9 / 0.25e-3 / 1e12 * 16 / 36, 2)


def test_signature_sees_replaced_tensors():
    """Advisor (round 2): a new nn.Parameter, a replaced submodule or a child's load_state_dict(assign=True) must change
    the module signature, or the forward would keep running on the stale device copy of the weights."""
    import copy
    import torch.nn as nn
    from spec_amd.modules import CameraRegressorNetwork
    m = CameraRegressorNetwork()
    s0 = m._signature()
    assert m._signature() == s0                                   # stable when nothing changed
    name, child = next((n, c) for n, c in m.named_modules() if n and isinstance(getattr(c, 'weight', None), nn.Parameter))
    child.weight = nn.Parameter(child.weight.detach().clone())    # new Parameter object on a CHILD module
    s1 = m._signature()
    assert s1 != s0
    top = next(n for n, _ in m.named_children())
    setattr(m, top, copy.deepcopy(getattr(m, top)))               # replaced submodule
    s2 = m._signature()
    assert s2 != s1
    name, child = next((n, c) for n, c in m.named_modules() if n and isinstance(getattr(c, 'weight', None), nn.Parameter))
    sd = {k: v.clone() for k, v in child.state_dict().items()}
    child.load_state_dict(sd, assign=True)                        # child-level assign: tensors swapped under the parent
    s3 = m._signature()
    assert s3 != s2
    with torch.no_grad():
        child.weight.add_(1.0)                                    # in-place edit bumps _version
    assert m._signature() != s3
