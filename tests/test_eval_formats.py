"""Config-5 plumbing without a GPU: the reference's files in their REAL container formats (Lightning checkpoint with
``model.`` prefix + foreign pickled classes + trainer-level keys, SMPL pickle with chumpy / scipy-sparse members,
.npy / .npz side files, yacs-style YAML) are written by ``write_standin_data_tree`` and read back by the loaders
``scripts/spec_eval.py`` uses."""
import os
import pickle

import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    from spec_amd import evaluation
    d = str(tmp_path_factory.mktemp('standin'))
    gt = evaluation.write_standin_data_tree(d, n_images=3)
    return d, gt


def test_lightning_checkpoint_real_format(tree):
    from spec_amd.checkpoint import load_pretrained_model, read_checkpoint
    from spec_amd import assets
    from spec_amd.modules import HMR
    d, gt = tree
    path = os.path.join(d, 'data/spec/checkpoints/spec_checkpoint.ckpt')
    with pytest.raises(pickle.UnpicklingError):                 # really contains a class torch's safe loader refuses
        torch.load(path, map_location='cpu', weights_only=True)
    ck = read_checkpoint(path)
    sd = ck['state_dict']
    assert 'model.head.init_pose' in sd                        # scripts/spec_eval.py:57 of the reference checks this key
    assert any(k.startswith('smpl_native.') for k in sd) and 'J_regressor' in sd
    assets.use_synthetic_assets(1003)
    hm = HMR(use_cam=True, use_cam_feats=True)
    load_pretrained_model(hm, sd, overwrite_shape_mismatch=True, remove_lightning=True)   # spec/tester.py:70
    for k in ('head.fc1.weight', 'backbone.layer3.2.conv2.weight', 'backbone.bn1.running_var', 'head.init_pose'):
        assert np.array_equal(hm.state_dict()[k].numpy().reshape(-1), gt['hmr_state'][k].reshape(-1)), k
    # 2205-column checkpoint into a use_cam_feats model: the reference's growth patch (copies of the last 7 columns)
    sd2 = dict(sd)
    sd2['model.head.fc1.weight'] = sd['model.head.fc1.weight'][:, :2205].clone()
    hm2 = HMR(use_cam=True, use_cam_feats=True)
    load_pretrained_model(hm2, sd2, overwrite_shape_mismatch=True, remove_lightning=True)
    w = hm2.state_dict()['head.fc1.weight']
    assert w.shape == (1024, 2212) and torch.equal(w[:, 2205:], w[:, 2198:2205])


def test_read_checkpoint_does_not_mask_io_errors(tmp_path):
    from spec_amd.checkpoint import read_checkpoint
    with pytest.raises(FileNotFoundError):
        read_checkpoint(str(tmp_path / 'missing.ckpt'))
    bad = tmp_path / 'corrupt.ckpt'
    bad.write_bytes(b'PK\x03\x04 this is not a checkpoint')
    with pytest.raises(Exception) as e:
        read_checkpoint(str(bad))
    assert not isinstance(e.value, AttributeError)


def test_tolerant_unpickler_runs_nothing_outside_allow_list(tmp_path):
    """A checkpoint naming os.system must come back as an inert stub, not execute."""
    from spec_amd.checkpoint import _TolerantUnpickler
    import io

    class Evil:
        def __reduce__(self):
            return (os.system, ('echo pwned > ' + str(tmp_path / 'pwned'),))
    buf = io.BytesIO(pickle.dumps({'x': Evil()}))
    out = _TolerantUnpickler(buf).load()
    assert not (tmp_path / 'pwned').exists() and type(out['x']).__name__ == '_AnyStub'


def test_tolerant_unpickler_does_not_trust_whole_packages(tmp_path):
    """Callables that live INSIDE torch / numpy / builtins but are not tensor reconstructors must be stubbed too
    (advisor, round 2: torch.utils.collect_env.run is a shell, numpy.load can unpickle, builtins.map / type ...)."""
    from spec_amd.checkpoint import _TolerantUnpickler, _ALLOWED
    import io
    import builtins
    import torch.utils.collect_env as ce

    marker = tmp_path / 'pwned2'

    class ViaTorch:
        def __reduce__(self):
            return (ce.run, ('echo pwned > ' + str(marker),))

    class ViaNumpy:
        def __reduce__(self):
            return (np.load, (str(tmp_path / 'x.npy'),))

    class ViaBuiltins:
        def __reduce__(self):
            return (builtins.vars, ())

    out = _TolerantUnpickler(io.BytesIO(pickle.dumps({'a': ViaTorch(), 'b': ViaNumpy(), 'c': ViaBuiltins()}))).load()
    assert not marker.exists()
    assert all(type(v).__name__ == '_AnyStub' for v in out.values())
    for bad in (('torch.utils.collect_env', 'run'), ('numpy', 'load'), ('torch', 'load'), ('torch.hub', 'load'),
                ('builtins', 'eval'), ('builtins', 'type'), ('builtins', 'map'), ('builtins', 'globals'),
                ('torch.storage', '_load_from_bytes')):
        assert bad not in _ALLOWED
    # and real tensors / arrays / ordered dicts still come through
    from collections import OrderedDict
    good = {'w': torch.arange(6.).reshape(2, 3), 'n': np.arange(4, dtype=np.float32), 'o': OrderedDict(a=1)}
    buf = io.BytesIO()
    torch.save(good, buf)
    buf.seek(0)
    from spec_amd.checkpoint import _tolerant_pickle
    back = torch.load(buf, weights_only=False, pickle_module=_tolerant_pickle)
    assert torch.equal(back['w'], good['w']) and np.array_equal(back['n'], good['n']) and back['o'] == good['o']


def test_strict_load_keeps_reference_failure_mode():
    from spec_amd.checkpoint import load_pretrained_model
    from spec_amd.modules import CameraRegressorNetwork
    from spec_amd import synth
    m = CameraRegressorNetwork()
    sd = {'model.' + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.camcalib_state(1001).items()}
    load_pretrained_model(m, sd, remove_lightning=True, strict=True)               # scripts/camcalib_demo.py:81
    sd_bad = dict(sd); sd_bad['model.fc_vfov.weight'] = torch.zeros(128, 2048)
    with pytest.raises(RuntimeError):
        load_pretrained_model(m, sd_bad, remove_lightning=True, strict=True)
    sd_missing = dict(sd); del sd_missing['model.fc_roll.bias']
    with pytest.raises(RuntimeError):
        load_pretrained_model(m, sd_missing, remove_lightning=True, strict=True)


def test_smpl_pickle_real_format(tree):
    from spec_amd import assets
    d, gt = tree
    raw = open(os.path.join(d, 'data/body_models/smpl/SMPL_NEUTRAL.pkl'), 'rb').read()
    assert b'chumpy' in raw and b'scipy.sparse' in raw          # the members really are chumpy / scipy objects
    cwd = os.getcwd()
    os.chdir(d)
    try:
        m = assets.load_assets()
        mp = assets.mean_params()
    finally:
        os.chdir(cwd)
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'J_regressor_extra', 'parents'):
        assert np.array_equal(m[k], gt['smpl_model'][k]), k
    assert mp['pose'].shape == (144,) and mp['cam'].shape == (3,)
    assets.use_synthetic_assets(1003)


def test_npz_smpl_refuses_pickled_members(tmp_path):
    from spec_amd import assets
    p = str(tmp_path / 'SMPL_NEUTRAL.npz')
    np.savez(p, v_template=np.array([{'a': 1}], dtype=object))
    with pytest.raises(ValueError):
        assets.load_smpl_file(p, np.zeros((9, 1), np.float32))


def test_config_and_annotations(tree):
    from spec_amd import evaluation
    d, gt = tree
    hp = evaluation.load_config(os.path.join(d, 'data/spec/checkpoints/spec_config.yaml'),
                                ['DATASET.VAL_DS', 'spec-syn_spec-mtp', 'TESTING.USE_GT_CAM', 'True', 'DATASET.BATCH_SIZE', '8'])
    assert hp['METHOD'] == 'hmr_cam' and hp['HMR']['USE_CAM_FEATS'] is True and hp['TESTING']['USE_GT_CAM'] is True
    assert hp['DATASET']['BATCH_SIZE'] == 8 and hp['DATASET']['VAL_DS'].split('_') == ['spec-syn', 'spec-mtp']
    ds = evaluation.EvalDataset('spec-syn', d)
    assert len(ds) == 3 and ds.data['pose'].shape == (3, 72)
    img = evaluation.read_image_rgb(os.path.join(ds.img_dir, str(ds.imgname[0])))
    assert img.dtype == np.uint8 and img.shape == (gt['frame_hw'][0], gt['frame_hw'][1], 3)
