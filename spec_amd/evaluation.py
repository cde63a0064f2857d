"""Config-5 evaluation flow on MI355X: what ``python scripts/spec_eval.py --cfg data/spec/checkpoints/spec_config.yaml
--opts DATASET.VAL_DS spec-syn`` does in the reference (scripts/spec_eval.py:35-82 -> ``pl.Trainer.test`` ->
``SPECTrainer.validation_step`` spec/trainer.py:230-364 -> ``compute_error`` spec/utils/compute_error.py:89-223),
reading the reference's files in their real container formats:

* ``spec_config.yaml``                      - the yacs dump of ``spec/config.py`` (keys below), plain YAML;
* ``TRAINING.PRETRAINED_LIT`` / ``--ckpt``  - Lightning checkpoint, ``['state_dict']`` with the ``model.`` prefix and
                                               pickled foreign classes (``spec_amd.checkpoint.read_checkpoint``);
* ``data/body_models/smpl/SMPL_NEUTRAL.pkl`` (chumpy / scipy-sparse members), ``data/J_regressor_extra.npy``,
  ``data/smpl_mean_params.npz``, ``data/J_regressor_h36m.npy``  (spec/config.py:34-37);
* ``data/dataset_folders/<ds>/annotations/test.npz`` + the images under ``data/dataset_folders/<ds>``
  (spec/config.py:39-56; keys ``imgname scale center pose shape cam_rotmat cam_int camcalib_pitch camcalib_roll
  camcalib_vfov camcalib_f_pix``, spec/dataset/cam_dataset.py:56-146).

Images are decoded with Pillow on the host (the reference uses cv2.imread), uploaded as uint8 frames and cropped /
normalised on the device the way the reference dataset does it (``specmi_crop_resize_normalize``: PARE's ``crop`` = integer box
copy + cv2.resize bilinear, clip, / 255, Normalize).  Everything after that stays in HBM: forward, metrics.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import assets, io_formats, metrics
from .cam_utils import cam_params_from_angles
from .checkpoint import load_pretrained_model, read_checkpoint
from .preprocess import dataset_crops

# spec/config.py:34-56
DATASET_FOLDERS = {'spec-mtp': 'data/dataset_folders/spec-mtp', 'spec-syn': 'data/dataset_folders/spec-syn',
                   '3dpw-test-cam': 'data/dataset_folders/3dpw'}
DATASET_FILES = {'spec-mtp': 'data/dataset_folders/spec-mtp/annotations/test.npz',
                 'spec-syn': 'data/dataset_folders/spec-syn/annotations/test.npz',
                 '3dpw-test-cam': 'data/dataset_extras/3dpw_test_0yaw_inverseyz_w_camcalib.npz'}
README_TABLE = {'spec-mtp': (124.3, 71.8, 147.1), 'spec-syn': (74.9, 54.5, 90.5), '3dpw-test-cam': (106.7, 53.3, 124.7)}

DEFAULTS = {   # the hparams of spec/config.py the evaluation reads
    'LOG_DIR': 'logs/experiments', 'METHOD': 'hmr_cam',
    'DATASET': {'BATCH_SIZE': 64, 'VAL_DS': 'spec-syn_spec-mtp_3dpw-test-cam', 'IMG_RES': 224},
    'TRAINING': {'PRETRAINED_LIT': None},
    'TESTING': {'USE_GT_CAM': False, 'SAVE_RESULTS': True},
    'HMR': {'BACKBONE': 'resnet50', 'USE_CAM_FEATS': False},
}


def load_config(cfg_path: Optional[str], opts: Optional[List[str]] = None) -> dict:
    """YAML config (yacs dump) merged over the defaults, then ``--opts KEY.SUB value ...`` overrides."""
    import copy
    import yaml
    hp = copy.deepcopy(DEFAULTS)

    def merge(dst, src):
        for k, v in (src or {}).items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                merge(dst[k], v)
            else:
                dst[k] = v
    if cfg_path:
        with open(cfg_path) as f:
            merge(hp, yaml.safe_load(f))
    opts = list(opts or [])
    if len(opts) % 2:
        raise ValueError('--opts takes KEY value pairs')
    for k, v in zip(opts[0::2], opts[1::2]):
        node = hp
        parts = k.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(v) if isinstance(v, str) else v
    return hp


def read_image_rgb(path: str) -> np.ndarray:
    """(H,W,3) uint8 RGB (the reference's ``read_img`` = cv2.imread + BGR->RGB)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert('RGB'))       # a writable copy (torch.from_numpy needs one)


class EvalDataset:
    """The evaluation half of ``CamDataset`` (spec/dataset/cam_dataset.py): annotation arrays + image loading; no
    augmentation (is_train=False: flip 0, rot 0, sc 1, pn 1)."""

    def __init__(self, name: str, data_root: str = '.', dataset_file: Optional[str] = None, img_dir: Optional[str] = None):
        self.name = name
        self.img_dir = img_dir or os.path.join(data_root, DATASET_FOLDERS[name])
        self.data = dict(np.load(dataset_file or os.path.join(data_root, DATASET_FILES[name])))
        self.imgname = self.data['imgname']
        self.n = len(self.imgname)

    def __len__(self):
        return self.n

    def batch(self, idx, device, img_res=224, use_gt_cam=False) -> Dict[str, torch.Tensor]:
        d = self.data
        crops, shapes = [], []
        for i in idx:
            frame = torch.from_numpy(read_image_rgb(os.path.join(self.img_dir, str(self.imgname[i])))).to(device)
            H, W = frame.shape[:2]
            crops.append(dataset_crops(frame, d['center'][i:i + 1], d['scale'][i:i + 1], img_res))   # cam_dataset.py:367-377
            shapes.append((H, W))
        shapes = np.asarray(shapes, np.float32)
        f = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).to(device)
        img_h, img_w = f(shapes[:, 0]), f(shapes[:, 1])
        if use_gt_cam:
            R = f(d['cam_rotmat'][idx])
            if 'cam_int' in d:
                K = f(d['cam_int'][idx])
            else:                                     # cam_dataset.py:586-604: built from focal_length, K[2,2] = 0
                fl = np.asarray(d['focal_length'][idx], np.float32).reshape(len(idx), -1)
                K = torch.zeros(len(idx), 3, 3, device=device)
                K[:, 0, 0], K[:, 1, 1] = f(fl[:, 0]), f(fl[:, -1])
                K[:, 0, 2], K[:, 1, 2] = img_w / 2, img_h / 2
        else:                                         # cam_dataset.py:617-653: precomputed CamCalib predictions
            R, K = cam_params_from_angles(d['camcalib_pitch'][idx], d['camcalib_roll'][idx], d['camcalib_f_pix'][idx],
                                          shapes[:, 1], shapes[:, 0], device=device)
        return {'img': torch.cat(crops), 'cam_rotmat': R, 'cam_int': K, 'scale': f(d['scale'][idx]),
                'center': f(d['center'][idx]), 'img_h': img_h, 'img_w': img_w,
                'imgname': [os.path.join(self.img_dir, str(self.imgname[i])) for i in idx]}


@torch.no_grad()
def run_evaluation(hparams: dict, data_root: str = '.', ckpt: Optional[str] = None, log=print, limit: Optional[int] = None,
                   device='cuda') -> Dict[str, dict]:
    """Test + compute_error for every dataset of ``DATASET.VAL_DS``; returns {dataset: compute_error result}."""
    from .modules import HMR
    dev = torch.device(device)
    if hparams.get('METHOD') != 'hmr_cam':
        raise ValueError(f"METHOD {hparams.get('METHOD')!r} is undefined (spec/trainer.py:47-69 builds HMR for 'hmr_cam' only)")
    cwd = os.getcwd()
    os.chdir(data_root)                     # the reference's asset paths are relative to the repo root
    try:
        assets.load_assets()
        hm = HMR(backbone=hparams['HMR']['BACKBONE'], img_res=hparams['DATASET']['IMG_RES'], pretrained=None,
                 use_cam_feats=bool(hparams['HMR']['USE_CAM_FEATS']), use_cam=True)
        ckpt = ckpt or hparams['TRAINING']['PRETRAINED_LIT']
        if ckpt is None:
            raise ValueError('no checkpoint: set TRAINING.PRETRAINED_LIT in the config or pass --ckpt')
        state = read_checkpoint(ckpt)['state_dict']
        load_pretrained_model(hm, state, overwrite_shape_mismatch=True, remove_lightning=True)
        Jh36m = np.load('data/J_regressor_h36m.npy')
    finally:
        os.chdir(cwd)
    hm.to(dev).eval().commit(dev, freeze=True)
    body = metrics.BodyModel(assets.smpl_model(), device=dev)
    bs = int(hparams['DATASET']['BATCH_SIZE'])
    log_dir = hparams['LOG_DIR'] if os.path.isabs(hparams['LOG_DIR']) else os.path.join(data_root, hparams['LOG_DIR'])
    results = {}
    for name in str(hparams['DATASET']['VAL_DS']).split('_'):
        ds = EvalDataset(name, data_root)
        n = len(ds) if limit is None else min(limit, len(ds))
        dump = io_formats.EvalDump()
        for b0 in range(0, n, bs):
            idx = np.arange(b0, min(n, b0 + bs))
            b = ds.batch(idx, dev, hparams['DATASET']['IMG_RES'], bool(hparams['TESTING']['USE_GT_CAM']))
            # positional call of spec/trainer.py:139
            pred = hm(b['img'], b['cam_rotmat'], b['cam_int'], b['scale'], b['center'], b['img_w'], b['img_h'])
            dump.add(pred, imgnames=b['imgname'], dataset_name=name)
        path = dump.write(log_dir, name)
        log(f'wrote {path}')
        if n != len(ds):                            # a truncated run scores the truncated annotations
            sub = os.path.join(log_dir, f'_annotations_{name}_first{n}.npz')
            np.savez(sub, **{k: v[:n] for k, v in ds.data.items() if getattr(v, 'shape', ()) and v.shape[0] == len(ds)})
            res = metrics.compute_error(path, dataset_file=sub, data_root=data_root, body_model=body,
                                        J_regressor_h36m=Jh36m, log=log)
        else:
            res = metrics.compute_error(path, data_root=data_root, body_model=body, J_regressor_h36m=Jh36m, log=log)
        m = res['mean']
        ours = ((m['wmpjpe'], m['pampjpe'], m['wv2v']) if name == '3dpw-test-cam'
                else (m['wmpjpe_24'], m['pampjpe_24'], m['wv2v']))
        ref = README_TABLE.get(name)
        if ref:
            log(f'{name}: W-MPJPE {ours[0]:.1f} (README {ref[0]})  PA-MPJPE {ours[1]:.1f} (README {ref[1]})  '
                f'W-PVE {ours[2]:.1f} (README {ref[2]})')
            # the number BASELINE.json asks for: |delta W-MPJPE| vs the reference's published table (README.md:155-159);
            # meaningful only on the real assets - a stand-in tree holds synthetic weights
            d = [ours[i] - ref[i] for i in range(3)]
            log(f'{name}: delta vs README  W-MPJPE {d[0]:+.2f} mm  PA-MPJPE {d[1]:+.2f} mm  W-PVE {d[2]:+.2f} mm  '
                f'(target |delta W-MPJPE| <= 0.1 mm: {"met" if abs(d[0]) <= 0.1 else "NOT met"})')
            res['readme_delta_mm'] = {'wmpjpe': d[0], 'pampjpe': d[1], 'wpve': d[2]}
        results[name] = res
    return results


# ------------------------------------------------------------------------------------------------------------
# stand-in ``data/`` tree in the REAL container formats (tests / dry runs; the licensed files cannot ship)
# ------------------------------------------------------------------------------------------------------------
def write_standin_data_tree(root: str, n_images: int = 6, seed: int = 7, dataset: str = 'spec-syn',
                            hmr_seed: int = 1002, smpl_seed: int = 1003) -> dict:
    """Writes under ``root`` everything run_evaluation reads, with synthetic numbers but the real formats: a Lightning
    ``.ckpt`` (``model.``-prefixed keys, trainer-level ``smpl.*`` / ``J_regressor`` keys that must be ignored, a pickled
    foreign hyper-parameter class), an SMPL ``.pkl`` with chumpy-pickled and scipy-sparse members, the ``.npy`` / ``.npz``
    side files, the yacs-style YAML, annotations and PNG frames.  Returns the ground truth it used."""
    import pickle
    import scipy.sparse as sp
    import yaml
    from PIL import Image
    from . import synth
    rng = np.random.default_rng(seed)
    j = lambda *p: os.path.join(root, *p)
    for d in ('data/body_models/smpl', 'data/spec/checkpoints', f'data/dataset_folders/{dataset}/annotations',
              f'data/dataset_folders/{dataset}/images', 'data/camcalib/checkpoints', 'data/sample_images'):
        os.makedirs(j(d), exist_ok=True)
    model = synth.smpl_model(smpl_seed)
    V = model['v_template'].shape[0]

    # ---- SMPL pickle: chumpy arrays (class chumpy.ch.Ch, state {'x': array}) + scipy sparse J_regressor -------
    import sys
    import types
    ch_mod, ch_sub = types.ModuleType('chumpy'), types.ModuleType('chumpy.ch')

    class Ch:                                      # pickles as chumpy.ch.Ch, like the official SMPL files
        def __init__(self, x):
            self.x = np.asarray(x)

        def __getstate__(self):
            return {'x': self.x}
    Ch.__module__, Ch.__qualname__ = 'chumpy.ch', 'Ch'
    ch_sub.Ch = Ch
    ch_mod.ch = ch_sub
    saved = {k: sys.modules.get(k) for k in ('chumpy', 'chumpy.ch')}
    sys.modules['chumpy'], sys.modules['chumpy.ch'] = ch_mod, ch_sub
    try:
        kintree = np.stack([np.where(model['parents'] < 0, 2 ** 32 - 1, model['parents']).astype(np.uint32),
                            np.arange(24, dtype=np.uint32)])
        smpl_pkl = {
            'v_template': Ch(model['v_template'].astype(np.float64)),
            'shapedirs': Ch(model['shapedirs'].astype(np.float64)),
            'posedirs': Ch(model['posedirs'].T.reshape(V, 3, 207).astype(np.float64)),
            'J_regressor': sp.csc_matrix(model['J_regressor'].astype(np.float64)),
            'weights': Ch(model['lbs_weights'].astype(np.float64)),
            'kintree_table': kintree, 'f': np.zeros((13776, 3), np.uint32),
            'bs_type': 'lrotmin', 'bs_style': 'lbs',
        }
        with open(j('data/body_models/smpl/SMPL_NEUTRAL.pkl'), 'wb') as f:
            pickle.dump(smpl_pkl, f, protocol=2)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    np.save(j('data/J_regressor_extra.npy'), model['J_regressor_extra'])
    np.save(j('data/J_regressor_h36m.npy'), synth.h36m_regressor(smpl_seed))
    mean = {'pose': np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), 24), 'shape': np.zeros(10, np.float32),
            'cam': np.array([0.9, 0., 0.], np.float32)}
    np.savez(j('data/smpl_mean_params.npz'), **mean)

    # ---- Lightning checkpoint -------------------------------------------------------------------------------
    hs = synth.hmr_state(hmr_seed, True)
    sd = {'model.' + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in hs.items()}
    sd['model.backbone.bn1.num_batches_tracked'] = torch.tensor(7)
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights'):      # smplx buffers of model.smpl.smpl
        sd['model.smpl.smpl.' + k] = torch.from_numpy(np.ascontiguousarray(model[k]))
        sd['smpl.' + k] = sd['model.smpl.smpl.' + k]                                      # trainer-level copies: ignored
        sd['smpl_native.' + k] = sd['model.smpl.smpl.' + k]
    sd['J_regressor'] = torch.from_numpy(synth.h36m_regressor(smpl_seed))
    fake_mod = types.ModuleType('yacs.config')

    class CfgNode(dict):
        pass
    CfgNode.__module__, CfgNode.__qualname__ = 'yacs.config', 'CfgNode'
    fake_mod.CfgNode = CfgNode
    yacs_pkg = types.ModuleType('yacs')
    yacs_pkg.config = fake_mod
    saved = {k: sys.modules.get(k) for k in ('yacs', 'yacs.config')}
    sys.modules['yacs'], sys.modules['yacs.config'] = yacs_pkg, fake_mod
    try:
        torch.save({'epoch': 3, 'global_step': 1234, 'pytorch-lightning_version': '1.1.8', 'state_dict': sd,
                    'hyper_parameters': CfgNode(METHOD='hmr_cam', HMR=CfgNode(USE_CAM_FEATS=True))},
                   j('data/spec/checkpoints/spec_checkpoint.ckpt'))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    # CamCalib Lightning checkpoint (scripts/camcalib_demo.py:39,80-81: strict load after stripping 'model.')
    cs = synth.camcalib_state(1001)
    torch.save({'epoch': 26, 'global_step': 337742,
                'state_dict': {'model.' + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in cs.items()}},
               j('data/camcalib/checkpoints/camcalib_sa_biased_l2.ckpt'))
    for i in range(2):                                             # data/sample_images of the demo (README.md:100-104)
        Image.fromarray((rng.random((300, 420, 3)) * 255).astype(np.uint8)).save(j('data/sample_images', f'im{i}.png'))
    with open(j('data/spec/checkpoints/spec_config.yaml'), 'w') as f:
        yaml.safe_dump({'METHOD': 'hmr_cam', 'LOG_DIR': 'logs/eval_standin',
                        'DATASET': {'BATCH_SIZE': 4, 'VAL_DS': dataset, 'IMG_RES': 224},
                        'TRAINING': {'PRETRAINED_LIT': 'data/spec/checkpoints/spec_checkpoint.ckpt'},
                        'HMR': {'BACKBONE': 'resnet50', 'USE_CAM_FEATS': True}}, f)

    # ---- annotations + frames -------------------------------------------------------------------------------
    H, W = 360, 480
    names = []
    for i in range(n_images):
        img = (rng.random((H, W, 3)) * 255).astype(np.uint8)
        names.append(f'images/frame_{i:04d}.png')
        Image.fromarray(img).save(j(f'data/dataset_folders/{dataset}', names[-1]))
    pose = (rng.standard_normal((n_images, 72)) * 0.25).astype(np.float64)
    pose[0] = 0.0                                              # zero pose: Rodrigues at the 1e-8 guard
    shape = (rng.standard_normal((n_images, 10)) * 0.5).astype(np.float64)
    pitch = rng.uniform(-0.4, 0.4, n_images).astype(np.float32)
    roll = rng.uniform(-0.2, 0.2, n_images).astype(np.float32)
    vfov = rng.uniform(0.6, 1.4, n_images).astype(np.float32)
    f_pix = (H / 2. / np.tan(vfov / 2.)).astype(np.float64)
    def rx_rz(p, r):                                           # any rotation serves as stand-in ground truth
        cp, sp_, cr, sr = np.cos(p), np.sin(p), np.cos(r), np.sin(r)
        Rx = np.array([[1, 0, 0], [0, cp, -sp_], [0, sp_, cp]])
        Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        return Rx @ Rz
    Rgt = np.stack([rx_rz(p + 0.03, r - 0.02) for p, r in zip(pitch, roll)])
    ann = {'imgname': np.array(names), 'scale': rng.uniform(0.9, 1.5, n_images), 'pose': pose, 'shape': shape,
           'center': np.stack([rng.uniform(150, 330, n_images), rng.uniform(120, 240, n_images)], 1),
           'cam_rotmat': Rgt.astype(np.float64), 'camcalib_pitch': pitch, 'camcalib_roll': roll,
           'camcalib_vfov': vfov, 'camcalib_f_pix': f_pix}
    if dataset != 'spec-syn':
        ann['pose_cam'] = (rng.standard_normal((n_images, 72)) * 0.25).astype(np.float64)
        import joblib
        joblib.dump(torch.from_numpy(Rgt), j(f'data/camcalib/{dataset}_cam_rotmat.pkl'))
    np.savez(j(DATASET_FILES[dataset]), **ann)
    return {'annotations': ann, 'smpl_model': model, 'hmr_state': hs, 'camcalib_state': cs, 'frame_hw': (H, W)}
