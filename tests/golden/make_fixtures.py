#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ -- run ONLY in the build container.

The reference (``/root/reference``) holds no tests or golden vectors, so
the vectors are produced here by importing the reference's OWN in-tree modules

    spec/models/hmr.py, camcalib/model.py, camcalib/cam_utils.py,
    spec/utils/cam_params.py, spec/constants.py

over the name shim in ``oracle/refshim.py`` (the un-vendored ``pare``/``smplx`` leaf modules
are bound to the oracle's restatements) and running them on seeded synthetic tensors
(``spec_amd.synth``).  Only seeds, small inputs and the expected outputs are stored; the
reference's source never enters the repo.  Usage:

    python tests/golden/make_fixtures.py                 # (re)write tests/golden/*.npz over the shim
    python tests/golden/make_fixtures.py --selfcheck     # regenerate into a temp dir, diff against the committed
                                                         # files (must be bit-identical; what CI runs)
    python tests/golden/make_fixtures.py --upstream      # the ONE-COMMAND UPSTREAM PIN: wherever the reference's real leaf
                                                         # packages import (pip install smplx==0.1.28 loguru yacs + the
                                                         # pare checkout of requirements.txt:28; opencv-python for the
                                                         # crops), bind THEM instead of oracle/refshim.py, regenerate every
                                                         # fixture into a temp dir and print the max deviation per array
                                                         # against the committed ones; add --write to replace them
    python tests/golden/make_fixtures.py --upstream --data-root /path/with/data    # real SMPL assets instead of the
                                                         # synthetic stand-in tree (outputs then differ by construction:
                                                         # only shapes / keys are compared)

With ``--upstream`` the real ``pare`` heads load the body model from ``data/body_models/smpl`` relative to the working
directory; unless ``--data-root`` is given the script writes a stand-in tree there first (the SAME synthetic SMPL model
the committed fixtures were made with, in the official pickle format: ``spec_amd.evaluation.write_standin_data_tree``),
so the regenerated arrays are directly comparable: any deviation is upstream leaf arithmetic vs this build's restatement.
"""
import argparse
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT_DIR = os.path.dirname(os.path.abspath(__file__))

from spec_amd import synth  # noqa: E402
from oracle import refshim, heads  # noqa: E402
from oracle.models import load_numpy_state  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)

SEED_CAMCALIB, SEED_HMR, SEED_SMPL, SEED_IMG = 1001, 1002, 1003, 20210001
PL_SEED_CC, PL_SEED_HM, PL_SEED_IMG, PL_DEC_GAIN, PL_CAM_GAIN = 2101, 2102, 2103, 2.0, 1.0     # = tests/util.py


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def generate(OUT, upstream=False):
    smpl_model = synth.smpl_model(SEED_SMPL)
    heads.set_assets(smpl_model=smpl_model)
    ref = refshim.import_reference(upstream=upstream)
    print('leaf packages bound:', dict(refshim.BOUND))
    RC = ref['constants']

    # ---- (1) index tables and constants, from the reference's spec/constants.py ----------
    joint_map = np.array([RC.JOINT_MAP[n] for n in RC.JOINT_NAMES], dtype=np.int32)
    np.savez(os.path.join(OUT, 'constants.npz'),
             joint_map=joint_map,
             joint_names=np.array(RC.JOINT_NAMES),
             h36m_to_j14=np.array(RC.H36M_TO_J14, dtype=np.int32),
             j24_to_j14=np.array(RC.J24_TO_J14, dtype=np.int32),
             h36m_to_j17=np.array(RC.H36M_TO_J17, dtype=np.int32),
             j24_to_j17=np.array(RC.J24_TO_J17, dtype=np.int32),
             img_norm_mean=np.array(RC.IMG_NORM_MEAN), img_norm_std=np.array(RC.IMG_NORM_STD))

    # ---- (2) CamCalib decode through the reference's camcalib/cam_utils.py ---------------
    CU = ref['cam_utils']
    nb = 256
    rows = []
    for k in (0, 1, 100, 128, 254, 255):                     # one-hot-ish (closed form k/255*2-1)
        r = np.full(nb, -80.0, np.float32); r[k] = 40.0; rows.append(r)
    rows.append(np.zeros(nb, np.float32))                    # uniform -> centre of range
    r = np.full(nb, -60.0, np.float32); r[10] = 5.0; r[200] = 5.0; rows.append(r)   # two peaks
    rnd = synth.normal(77, 'decode.logits', (24, nb), std=3.0)
    logits = np.concatenate([np.stack(rows), rnd], 0)
    lv, lp, lr = logits, np.roll(logits, 3, axis=0), np.roll(logits, 7, axis=0)
    vf, pi, ro = CU.convert_preds_to_angles(t(lv), t(lp), t(lr), loss_type='softargmax_l2')
    vf2, pi2, ro2 = CU.convert_preds_to_angles(t(lv), t(lp), t(lr), loss_type='softargmax_biased_l2')
    assert torch.equal(vf, vf2) and torch.equal(pi, pi2) and torch.equal(ro, ro2)
    np.savez(os.path.join(OUT, 'camcalib_decode.npz'),
             logits_vfov=lv, logits_pitch=lp, logits_roll=lr,
             vfov=vf.numpy(), pitch=pi.numpy(), roll=ro.numpy(),
             vfov_range=np.array([np.min(CU.vfov_bins), np.max(CU.vfov_bins)]),
             pitch_range=np.array([np.min(CU.pitch_bins), np.max(CU.pitch_bins)]),
             roll_range=np.array([-0.6, 0.6]))

    # ---- (3) R, K hand-off through the reference's spec/utils/cam_params.py --------------
    import joblib
    CP = ref['cam_params']
    cases = [(0.1, -0.05, 0.9, 480, 640), (-0.3, 0.2, 1.4, 1080, 1920), (0.0, 0.0, 0.5, 224, 224),
             (0.55, -0.6, 2.0, 500, 450)]
    Rs, Ks, meta = [], [], []
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, 'camcalib'))
        for i, (pitch, roll, vfov, h, w) in enumerate(cases):
            f_pix = np.float64(h / 2. / np.tan(np.float32(vfov) / 2.))   # scripts/camcalib_demo.py:129 (fp32 math)
            joblib.dump({'vfov': np.float32(vfov), 'f_pix': f_pix, 'pitch': np.float32(pitch),
                         'roll': np.float32(roll)}, os.path.join(td, 'camcalib', f'im{i}.jpg.pkl'))
            R, K, *_ = CP.read_cam_params(td, f'/x/im{i}.jpg', (h, w))
            Rs.append(R.numpy()); Ks.append(K.numpy()); meta.append([pitch, roll, vfov, h, w, f_pix])
    np.savez(os.path.join(OUT, 'cam_params.npz'), R=np.stack(Rs), K=np.stack(Ks),
             meta=np.array(meta, dtype=np.float64))

    # ---- (4) CamCalib network through the reference's camcalib/model.py ------------------
    B = 2
    imgs = synth.images(SEED_IMG, B)
    net = ref['camcalib_model'].CameraRegressorNetwork(backbone='resnet50', num_fc_layers=1,
                                                       num_fc_channels=1024).eval()
    load_numpy_state(net, synth.camcalib_state(SEED_CAMCALIB))
    keys = list(net.state_dict().keys())
    lg = net(t(imgs))
    assert isinstance(lg, list) and len(lg) == 3
    ang = CU.convert_preds_to_angles(*lg, loss_type='softargmax_biased_l2')
    np.savez(os.path.join(OUT, 'camcalib_e2e.npz'), seed_weights=SEED_CAMCALIB, seed_images=SEED_IMG,
             batch=B, logits_vfov=lg[0].numpy(), logits_pitch=lg[1].numpy(), logits_roll=lg[2].numpy(),
             vfov=ang[0].numpy(), pitch=ang[1].numpy(), roll=ang[2].numpy(),
             state_keys=np.array(keys))

    # ---- (5) SPEC network through the reference's spec/models/hmr.py ---------------------
    HMR = ref['hmr'].HMR
    for tag, kw, B in (('camfeats', dict(use_cam=True, use_cam_feats=True), 3),
                       ('cam', dict(use_cam=True, use_cam_feats=False), 2),
                       ('nocam', dict(use_cam=False, use_cam_feats=False), 2)):
        imgs = synth.images(SEED_IMG + 1, B)
        scale, center, img_w, img_h = synth.bbox_inputs(SEED_IMG + 1, B, img_w=640., img_h=480.)
        pitch = synth.uniform(5, 'pitch', (B,), -0.5, 0.5)
        roll = synth.uniform(5, 'roll', (B,), -0.4, 0.4)
        fpix = synth.uniform(5, 'fpix', (B,), 300., 900.)
        Rl, Kl = [], []
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, 'camcalib'))
            for i in range(B):
                joblib.dump({'vfov': np.float32(1.0), 'f_pix': np.float64(fpix[i]),
                             'pitch': np.float32(pitch[i]), 'roll': np.float32(roll[i])},
                            os.path.join(td, 'camcalib', f'im{i}.jpg.pkl'))
                R, K, *_ = CP.read_cam_params(td, f'im{i}.jpg', (480, 640))
                Rl.append(R); Kl.append(K)
        R, K = torch.stack(Rl), torch.stack(Kl)
        model = HMR(backbone='resnet50', img_res=224, pretrained=None, **kw).eval()
        load_numpy_state(model, synth.hmr_state(SEED_HMR, use_cam_feats=kw['use_cam_feats']))
        if kw['use_cam']:
            out = model(t(imgs), cam_rotmat=R, cam_intrinsics=K, bbox_scale=t(scale),
                        bbox_center=t(center), img_w=t(img_w), img_h=t(img_h))
        else:
            out = model(t(imgs))
        keys = [k for k in model.state_dict().keys()]
        np.savez_compressed(
            os.path.join(OUT, f'hmr_e2e_{tag}.npz'), seed_weights=SEED_HMR, seed_smpl=SEED_SMPL,
            seed_images=SEED_IMG + 1, batch=B, cam_rotmat=R.numpy(), cam_intrinsics=K.numpy(),
            bbox_scale=scale, bbox_center=center, img_w=img_w, img_h=img_h,
            out_keys=np.array(list(out.keys())), state_keys=np.array(keys),
            **{f'out_{k}': v.numpy() for k, v in out.items()})
        print(tag, {k: tuple(v.shape) for k, v in out.items()})

    # ---- (6) evaluation metrics through the reference's spec/utils/compute_error.py ---------
    CE = ref['compute_error']
    Bm, V = 5, 6890
    rng_seed = 4242
    Jh36m = synth.smpl_model(SEED_SMPL)['J_regressor']            # (24,V) rows sum to 1
    J17 = synth.h36m_regressor(SEED_SMPL)                         # a 17-row regressor
    gt_v = synth.normal(rng_seed, 'gt_verts', (Bm, V, 3), std=0.3)
    pr_v = gt_v + synth.normal(rng_seed, 'noise', (Bm, V, 3), std=0.03) + synth.normal(rng_seed, 'shift', (Bm, 1, 3), std=0.2)
    Jb = t(J17)[None].expand(Bm, -1, -1)
    mpjpe, pampjpe, v2v = CE.eval_single(t(pr_v), t(gt_v), Jb)
    pj = torch.einsum('bik,ji->bjk', t(pr_v), t(Jh36m))
    gj = torch.einsum('bik,ji->bjk', t(gt_v), t(Jh36m))
    mpjpe24, pampjpe24 = CE.eval_j_24(pj, gj)
    np.savez(os.path.join(OUT, 'metrics.npz'), seed=rng_seed, batch=Bm, seed_smpl=SEED_SMPL,
             mpjpe=mpjpe, pampjpe=pampjpe, v2v=v2v, mpjpe24=mpjpe24, pampjpe24=pampjpe24)

    # ---- (7) the same two networks with released-checkpoint-like statistics (synth stats='pretrained_like') -------------
    # BN variances over six decades, zero / negative gammas, dead filters, Student-t filters, O(1)-gain decoders, saturated
    # crops: call sites that load such checkpoints are spec/tester.py:63-71 and scripts/camcalib_demo.py:74-81
    B = 2
    imgs = synth.images(PL_SEED_IMG + 100, B, saturate=True)
    net = ref['camcalib_model'].CameraRegressorNetwork(backbone='resnet50', num_fc_layers=1, num_fc_channels=1024).eval()
    load_numpy_state(net, synth.camcalib_state(PL_SEED_CC, stats='pretrained_like'))
    lg = net(t(imgs))
    ang = CU.convert_preds_to_angles(*lg, loss_type='softargmax_biased_l2')
    np.savez(os.path.join(OUT, 'camcalib_e2e_pl.npz'), seed_weights=PL_SEED_CC, seed_images=PL_SEED_IMG + 100, batch=B,
             logits_vfov=lg[0].numpy(), logits_pitch=lg[1].numpy(), logits_roll=lg[2].numpy(),
             vfov=ang[0].numpy(), pitch=ang[1].numpy(), roll=ang[2].numpy())
    scale, center, img_w, img_h = synth.bbox_inputs(PL_SEED_IMG + 100, B, img_w=640., img_h=480.)
    Rl, Kl = [], []
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, 'camcalib'))
        for i in range(B):
            f_pix = np.float64(480 / 2. / np.tan(ang[0].numpy()[i] / 2.))              # scripts/camcalib_demo.py:129
            joblib.dump({'vfov': ang[0].numpy()[i], 'f_pix': f_pix, 'pitch': ang[1].numpy()[i], 'roll': ang[2].numpy()[i]},
                        os.path.join(td, 'camcalib', f'im{i}.jpg.pkl'))
            R, K, *_ = CP.read_cam_params(td, f'im{i}.jpg', (480, 640))
            Rl.append(R); Kl.append(K)
    R, K = torch.stack(Rl), torch.stack(Kl)
    model = HMR(backbone='resnet50', img_res=224, pretrained=None, use_cam=True, use_cam_feats=True).eval()
    load_numpy_state(model, synth.hmr_state(PL_SEED_HM, True, dec_gain=PL_DEC_GAIN, cam_gain=PL_CAM_GAIN, stats='pretrained_like'))
    out = model(t(imgs), cam_rotmat=R, cam_intrinsics=K, bbox_scale=t(scale), bbox_center=t(center), img_w=t(img_w), img_h=t(img_h))
    np.savez_compressed(os.path.join(OUT, 'hmr_e2e_pl.npz'), seed_weights=PL_SEED_HM, seed_smpl=SEED_SMPL,
                        seed_images=PL_SEED_IMG + 100, batch=B, dec_gain=PL_DEC_GAIN, cam_gain=PL_CAM_GAIN, cam_rotmat=R.numpy(),
                        cam_intrinsics=K.numpy(), bbox_scale=scale, bbox_center=center, img_w=img_w, img_h=img_h,
                        out_keys=np.array(list(out.keys())), **{f'out_{k}': v.numpy() for k, v in out.items()})
    print('pretrained_like', {k: (tuple(v.shape), float(v.abs().max())) for k, v in out.items()})

    # ---- (8) HMR(estimate_var=True) through the reference's spec/models/hmr.py: both decoder layouts of pare's HMRHead ----------
    B = 2
    imgs = synth.images(SEED_IMG + 2, B)
    scale, center, img_w, img_h = synth.bbox_inputs(SEED_IMG + 2, B, img_w=640., img_h=480.)
    from oracle.models import cam_params as _cam_params
    R, K = _cam_params(t(synth.uniform(6, 'pitch', (B,), -0.5, 0.5)), t(synth.uniform(6, 'roll', (B,), -0.4, 0.4)),
                       synth.uniform(6, 'fpix', (B,), 300., 900.), t(img_w), t(img_h))
    for tag, separate, act in (('doubled', False, 'softplus'), ('separate', True, 'sigmoid')):
        model = HMR(backbone='resnet50', img_res=224, pretrained=None, use_cam=True, use_cam_feats=True, estimate_var=True,
                    use_separate_var_branch=separate, uncertainty_activation=act).eval()
        load_numpy_state(model, synth.hmr_state(SEED_HMR, True, estimate_var=True, use_separate_var_branch=separate))
        out = model(t(imgs), cam_rotmat=R, cam_intrinsics=K, bbox_scale=t(scale), bbox_center=t(center), img_w=t(img_w), img_h=t(img_h))
        np.savez_compressed(os.path.join(OUT, f'hmr_e2e_var_{tag}.npz'), seed_weights=SEED_HMR, seed_smpl=SEED_SMPL, seed_images=SEED_IMG + 2,
                            batch=B, uncertainty_activation=np.array(act), cam_rotmat=R.numpy(), cam_intrinsics=K.numpy(), bbox_scale=scale,
                            bbox_center=center, img_w=img_w, img_h=img_h, out_keys=np.array(list(out.keys())),
                            state_keys=np.array([k for k in model.state_dict().keys() if k.startswith('head.')]),
                            **{f'out_{k}': v.numpy() for k, v in out.items()})
        print('estimate_var', tag, {k: tuple(v.shape) for k, v in out.items() if k.endswith('_var')})

    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith('.npz'))
    print('fixtures written to', OUT, 'total bytes', sz)
    return sorted(f for f in os.listdir(OUT) if f.endswith('.npz'))


def compare(new_dir, old_dir, files, values=True, records=None):
    """Per array: max |new - committed| and that over max |committed|.  Returns (worst relative deviation, problems);
    ``records`` (a list) receives one dict per array for the machine-readable report."""
    worst, problems = 0.0, []
    print(f'{"file":24s} {"array":22s} {"shape":18s} {"max abs dev":>12s} {"max rel dev":>12s}')
    for f in files:
        a, b = np.load(os.path.join(new_dir, f), allow_pickle=False), np.load(os.path.join(old_dir, f), allow_pickle=False)
        if sorted(a.files) != sorted(b.files):
            problems.append(f'{f}: array names differ: {sorted(set(a.files) ^ set(b.files))}')
        for k in b.files:
            if k not in a.files:
                continue
            x, y = a[k], b[k]
            if x.shape != y.shape or x.dtype.kind != y.dtype.kind:
                problems.append(f'{f}:{k}: shape / kind {x.shape} {x.dtype} vs committed {y.shape} {y.dtype}')
                continue
            if y.dtype.kind in 'US':
                if not np.array_equal(x, y):
                    problems.append(f'{f}:{k}: strings differ')
                continue
            if not values:
                continue
            x64, y64 = x.astype(np.float64), y.astype(np.float64)
            dev = float(np.abs(x64 - y64).max()) if x.size else 0.0
            rel = dev / max(float(np.abs(y64).max()), 1e-30) if x.size else 0.0
            if y.dtype.kind in 'iub' and dev != 0:
                problems.append(f'{f}:{k}: integer / index array differs')
            worst = max(worst, rel)
            if records is not None:
                records.append({'file': f, 'array': k, 'shape': list(x.shape), 'dtype': str(y.dtype), 'max_abs_dev': dev, 'max_rel_dev': rel,
                                'exact_required': y.dtype.kind in 'iub'})
            if dev != 0 or x.size > 1:
                print(f'{f:24s} {k:22s} {str(x.shape):18s} {dev:12.3e} {rel:12.3e}')
    return worst, problems


def check_opencv():
    """When OpenCV imports: the two restated OpenCV leaves of the crop rows against the REAL binary on
    seeded frames - cv2.warpAffine (INTER_LINEAR, fixed-point path, spec/tester.py:116-128 via pare's
    get_single_image_crop_demo) and cv2.resize on float64 (spec/dataset/cam_dataset.py:253-287 via pare's crop)."""
    try:
        import cv2
    except Exception as e:                          # noqa: BLE001
        print(f'opencv: not importable ({type(e).__name__}) - cv2.warpAffine / cv2.resize restatements stay unpinned; '
              'pip install opencv-python and re-run --upstream')
        return None
    from oracle import preprocess as P
    rng = np.random.default_rng(99)
    worst = 0
    for (h, w) in ((480, 640), (1080, 1920), (333, 517)):
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for (cx, cy, bw, bh, sc) in ((w / 2, h / 2, 200.3, 300.7, 1.0), (20.0, 30.0, 150.0, 90.0, 1.2), (w - 5.0, h - 9.0, 400.0, 420.0, 1.0)):
            M = np.asarray(P.gen_trans_from_patch(cx, cy, bw, bh, 224, 224, sc), dtype=np.float64)
            ours = P.warp_affine_linear_u8(frame, M, 224, 224)
            theirs = cv2.warpAffine(frame, M, (224, 224), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT)
            worst = max(worst, int(np.abs(ours.astype(np.int32) - theirs.astype(np.int32)).max()))
        src = rng.random((h // 3, w // 3, 3))
        d = np.abs(P.cv2_resize_linear_f64(src, 224, 224) - cv2.resize(src, (224, 224), interpolation=cv2.INTER_LINEAR)).max()
        print(f'opencv {cv2.__version__}: cv2.resize float64 {src.shape[:2]} -> 224: max abs dev {d:.3e}')
        worst = max(worst, 0 if d == 0 else 1)
    print(f'opencv {cv2.__version__}: cv2.warpAffine u8 restatement: max grey-level deviation {worst} (0 = bit-exact)')
    return worst


def package_versions(names=('pare', 'smplx', 'loguru', 'opencv-python', 'torch', 'numpy', 'scipy', 'Pillow')):
    """{distribution: version | None} of what this run could have bound (importlib.metadata; a source checkout on
    PYTHONPATH, like pare's, reports 'path:<dir>')."""
    import importlib
    import importlib.metadata as md
    out = {}
    for n in names:
        try:
            out[n] = md.version(n)
        except Exception:                           # noqa: BLE001
            mod = {'opencv-python': 'cv2', 'Pillow': 'PIL'}.get(n, n)
            try:
                m = importlib.import_module(mod)
                out[n] = getattr(m, '__version__', None) or 'path:' + os.path.dirname(getattr(m, '__file__', '') or '')
            except Exception:                       # noqa: BLE001
                out[n] = None
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--upstream', action='store_true', help='bind the real pare / smplx / loguru where they import')
    ap.add_argument('--selfcheck', action='store_true', help='regenerate over the shim into a temp dir and diff (must be 0)')
    ap.add_argument('--write', action='store_true', help='with --upstream / --selfcheck: replace the committed fixtures')
    ap.add_argument('--data-root', default=None, help='directory holding the reference data/ tree (real SMPL assets)')
    ap.add_argument('--tolerance', type=float, default=1e-5,
                    help='--upstream: relative deviation of a floating-point array above which the run fails (index tables, uint8 '
                         'crops and the OpenCV checks must be exact)')
    ap.add_argument('--report', default=None, metavar='pin.json',
                    help='write the machine-readable pin report: bound leaves + package versions, per-array max abs / rel '
                         'deviation, OpenCV grey-level deviation, verdict')
    args = ap.parse_args()
    if not (args.upstream or args.selfcheck):
        generate(OUT_DIR)
        return 0
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        new_dir = os.path.join(td, 'golden')
        os.makedirs(new_dir)
        data_root = args.data_root
        if args.upstream and data_root is None:
            from spec_amd.evaluation import write_standin_data_tree
            data_root = os.path.join(td, 'tree')
            write_standin_data_tree(data_root, n_images=1, hmr_seed=SEED_HMR, smpl_seed=SEED_SMPL)
        if data_root:
            os.chdir(data_root)          # pare / smplx resolve 'data/...' against the working directory
        try:
            files = generate(new_dir, upstream=args.upstream)
        finally:
            os.chdir(cwd)
        bound = dict(refshim.BOUND)
        if args.upstream and 'upstream' not in bound.values():
            print('NOTE: none of pare / smplx / loguru is importable here - every leaf is still the shim, so this run is the '
                  'self-check.  Install them (see the module docstring) and re-run to pin the upstream leaves.')
        records = []
        worst, problems = compare(new_dir, OUT_DIR, files, values=args.data_root is None, records=records)
        print(f'worst relative deviation over all arrays: {worst:.3e}   leaves: {bound}')
        cv_dev = None
        if args.upstream:
            cv_dev = check_opencv()
            if cv_dev:
                problems.append(f'OpenCV restatement deviates from the installed binary (max {cv_dev})')
        for p_ in problems:
            print('PROBLEM', p_)
        if args.write:
            for f in files:
                shutil.copy(os.path.join(new_dir, f), os.path.join(OUT_DIR, f))
            print('committed fixtures replaced')
        exact = args.selfcheck or 'upstream' not in bound.values()
        failed = bool(problems or (exact and worst != 0.0) or (not exact and worst > args.tolerance))
        if args.report:
            import json
            rep = {'mode': 'upstream' if args.upstream else 'selfcheck', 'leaves': bound, 'packages': package_versions(),
                   'pinned_upstream': sorted(k for k, v in bound.items() if v == 'upstream'),
                   'tolerance_fp_rel': 0.0 if exact else args.tolerance, 'integer_and_byte_arrays': 'exact',
                   'worst_rel_dev': worst, 'arrays': records,
                   'opencv': {'checked': cv_dev is not None, 'max_grey_level_dev': cv_dev},
                   'data_root': args.data_root or 'stand-in tree (synthetic SMPL model in the official pickle format)',
                   'problems': problems, 'pass': not failed}
            with open(args.report, 'w') as f:
                json.dump(rep, f, indent=1)
            print(f'pin report written to {args.report}: pass = {not failed}')
        if failed:
            return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
