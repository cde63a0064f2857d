#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ -- run ONLY in the build container.

The reference (``/root/reference``) holds no tests or golden vectors, so
the vectors are produced here by importing the reference's OWN in-tree modules

    spec/models/hmr.py, camcalib/model.py, camcalib/cam_utils.py,
    spec/utils/cam_params.py, spec/constants.py

over the name shim in ``oracle/refshim.py`` (the un-vendored ``pare``/``smplx`` leaf modules
are bound to the oracle's restatements) and running them on seeded synthetic tensors
(``spec_amd.synth``).  Only seeds, small inputs and the expected outputs are stored; the
reference's source never enters the repo.  Usage:  python tests/golden/make_fixtures.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from spec_amd import synth  # noqa: E402
from oracle import refshim, heads  # noqa: E402
from oracle.models import load_numpy_state  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)

SEED_CAMCALIB, SEED_HMR, SEED_SMPL, SEED_IMG = 1001, 1002, 1003, 20210001


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    smpl_model = synth.smpl_model(SEED_SMPL)
    heads.set_assets(smpl_model=smpl_model)
    ref = refshim.import_reference()
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

    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith('.npz'))
    print('fixtures written, total bytes', sz)


if __name__ == '__main__':
    main()
