#!/usr/bin/env python
"""Build-owned counterpart of the reference's evaluation flow (``scripts/spec_eval.py`` ->
``SPECTrainer.validation_step`` spec/trainer.py:230-364 -> ``compute_error``
spec/utils/compute_error.py:89-223) on MI355X: batches of 64 (``DATASET.BATCH_SIZE``,
spec/config.py:85) go through the hot path with the *precomputed* CamCalib predictions of the
dataset (``pred_cam_rotmat`` / ``pred_cam_int``, spec/dataset/cam_dataset.py:617-653), the
metrics are computed on the device and the ``evaluation_results_<ds>.pkl`` dump is written.

Real data (``--npz``: arrays img (N,3,224,224) fp32 normalised crops, cam_rotmat (N,3,3),
cam_int (N,3,3), scale (N,), center (N,2), orig_shape (N,2)=[h,w], gt_vertices (N,6890,3))
needs the licensed assets + checkpoint; ``--synthetic N`` builds a stand-in dataset whose ground
truth is the model's own prediction plus noise, so the printed numbers exercise the full code path.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--npz', type=str, default=None)
    ap.add_argument('--ckpt', type=str, default=None)
    ap.add_argument('--synthetic', type=int, default=0)
    ap.add_argument('--batch_size', type=int, default=64)
    ap.add_argument('--log_dir', type=str, default='logs/eval')
    ap.add_argument('--dataset_name', type=str, default='spec-syn')
    args = ap.parse_args()

    from spec_amd import assets, synth, metrics, io_formats
    from spec_amd.checkpoint import load_pretrained_model, read_checkpoint
    from spec_amd.modules import HMR
    torch.set_grad_enabled(False)
    dev = torch.device('cuda')
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    if args.synthetic:
        N = args.synthetic
        smpl = assets.use_synthetic_assets(1003)
        hm = HMR(use_cam=True, use_cam_feats=True)
        hm.load_state_dict({k: t(v) for k, v in synth.hmr_state(1002, True).items()}, strict=False)
        scale, center, img_w, img_h = synth.bbox_inputs(5, N, 640., 480.)
        rng = np.random.default_rng(0)
        ang = rng.uniform(-0.4, 0.4, (N, 2)).astype(np.float32)
        from spec_amd.cam_utils import cam_params_from_angles
        R, K = cam_params_from_angles(ang[:, 0], ang[:, 1], rng.uniform(300, 900, N).astype(np.float32), img_w, img_h)
        data = {'img': synth.images(123, N), 'cam_rotmat': R.cpu().numpy(), 'cam_int': K.cpu().numpy(), 'scale': scale,
                'center': center, 'orig_shape': np.stack([img_h, img_w], 1), 'gt_vertices': None}
        J24 = t(smpl['J_regressor']).to(dev)
    else:
        assets.load_assets()
        smpl = assets.smpl_model()
        hm = HMR(backbone='resnet50', img_res=224, pretrained=None, use_cam_feats=True, use_cam=True)
        load_pretrained_model(hm, read_checkpoint(args.ckpt)['state_dict'], overwrite_shape_mismatch=True, remove_lightning=True)
        data = dict(np.load(args.npz))
        J24 = t(smpl['J_regressor']).to(dev)
    hm.to(dev).eval().commit(dev, freeze=True)

    N = data['img'].shape[0]
    dump = io_formats.EvalDump()
    acc = {'w_mpjpe_24': [], 'pa_mpjpe_24': [], 'w_v2v': []}
    for b0 in range(0, N, args.batch_size):
        sl = slice(b0, min(N, b0 + args.batch_size))
        x = t(data['img'][sl]).to(dev)
        shp = t(data['orig_shape'][sl]).float().to(dev)
        # positional call order of spec/trainer.py:139
        pred = hm(x, t(data['cam_rotmat'][sl]).to(dev), t(data['cam_int'][sl]).to(dev), t(data['scale'][sl]).to(dev),
                  t(data['center'][sl]).to(dev), shp[:, 1].contiguous(), shp[:, 0].contiguous())
        dump.add(pred)
        if data['gt_vertices'] is None:          # synthetic ground truth: prediction + 1 cm noise + a global offset
            g = torch.Generator(device=dev).manual_seed(b0)
            gt = pred['smpl_vertices'] + 0.01 * torch.randn(pred['smpl_vertices'].shape, device=dev, generator=g) + 0.05
        else:
            gt = t(data['gt_vertices'][sl]).to(dev)
        mp, pa, v2v = metrics.eval_single(pred['smpl_vertices'], gt, J24, joint_sel=range(24))
        acc['w_mpjpe_24'].append(mp); acc['pa_mpjpe_24'].append(pa); acc['w_v2v'].append(v2v)
    path = dump.write(args.log_dir, args.dataset_name)
    res = {k: float(torch.cat(v).mean()) for k, v in acc.items()}
    print(f'***** RESULTS ON {args.dataset_name.upper()} ({N} samples) *****')
    print(f"W-MPJPE-24: {res['w_mpjpe_24']:.3f}\nPA-MPJPE-24: {res['pa_mpjpe_24']:.3f}\nW-V2V: {res['w_v2v']:.3f}")
    print('dump:', path)


if __name__ == '__main__':
    main()
