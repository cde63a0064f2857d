#!/usr/bin/env python
"""Build-owned counterpart of the reference's ``scripts/spec_demo.py`` flow on MI355X:

    frame -> CamCalib (vfov, pitch, roll) -> decode -> (R, K) -> crop detections on device ->
    SPEC (HMR regressor + SMPL + projection) -> result pickles in the reference's formats.

Differences by design: CamCalib runs in-process (no ``os.system`` subprocess, spec/tester.py:86-88),
crops are cut on the device, nothing is rendered.  The person detector (YOLOv3 + tracker,
spec/tester.py:73-84) is out of scope: boxes come from ``--detections`` (joblib dict
``{image name: (n,4) [cx, cy, w, h]}``) or default to one centred square box per frame.

    python scripts/spec_demo.py --image_folder imgs --output_folder out \
        --ckpt data/spec/checkpoints/spec_checkpoint.ckpt --camcalib_ckpt data/camcalib/checkpoints/camcalib_sa_biased_l2.ckpt
    python scripts/spec_demo.py --synthetic 4 --output_folder /tmp/out      # random frames + random weights
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def camcalib_input(frame_u8, min_size=600, device='cuda'):
    """ImageFolder transform of camcalib/pano_dataset.py:156-162 (Resize(min side 600) of the PIL image,
    ToTensor, ImageNet Normalize) on the device: the uint8 frame goes to HBM once and
    ``specmi_resize_normalize`` reproduces Pillow's resample bit for bit."""
    from spec_amd.preprocess import camcalib_transform
    return camcalib_transform(torch.from_numpy(np.ascontiguousarray(frame_u8)).to(device), min_size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--image_folder', type=str, default=None)
    ap.add_argument('--output_folder', type=str, default='logs/demo_results')
    ap.add_argument('--ckpt', type=str, default=None, help='SPEC Lightning checkpoint')
    ap.add_argument('--camcalib_ckpt', type=str, default=None)
    ap.add_argument('--detections', type=str, default=None)
    ap.add_argument('--synthetic', type=int, default=0, help='run on N random frames with random weights')
    ap.add_argument('--no_save', action='store_true')
    args = ap.parse_args()

    from spec_amd import assets, synth, io_formats, cam_utils
    from spec_amd.checkpoint import load_pretrained_model, read_checkpoint
    from spec_amd.modules import HMR, CameraRegressorNetwork
    from spec_amd.preprocess import crop_detections
    torch.set_grad_enabled(False)
    dev = torch.device('cuda')
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    if args.synthetic:
        assets.use_synthetic_assets(1003)
        cc = CameraRegressorNetwork(); cc.load_state_dict({k: t(v) for k, v in synth.camcalib_state(1001).items()})
        hm = HMR(use_cam=True, use_cam_feats=True); hm.load_state_dict({k: t(v) for k, v in synth.hmr_state(1002, True).items()}, strict=False)
        rng = np.random.default_rng(0)
        frames = {f'synthetic_{i:03d}.jpg': (rng.random((480, 640, 3)) * 255).astype(np.uint8) for i in range(args.synthetic)}
    else:
        from PIL import Image
        assets.load_assets()
        cc = CameraRegressorNetwork(backbone='resnet50', num_fc_layers=1, num_fc_channels=1024)
        load_pretrained_model(cc, read_checkpoint(args.camcalib_ckpt)['state_dict'], remove_lightning=True, strict=True)
        hm = HMR(backbone='resnet50', img_res=224, pretrained=None, use_cam_feats=True, use_cam=True)
        load_pretrained_model(hm, read_checkpoint(args.ckpt)['state_dict'], overwrite_shape_mismatch=True, remove_lightning=True)
        names = sorted(f for f in os.listdir(args.image_folder) if f.lower().endswith(('.png', '.jpg', '.jpeg')))
        frames = {n: np.asarray(Image.open(os.path.join(args.image_folder, n)).convert('RGB')) for n in names}
    cc.to(dev).eval().commit(dev, freeze=True)
    hm.to(dev).eval().commit(dev, freeze=True)
    dets_all = None
    if args.detections:
        import joblib
        dets_all = joblib.load(args.detections)

    t0, nimg, nper = time.time(), 0, 0
    for name, frame in frames.items():
        H, W = frame.shape[:2]
        # ---- CamCalib on the whole frame (scripts/camcalib_demo.py:95-140)
        logits = cc(camcalib_input(frame).to(dev))
        dets = dets_all[name] if dets_all is not None else np.array([[W / 2, H / 2, 0.8 * min(H, W), 0.8 * min(H, W)]], np.float32)
        n = len(dets)
        if n < 1:
            continue
        img_h = torch.full((1,), float(H), device=dev); img_w = torch.full((1,), float(W), device=dev)
        cam = cam_utils.decode_camera(logits[0], logits[1], logits[2], img_h=img_h, img_w=img_w)
        # ---- crops on the device (spec/tester.py:116-128) and SPEC forward (:143-151)
        crops = crop_detections(t(frame).to(dev), t(np.asarray(dets, np.float32)).to(dev), scale=1.0, crop_size=224)
        out = hm(crops['inp_images'], cam_rotmat=cam['cam_rotmat'].repeat(n, 1, 1),
                 cam_intrinsics=cam['cam_intrinsics'].repeat(n, 1, 1), bbox_scale=crops['bbox_scale'],
                 bbox_center=crops['bbox_center'], img_w=img_w.repeat(n), img_h=img_h.repeat(n))
        if not args.no_save:
            io_formats.write_camcalib_result(args.output_folder, name, cam['vfov'][0], cam['pitch'][0], cam['roll'][0], H)
            io_formats.write_spec_result(args.output_folder, name, out)
        nimg += 1; nper += n
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'SPEC FPS: {nimg / dt:.2f} frames/s, {nper / dt:.2f} persons/s over {nimg} frames (results in {args.output_folder})')


if __name__ == '__main__':
    main()
