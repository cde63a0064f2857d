#!/usr/bin/env python
"""The reference's ``scripts/spec_demo.py`` on MI355X - same command line, same ``SPECTester`` calls, same files:

    python scripts/spec_demo.py --image_folder data/sample_images --output_folder logs/spec/sample_images \\
        [--cfg data/spec/checkpoints/spec_config.yaml] [--ckpt data/spec/checkpoints/spec_checkpoint.ckpt] \\
        --detections boxes.pkl

    frame -> CamCalib (vfov, pitch, roll) -> <out>/camcalib/<name>.pkl -> (R, K) -> crops on the device ->
    SPEC (HMR regressor + SMPL + projection) -> <out>/spec_results/<stem>.pkl

Differences by design (see spec_amd/tester.py): CamCalib runs in-process, crops are cut on the device, the person detector and
the renderer are outside the path - boxes come from ``--detections`` (joblib: list per image or {image name: (n,4) [cx, cy, w,
h]}); without it one centred square box per frame is used.  ``--synthetic N`` runs the same flow on N random frames with random
weights (no licensed assets needed)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = 'data/spec/checkpoints/spec_config.yaml'
CKPT = 'data/spec/checkpoints/spec_checkpoint.ckpt'


def default_boxes(image_folder):
    from PIL import Image
    from spec_amd.tester import list_images
    out = []
    for f in list_images(image_folder):
        with Image.open(f) as im:
            W, H = im.size
        out.append(np.array([[W / 2, H / 2, 0.8 * min(H, W), 0.8 * min(H, W)]], np.float32))
    return out


def main(args):
    from spec_amd.tester import SPECTester
    torch.set_grad_enabled(False)
    if args.mode != 'folder':
        raise NotImplementedError                                    # as in the reference (video / webcam)
    tmp = None
    if args.synthetic:
        from PIL import Image
        from spec_amd import assets, synth
        from spec_amd.modules import CameraRegressorNetwork
        tmp = tempfile.TemporaryDirectory()
        rng = np.random.default_rng(0)
        for i in range(args.synthetic):
            Image.fromarray((rng.random((480, 640, 3)) * 255).astype(np.uint8)).save(os.path.join(tmp.name, f'synthetic_{i:03d}.jpg'))
        args.image_folder = tmp.name
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        assets.use_synthetic_assets(1003)
        cc = CameraRegressorNetwork()
        cc.load_state_dict({k: t(v) for k, v in synth.camcalib_state(1001).items()})
        args.camcalib_model = cc.to('cuda').eval()
        args.ckpt = {'model.' + k: t(v) for k, v in synth.hmr_state(1002, True).items()}
        args.cfg = os.path.join(tmp.name, 'spec_config.yaml')          # what the released spec_config.yaml sets
        with open(args.cfg, 'w') as f:
            f.write('METHOD: hmr_cam\nHMR:\n  BACKBONE: resnet50\n  USE_CAM_FEATS: true\nDATASET:\n  IMG_RES: 224\n')
        output_path = args.output_folder
    else:
        input_image_folder = args.image_folder
        output_path = os.path.join(args.output_folder, input_image_folder.rstrip('/').split('/')[-1] + '_' + args.exp)
    os.makedirs(output_path, exist_ok=True)
    output_img_folder = os.path.join(output_path, 'spec_results')
    os.makedirs(output_img_folder, exist_ok=True)
    num_frames = len(os.listdir(args.image_folder))

    total_time = time.time()
    tester = SPECTester(args)
    print(f'Number of input frames {num_frames}')
    tester.run_camcalib(args.image_folder, output_path)                       # CamCalib
    detections = tester.run_detector(args.image_folder) if args.detections else default_boxes(args.image_folder)
    spec_time = time.time()
    tester.run_on_image_folder(args.image_folder, detections, output_path, output_img_folder)
    torch.cuda.synchronize()
    end = time.time()
    print(f'SPEC FPS: {num_frames / (end - spec_time):.2f}')
    total_time = time.time() - total_time
    print(f'Total time spent: {total_time:.2f} seconds (including model loading time).')
    print(f'Total FPS (including model loading time): {num_frames / total_time:.2f}.')
    print('================= END =================')
    if tmp is not None:
        tmp.cleanup()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--cfg', type=str, help='config file that defines model hyperparams', default=CFG)
    parser.add_argument('--ckpt', type=str, help='checkpoint path', default=CKPT)
    parser.add_argument('--camcalib_ckpt', type=str, default=None, help='CamCalib checkpoint (default: the path in scripts/camcalib_demo.py)')
    parser.add_argument('--exp', type=str, default='', help='short description of the experiment')
    parser.add_argument('--mode', default='folder', choices=['video', 'folder', 'webcam'], help='Demo type')
    parser.add_argument('--vid_file', type=str, help='input video path or youtube link')
    parser.add_argument('--image_folder', type=str, help='input image folder')
    parser.add_argument('--output_folder', type=str, default='logs/demo/demo_results', help='output folder to write results')
    parser.add_argument('--detections', type=str, default=None, help='joblib file with the person boxes (the detector is outside the path)')
    parser.add_argument('--tracking_method', type=str, default='bbox', choices=['bbox', 'pose'])
    parser.add_argument('--detector', type=str, default='yolo', choices=['yolo', 'maskrcnn'])
    parser.add_argument('--yolo_img_size', type=int, default=416)
    parser.add_argument('--tracker_batch_size', type=int, default=12)
    parser.add_argument('--batch_size', type=int, default=16, help='batch size of SPEC')
    parser.add_argument('--frame_batch', type=int, default=256, help='crops per SPEC forward, collected across frames (1 = one forward per frame, the reference structure; results are bit-identical)')
    parser.add_argument('--decode_threads', type=int, default=4, help='host threads decoding frames ahead of the GPU')
    parser.add_argument('--plan', type=str, default=None, choices=['throughput', 'latency', 'single', 'auto'],
                        help="execution plan of the trunk; default: 'throughput' for the whole run whatever --frame_batch (results are then "
                             "bit-identical for any batching); 'auto' = lowest latency per forward, last bits depend on the batch")
    parser.add_argument('--display', action='store_true')
    parser.add_argument('--smooth', action='store_true')
    parser.add_argument('--no_render', action='store_true', help='(rendering is never done by this build)')
    parser.add_argument('--no_save', action='store_true', help='disable final save of output results.')
    parser.add_argument('--save_obj', action='store_true')
    parser.add_argument('--sideview', action='store_true')
    parser.add_argument('--synthetic', type=int, default=0, help='run on N random frames with random weights')
    main(parser.parse_args())
