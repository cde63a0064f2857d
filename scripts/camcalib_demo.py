#!/usr/bin/env python
"""Drop-in for the reference's ``scripts/camcalib_demo.py`` (same flags: --img_folder --out_folder --loss --ckpt --show
--no_save) on MI355X: CamCalib on every image of a folder, one ``<image name>.pkl`` with ``{'vfov','f_pix','pitch','roll'}``
per image in ``--out_folder`` (what ``spec/utils/cam_params.py:28-35`` reads back).  The horizon-line visualisations
(--show / saved images, matplotlib + skimage in the reference) are outside the path: ``--no_save`` is implied."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CKPT = 'data/camcalib/checkpoints/camcalib_sa_biased_l2.ckpt'


def main(args):
    import torch
    from spec_amd.tester import run_camcalib_folder
    torch.set_grad_enabled(False)
    if args.img_folder in (None, '-'):
        sys.exit('only --img_folder input is built (the dataset modes need the Pano360 / SPEC datasets)')
    if args.show or not args.no_save:
        print('[camcalib_demo] visualisation output is not produced by this build (pickles only)', file=sys.stderr)
    res = run_camcalib_folder(args.img_folder, args.out_folder, ckpt=args.ckpt or CKPT, loss_type=args.loss)
    print(f'CamCalib: {len(res)} images -> {args.out_folder}')


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--img_folder', help='input image folder', type=str)
    parser.add_argument('--out_folder', help='output folder', type=str)
    parser.add_argument('--dataset', type=str, default=None)
    parser.add_argument('--loss', default='softargmax_l2')
    parser.add_argument('--ckpt', default=CKPT)
    parser.add_argument('--show', help='visualize raw network predictions', action='store_true')
    parser.add_argument('--no_save', help='do not save output images', action='store_true')
    main(parser.parse_args())
