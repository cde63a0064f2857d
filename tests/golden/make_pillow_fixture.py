#!/usr/bin/env python
"""Golden vectors for the CamCalib frame transform (camcalib/pano_dataset.py:156-162).

The reference hands the frame to torchvision ``Resize(600)`` which, for a PIL image, is
``img.resize((ow, oh), Image.BILINEAR)``.  torchvision is not installed in this image; Pillow is, so the
vectors are produced by the real Pillow binary on deterministic synthetic frames (the same generator the
tests use) with the Resize geometry of torchvision restated (shorter side -> size, longer ->
int(size * long / short)).  Output: tests/golden/camcalib_transform.npz (resized uint8 images).

    python tests/golden/make_pillow_fixture.py
"""
import os
import sys

import numpy as np
from PIL import Image
import PIL

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_preprocess import _frame, RESIZE_CASES  # noqa: E402


def main():
    out = {'pillow_version': PIL.__version__}
    for i, (seed, H, W, ms) in enumerate(RESIZE_CASES):
        img = _frame(seed, H, W)
        ow, oh = (ms, int(ms * H / W)) if W <= H else (int(ms * W / H), ms)
        out[f'case{i}'] = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    path = os.path.join(ROOT, 'tests', 'golden', 'camcalib_transform.npz')
    np.savez_compressed(path, **out)
    print('written', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
