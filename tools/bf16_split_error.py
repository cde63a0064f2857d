#!/usr/bin/env python
"""Numerical side of the split-precision experiment the round-1 verdict listed (item 9): what would running the 1x1
convolutions on the bf16 matrix cores cost in accuracy?  CPU emulation (no kernel): every 1x1 convolution of the oracle's
ResNet-50 trunk is computed as a sum of bf16 x bf16 products accumulated in fp32, with the operands split as
x = x_hi + x_lo (+ x_lo2), w = w_hi + w_lo (+ w_lo2):

    1 term  : x_hi w_hi                                              (plain bf16)
    3 terms : x_hi w_hi + x_hi w_lo + x_lo w_hi                       (~16 mantissa bits per product)
    6 terms : + x_lo w_lo + x_hi w_lo2 + x_lo2 w_hi                   (~24 bits: fp32-class)

and the trunk output (and the per-layer error of one mid layer) is compared with the fp32 path and an fp64 reference.
The build's headline path stays exact fp32 MFMA; this script only quantifies the alternative.  Usage: python tools/bf16_split_error.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spec_amd import synth  # noqa: E402
from oracle.models import load_numpy_state  # noqa: E402  (test infrastructure; this tool is not part of the product path)
from oracle.resnet import ResNet50Trunk  # noqa: E402

torch.set_grad_enabled(False)
_conv2d = F.conv2d


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = bf(r)
        parts.append(p)
        r = r - p
    return parts


def make_conv(terms):
    def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if terms == 0 or w.shape[2:] != (1, 1) or x.dtype != torch.float32:
            return _conv2d(x, w, b, stride, padding, dilation, groups)
        n = 1 if terms == 1 else (2 if terms == 3 else 3)
        xs, ws = split(x, n), split(w, n)
        pairs = {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]}[terms]
        y = None
        for i, j in reversed(pairs):                      # small terms first
            t = _conv2d(xs[i], ws[j], None, stride, padding, dilation, groups)
            y = t if y is None else y + t
        return y if b is None else y + b.view(1, -1, 1, 1)
    return conv


def main():
    sd = synth.resnet50_state(1002)
    trunk = load_numpy_state(ResNet50Trunk().eval(), sd)
    x = torch.from_numpy(synth.images(5, 2))
    ref64 = load_numpy_state(ResNet50Trunk().eval().double(), {k: v.astype(np.float64) if v.dtype == np.float32 else v for k, v in sd.items()})(x.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    print(f'{"1x1 convs computed as":34s} {"trunk features vs fp64":>24s}')
    for terms, name in ((0, 'fp32 (the build)'), (6, '6 x bf16 products'), (3, '3 x bf16 products'), (1, '1 x bf16 product')):
        F.conv2d = make_conv(terms)          # nn.Conv2d.forward resolves F.conv2d at call time
        try:
            y = trunk(x)
        finally:
            F.conv2d = _conv2d
        print(f'{name:34s} {rel(y, ref64):24.2e}')
    # one layer in isolation: layer3.*.conv1 shape (K = 1024 -> 256) on post-ReLU activations
    g = torch.Generator().manual_seed(0)
    a = torch.randn(2, 1024, 14, 14, generator=g).relu()
    w = torch.randn(256, 1024, 1, 1, generator=g) * (2.0 / 1024) ** 0.5
    r64 = _conv2d(a.double(), w.double())
    print('single 1x1 layer (K = 1024):', {name: f'{rel(make_conv(t)(a, w), r64):.1e}' for t, name in ((0, 'fp32'), (6, '6xbf16'), (3, '3xbf16'), (1, 'bf16'))})


if __name__ == '__main__':
    main()
