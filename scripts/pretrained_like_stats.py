#!/usr/bin/env python
"""Activation statistics of the 'pretrained_like' synthetic trunks under the CPU oracle (test infrastructure; CPU only).

For every BatchNorm of the trunk: r = std of the pre-BN activations over (batch, pixels) / sqrt(running_var), channel by
channel (a released checkpoint has r ~ 1 because training made it so), the second moment of what each convolution reads, and
the range of the layer-4 map.  The ``_PL_M_*`` constants of ``spec_amd/synth.py`` are rounded from this script's output.

    python scripts/pretrained_like_stats.py [--seed 2101] [--batch 2] [--double]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from spec_amd import synth  # noqa: E402
from oracle.models import load_numpy_state  # noqa: E402
from oracle.resnet import ResNet50Trunk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seed', type=int, default=2101)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--double', action='store_true')
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    net = load_numpy_state(ResNet50Trunk().eval(), synth.resnet50_state(args.seed, stats='pretrained_like'))
    x = torch.from_numpy(synth.images(args.seed + 1, args.batch, saturate=True))
    if args.double:
        net, x = net.double(), x.double()
    rows = []

    def hook(name):
        def fn(mod, inp, out):
            pre = inp[0]
            s = pre.transpose(0, 1).reshape(pre.shape[1], -1)
            r = (s.std(dim=1) / mod.running_var.sqrt()).numpy()
            live = r > 0
            rows.append((name, float(np.median(r[live])), float(r[live].min()), float(r.max()), int((~live).sum()),
                         float(out.abs().max()), float((out.clamp(min=0) ** 2).mean())))
        return fn

    def conv_hook(name):
        def fn(mod, inp, out):
            print(f'{name:28s} reads second moment {float((inp[0] ** 2).mean()):8.3f}   max |x| {float(inp[0].abs().max()):9.2f}')
        return fn

    for n, m in net.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.register_forward_hook(hook(n))
        if isinstance(m, torch.nn.Conv2d):
            m.register_forward_hook(conv_hook(n))
    y = net(x)
    print(f'\n{"bn":28s} {"median r":>9s} {"min r":>9s} {"max r":>9s} {"dead":>5s} {"max |post-BN|":>13s} {"E relu^2":>9s}')
    for row in rows:
        print(f'{row[0]:28s} {row[1]:9.3f} {row[2]:9.3f} {row[3]:9.3f} {row[4]:5d} {row[5]:13.2f} {row[6]:9.3f}')
    yc = y.transpose(0, 1).reshape(y.shape[1], -1)
    cm = yc.abs().amax(dim=1)
    print(f'\nlayer-4 map: max {float(y.max()):.2f}, mean {float(y.mean()):.3f}, zero fraction {float((y == 0).double().mean()):.3f}; '
          f'per-channel max-norm: min {float(cm.min()):.3e} median {float(cm.median()):.3f} max {float(cm.max()):.2f}; '
          f'all-zero channels {int((cm == 0).sum())}')


if __name__ == '__main__':
    main()
