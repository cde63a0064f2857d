#!/usr/bin/env python
"""Throughput of the non-headline model variants of the path, one JSON line each:
HMR on the HRNet-W32 / W48 trunks (spec/models/hmr.py:44-51) and CamCalib on ResNet-34 (camcalib/config.py:81).
Not the BASELINE.json metric (bench.py measures that); numbers quoted in DESIGN.md section 7."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--only', default='', help='comma-separated backbones (default: all)')
    ap.add_argument('--labels', type=int, default=0, help='also print the N most expensive (kernel, layer group) rows')
    args = ap.parse_args()
    from spec_amd import assets, synth
    from spec_amd.cam_utils import cam_params_from_angles
    from spec_amd.modules import HMR, CameraRegressorNetwork
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    assets.use_synthetic_assets(1003)
    B = args.batch
    x = t(synth.images(3, 16)).to(dev).repeat(B // 16 + 1, 1, 1, 1)[:B].contiguous()
    sc, ce, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(3, B, 224., 224., jitter=False)]
    R, K = cam_params_from_angles(np.full(B, 0.1, np.float32), np.full(B, -0.05, np.float32), np.full(B, 300., np.float32), iw, ih)
    for backbone in ('hrnet_w32-conv', 'hrnet_w32-interp', 'hrnet_w48-conv', 'resnet50'):
        if args.only and backbone not in args.only.split(','):
            continue
        hm = HMR(backbone=backbone, use_cam=True, use_cam_feats=True)
        hm.load_state_dict({k: t(v) for k, v in synth.hmr_state(1002, True, backbone=backbone).items()}, strict=False)
        hm.to(dev).eval().commit(dev, freeze=True)
        eng = hm.engine(dev)
        ms = timed(lambda: hm(x, R, K, sc, ce, iw, ih), args.steps)
        eng.profile(True)
        hm(x, R, K, sc, ce, iw, ih)
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
        by = {}
        for e in prof:
            by[e['kernel']] = by.get(e['kernel'], 0.0) + e['ms']
        top = sorted(by.items(), key=lambda kv: -kv[1])[:5]
        print(json.dumps({'variant': f'HMR({backbone}) forward', 'batch': B, 'ms_per_step': round(ms, 3),
                          'images_per_s': round(B * 1e3 / ms, 1), 'launches': len(prof) and sum(e['launches'] for e in prof),
                          'top_kernels_ms': {k: round(v, 3) for k, v in top}}), flush=True)
        if args.labels:
            import re
            grp = {}
            for e in prof:     # group layers that differ only in their indices
                key = (e['kernel'], re.sub(r'\d+', '#', e['label']))
                g = grp.setdefault(key, [0.0, 0, 0.0])
                g[0] += e['ms']; g[1] += e['launches']; g[2] += e['flops']
            for (kern, lab), (ms_, n, fl) in sorted(grp.items(), key=lambda kv: -kv[1][0])[:args.labels]:
                print(f'    {ms_:8.3f} ms x{n:<3d} {fl / max(ms_, 1e-9) / 1e9:7.1f} TF/s  {kern:<40s} {lab}', file=sys.stderr)
        del hm, eng
        torch.cuda.empty_cache()
    if args.only:
        return
    cc = CameraRegressorNetwork(backbone='resnet34')
    cc.load_state_dict({k: t(v) for k, v in synth.camcalib_state(1001, backbone='resnet34').items()})
    cc.to(dev).eval().commit(dev, freeze=True)
    ms = timed(lambda: cc(x), args.steps)
    print(json.dumps({'variant': 'CameraRegressorNetwork(resnet34) forward', 'batch': B, 'ms_per_step': round(ms, 3),
                      'images_per_s': round(B * 1e3 / ms, 1)}), flush=True)


if __name__ == '__main__':
    main()
