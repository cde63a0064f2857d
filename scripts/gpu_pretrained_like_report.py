#!/usr/bin/env python
"""Parity report on released-checkpoint-like statistics (GPU box; test infrastructure - uses the CPU oracle as the checker).

For every execution plan x trunk structure: the layer-4 map of the GPU trunk against a FLOAT64 oracle, channel by channel, next
to the CPU fp32 oracle's own error on the same channel (tests/test_gpu_pretrained_like.py asserts what this prints), then the
whole path against the CPU oracle at B = 1 and 8.

    python scripts/gpu_pretrained_like_report.py > gpurun_out/r06_parity_report.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from spec_amd import synth  # noqa: E402
from tests import test_gpu_pretrained_like as T  # noqa: E402
from tests.util import PL_SEED_IMG, pinned_plan, rel_err, t  # noqa: E402

torch.set_grad_enabled(False)
DEV = 'cuda:0'


def main():
    pl = T.build_pl()
    f64, f32 = pl['f64'], pl['f32']
    e_cpu = T.per_channel_errors(f32, f64)
    cmax = np.abs(f64).reshape(-1, f64.shape[-1]).max(axis=0)
    live = cmax > 0
    print('stand-in checkpoint: spec_amd.synth stats=pretrained_like (BN var over six decades, zero / negative / loud gammas, dead filters,')
    print('Student-t(3) filters, calibrated running statistics), two saturated 224 x 224 crops, layer-4 map (2, 7, 7, 2048)')
    print(f'layer-4 channel maxima: min non-zero {cmax[live].min():.3e}  median {np.median(cmax):.3f}  max {cmax.max():.2f}  all-zero {int((~live).sum())}')
    print(f'CPU fp32 oracle vs float64: tensor max-norm rel {rel_err(f32, f64):.3e}; per-channel rel (err / channel max): '
          f'median {np.median(e_cpu[live] / cmax[live]):.3e}  max {np.max(e_cpu[live] / cmax[live]):.3e}')
    e_b = T.per_channel_errors(pl['f32b'], f64)
    ok = e_cpu > 0
    q = e_b[ok] / e_cpu[ok]
    print(f'a SECOND run of the CPU fp32 oracle (channels_last, 1 thread: another summation order) against the first, e_2 / e_1 per channel: '
          f'median {np.median(q):.3f}  p99 {np.quantile(q, 0.99):.3f}  max {q.max():.3f}  channels over 2x: {int((q > 2).sum())}')
    print(f'=> per channel, e_cpu below is the larger of the two CPU runs; asserted: median <= {T.CHANNEL_MEDIAN}, p99 <= {T.CHANNEL_P99}, every channel <= {T.CHANNEL_HARD:.0f} x max(e_cpu, the CPU median relative error x the channel maximum) or {T.CHANNEL_FLOOR:.0f} ulp of the channel maximum')
    print()
    print(f'{"plan":11s} {"wino":>4s} {"fuse":>4s} {"tensor rel vs f64":>18s} {"vs CPU fp32":>12s} {"median e_gpu/e_cpu":>19s} {"p99":>7s} {"max":>7s} '
          f'{"over 2x":>7s} {"worst / bound":>13s} {"over":>5s}  worst channel (e_gpu, e_cpu, max |ref|)')
    for plan in T.PLANS:
        for wino, fuse in T.STRUCTURES:
            feat = T.gpu_trunk(pl['hm'], pl['x'], plan, wino, fuse)
            rep = T.channel_report(feat, (f32, pl['f32b']), f64)
            c = rep['argworst']
            print(f'{plan:11s} {wino:4d} {fuse:4d} {rel_err(feat, f64):18.3e} {rel_err(feat, f32):12.3e} {rep["median"]:19.3f} '
                  f'{rep["p99"]:7.3f} {rep["max"]:7.3f} {rep["n_over2"]:7d} {rep["worst"]:13.3f} {rep["n_over"]:5d}  '
                  f'c={c} ({rep["e_gpu"][c]:.3e}, {rep["e_cpu"][c]:.3e}, {rep["cmax"][c]:.3e})')
    print()
    from oracle.models import full_pipeline
    from spec_amd.pipeline import SpecPipeline
    cc, hm = pl['cc'], pl['hm']
    print(f'{"B":>3s} {"plan":11s} ' + ' '.join(f'{k:>14s}' for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'cam angles')) + '  dW-MPJPE mm')
    for B in (1, 8):
        x = t(synth.images(PL_SEED_IMG + B, B, saturate=True))
        sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(PL_SEED_IMG + B, B, 640., 480.)]
        ref = full_pipeline(pl['occ'], pl['ohm'], x, sc, ce, iw, ih)
        for plan in ('auto', 'latency', 'throughput'):
            with pinned_plan(plan, cc, hm):
                out = SpecPipeline(cc, hm)(x.to(DEV), sc.to(DEV), ce.to(DEV), iw.to(DEV), ih.to(DEV))
            errs = [rel_err(out[k].cpu().numpy(), ref[k].numpy()) for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose')]
            ang = max(float(np.abs(out[k].cpu().numpy() - ref[k].numpy()).max()) for k in ('cam_vfov', 'cam_pitch', 'cam_roll'))
            d = T._wmpjpe_mm(out['smpl_vertices'].cpu().numpy().astype(np.float64), ref['smpl_vertices'].numpy().astype(np.float64))
            print(f'{B:3d} {plan:11s} ' + ' '.join(f'{e:14.3e}' for e in errs) + f' {ang:14.3e}  {d:.5f}')


if __name__ == '__main__':
    main()
