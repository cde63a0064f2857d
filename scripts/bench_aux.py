#!/usr/bin/env python
"""Roofline of the kernels either side of the hot path (SURVEY.md 8f rows 1, 2, 4 and the C5 helpers): device crop /
resize pre-processing, evaluation metrics, the SMPL-only body model, CamCalib bin reductions.

Every kernel is timed by the library's own per-launch HIP events (``specmi_profile_enable`` - events recorded on the
launch stream around each kernel), averaged over ``--iters`` launches after a warm-up, and priced against the HBM roof
(8 TB/s) with the ALGORITHMIC bytes the launch reports (inputs read once + outputs written once, the figures DESIGN.md
section 3 states).  All of them are byte / gather work: none is reshaped into a GEMM.

    python scripts/bench_aux.py [--iters 20] [--out gpurun_out/aux_bench.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK = 8.0e12


def timed(eng, fn, iters):
    """-> {kernel: (ms per launch, algorithmic bytes per launch, flops per launch)} for the launches ``fn`` makes."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    rows = {}
    for e in eng.profile_read(4096):
        r = rows.setdefault(e['kernel'], [0.0, 0.0, 0.0, 0])
        r[0] += e['ms']; r[1] += e['bytes']; r[2] += e['flops']; r[3] += e['launches']
    eng.profile(False)
    return {k: (v[0] / v[3], v[1] / v[3], v[2] / v[3], v[3] // iters) for k, v in rows.items() if v[3]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--out', default='gpurun_out/aux_bench.json')
    a = ap.parse_args()
    from spec_amd import assets, preprocess, metrics
    from spec_amd.cam_utils import _engine
    assets.use_synthetic_assets(1003)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    eng = _engine(dev)
    g = torch.Generator().manual_seed(0)
    B = a.batch
    table = []

    def add(name, workload, res, per_unit=None):
        for k, (ms, by, fl, n) in res.items():
            row = {'call': name, 'kernel': k, 'workload': workload, 'launches_per_call': n, 'ms_per_launch': round(ms, 5),
                   'algorithmic_MB': round(by / 1e6, 3), 'achieved_GBps': round(by / (ms * 1e-3) / 1e9, 1),
                   'frac_of_hbm_peak': round(by / (ms * 1e-3) / HBM_PEAK, 4)}
            if fl:
                row['TFLOPs'] = round(fl / (ms * 1e-3) / 1e12, 2)
            if per_unit:
                row[per_unit[0]] = round(per_unit[1] / (ms * 1e-3), 1)
            table.append(row)

    # ---- 8f-1: crops from one 1080p frame ---------------------------------------------------------------------------
    frame = torch.randint(0, 256, (1080, 1920, 3), generator=g, dtype=torch.uint8).to(dev)
    cx = torch.rand(B, generator=g) * 1920; cy = torch.rand(B, generator=g) * 1080
    wh = 150 + torch.rand(B, 2, generator=g) * 500
    dets = torch.stack([cx, cy, wh[:, 0], wh[:, 1]], 1)
    add('preprocess.crop_detections', f'{B} person boxes of one 1920x1080 uint8 frame -> ({B},3,224,224) fp32',
        timed(eng, lambda: preprocess.crop_detections(frame, dets), a.iters), ('crops_per_s', B))
    centers = torch.stack([cx, cy], 1).numpy(); scales = (wh.max(1).values / 200).numpy()
    boxes = torch.from_numpy(preprocess.pare_crop_boxes(centers, scales, 224)).to(dev)
    out = torch.empty(B, 3, 224, 224, device=dev)
    from spec_amd import _lib
    from spec_amd.engine import _ptr

    def ds():
        _lib.check(eng.h, eng.lib.specmi_crop_resize_normalize(eng.h, _ptr(frame), 1080, 1920, _ptr(boxes), B, 224, _ptr(out),
                                                               eng._stream()))
    add('preprocess.dataset_crops (kernel only, boxes precomputed)', f'{B} pare crop boxes of one 1920x1080 frame -> ({B},3,224,224) fp32',
        timed(eng, ds, a.iters), ('crops_per_s', B))
    add('preprocess.camcalib_transform', '1920x1080 uint8 frame -> Resize(600) antialiased -> (1,3,600,1066) fp32',
        timed(eng, lambda: preprocess.camcalib_transform(frame), a.iters), ('frames_per_s', 1))
    big = torch.randint(0, 256, (2160, 3840, 3), generator=g, dtype=torch.uint8).to(dev)
    add('preprocess.camcalib_transform', '3840x2160 uint8 frame -> Resize(600) antialiased -> (1,3,600,1066) fp32',
        timed(eng, lambda: preprocess.camcalib_transform(big), a.iters), ('frames_per_s', 1))

    # ---- 8f-2: evaluation metrics -----------------------------------------------------------------------------------
    V = 6890
    pv = torch.randn(B, V, 3, generator=g).to(dev); gv = (pv.cpu() + 0.01 * torch.randn(B, V, 3, generator=g)).to(dev)
    sm = assets.smpl_model()
    Jh = torch.from_numpy(np.ascontiguousarray(sm['J_regressor_extra'])).float()
    J17 = torch.cat([Jh, Jh[:8]], 0).to(dev)                      # 17 x 6890 stand-in with the H36M regressor's shape
    J24 = torch.from_numpy(np.ascontiguousarray(sm['J_regressor'])).float().to(dev)
    add('metrics.eval_single', f'MPJPE / PA-MPJPE / V2V of {B} meshes (6890 verts, 17x6890 regressor, 14 joints)',
        timed(eng, lambda: metrics.eval_single(pv, gv, J17), a.iters), ('meshes_per_s', B))
    pj = torch.randn(B, 24, 3, generator=g).to(dev); gj = torch.randn(B, 24, 3, generator=g).to(dev)
    add('metrics.eval_j_24', f'MPJPE / PA-MPJPE of {B} x 24 joints', timed(eng, lambda: metrics.eval_j_24(pj, gj), a.iters),
        ('poses_per_s', B))
    add('metrics.regress_joints', f'24x6890 regressor on {B} meshes', timed(eng, lambda: metrics.regress_joints(pv, J24), a.iters),
        ('meshes_per_s', B))
    R = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0].to(dev)
    add('metrics.rotate_points', f'{B} meshes x 6890 points by a per-mesh 3x3', timed(eng, lambda: metrics.rotate_points(R, pv), a.iters),
        ('meshes_per_s', B))

    # ---- C5 helper: the body model alone -----------------------------------------------------------------------------
    bm = metrics.BodyModel(device=dev)
    pose = (0.3 * torch.randn(B, 72, generator=g)).to(dev); betas = torch.randn(B, 10, generator=g).to(dev)
    add('BodyModel.native', f'SMPL (Rodrigues + blend shapes + LBS) for {B} bodies -> vertices + 24 joints',
        timed(bm.engine, lambda: bm.native(pose, betas), a.iters), ('bodies_per_s', B))

    # ---- CamCalib bin reductions ----------------------------------------------------------------------------------
    logits = torch.randn(3 * B, 256, generator=g).to(dev)
    add('cam_utils bins (arg-max + soft-argmax)', f'{3 * B} rows x 256 bins',
        timed(eng, lambda: eng.camcalib_bins(logits, argmax=True, soft=True), a.iters), ('rows_per_s', 3 * B))

    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump({'hbm_peak_TBps': HBM_PEAK / 1e12, 'iters': a.iters, 'batch': B, 'timing': 'per-launch HIP events on the launch stream '
                   '(library profiler)', 'rows': table}, f, indent=1)
    w = max(len(r['call']) for r in table)
    for r in table:
        print(f"{r['call']:<{w}}  {r['kernel']:<22} {r['ms_per_launch'] * 1e3:9.1f} us  {r['algorithmic_MB']:9.2f} MB  "
              f"{r['achieved_GBps']:8.1f} GB/s  frac {r['frac_of_hbm_peak']:.3f}   {r['workload']}")


if __name__ == '__main__':
    main()
