#!/usr/bin/env python
"""bench.py - images/sec of the fused CamCalib + SPEC + SMPL forward at batch 256 per GPU
(BASELINE.json config 3; config 4 = the same per-GPU work on N GPUs + one RCCL all-gather).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the whole hot path over one synthetic batch that is already resident in
HBM: CamCalib trunk + 3 FC heads -> soft-argmax decode -> (R, K) -> SPEC trunk -> 3-iteration
regressor -> SMPL LBS (6890 vertices) -> 49 joints -> perspective projection (+ for N > 1 the
all-gather of the packed per-image records).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_TBS = 8.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_models(device):
    from spec_amd import synth, assets
    from spec_amd.modules import HMR, CameraRegressorNetwork
    t0 = time.time()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    cs, hs = synth.camcalib_state(1001), synth.hmr_state(1002, True)
    assets.use_synthetic_assets(1003)
    cc = CameraRegressorNetwork()
    cc.load_state_dict({k: t(v) for k, v in cs.items()})
    hm = HMR(use_cam=True, use_cam_feats=True)
    hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
    cc.to(device).eval().commit(device, freeze=True)
    hm.to(device).eval().commit(device, freeze=True)
    log(f'[bench] models built + packed in {time.time() - t0:.1f}s')
    return cc, hm, cs, hs


def make_inputs(B, device, seed):
    from spec_amd import constants as C
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.rand(B, 3, 224, 224, device=device, generator=g)
    mean = torch.tensor(C.IMG_NORM_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(C.IMG_NORM_STD, device=device).view(1, 3, 1, 1)
    x = ((x - mean) / std).contiguous()
    scale = torch.full((B,), 224.0 / 200.0, device=device)            # bbox[2]/200, spec/tester.py:127
    center = torch.full((B, 2), 112.0, device=device)
    img_w = torch.full((B,), 224.0, device=device)
    img_h = torch.full((B,), 224.0, device=device)
    return x, scale, center, img_w, img_h


def pmc_traffic():
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs,
    gfx950-corrected by scripts/rocprof_summary.py); None when no summary is committed."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic_latest.json')
    try:
        with open(path) as f:
            return json.load(f).get('traffic_bytes_per_launch')
    except Exception:
        return None


def roofline_from_profile(entries):
    conv = [e for e in entries if e['kernel'].startswith('conv_igemm_f32')]
    ms = sum(e['ms'] for e in conv)
    fl = sum(e['flops'] for e in conv)
    by = sum(e['bytes'] for e in conv)
    n = sum(e['launches'] for e in conv)
    total_ms = sum(e['ms'] for e in entries)
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    # second MFMA kernel family: the fused Winograd F(2x2,3x3) 3x3 convolutions.  Its profiler flops are the
    # ALGORITHMIC (direct-convolution) flops; the matrix cores execute 16/36 of them, which is what the
    # MFMA roofline bounds.
    wino = [e for e in entries if e['kernel'].startswith('conv_wino_f32')]
    wms = sum(e['ms'] for e in wino)
    wfl = sum(e['flops'] for e in wino)
    wn = sum(e['launches'] for e in wino)
    wino_info = None
    if wms > 0:
        walg = wfl / (wms * 1e-3) / 1e12
        wino_info = {
            'kernel': 'conv_wino_f32 (F(2x2,3x3), 3x3 stride-1 layers)', 'launches_per_step': wn,
            'avg_launch_ms': round(wms / max(wn, 1), 4),
            'algorithmic_TFLOPs': round(walg, 2), 'executed_mfma_TFLOPs': round(walg * 16.0 / 36.0, 2),
            'frac_executed_of_peak': round(walg * 16.0 / 36.0 / PEAK_FP32_MFMA_TFLOPS, 4),
            'share_of_step_kernel_time': round(wms / total_ms, 4) if total_ms > 0 else None,
        }
    return {
        'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': pmc_traffic(),
        'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, profiles/pmc_traffic_latest.json)',
        'measured': 'HIP events on the launch stream around every kernel of a second region of the same K steps run '
                    'single-stream / eager right after the timed region (the timed region replays the step as a '
                    '2-stream hipGraph, where per-kernel events would time co-running kernels); agrees with '
                    'profiles/*_rocprof_kernel_stats.txt',
        'algorithmic_bytes_per_launch': round(by / max(n, 1)),
        'kernel': 'conv_igemm_f32 (all tile variants)', 'launches_per_step': n,
        'avg_launch_ms': round(ms / max(n, 1), 4),
        'algorithmic_gflop_per_launch': round(fl / max(n, 1) / 1e9, 3),
        'algorithmic_hbm_GBps': round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0,
        'share_of_step_kernel_time': round(ms / total_ms, 4) if total_ms > 0 else None,
        'second_kernel': wino_info,
    }


def cpu_baseline(cs, hs, budget_s=12.0, batch=16):
    """The CPU oracle (PyTorch-CPU fp32 restatement of the reference forward) on the host cores."""
    from spec_amd import synth
    from oracle import heads
    from oracle.models import CamCalibOracle, HMROracle, load_numpy_state, full_pipeline
    torch.set_grad_enabled(False)
    heads.set_assets(smpl_model=synth.smpl_model(1003))
    occ = load_numpy_state(CamCalibOracle().eval(), cs)
    ohm = load_numpy_state(HMROracle(use_cam=True, use_cam_feats=True).eval(), hs)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    x = t(synth.images(3, batch))
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(3, batch, jitter=False)]
    full_pipeline(occ, ohm, x[:2], sc[:2], ce[:2], iw[:2], ih[:2])     # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        full_pipeline(occ, ohm, x, sc, ce, iw, ih)
        n += batch
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 20 * batch:
            break
    return {'value': round(n / el, 2), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} synthetic 224x224 images in batches of {batch}, full CamCalib+SPEC+SMPL forward, '
                      f'PyTorch-CPU fp32 oracle, {el:.1f}s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-overlap', action='store_true', help='run CamCalib and SPEC back to back on one stream')
    ap.add_argument('--force-variant', type=int, default=0, help='debug: force a conv tile (1:128x128 2:128x64 3:64x64)')
    ap.add_argument('--graph', action='store_true', help='(default) capture the step in a hipGraph and replay it')
    ap.add_argument('--no-graph', action='store_true', help='launch the ~130 kernels of a step eagerly')
    ap.add_argument('--subbatch', type=int, default=-1, help='trunk sub-batch for the early stages (0 = off, -1 = library default)')
    ap.add_argument('--subbatch-layers', type=int, default=-1)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        log('[bench] --gpus > 1 needs a torch.distributed.run launch; running 1 GPU')
    n_gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)

    from spec_amd.pipeline import SpecPipeline, AsyncGather
    torch.set_grad_enabled(False)
    cc, hm, cs, hs = build_models(device)
    pipe = SpecPipeline(cc, hm, overlap=not args.no_overlap)
    seq_pipe = SpecPipeline(cc, hm, overlap=False)    # per-kernel profiling pass runs serially
    for m in (cc, hm):
        if args.subbatch >= 0:
            m._engine.set_option('trunk_subbatch', args.subbatch)
        if args.subbatch_layers >= 0:
            m._engine.set_option('trunk_subbatch_layers', args.subbatch_layers)
    if args.force_variant:
        cc._engine.set_option('force_conv_variant', args.force_variant)
    B = args.batch
    x, scale, center, img_w, img_h = make_inputs(B, device, 20210001 + rank)

    run = pipe
    launch_mode = 'eager launches'
    if not args.no_graph:
        # the step is ~130 dependent launches: replaying them as one hipGraph removes the inter-launch gaps
        # (+1 % at B=256).  Same kernels, same work; falls back to eager launches if capture is unavailable.
        try:
            from spec_amd.pipeline import GraphedPipeline
            run = GraphedPipeline(pipe, x, scale, center, img_w, img_h)
            launch_mode = 'hipGraph replay'
        except Exception as e:
            log('[bench] hipGraph capture failed, launching eagerly:', repr(e))
            run = pipe

    # N > 1: one all-gather of the packed records per step, started asynchronously so that RCCL moves step s over xGMI
    # while the kernels of step s+1 run (at most 2 in flight; everything is drained inside the timed region)
    gather = AsyncGather(depth=2) if world > 1 else None

    def step():
        out = run(x, scale, center, img_w, img_h)
        if gather is not None:
            return gather.submit(out)
        return out

    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = B * n_gpus * args.steps / elapsed

    roof, stages = None, None
    if rank == 0 and not args.no_profile:
        for m in (cc, hm):
            m._engine.profile(True)
        cam_eng = None
        try:
            from spec_amd import cam_utils
            cam_eng = cam_utils._engine(device)
            cam_eng.profile(True)
        except Exception:
            pass
        # second region: the same K steps, one stream, eager launches, HIP events around every kernel
        nprof = max(args.steps, 1)
        seq_pipe(x, scale, center, img_w, img_h)          # (workspace / table warm-up, not recorded)
        torch.cuda.synchronize()
        for e in (cc._engine, hm._engine, cam_eng):
            if e is not None:
                e.profile_read()                          # drop the warm-up records
        t1 = time.perf_counter()
        for _ in range(nprof):
            seq_pipe(x, scale, center, img_w, img_h)
        torch.cuda.synchronize()
        prof_ms_per_step = (time.perf_counter() - t1) / nprof * 1e3
        entries = []
        for tag, e in (('camcalib', cc._engine), ('spec', hm._engine), ('decode', cam_eng)):
            if e is None:
                continue
            for r in e.profile_read():
                r['model'] = tag
                r['ms'] /= nprof; r['flops'] /= nprof; r['bytes'] /= nprof; r['launches'] //= nprof
                entries.append(r)
            e.profile(False)
        roof = roofline_from_profile(entries)
        roof['region'] = {'steps': nprof, 'ms_per_step': round(prof_ms_per_step, 3), 'streams': 1, 'launch': 'eager launches',
                          'images_per_s': round(B * 1e3 / prof_ms_per_step, 1)}
        agg = {}
        for r in entries:
            a = agg.setdefault(r['kernel'], {'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'launches': 0})
            for k in ('ms', 'flops', 'bytes', 'launches'):
                a[k] += r[k]
        stages = {k: {'ms': round(v['ms'], 3), 'launches': v['launches'],
                      'TFLOPs': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 else 0,
                      'GBps': round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else 0}
                  for k, v in agg.items()}
        log('[bench] per-kernel (HIP events, one step):')
        for k, v in sorted(stages.items(), key=lambda kv: -kv[1]['ms']):
            log(f'    {k:34s} {v["ms"]:9.3f} ms  x{v["launches"]:<4d} {v["TFLOPs"]:7.2f} TF/s {v["GBps"]:9.1f} GB/s(alg)')
        outdir = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(outdir):
            with open(os.path.join(outdir, 'bench_profile.json'), 'w') as f:
                json.dump({'entries': entries, 'stages': stages, 'ms_per_step': ms_per_step}, f, indent=1)

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(cs, hs)
        except Exception as e:  # the baseline must never sink the bench line
            log('[bench] cpu baseline failed:', repr(e))

    if rank == 0:
        line = {
            'metric': 'images/sec (CamCalib+SPEC+SMPL fwd) @ bs256',
            'value': round(value, 2), 'unit': 'images/s', 'n_gpus': n_gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'C3: full SPEC forward (CamCalib ResNet-50 + decode + SPEC ResNet-50/HMR '
                                   'regressor + SMPL LBS 6890 verts + projection), random weights, '
                                   'synthetic 224x224 crops resident in HBM',
                       'batch_per_gpu': B, 'global_batch': B * n_gpus,
                       'streams': 1 if args.no_overlap else 2, 'launch': launch_mode,
                       'parallelism': f'images sharded over {n_gpus} GPU(s), 1 all-gather' if n_gpus > 1 else 'single GPU'},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        if stages is not None:
            line['stages_ms'] = {k: v['ms'] for k, v in stages.items()}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()          # rank 0 may still be in its profiling pass
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
