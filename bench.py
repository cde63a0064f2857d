#!/usr/bin/env python
"""bench.py - images/sec of the fused CamCalib + SPEC + SMPL forward at batch 256 per GPU
(BASELINE.json config 3; config 4 = the same per-GPU work on N GPUs + one RCCL all-gather).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 256]

``--gpus N`` with N > 1 launches the N ranks itself (re-exec under ``torch.distributed.run``, one process per
GPU, rendezvous on 127.0.0.1) unless it already runs under a launcher (WORLD_SIZE set); it REFUSES to run
when fewer than N devices are visible - it never falls back to fewer GPUs silently.

A step = one pass of the whole hot path over one synthetic batch that is already resident in
HBM: CamCalib trunk + 3 FC heads -> soft-argmax decode -> (R, K) -> SPEC trunk -> regressor
-> SMPL LBS (6890 vertices) -> 49 joints -> perspective projection (+ for N > 1 the
all-gather of the packed per-image records, which the kernels write in place).  Rank 0 prints ONE JSON line on stdout and,
as the LAST line of stderr, the compact ``[bench] summary {...}`` (headline + small-batch / C2 / demo scalars).

``--backend gloo --fake-forward`` is the hardware-free dry run of the N > 1 control flow (tests/test_bench_dryrun.py): the same
main() - self-launch, process group, 16-image probe incl. its ragged pad path, two alternating record buffers + AsyncGather,
per-rank timing gather, the ``comm`` block, the CPU baseline on rank 0 - with the GPU forward replaced by a per-image stand-in on
CPU tensors.  Its numbers mean nothing; its JSON line carries ``"dry_run": true``.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_TBS = 8.0
TRUNK_GFLOP_PER_IMAGE = 8.174272512   # 4.087136256 GMAC per ResNet-50 trunk at 224x224
STREAM_HBM_TBS = 6.3                  # what a pure read stream reaches on this part (MI355X_MICROARCH.md: ldsdma-fill row, chip 6.4-6.8 TB/s)
NODE_US = 1.45                        # a dependent kernel boundary (same guide, row "boundary")


def resnet50_layers():
    """(name, Cin, Cout, k, stride, output side at 224 x 224) of the 53 convolutions of a ResNet-50 trunk (torchvision order:
    stride on conv2; the downsample branch as its own entry)."""
    out = [('conv1', 3, 64, 7, 2, 112)]
    inpl, side = 64, 56
    for li, (nb, planes) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), start=1):
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 1) else 1
            oside = side // stride
            out.append((f'layer{li}.{bi}.conv1', inpl, planes, 1, 1, side))
            out.append((f'layer{li}.{bi}.conv2', planes, planes, 3, stride, oside))
            out.append((f'layer{li}.{bi}.conv3', planes, planes * 4, 1, 1, oside))
            if bi == 0:
                out.append((f'layer{li}.{bi}.downsample', inpl, planes * 4, 1, stride, oside))
            inpl, side = planes * 4, oside
    return out


def small_batch_floor(batch, nodes):
    """A reachable lower bound for one small-batch step of BOTH trunks, per layer: the later of (a) the matrix-core time of the
    layer's MACs with M and N rounded up to the 32 x 32 MFMA tile - the finest unit any kernel here can compute - at the fp32
    MFMA peak of the whole chip, and (b) the layer's weights read once from HBM at the streaming rate; summed over the 53
    convolutions x 2 networks (layers are sequential: each needs the previous one's complete output), plus the FC / SMPL
    operands read once, plus one kernel boundary per graph node.  'frac' of a measured step = floor / measured."""
    t_mfma = t_w = t_layer = 0.0
    for _, cin, cout, k, _s, oside in resnet50_layers():
        m = batch * oside * oside
        mp, np_ = (m + 31) // 32 * 32, (cout + 31) // 32 * 32
        a = 2.0 * mp * np_ * cin * k * k * 2 / (PEAK_FP32_MFMA_TFLOPS * 1e12)
        w = cin * k * k * cout * 4.0 * 2 / (STREAM_HBM_TBS * 1e12)
        t_mfma += a; t_w += w; t_layer += max(a, w)
    heads_bytes = 4.0 * (3 * 256 * 2048 + 157 * 2240 + 6890 * 3 * 224 + 6890 * 24 + 9 * 6890)   # CamCalib FC, composed IEF map, SMPL dirs / weights / J_extra
    t_tail = heads_bytes / (STREAM_HBM_TBS * 1e12)
    floor = t_layer + t_tail
    return {'floor_ms': round(floor * 1e3, 4), 'floor_with_boundaries_ms': round((floor + nodes * NODE_US * 1e-6) * 1e3, 4),
            'mfma_part_ms': round(t_mfma * 1e3, 4), 'weight_stream_part_ms': round(t_w * 1e3, 4),
            'model': 'sum over the 53 convolutions x 2 trunks of max(MACs with M, N rounded to 32 at 157.3 TF/s, weights once at 6.3 TB/s) + '
                     'FC / SMPL operands once; the second figure adds one 1.45 us kernel boundary per graph node'}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------
# multi-GPU self-launch
# ------------------------------------------------------------------------------------------------
def self_launch(n: int, need_gpus: bool = True) -> int:
    """Re-exec this script as n ranks under torch.distributed.run (one process per GPU, RCCL)."""
    have = torch.cuda.device_count()
    if need_gpus and have < n:
        log(f'[bench] ERROR: --gpus {n} requested but only {have} GPU(s) are visible; refusing to run on fewer '
            f'devices (use --gpus {max(have, 1)})')
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('[bench] launching', n, 'ranks:', ' '.join(cmd))
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------
# models / inputs
# ------------------------------------------------------------------------------------------------
ERR_KEYS = ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'pred_shape', 'pred_cam',
            'cam_vfov', 'cam_pitch', 'cam_roll')


def build_models(device, conv_precision=0):
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
    cc.conv_precision = hm.conv_precision = int(conv_precision)
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


# ------------------------------------------------------------------------------------------------
# roofline bookkeeping
# ------------------------------------------------------------------------------------------------
def pmc_traffic():
    """HBM bytes per conv launch from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950-corrected by
    scripts/rocprof_summary.py; scripts/gpu_prof.sh regenerates the file and stamps it with the hash of the kernel sources
    it measured) -> (bytes per launch | None, stale | None): stale = the stamp differs from the kernels this run loads."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic_latest.json')
    try:
        with open(path) as f:
            d = json.load(f)
        from spec_amd import _lib
        stamp = d.get('source_hash')
        return d.get('traffic_bytes_per_launch'), (stamp is None or stamp != _lib.source_hash()), stamp
    except Exception:
        return None, None, None


def executed_flops(e):
    """FLOPs the matrix cores execute for a profiler entry: the Winograd F(2x2,3x3) kernels run 16/36 of the
    algorithmic (direct-convolution) count their entries carry."""
    return e['flops'] * (16.0 / 36.0 if e['kernel'].startswith('conv_wino') else 1.0)


_LAYER_RE = re.compile(r'backbone\.(layer\d)\.\d+\.(conv\d(?:\+downsample)?|downsample)')


def stage_name(e):
    """Group the ~130 launches of a step into the stages of SURVEY.md App. B (both trunks together)."""
    lab = e['label']
    m = _LAYER_RE.search(lab)
    if m:
        return f'{m.group(1)}.{m.group(2)}'
    if lab.endswith('backbone.conv1'):
        return 'stem.conv7x7+bn+relu'
    if lab.endswith('maxpool'):
        return 'stem.maxpool'
    if lab.startswith('fc_') or lab == 'avgpool':
        return 'camcalib.avgpool+fc'
    if lab.startswith('head.'):
        return 'hmr.regressor'
    if lab == 'smpl':
        return 'smpl.' + e['kernel'].replace('smpl_', '')
    return lab or e['kernel']


def stage_table(entries):
    """per stage, frac = max(bytes_alg / BW_peak, flops / FLOP_peak) / t_measured, naming the
    resource that binds.  flops = executed MFMA FLOPs (Winograd: 16/36 of the direct count)."""
    agg = {}
    for e in entries:
        a = agg.setdefault(stage_name(e), {'ms': 0.0, 'flops': 0.0, 'alg_flops': 0.0, 'bytes': 0.0, 'launches': 0, 'kernels': set()})
        a['ms'] += e['ms']; a['flops'] += executed_flops(e); a['alg_flops'] += e['flops']
        a['bytes'] += e['bytes']; a['launches'] += e['launches']; a['kernels'].add(e['kernel'])
    rows = []
    for name, a in agg.items():
        t = a['ms'] * 1e-3
        if t <= 0:
            continue
        t_hbm = a['bytes'] / (PEAK_HBM_TBS * 1e12)
        t_mfma = a['flops'] / (PEAK_FP32_MFMA_TFLOPS * 1e12)
        # flop-bound stages whose kernels use no matrix instruction (joints, heads' element-wise kernels) are
        # priced against the same 157.3 TF/s - gfx950's vector fp32 peak equals its fp32 MFMA peak - but named 'valu'
        uses_mfma = any(k.startswith(('conv_igemm', 'conv_wino', 'stem_conv', 'smpl_skin')) for k in a['kernels'])
        rows.append({'stage': name, 'kernel': '|'.join(sorted(a['kernels'])), 'launches': a['launches'],
                     'ms': round(a['ms'], 4), 'bound': 'hbm' if t_hbm >= t_mfma else ('mfma' if uses_mfma else 'valu'),
                     'frac': round(max(t_hbm, t_mfma) / t, 4),
                     'TFLOPs': round(a['flops'] / t / 1e12, 2), 'alg_TBps': round(a['bytes'] / t / 1e12, 3)})
    rows.sort(key=lambda r: -r['ms'])
    return rows


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
    # whole step against both roofs (every kernel of the step, executed FLOPs, algorithmic bytes)
    all_fl = sum(executed_flops(e) for e in entries)
    all_by = sum(e['bytes'] for e in entries)
    traffic, traffic_stale, traffic_stamp = pmc_traffic()
    return {
        'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': traffic,
        'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC passes of scripts/gpu_prof.sh, profiles/pmc_traffic_latest.json)',
        'traffic_stale': traffic_stale, 'traffic_source_hash': traffic_stamp,
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
        'whole_step': {'kernel_ms': round(total_ms, 3),
                       'executed_mfma_TFLOPs': round(all_fl / (total_ms * 1e-3) / 1e12, 2) if total_ms > 0 else None,
                       'frac_of_mfma_peak': round(all_fl / (total_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if total_ms > 0 else None,
                       'algorithmic_TBps': round(all_by / (total_ms * 1e-3) / 1e12, 3) if total_ms > 0 else None},
    }


# ------------------------------------------------------------------------------------------------
# CPU baseline (oracle = test infrastructure; imported here and nowhere in the product path)
# ------------------------------------------------------------------------------------------------
def cpu_baseline(cs, hs, budget_s=22.0, batch=16):
    """The CPU oracle (PyTorch-CPU fp32 restatement of the reference forward) on the host cores: thread sweep
    (oversubscribing SMT siblings costs 2x on this path), best setting timed longer, plus batch-1 latency and the
    single-thread rate."""
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
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    t_start = time.perf_counter()

    def rate(nthreads, nimg, reps=1):
        torch.set_num_threads(nthreads)
        full_pipeline(occ, ohm, x[:2], sc[:2], ce[:2], iw[:2], ih[:2])     # warm-up at this thread count
        t0 = time.perf_counter()
        for _ in range(reps):
            full_pipeline(occ, ohm, x[:nimg], sc[:nimg], ce[:nimg], iw[:nimg], ih[:nimg])
        return nimg * reps / (time.perf_counter() - t0)

    sweep = {}
    for nt in sorted({n for n in (8, 16, 32, 64, physical, logical) if 1 <= n <= logical}):
        if time.perf_counter() - t_start > budget_s * 0.5:
            break
        sweep[nt] = round(rate(nt, batch), 2)
        if sweep[nt] < 0.6 * max(sweep.values()):      # past the knee: more threads only oversubscribe (SMT, NUMA)
            break
    best_nt = max(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    n, t0 = 0, time.perf_counter()
    while True:
        full_pipeline(occ, ohm, x, sc, ce, iw, ih)
        n += batch
        el = time.perf_counter() - t0
        if time.perf_counter() - t_start >= budget_s * 0.85 or n >= 16 * batch:
            break
    best = n / el
    b1 = rate(best_nt, 1, reps=3)
    single = rate(1, 1, reps=1)
    torch.set_num_threads(best_nt)
    return {'value': round(best, 2), 'unit': 'images/s', 'cores': best_nt, 'kind': 'port',
            'sample': f'{n} synthetic 224x224 images in batches of {batch}, full CamCalib+SPEC+SMPL forward, '
                      f'PyTorch-CPU fp32 oracle, {el:.1f}s at the best thread count of the sweep',
            'thread_sweep_images_per_s': {str(k): v for k, v in sweep.items()},
            'host': {'logical_cpus': logical, 'physical_cores': physical},
            'batch1_images_per_s': round(b1, 2), 'batch1_threads': best_nt,
            'single_thread_images_per_s': round(single, 3)}


def _cpu_worker(idx, nthreads, t_go, seconds, batch=16):
    """One process of the whole-host CPU baseline: pins itself to its own block of logical CPUs, builds the oracle, waits
    for the common start time and counts the batches it completes inside the window.  Prints one JSON line."""
    try:
        os.sched_setaffinity(0, set(range(idx * nthreads, (idx + 1) * nthreads)))
    except Exception:
        pass
    torch.set_num_threads(nthreads)
    torch.set_grad_enabled(False)
    from spec_amd import synth
    from oracle import heads
    from oracle.models import CamCalibOracle, HMROracle, load_numpy_state, full_pipeline
    heads.set_assets(smpl_model=synth.smpl_model(1003))
    occ = load_numpy_state(CamCalibOracle().eval(), synth.camcalib_state(1001))
    ohm = load_numpy_state(HMROracle(use_cam=True, use_cam_feats=True).eval(), synth.hmr_state(1002, True))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    x = t(synth.images(3 + idx, batch))
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(3, batch, jitter=False)]
    full_pipeline(occ, ohm, x[:2], sc[:2], ce[:2], iw[:2], ih[:2])
    ready = time.time()
    while time.time() < t_go:
        time.sleep(0.01)
    n, t0 = 0.0, time.time()
    last = t0
    while True:
        full_pipeline(occ, ohm, x, sc, ce, iw, ih)
        now = time.time()
        if now - t0 > seconds:
            # the batch that straddles the end of the window counts for the part of it that lies inside
            n += batch * max(0.0, (t0 + seconds - last) / max(now - last, 1e-9))
            break
        n += batch
        last = now
    print(json.dumps({'idx': idx, 'images': n, 'late_s': round(max(0.0, ready - t_go), 2)}), flush=True)


def cgroup_cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except Exception:
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            q = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline_whole_host(physical, threads_per_proc=16, seconds=10.0, startup_s=45.0):
    """P = physical_cores / 16 oracle processes x 16 threads, each pinned to its own cores, all timed over the same
    window: the reference's CPU forward on ALL of the box's host cores (one process cannot use them: the thread sweep
    peaks at 16)."""
    import subprocess
    P = max(1, physical // threads_per_proc)
    t_go = time.time() + startup_s
    env = dict(os.environ, OMP_NUM_THREADS=str(threads_per_proc), MKL_NUM_THREADS=str(threads_per_proc),
               HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(i), str(threads_per_proc),
                               repr(t_go), repr(seconds)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
             for i in range(P)]
    total, late, done = 0.0, 0.0, 0
    for p_ in procs:
        try:
            out, _ = p_.communicate(timeout=startup_s + seconds + 60)
            r = json.loads(out.strip().splitlines()[-1])
            total += r['images']; late = max(late, r['late_s']); done += 1
        except Exception:
            p_.kill()
    if done == 0:
        return None
    return {'value': round(total / seconds, 2), 'unit': 'images/s', 'processes': done, 'threads_per_process': threads_per_proc,
            'cores': done * threads_per_proc, 'window_s': seconds, 'pinned': f'process i -> logical CPUs [{threads_per_proc} i, {threads_per_proc} i + {threads_per_proc})',
            'late_start_s': late}


# ------------------------------------------------------------------------------------------------
# --fake-forward: stand-ins that keep the N > 1 control flow runnable without a GPU (dry run; never measured)
# ------------------------------------------------------------------------------------------------
class FakeModule:
    """set_plan / plan of the drop-in modules, nothing else."""
    plan = 'auto'
    _engine = None

    def set_plan(self, plan):
        self.plan = plan
        return self


class FakePipeline:
    """SpecPipeline's call contract on any device: one (B, 21294)-float packed record per step whose row b depends on image b
    only (so rank-sharded == unsharded, bit for bit, as the real kernels guarantee within a plan)."""
    RECORD = 21294

    def __call__(self, images, bbox_scale, bbox_center, img_w, img_h, record=None):
        B = images.shape[0]
        key = images.reshape(B, -1)[:, :97].double().sum(dim=1).float() + bbox_scale + img_w * 1e-3
        cols = torch.arange(self.RECORD, dtype=torch.float32, device=images.device) * 1e-3
        rec = key[:, None] + cols[None, :]
        if record is not None:
            record.copy_(rec)
            rec = record
        from spec_amd.pipeline import unpack_outputs
        out = unpack_outputs(rec, 6890)          # the per-key views of the packed record, as SpecPipeline(packed=True) returns them
        out['record'] = rec
        return out


class FakeGraphed:
    """GraphedPipeline's contract: static inputs, ``buffers`` alternating static records marked ``specmi_static_buffers``."""

    def __init__(self, pipeline, images, bbox_scale, bbox_center, img_w, img_h, buffers=1):
        self.pipeline = pipeline
        self.static_in = [t.clone() for t in (images, bbox_scale, bbox_center, img_w, img_h)]
        self.records = [torch.empty(images.shape[0], FakePipeline.RECORD, device=images.device) for _ in range(max(1, buffers))]
        for r in self.records:
            r.specmi_static_buffers = max(1, buffers)
        self.turn = 0

    def __call__(self, images, bbox_scale, bbox_center, img_w, img_h):
        for dst, src in zip(self.static_in, (images, bbox_scale, bbox_center, img_w, img_h)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        i = self.turn
        self.turn = (i + 1) % len(self.records)
        return self.pipeline(*self.static_in, record=self.records[i])


# ------------------------------------------------------------------------------------------------
def main():
    if len(sys.argv) >= 6 and sys.argv[1] == '--cpu-worker':
        _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-sustained', action='store_true', help='skip the >= 5 s sustained region')
    ap.add_argument('--sustained-seconds', type=float, default=5.0)
    ap.add_argument('--no-small-batch', action='store_true', help='skip the batch-1 / batch-8 latency lines')
    ap.add_argument('--no-split-bf16', action='store_true', help='skip the labelled secondary line (1x1 convs as bf16 piece products)')
    ap.add_argument('--no-e2e', action='store_true', help='skip the real-input line (1080p frames from pinned host memory)')
    ap.add_argument('--no-c2', action='store_true', help='skip the config-2 line (CamCalib trunk only, batch 64)')
    ap.add_argument('--no-overlap', action='store_true', help='run CamCalib and SPEC back to back on one stream')
    ap.add_argument('--force-variant', type=int, default=0, help='debug: force a conv tile (1:128x128 2:128x64 3:64x64)')
    ap.add_argument('--graph', action='store_true', help='(default) capture the step in a hipGraph and replay it')
    ap.add_argument('--no-graph', action='store_true', help='launch the ~120 kernels of a step eagerly')
    ap.add_argument('--dist', action='store_true', help='run the N > 1 code path (torch.distributed.run launch, RCCL process '
                                                        'group, asynchronous all-gather, two record buffers) even with --gpus 1')
    ap.add_argument('--gather', choices=('full', 'joints'), default='full',
                    help='N > 1 all-gather payload: the full 85,176-byte record per image, or the 2,496 bytes without vertices')
    ap.add_argument('--backend', choices=('nccl', 'gloo'), default='nccl', help="process-group backend; 'gloo' only with --fake-forward")
    ap.add_argument('--fake-forward', action='store_true', help='dry run of the N > 1 control flow on CPU tensors (no GPU, no kernels, '
                                                                 'numbers meaningless): see the module docstring')
    ap.add_argument('--cpu-baseline-seconds', type=float, default=22.0, help='budget of the CPU oracle timing on rank 0')
    ap.add_argument('--subbatch', type=int, default=-1, help='trunk sub-batch for the early stages (0 = off, -1 = library default)')
    ap.add_argument('--subbatch-layers', type=int, default=-1)
    args = ap.parse_args()

    if args.gpus < 1:
        log('[bench] ERROR: --gpus must be >= 1')
        sys.exit(2)
    fake = args.fake_forward
    if (args.backend == 'gloo') != fake:
        log('[bench] ERROR: --backend gloo and --fake-forward go together (the dry run of the N > 1 control flow); the measured path is nccl = RCCL')
        sys.exit(2)
    if fake:     # nothing below the distributed control flow exists without a GPU
        args.no_profile = args.no_c2 = args.no_small_batch = args.no_e2e = args.no_split_bf16 = True
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.dist):
        sys.exit(self_launch(args.gpus, need_gpus=not fake))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        log(f'[bench] ERROR: launched with WORLD_SIZE={world} but --gpus {args.gpus}; they must agree')
        sys.exit(2)
    if not fake and (not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank):
        log(f'[bench] ERROR: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible (no CPU path)')
        sys.exit(2)
    n_gpus = world
    if fake:
        device = torch.device('cpu')
        torch.set_num_threads(2)
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    sync = (lambda: None) if fake else torch.cuda.synchronize

    def timed_ms(fn, n):
        """Average time of n calls of fn in ms: HIP events on the current stream (wall clock in the dry run)."""
        if fake:
            t0_ = time.perf_counter()
            for _ in range(n):
                fn()
            return (time.perf_counter() - t0_) / n * 1e3
        e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0_.record()
        for _ in range(n):
            fn()
        e1_.record()
        torch.cuda.synchronize()
        return e0_.elapsed_time(e1_) / n

    dist = None
    use_dist = world > 1 or args.dist
    if use_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:     # (torch.distributed.run always sets it; a bare single-rank --dist run picks a free one)
            if world > 1:
                log('[bench] ERROR: MASTER_PORT is not set for a multi-rank job (launch through torch.distributed.run or `bench.py --gpus N`)')
                sys.exit(2)
            with socket.socket() as s_:
                s_.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(s_.getsockname()[1])
        # (the other ranks wait at the final barrier while rank 0 runs its single-GPU extras and the CPU baseline: minutes)
        kw = {} if fake else {'device_id': device}
        dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30), **kw)

    from spec_amd.pipeline import SpecPipeline, AsyncGather, gather_outputs
    torch.set_grad_enabled(False)
    if fake:
        cc, hm, cs, hs = FakeModule(), FakeModule(), None, None
        pipe = seq_pipe = FakePipeline()
    else:
        cc, hm, cs, hs = build_models(device)
        pipe = SpecPipeline(cc, hm, overlap=not args.no_overlap)
        seq_pipe = SpecPipeline(cc, hm, overlap=False)    # per-kernel profiling pass runs serially
        if args.subbatch >= 0 or args.subbatch_layers >= 0 or args.force_variant:
            os.environ['SPECMI_EXPERIMENTAL'] = '1'       # debug flags: names of the experimental list (include/specmi.h)
        for m in (cc, hm):
            if args.subbatch >= 0:
                m._engine.set_option('trunk_subbatch', args.subbatch)
            if args.subbatch_layers >= 0:
                m._engine.set_option('trunk_subbatch_layers', args.subbatch_layers)
            if args.force_variant:
                m._engine.set_option('force_conv_variant', args.force_variant)
    B = args.batch
    x, scale, center, img_w, img_h = make_inputs(B, device, 20210001 + rank)
    if use_dist:
        # config 4: every rank runs its shard of the global batch and the gathered records must equal the unsharded forward bit
        # for bit - an image's bits are batch-invariant WITHIN a plan, so the sharded path pins it (auto would pick by shard size)
        for m in (cc, hm):
            m.set_plan('throughput')

    # N > 1: one all-gather of the packed records per step, started asynchronously so that RCCL moves step s over xGMI
    # while the kernels of step s+1 run (at most 2 in flight; everything is drained inside the timed region).  The
    # kernels write the record in place; under graph replay two record buffers alternate so that step s+1 never
    # writes the buffer the collective of step s still reads.
    gather = AsyncGather(depth=2, payload=args.gather) if use_dist else None

    run = pipe
    launch_mode = 'eager launches'
    if not args.no_graph:
        # the step is ~120 dependent launches: replaying them as one hipGraph removes the inter-launch gaps
        # (+1 % at B=256).  Same kernels, same work; falls back to eager launches if capture is unavailable.
        try:
            from spec_amd.pipeline import GraphedPipeline
            run = (FakeGraphed if fake else GraphedPipeline)(pipe, x, scale, center, img_w, img_h, buffers=2 if use_dist else 1)
            launch_mode = 'hipGraph replay'
        except Exception as e:
            log('[bench] hipGraph capture failed, launching eagerly:', repr(e))
            run = pipe

    def step(collect=True):
        if gather is not None and collect:
            gather.reserve()        # the buffer this step writes must not be read by a pending collective
        out = run(x, scale, center, img_w, img_h)
        if gather is not None and collect:
            # zero-copy send of the record the kernels wrote: legal because two graph buffers alternate and reserve() ran
            return gather.submit(out, inplace=True)
        return out

    def timed(nsteps, collect=True):
        sync()
        if use_dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step(collect)
        if gather is not None:
            gather.drain()
        sync()
        if use_dist:
            dist.barrier()
        sync()
        return time.perf_counter() - t0

    # N > 1 start-up probe (outside the timed region): a common 16-image batch, each rank runs its shard_range slice, one
    # all-gather - the gathered records must equal the unsharded forward of all 16 images bit for bit (plan pinned: an
    # image's bits are batch-invariant within a plan)
    probe_ok = None
    if use_dist:
        from spec_amd.pipeline import shard_range
        px, psc, pce, piw, pih = make_inputs(16, device, 424242)          # same seed on every rank
        for m in (cc, hm):
            m.set_plan('throughput')
        whole = pipe(px, psc, pce, piw, pih)['record'].clone()
        lo, hi = shard_range(16, rank, world)
        mine = pipe(px[lo:hi], psc[lo:hi], pce[lo:hi], piw[lo:hi], pih[lo:hi])
        counts = [shard_range(16, r, world)[1] - shard_range(16, r, world)[0] for r in range(world)]
        if len(set(counts)) == 1:
            got = gather_outputs(mine)
        else:                                    # ragged shards (16 % world != 0): pad to the largest, cut after the gather
            pad = torch.zeros(max(counts), mine['record'].shape[1], device=device)
            pad[:hi - lo] = mine['record']
            full = torch.empty(world * max(counts), pad.shape[1], device=device)
            dist.all_gather_into_tensor(full, pad)
            got = torch.cat([full[r * max(counts): r * max(counts) + counts[r]] for r in range(world)], 0)
        flag = torch.tensor([int(torch.equal(got, whole))], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        probe_ok = bool(flag.item())
        for m in (cc, hm):
            m.set_plan('auto')
        if not probe_ok:
            log(f'[bench] WARNING rank {rank}: gathered 16-image probe differs from the unsharded forward')
        del whole, mine, got

    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.drain()
    local_elapsed = timed(args.steps)
    elapsed = local_elapsed
    per_rank = None
    if use_dist:
        tt = torch.tensor([local_elapsed], device=device, dtype=torch.float64)
        allt = torch.empty(world, device=device, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, tt)
        elapsed = float(allt.max().item())
        per_rank = [round(B * args.steps / float(v), 1) for v in allt.tolist()]
    ms_per_step = elapsed / args.steps * 1e3
    value = B * n_gpus * args.steps / elapsed

    # ---- sustained region: the same step for >= 5 s (clock / thermal steady state) -------------------------
    sustained = None
    if not args.no_sustained:
        n_sus = max(args.steps, int(math.ceil(args.sustained_seconds / (ms_per_step * 1e-3))))
        el = timed(n_sus)
        if use_dist:
            tt = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        sustained = {'steps': n_sus, 'seconds': round(el, 3), 'ms_per_step': round(el / n_sus * 1e3, 3),
                     'images_per_s': round(B * n_gpus * n_sus / el, 2)}

    # ---- N > 1: the collective alone ------------------------------------------------------------------------
    comm = None
    if use_dist:
        out = pipe(x, scale, center, img_w, img_h)
        sync()
        dist.barrier()
        full = gather_outputs(out)
        sync()
        ag_ms = timed_ms(lambda: gather_outputs(out), 5)
        # the same K steps with no collective at all: what the asynchronous gather costs on top of the compute
        el_nc = timed(args.steps, collect=False)
        tt = torch.tensor([el_nc], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms_nc = float(tt.item()) / args.steps * 1e3
        comm = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'payload': args.gather,
                'probe_gathered_16_images_equal_unsharded': probe_ok,
                'record_bytes_per_image': int(out['record'].shape[1] * 4),
                'sent_bytes_per_image': 2496 if args.gather == 'joints' else int(out['record'].shape[1] * 4),
                'all_gather_ms_blocking_full_record': round(ag_ms, 3),
                'gathered_MB_per_rank_full_record': round(full.numel() * 4 / 1e6, 1),
                'ms_per_step_without_gather': round(ms_nc, 3), 'ms_per_step_with_async_gather': round(ms_per_step, 3),
                'overlap_efficiency': round(ms_nc / ms_per_step, 4),
                'receive_buffers': 'persistent x2', 'send': 'in place (record written by the kernels)' if args.gather == 'full' else '624-float slice copy',
                'per_rank_images_per_s': per_rank}

    # everything up to here - the timed headline, the sustained region, the collective - ran on the STABLE option surface; the
    # informational extras below flip opt-in paths and the secondary arithmetic mode
    os.environ['SPECMI_EXPERIMENTAL'] = '1'
    roof, stages, c2 = None, None, None
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
        stages = stage_table(entries)
        log('[bench] per-stage (HIP events, one step, both trunks):')
        for r in stages:
            log(f'    {r["stage"]:28s} {r["ms"]:8.3f} ms x{r["launches"]:<3d} {r["bound"]:4s} frac {r["frac"]:.3f} '
                f'{r["TFLOPs"]:7.2f} TF/s {r["alg_TBps"]:6.3f} TB/s(alg)  {r["kernel"]}')
        outdir = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(outdir):
            with open(os.path.join(outdir, 'bench_profile.json'), 'w') as f:
                json.dump({'entries': entries, 'stages': stages, 'ms_per_step': ms_per_step}, f, indent=1)

    # ---- config 2: CamCalib ResNet-50 trunk only, batch 64 -----------------------------------
    if rank == 0 and not args.no_c2:
        x64 = x[:64].contiguous() if B >= 64 else make_inputs(64, device, 20210001)[0]
        eng = cc._engine
        for _ in range(3):
            eng.trunk(x64)
        torch.cuda.synchronize()
        n2 = max(20, args.steps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n2):
            eng.trunk(x64)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / n2
        tf2 = 64 * TRUNK_GFLOP_PER_IMAGE / ms2                     # GFLOP / ms = TFLOP/s (algorithmic, direct-conv count)
        c2 = {'workload': 'C2: CamCalib ResNet-50 trunk only, synthetic 224x224, batch 64, 1 stream, eager launches',
              'batch': 64, 'steps': n2, 'ms_per_step': round(ms2, 3), 'images_per_s': round(64e3 / ms2, 1),
              'algorithmic_TFLOPs': round(tf2, 2), 'frac_of_mfma_peak_algorithmic': round(tf2 / PEAK_FP32_MFMA_TFLOPS, 4)}

    # ---- small batches: latency of one whole step (the reference's own operating point: spec/tester.py:109-151 runs the path at
    # batch = #detections of a frame, scripts/camcalib_demo.py:95-102 at batch 1) -------------------------------------------------
    small = None
    if rank == 0 and not args.no_small_batch and not args.no_graph:
        small = []
        try:
            from spec_amd.pipeline import GraphedPipeline

            def step_ms(pp, b, iters=100):
                g = GraphedPipeline(pp, x[:b].contiguous(), scale[:b].contiguous(), center[:b].contiguous(),
                                    img_w[:b].contiguous(), img_h[:b].contiguous())
                ins = g.static_in
                for _ in range(5):
                    g(*ins)
                torch.cuda.synchronize()
                best = None
                for _ in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        g(*ins)
                    e1.record()
                    torch.cuda.synchronize()
                    ms_ = e0.elapsed_time(e1) / iters
                    best = ms_ if best is None else min(best, ms_)
                nodes = None
                try:    # graph nodes of the captured step (torch >= 2.1 keeps the raw graph for debug dumps only: count launches instead)
                    for m in (cc, hm):
                        m._engine.profile(True)
                    pp(*ins)
                    torch.cuda.synchronize()
                    nodes = sum(r['launches'] for m in (cc, hm) for r in m._engine.profile_read())
                except Exception:
                    pass
                finally:    # the launch profiler must not stay on: it adds events per launch and disables the opt-in fused paths
                    for m in (cc, hm):
                        try:
                            m._engine.profile(False)
                        except Exception:
                            pass
                del g
                return round(best, 4), nodes

            for b in (1, 2, 4, 8):
                row = {'batch': b}
                auto_pipe = SpecPipeline(cc, hm)
                for m in (cc, hm):
                    m.set_plan('auto')
                what = auto_pipe.launch_structure((b, 3, 224, 224))          # asked, not re-derived: the pipeline's own rule + the library's plan
                row['auto_ms'], nodes = step_ms(auto_pipe, b)
                if b in (1, 8):
                    for tag, pp, plan in (('grouped', SpecPipeline(cc, hm, grouped=True), 'auto'),
                                          ('two_streams', SpecPipeline(cc, hm, overlap=True, grouped=False), 'auto'),
                                          ('grouped_latency_plan_r4_kernels', SpecPipeline(cc, hm, grouped=True), 'latency'),
                                          ('grouped_throughput_plan', SpecPipeline(cc, hm, grouped=True), 'throughput')):
                        for m in (cc, hm):
                            m.set_plan(plan)
                            m._engine.set_option('wsplit', 0 if 'r4_kernels' in tag else 1)
                        row[tag + '_ms'] = step_ms(pp, b)[0]
                    for m in (cc, hm):
                        m.set_plan('auto')
                        m._engine.set_option('wsplit', 1)
                ms = row['auto_ms']
                fl = small_batch_floor(b, nodes or 0)
                row.update({'ms_per_step': ms, 'images_per_s': round(b * 1e3 / ms, 1), 'plan': what['plan'], 'structure': what['structure'],
                            'graph_nodes': nodes,
                            'algorithmic_TFLOPs': round(b * 2 * TRUNK_GFLOP_PER_IMAGE / ms, 2),
                            'frac_of_mfma_peak_algorithmic': round(b * 2 * TRUNK_GFLOP_PER_IMAGE / ms / PEAK_FP32_MFMA_TFLOPS, 4),
                            'floor': fl, 'frac_of_floor': round(fl['floor_ms'] / ms, 4),
                            'frac_of_floor_with_boundaries': round(fl['floor_with_boundaries_ms'] / ms, 4)})
                if 'grouped_throughput_plan_ms' in row:
                    row['speedup_vs_throughput_plan'] = round(row['grouped_throughput_plan_ms'] / ms, 3)
                    row['speedup_vs_round4_latency_plan'] = round(row['grouped_latency_plan_r4_kernels_ms'] / ms, 3)
                small.append(row)
        except Exception as e:
            log('[bench] small-batch latency failed:', repr(e))

    # ---- host hand-over: what the step costs when the crops arrive in (pinned) host memory instead of HBM ----------------
    # informational only - `value` is measured with the inputs resident in HBM, as the contract requires
    pcie = None
    if rank == 0 and not args.no_profile:
        try:
            host = torch.empty(x.shape, dtype=x.dtype).pin_memory()
            host.copy_(x)
            dst = torch.empty_like(x)
            dst.copy_(host, non_blocking=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(host, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            h2d_ms = e0.elapsed_time(e1) / 5
            nbytes = x.numel() * 4
            pcie = {'h2d_batch_MB': round(nbytes / 1e6, 1), 'h2d_ms': round(h2d_ms, 3), 'h2d_GBps': round(nbytes / h2d_ms / 1e6, 1),
                    'images_per_s_if_serial': round(B * 1e3 / (ms_per_step + h2d_ms), 1),
                    'note': 'fp32 crops from pinned host memory; serial = copy then step, no overlap (a double-buffered '
                            'copy on its own stream hides it: h2d_ms < ms_per_step)'}
            del host, dst
        except Exception as e:
            log('[bench] pcie measurement failed:', repr(e))

    # ---- real-input line: uint8 1080p frames in pinned HOST memory, K detections each, uploaded on a copy
    # stream into two alternating device slabs, ONE batched crop launch into the step's input buffers, the same step.
    # Informational: `value` above stays the HBM-resident number the contract asks for.
    e2e = None
    if rank == 0 and not use_dist and not args.no_e2e:
        try:
            from spec_amd.frames import FrameStream
            Hf, Wf, Kdet = 1080, 1920, 8
            F = max(1, B // Kdet)
            fs = FrameStream(run, device, (Hf, Wf), F, B)
            gcpu = torch.Generator().manual_seed(77)
            hosts = []
            for _ in range(2):
                hf, hb, hi = fs.host_buffers()
                hf.copy_(torch.randint(0, 256, hf.shape, dtype=torch.uint8, generator=gcpu))
                cx = torch.rand(B, generator=gcpu) * Wf
                cy = torch.rand(B, generator=gcpu) * Hf
                bw = 150 + torch.rand(B, generator=gcpu) * 250          # person-sized boxes, some leaving the frame
                bh = 300 + torch.rand(B, generator=gcpu) * 500
                hb.copy_(torch.stack([cx, cy, bw, bh], 1))
                hi.copy_((torch.arange(B) // Kdet).clamp_(max=F - 1).to(torch.int32))
                hosts.append((hf, hb, hi))
            for s_ in range(3):
                fs.submit(*hosts[s_ % 2])
            fs.drain()
            fs.h2d_bytes = 0
            t1 = time.perf_counter()
            for s_ in range(args.steps):
                fs.submit(*hosts[s_ % 2])
            fs.drain()
            el = time.perf_counter() - t1
            # the upload alone (nothing else on the GPU): what PCIe gives for this slab
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fs.slabs[0].copy_(hosts[0][0], non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            up_ms = e0.elapsed_time(e1) / 3
            slab_bytes = hosts[0][0].numel()
            e2e = {'frames_per_s': round(F * args.steps / el, 1), 'crops_per_s': round(B * args.steps / el, 1),
                   'ms_per_step': round(el / args.steps * 1e3, 3), 'frame': f'{Wf}x{Hf} uint8 RGB', 'frames_per_step': F,
                   'detections_per_frame': Kdet, 'crops_per_step': B,
                   'h2d_MB_per_step': round(fs.h2d_bytes / args.steps / 1e6, 1),
                   'h2d_GBps_sustained_while_overlapped': round(fs.h2d_bytes / el / 1e9, 2),
                   'h2d_GBps_upload_alone': round(slab_bytes / up_ms / 1e6, 1), 'upload_alone_ms': round(up_ms, 3),
                   'overlap': round(B * args.steps / el / value, 4),
                   'copy_stream_probe': getattr(fs, 'copy_probe', None),
                   'flow': 'pinned host slab -> copy stream -> 2 alternating device slabs -> specmi_crop_normalize_batch into '
                           'the static inputs of the step -> ' + launch_mode,
                   'note': 'overlap = crops/s of this line / value (inputs resident in HBM); CamCalib sees the same crops as in '
                           'the headline configuration'}
            del fs, hosts
        except Exception as e:
            log('[bench] e2e_frames measurement failed:', repr(e))

    # ---- demo-shaped line: what a `scripts/spec_demo.py` user runs (camcalib/pano_dataset.py:156-162, scripts/camcalib_demo.py:
    # 95-129, spec/tester.py:86-88,109-151): CamCalib sees the FULL frame at short side 600 once per frame, SPEC sees the K
    # crops of that frame with the frame's camera.  1080p frames -> specmi_resize_normalize (600 x 1066) -> CamCalib, batch =
    # frames -> decode -> K = 8 crops per frame -> SPEC + SMPL.  Informational; `value` above stays the C3 number.
    demo = None
    if rank == 0 and not use_dist and not args.no_e2e:
        try:
            from spec_amd.pipeline import DemoPipeline, GraphedStep
            from spec_amd.preprocess import camcalib_transform_batch, resize_output_size
            from spec_amd.streams import concurrent_stream
            Hf, Wf, Kdet = 1080, 1920, 8
            F = max(1, B // Kdet)
            N = F * Kdet
            ow6, oh6 = resize_output_size(Wf, Hf, 600)
            gcpu = torch.Generator().manual_seed(78)
            hosts = []
            for _ in range(2):
                hf = torch.randint(0, 256, (F, Hf, Wf, 3), dtype=torch.uint8, generator=gcpu).pin_memory()
                cx, cy = torch.rand(N, generator=gcpu) * Wf, torch.rand(N, generator=gcpu) * Hf
                bw, bh = 150 + torch.rand(N, generator=gcpu) * 250, 300 + torch.rand(N, generator=gcpu) * 500
                hosts.append((hf, torch.stack([cx, cy, bw, bh], 1).pin_memory(), (torch.arange(N) // Kdet).to(torch.int32).pin_memory()))
            dp = DemoPipeline(cc, hm)
            dev_in = [[a.to(device) for a in h_] for h_ in hosts]
            # (a) CamCalib at 600 x 1066 alone, batch = F frames: per-stage table (HIP events, eager, one stream)
            cam_in = camcalib_transform_batch(dev_in[0][0], 600)
            for _ in range(2):
                cc(cam_in)
            torch.cuda.synchronize()
            cc._engine.profile(True)
            cc._engine.profile_read()
            n6 = 3
            t1 = time.perf_counter()
            for _ in range(n6):
                cc(cam_in)
            torch.cuda.synchronize()
            cam_ms = (time.perf_counter() - t1) / n6 * 1e3
            ents6 = []
            for r in cc._engine.profile_read():
                r['model'] = 'camcalib'
                r['ms'] /= n6; r['flops'] /= n6; r['bytes'] /= n6; r['launches'] //= n6
                ents6.append(r)
            cc._engine.profile(False)
            st6 = stage_table(ents6)
            k_ms = sum(e['ms'] for e in ents6)
            ex_fl = sum(executed_flops(e) for e in ents6)
            alg_fl = sum(e['flops'] for e in ents6)
            cam600 = {'input': f'{F} x 3 x {oh6} x {ow6} (1080p -> Resize(600), Pillow-exact on the device)', 'frames': F,
                      'ms_per_step': round(cam_ms, 3), 'ms_per_frame': round(cam_ms / F, 4), 'kernel_ms': round(k_ms, 3),
                      'frames_per_s': round(F * 1e3 / cam_ms, 1),
                      'algorithmic_gflop_per_frame': round(alg_fl / F / 1e9, 2),
                      'executed_mfma_TFLOPs': round(ex_fl / (k_ms * 1e-3) / 1e12, 2),
                      'frac_of_mfma_peak_executed': round(ex_fl / (k_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                      'algorithmic_TFLOPs': round(alg_fl / (k_ms * 1e-3) / 1e12, 2),
                      'final_map': '19 x 34 (ragged: M = F x 646 rows, Winograd rows of 32 tiles half empty at the right edge)',
                      'stages': st6}
            del cam_in
            # (b) the whole demo step, frames resident in HBM, hipGraph replay, CamCalib on a second stream beside the SPEC trunk
            steps = [GraphedStep(dp, *dev_in[i]) for i in range(2)]
            for i in range(4):
                steps[i % 2](*steps[i % 2].static_in)
            torch.cuda.synchronize()
            nd = max(6, args.steps // 2)
            t1 = time.perf_counter()
            for i in range(nd):
                steps[i % 2](*steps[i % 2].static_in)
            torch.cuda.synchronize()
            res_ms = (time.perf_counter() - t1) / nd * 1e3
            # (c) the same with every slab uploaded from pinned host memory on a copy stream while the other slab's step runs
            cs_probe = {}
            copy_stream = concurrent_stream(device, lambda: steps[0](*steps[0].static_in), probe=cs_probe)
            ready = [torch.cuda.Event() for _ in range(2)]
            done = [None, None]
            main = torch.cuda.current_stream(device)

            def submit(i, host):
                with torch.cuda.stream(copy_stream):
                    if done[i] is not None:
                        copy_stream.wait_event(done[i])            # the replay that last read this slab has finished
                    for dst, src in zip(steps[i].static_in, host):
                        dst.copy_(src, non_blocking=True)
                    ready[i].record(copy_stream)
                main.wait_event(ready[i])
                steps[i].graph.replay()
                ev = torch.cuda.Event()
                ev.record(main)
                done[i] = ev
            for i in range(4):
                submit(i % 2, hosts[i % 2])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(nd):
                submit(i % 2, hosts[i % 2])
            torch.cuda.synchronize()
            up_ms = (time.perf_counter() - t1) / nd * 1e3
            slab_mb = hosts[0][0].numel() / 1e6
            # (d) the reference's own granularity: ONE frame per step (spec/tester.py:109-151 - batch = the frame's K detections,
            # CamCalib at batch 1), frame resident in HBM, hipGraph replay: the latency a spec_demo.py user sees per frame
            single = {}
            one = [a[:1].contiguous() if a.dim() == 4 else a[:Kdet].contiguous() for a in (hosts[0][0], hosts[0][1], hosts[0][2])]
            one = [a.to(device) for a in one]
            for plan in ('auto', 'throughput', 'latency'):
                for m in (cc, hm):
                    m.set_plan(plan)
                g1 = GraphedStep(dp, *one)
                for _ in range(5):
                    g1(*g1.static_in)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(50):
                    g1(*g1.static_in)
                torch.cuda.synchronize()
                single[plan + '_plan_ms'] = round((time.perf_counter() - t1) / 50 * 1e3, 3)
                del g1
            for m in (cc, hm):
                m.set_plan('auto')
            single.update({'frames_per_s': round(1e3 / single['auto_plan_ms'], 1), 'detections': Kdet,
                           'note': 'one 1080p frame per step: CamCalib at 1 x 3 x 600 x 1066 (12.7 crops worth of rows; auto: latency plan up '
                                   'to 16 for a single trunk) beside the SPEC trunk on its 8 crops (auto: latency plan)'})
            demo = {'frame': f'{Wf}x{Hf} uint8 RGB', 'frames_per_step': F, 'detections_per_frame': Kdet, 'crops_per_step': N,
                    'frames_per_s': round(F * 1e3 / up_ms, 1), 'crops_per_s': round(N * 1e3 / up_ms, 1), 'ms_per_step': round(up_ms, 3),
                    'frames_resident_in_hbm': {'ms_per_step': round(res_ms, 3), 'frames_per_s': round(F * 1e3 / res_ms, 1)},
                    'overlap': round(res_ms / up_ms, 4), 'h2d_MB_per_step': round(slab_mb, 1),
                    'h2d_GBps_sustained_while_overlapped': round(slab_mb / up_ms, 2), 'copy_stream_probe': cs_probe,
                    'sum_of_parts_ms': round(cam_ms + ms_per_step / 2, 3), 'single_frame': single,
                    'camcalib_at_600': cam600,
                    'flow': 'pinned host slab -> copy stream -> 2 alternating device slabs (each the static input of its own hipGraph) -> '
                            'specmi_resize_normalize x F -> CamCalib (batch F, second stream) || specmi_crop_normalize_batch -> SPEC '
                            'trunk -> join -> regressor head with the frame camera of every crop -> SMPL -> projection',
                    'note': 'frames/s of the configuration scripts/spec_demo.py runs; overlap = resident / uploaded step time; '
                            'sum_of_parts = CamCalib-at-600 alone + half the C3 step (SPEC trunk + heads of 256 crops)'}
            del steps, dev_in, hosts
        except Exception as e:
            log('[bench] e2e_demo measurement failed:', repr(e))

    # ---- labelled secondary line: the plain 1x1 convolutions as bf16 piece products on the bf16 matrix cores ----------
    # NOT `value` (which is exact fp32 MFMA arithmetic): a different algorithm for the same contractions, reported with its
    # deviation from the fp32 path on the same inputs (tests/test_gpu_bf16split.py holds it to the 1e-4 fixtures)
    split = None
    if rank == 0 and not use_dist and not args.no_split_bf16:
        split = []
        try:
            ref_out = {k: v.clone() for k, v in pipe(x, scale, center, img_w, img_h).items() if k in ERR_KEYS}
            for terms in (6, 3):
                cc2, hm2, _, _ = build_models(device, conv_precision=terms)
                pipe2 = SpecPipeline(cc2, hm2, overlap=not args.no_overlap)
                out2 = pipe2(x, scale, center, img_w, img_h)
                errs = {k: float((out2[k].double() - ref_out[k].double()).abs().max() / ref_out[k].double().abs().max())
                        for k in ERR_KEYS}
                run2, mode2 = pipe2, 'eager launches'
                if not args.no_graph:
                    try:
                        from spec_amd.pipeline import GraphedPipeline
                        run2, mode2 = GraphedPipeline(pipe2, x, scale, center, img_w, img_h), 'hipGraph replay'
                    except Exception as e:
                        log('[bench] split-bf16 graph capture failed:', repr(e))
                for _ in range(args.warmup):
                    run2(x, scale, center, img_w, img_h)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    run2(x, scale, center, img_w, img_h)
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
                split.append({'terms': terms, 'images_per_s': round(B * args.steps / el, 1), 'ms_per_step': round(el / args.steps * 1e3, 3),
                              'speedup_vs_value': round(B * args.steps / el / value, 3),
                              'max_rel_err_vs_fp32_path': {k: float(f'{v:.3e}') for k, v in errs.items()},
                              'worst_rel_err_vs_fp32_path': float(f'{max(errs.values()):.3e}'),
                              'fixtures_within_1e-4': 'tests/test_gpu_bf16split.py (3 HMR fixtures + CamCalib fixture, both term counts)',
                              'layers': ('every 1x1 and 3x3 convolution of both ResNet-50 trunks (48 per trunk)' if terms == 3 else
                                         'the 32 1x1 convolutions and the 3 stride-2 3x3 convolutions of each ResNet-50 trunk (the 13 stride-1 3x3 '
                                         'layers stay on the fp32 Winograd kernel, which is faster than six bf16 products)') +
                                        '; stem and FC layers stay on the exact fp32 kernels',
                              'launch': mode2})
                del cc2, hm2, pipe2, run2, out2
        except Exception as e:
            log('[bench] split-bf16 line failed:', repr(e))

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:      # (N > 1 too: north_star wants it "in the same run"; the other ranks wait at the final barrier)
        try:
            if cs is None:
                from spec_amd import synth
                cs, hs = synth.camcalib_state(1001), synth.hmr_state(1002, True)
            cpu = cpu_baseline(cs, hs, budget_s=args.cpu_baseline_seconds)
            if n_gpus > 1:
                cpu['ranks_waiting_at_the_barrier_meanwhile'] = n_gpus - 1
        except Exception as e:  # the baseline must never sink the bench line
            log('[bench] cpu baseline failed:', repr(e))
        if cpu is not None:
            try:
                quota = cgroup_cpu_quota()
                cpu['host']['cgroup_cpu_quota'] = quota
                usable = cpu['host']['physical_cores'] if quota is None else min(cpu['host']['physical_cores'], int(quota))
                cpu['host']['usable_cores'] = usable
                whole = None
                if usable >= 32 and not fake:   # more than one 16-thread process fits: cover the host with pinned processes
                    whole = cpu_baseline_whole_host(usable)
                else:
                    cpu['whole_host'] = (f'the container may use {usable} CPUs at once (cgroup cpu.max; {cpu["host"]["physical_cores"]} '
                                         'physical cores are visible but throttled beyond that - which is why the thread sweep peaks '
                                         f'at {cpu["cores"]}): the single-process figure IS the whole-host figure; a run of 8 pinned '
                                         'processes x 16 threads on this box measured 8.8 images/s (profiles/r03_g_bench.json)')
                if whole is not None:
                    # headline = the whole host; the single-process figures stay beside it
                    cpu['single_process'] = {'value': cpu['value'], 'cores': cpu['cores'], 'sample': cpu['sample']}
                    cpu['whole_host'] = whole
                    cpu['value'], cpu['cores'] = whole['value'], whole['cores']
                    cpu['sample'] = (f"{whole['processes']} processes x {whole['threads_per_process']} threads, each pinned to its own "
                                     f"cores, batches of 16 synthetic 224x224 images through the PyTorch-CPU fp32 oracle (full "
                                     f"CamCalib+SPEC+SMPL forward) for {whole['window_s']:.0f} s; single process: "
                                     + cpu['single_process']['sample'])
            except Exception as e:
                log('[bench] whole-host cpu baseline failed:', repr(e))

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
                       'parallelism': (f'images sharded over {n_gpus} GPUs (one process per GPU), 1 asynchronous RCCL '
                                       f'all-gather of the packed records per step') if n_gpus > 1 else 'single GPU'},
            'roofline': roof, 'cpu_baseline': cpu,
            # the per-stage tables are the bulk of the line: they come FIRST among the extras, the compact results LAST (a
            # truncated tail of this line then still holds them)
            'stages': stages, 'split_bf16': split, 'e2e_frames': e2e, 'e2e_demo': demo, 'sustained': sustained, 'pcie': pcie, 'comm': comm,
            'c2': c2, 'small_batch': small,
        }
        summary = {'value_images_per_s': line['value'], 'ms_per_step': line['ms_per_step'],
                   'roofline_frac': (roof or {}).get('frac'), 'cpu_baseline_images_per_s': (cpu or {}).get('value')}
        try:
            for row in small or []:
                summary[f"small_batch_b{row['batch']}_ms"] = row['ms_per_step']
                summary[f"small_batch_b{row['batch']}_frac_of_floor"] = row['frac_of_floor']
                summary[f"small_batch_b{row['batch']}_nodes"] = row.get('graph_nodes')
            if c2:
                summary['c2_images_per_s'] = c2.get('images_per_s')
            if demo:
                summary['e2e_demo_frames_per_s'] = demo.get('frames_per_s')
                summary['e2e_demo_single_frame_auto_plan_ms'] = (demo.get('single_frame') or {}).get('auto_plan_ms')
            if e2e:
                summary['e2e_frames_per_s'] = e2e.get('frames_per_s')
        except Exception as e:
            log('[bench] summary incomplete:', repr(e))
        line['summary'] = summary
        if fake:
            line['dry_run'] = True
            line['data'] = 'none: --fake-forward dry run of the N > 1 control flow on CPU tensors (gloo); not a measurement'
        print(json.dumps(line), flush=True)
        # the compact scalars once more as the LAST line of stderr (the driver keeps the tail of the streams, and drops the extra
        # keys of the stdout line): <= 1.5 KB
        if comm:
            summary.update({'comm_world_size': comm.get('world_size'), 'comm_probe_equal_unsharded': comm.get('probe_gathered_16_images_equal_unsharded'),
                            'comm_overlap_efficiency': comm.get('overlap_efficiency'), 'comm_all_gather_ms': comm.get('all_gather_ms_blocking_full_record')})
        sys.stdout.flush()
        log('[bench] summary ' + json.dumps(summary)[:1500])
    if use_dist:
        dist.barrier()          # rank 0 may still be in its profiling pass
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
