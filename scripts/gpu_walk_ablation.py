"""Round-6 bound for the last small-batch lever: software pipelining ACROSS the tiles a resident workgroup walks
inside one layer of the latency plan.  Measured before anything is written, the way the walker and Winograd were bounded:

  perlayer   today's latency plan: one launch per layer, one workgroup per (tile, K slice)
  walk       every implicit-GEMM layer as its own launch of N resident workgroups that WALK the layer's (tile, K slice) items
             (option persist = 1 with persist_max_run = 1: no in-launch waits, correct results) - tile walking as it exists
  walkplain  the same built with -DSPECMI_PERSIST_AUXA=0: plain activation loads (no producer inside the launch any more)
  walkabl    -DSPECMI_PERSIST_AUXA=0 -DSPECMI_WALK_ABLATE=1: every item after a workgroup's first gets its first two chunks' operands
             FOR FREE and pays no hand-off - the UPPER BOUND of "issue the next tile's first two chunks under the current epilogue"
             (results wrong, timing only)

for the trunk pair (grouped launches) and for a single trunk, batch 8 and 16, hipGraph replay.  Each variant runs in its own process
(SPECMI_LIB selects the library; scripts/build_variants.sh builds them).  Output: gpurun_out/walk_ablation.jsonl"""
import json
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from spec_amd import synth, assets
from spec_amd.modules import HMR, CameraRegressorNetwork
from spec_amd.pipeline import SpecPipeline, GraphedPipeline
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
torch.set_grad_enabled(False)
dev = 'cuda:0'
cs, hs = synth.camcalib_state(1001), synth.hmr_state(1002, True)
assets.use_synthetic_assets(1003)
cc = CameraRegressorNetwork(); cc.load_state_dict({k: t(v) for k, v in cs.items()})
hm = HMR(use_cam=True, use_cam_feats=True); hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
cc = cc.to(dev).eval(); hm = hm.to(dev).eval()
cc.commit(dev, freeze=True); hm.commit(dev, freeze=True)
ce, he = cc.engine(dev), hm.engine(dev)
x = t(synth.images(9, 16)).to(dev)
sc, cen, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(9, 16, 640., 480.)]
def opt(n, v):
    ce.set_option(n, v); he.set_option(n, v)
def time_fn(fn, iters=150):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return round(best, 4)
def graphed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        keep = fn()
    ms = time_fn(g.replay)
    del g
    return ms
def pair_ms(b):
    xb = x[:b].contiguous()
    return graphed(lambda: ce.trunk_pair(he, xb, xb))
def single_ms(b):
    xb = x[:b].contiguous()
    return graphed(lambda: he.trunk(xb))
def step_ms(b, grouped):
    pp = SpecPipeline(cc, hm, grouped=grouped, overlap=True)
    g = GraphedPipeline(pp, x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
    ms = time_fn(lambda: g(*g.static_in))
    del g
    return ms
cc.set_plan('latency'); hm.set_plan('latency')
for b in (8, 16):
    row = {'variant': %(name)r, 'batch': b, 'plan': 'latency'}
    opt('persist', 0)
    row['perlayer_pair'] = pair_ms(b); row['perlayer_single'] = single_ms(b)
    row['perlayer_step_grouped'] = step_ms(b, True); row['perlayer_step_two_streams'] = step_ms(b, False)
    opt('persist', 1); opt('persist_min_run', 1); opt('persist_max_run', 1); opt('persist_allow_full', 1)
    for nwg in %(nwgs)r:
        opt('persist_wgs', nwg)
        for tag, fn in (('pair', pair_ms), ('single', single_ms)):
            try:
                row['walk_w%%d_%%s' %% (nwg, tag)] = fn(b)
            except Exception as e:
                row['walk_w%%d_%%s' %% (nwg, tag)] = repr(e)[:80]
    best = min((v, k) for k, v in row.items() if k.startswith('walk_w') and k.endswith('_pair') and isinstance(v, float))
    opt('persist_wgs', int(best[1].split('_')[1][1:]))
    row['walk_best_pair_wgs'] = int(best[1].split('_')[1][1:])
    row['walk_step_grouped'] = step_ms(b, True)
    best = min((v, k) for k, v in row.items() if k.startswith('walk_w') and k.endswith('_single') and isinstance(v, float))
    opt('persist_wgs', int(best[1].split('_')[1][1:]))
    row['walk_best_single_wgs'] = int(best[1].split('_')[1][1:])
    row['walk_step_two_streams'] = step_ms(b, False)
    row['sync'] = (ce.sync_status(), he.sync_status())
    print('ROW ' + json.dumps(row), flush=True)
'''


def main():
    out = os.path.join(ROOT, 'gpurun_out', 'walk_ablation.jsonl')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    variants = [('default', None)]
    vdir = os.path.join(ROOT, 'spec_amd', 'lib', 'variants')
    for name in ('walkplain', 'walkabl'):
        f = os.path.join(vdir, f'libspecmi_{name}.so')
        if os.path.exists(f):
            variants.append((name, f))
    nwgs = [256, 384, 512, 768, 1024]
    with open(out, 'a') as fo:
        for name, lib in variants:
            env = dict(os.environ)
            if lib:
                env['SPECMI_LIB'] = lib
            code = WORKER % {'root': ROOT, 'name': name, 'nwgs': nwgs}
            try:
                r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=900)
            except subprocess.TimeoutExpired:
                print('TIMEOUT', name, flush=True)
                continue
            for line in r.stdout.splitlines():
                if line.startswith('ROW '):
                    print(line[4:], flush=True)
                    fo.write(line[4:] + '\n')
            if r.returncode:
                print('FAILED', name, r.stderr[-1500:], flush=True)


if __name__ == '__main__':
    main()
