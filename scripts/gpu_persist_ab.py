"""A/B timing of the persistent walker under compile-time variants (scripts/build_variants.sh) and runtime knobs.
Each variant runs in its own process (SPECMI_LIB selects the library).  Output: gpurun_out/persist_ab.jsonl"""
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
def time_fn(fn, iters=200):
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
def step_ms(b, grouped=True):
    pp = SpecPipeline(cc, hm, grouped=grouped, overlap=True)
    g = GraphedPipeline(pp, x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
    ms = time_fn(lambda: g(*g.static_in))
    del g
    return ms
def trunk_ms(b):
    xb = x[:b].contiguous()
    ce.trunk_pair(he, xb, xb); ce.trunk_pair(he, xb, xb)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        fa, fb = ce.trunk_pair(he, xb, xb)
    return time_fn(g.replay)
rows = []
for b, plan in %(cases)r:
    cc.set_plan(plan); hm.set_plan(plan)
    row = {'variant': %(name)r, 'batch': b, 'plan': plan}
    opt('persist', 0)
    row['perlayer_step'] = step_ms(b); row['perlayer_trunks'] = trunk_ms(b)
    opt('persist', 1); opt('persist_allow_full', 1)
    for nwg in %(nwgs)r:
        for pf in (0, 1):
            opt('persist_wgs', nwg); opt('persist_l2_prefetch', pf)
            try:
                row['persist_w%%d_pf%%d_trunks' %% (nwg, pf)] = trunk_ms(b)
                if pf == 0: row['persist_w%%d_step' %% nwg] = step_ms(b)
            except Exception as e:
                row['persist_w%%d_pf%%d_trunks' %% (nwg, pf)] = repr(e)[:60]
    row['sync'] = (ce.sync_status(), he.sync_status())
    print('ROW ' + json.dumps(row), flush=True)
'''


def main():
    out = os.path.join(ROOT, 'gpurun_out', 'persist_ab.jsonl')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    variants = [('default', None)]
    vdir = os.path.join(ROOT, 'spec_amd', 'lib', 'variants')
    if os.path.isdir(vdir):
        for f in sorted(os.listdir(vdir)):
            if f.startswith('libspecmi_') and f.endswith('.so'):
                variants.append((f[len('libspecmi_'):-3], os.path.join(vdir, f)))
    cases = [(1, 'single'), (2, 'single'), (4, 'latency'), (8, 'latency')]
    nwgs = [256, 512, 768, 1024]
    with open(out, 'a') as fo:
        for name, lib in variants:
            env = dict(os.environ)
            if lib:
                env['SPECMI_LIB'] = lib
            code = WORKER % {'root': ROOT, 'cases': cases, 'name': name, 'nwgs': nwgs}
            try:
                r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
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
