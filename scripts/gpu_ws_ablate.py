"""Timing ablation of the wave-split kernel at a mid batch (WRONG results by design): variants that make every tile read the same
A rows / B columns (spec_amd/lib/variants/libspecmi_ws{A,B,AB}.so, scripts/build_variants.sh) - is the unit operand-bandwidth bound
when 3 workgroups share a CU?  Per-layer HIP-event times, trunk pair, wsplit forced (all groups per workgroup)."""
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r'''
import os, sys, json
import numpy as np, torch
sys.path.insert(0, %(root)r)
from spec_amd import synth, assets
from spec_amd.modules import HMR, CameraRegressorNetwork
from spec_amd.pipeline import SpecPipeline
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
cc.set_plan('latency'); hm.set_plan('latency')
b = %(b)d
x = t(synth.images(9, b)).to(dev)
sc, cen, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(9, b, 640., 480.)]
pipe = SpecPipeline(cc, hm, overlap=False, grouped=True)
res = {}
for tag, ws in (('k64', 0), ('ws_all', 3), ('ws_group', 2)):
    for e in (ce, he): e.set_option('wsplit', ws)
    for _ in range(3): pipe(x, sc, cen, iw, ih)
    torch.cuda.synchronize()
    ce.profile(True)
    for _ in range(20): pipe(x, sc, cen, iw, ih)
    torch.cuda.synchronize()
    rows = ce.profile_read(); ce.profile(False)
    res[tag] = {r['label'][9:]: round(r['ms'] / 20 * 1e3, 1) for r in rows if r['label'].startswith('backbone.')}
print('ROW ' + json.dumps({'variant': %(name)r, 'batch': b, 'layers': {l: [res[t_].get(l) for t_ in ('k64', 'ws_all', 'ws_group')] for l in ('layer3.1.conv1', 'layer3.1.conv2', 'layer4.1.conv1', 'layer4.1.conv2', 'layer4.1.conv3', 'layer2.1.conv1')}}))
'''
out = os.path.join(ROOT, 'gpurun_out', 'ws_ablate.jsonl')
with open(out, 'a') as fo:
    for name in ('default', 'wsA', 'wsB', 'wsAB'):
        env = dict(os.environ)
        if name != 'default':
            env['SPECMI_LIB'] = os.path.join(ROOT, 'spec_amd', 'lib', 'variants', f'libspecmi_{name}.so')
        for b in (8, 4):
            r = subprocess.run([sys.executable, '-c', W % {'root': ROOT, 'b': b, 'name': name}], env=env, capture_output=True, text=True, timeout=300)
            for line in r.stdout.splitlines():
                if line.startswith('ROW '):
                    print(line[4:], flush=True); fo.write(line[4:] + '\n')
            if r.returncode: print('FAILED', name, r.stderr[-500:])
