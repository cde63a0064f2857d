"""Whole-step time (hipGraph replay, auto structure) vs option latency_fill_wgs at batch 4-16."""
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
def step_ms(pp, b, iters=150):
    g = GraphedPipeline(pp, x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
    ins = g.static_in
    for _ in range(10): g(*ins)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): g(*ins)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    del g
    return round(best, 4)
out = open(os.path.join(ROOT, 'gpurun_out', 'fill_sweep.jsonl'), 'a')
for b in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4,6,8,10,12,16').split(',')]:
    row = {'batch': b}
    for fill in [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '120,160,200,240,280,400').split(',')]:
        # fill > 0: the threshold rule; fill < 0: the round model with -fill slots (option latency_unit_model)
        for e in (ce, he):
            e.set_option('latency_unit_model', 1 if fill < 0 else 0)
            e.set_option('latency_unit_slots', -fill if fill < 0 else 256)
            e.set_option('latency_fill_wgs', fill if fill > 0 else 240)
        row[f'fill{fill}' if fill > 0 else f'model{-fill}'] = step_ms(SpecPipeline(cc, hm), b)
    for e in (ce, he):
        e.set_option('latency_fill_wgs', 240); e.set_option('latency_unit_model', 0)
    line = json.dumps(row); print(line, flush=True); out.write(line + '\n'); out.flush()
