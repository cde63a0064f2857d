"""Whole-step time (hipGraph replay) around the plan / launch-structure switch points of plan 'auto': plans single (batch <= 4),
latency and throughput, both trunks grouped per layer and two trunks on two streams.  Every row carries the box it ran on (host name
+ GPU unique id), so sweeps of different gpurun calls can be told apart (round 6: the rules are kept only where two boxes agree)."""
import json, os, socket, subprocess, sys
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
x = t(synth.images(9, 32)).to(dev)
sc, cen, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(9, 32, 640., 480.)]
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
out = open(os.path.join(ROOT, 'gpurun_out', 'plan_crossover.jsonl'), 'a')
def _box_id():
    """GPU unique id: sysfs first, rocm-smi's table as a fallback."""
    import glob
    for f in sorted(glob.glob('/sys/class/drm/card*/device/unique_id')):
        try:
            v = open(f).read().strip()
            if v:
                return v
        except Exception:
            pass
    try:
        txt = subprocess.run(['rocm-smi', '--showuniqueid'], capture_output=True, text=True, timeout=30).stdout
        for l in txt.splitlines():
            if l.startswith('GPU[') and 'Unique ID' in l:
                return l.split(':')[-1].strip()
    except Exception:
        pass
    return None
uid = [_box_id()]
BOX = {'host': socket.gethostname(), 'gpu_unique_id': uid[0] if uid else None}
for b in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '8,10,11,12,14,16,20,24').split(',')]:
    row = {'batch': b, 'box': BOX}
    for plan in (('single',) if b <= 4 else ()) + ('latency', 'throughput'):
        cc.set_plan(plan); hm.set_plan(plan)
        row[f'{plan}_grouped'] = step_ms(SpecPipeline(cc, hm, grouped=True), b)
        row[f'{plan}_two_streams'] = step_ms(SpecPipeline(cc, hm, overlap=True, grouped=False), b)
    cc.set_plan('auto'); hm.set_plan('auto')
    row['auto'] = step_ms(SpecPipeline(cc, hm), b)
    line = json.dumps(row); print(line, flush=True); out.write(line + '\n'); out.flush()
