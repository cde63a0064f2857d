"""Experiment: a small batch as R independent sub-batches, each on its own model replicas (own handles, own workspaces) and its own
pair of streams, inside ONE captured graph - do their per-layer latencies overlap?  (bits: batch-invariant within a plan.)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spec_amd import synth, assets
from spec_amd.modules import HMR, CameraRegressorNetwork
from spec_amd.pipeline import SpecPipeline, GraphedStep
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cs, hs = synth.camcalib_state(1001), synth.hmr_state(1002, True)
assets.use_synthetic_assets(1003)
def build():
    cc = CameraRegressorNetwork(); cc.load_state_dict({k: t(v) for k, v in cs.items()})
    hm = HMR(use_cam=True, use_cam_feats=True); hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
    cc = cc.to(dev).eval(); hm = hm.to(dev).eval()
    cc.commit(dev, freeze=True); hm.commit(dev, freeze=True)
    return cc, hm
R = int(os.environ.get('REPLICAS', '4'))
ONLY = os.environ.get('ONLY')   # e.g. '8:2:auto'
reps = [build() for _ in range(R)]
x = t(synth.images(9, 16)).to(dev)
sc, cen, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(9, 16, 640., 480.)]
KEYS = ('smpl_vertices', 'smpl_joints2d', 'pred_cam_t', 'cam_vfov')

def make_step(nrep, grouped):
    pipes = [SpecPipeline(cc, hm, grouped=grouped) for cc, hm in reps[:nrep]]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nrep)]
    def step(xx, s_, c_, w_, h_):
        b = xx.shape[0]
        bounds = [(i * b) // nrep for i in range(nrep + 1)]
        main = torch.cuda.current_stream(dev)
        outs = []
        for i in range(nrep):
            lo, hi = bounds[i], bounds[i + 1]
            if i == 0:
                outs.append(pipes[0](xx[lo:hi], s_[lo:hi], c_[lo:hi], w_[lo:hi], h_[lo:hi]))
                continue
            st = streams[i]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(pipes[i](xx[lo:hi], s_[lo:hi], c_[lo:hi], w_[lo:hi], h_[lo:hi]))
        for i in range(1, nrep):
            main.wait_stream(streams[i])
        return {k: torch.cat([o[k] for o in outs], 0) for k in KEYS}
    return step

def time_fn(g, ins, iters=200):
    for _ in range(10): g(*ins)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): g(*ins)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return round(best, 4)

out = open(os.path.join(ROOT, 'gpurun_out', 'subbatch_overlap.jsonl'), 'a')
for b in ([int(ONLY.split(':')[0])] if ONLY else (4, 6, 8, 10, 12, 16)):
    ins = tuple(a[:b].contiguous() for a in (x, sc, cen, iw, ih))
    row = {'batch': b}
    ref = None
    for nrep in ([int(ONLY.split(':')[1])] if ONLY else (1, 2, 4)):
        if b % nrep or nrep > R:
            continue
        for grouped in ([{'auto': 'auto', '1': True, '0': False}[ONLY.split(':')[2]]] if ONLY else ('auto', True, False)):
            try:
                g = GraphedStep(make_step(nrep, grouped), *ins)
                o = g(*ins)
                torch.cuda.synchronize()
                if ref is None:
                    ref = {k: v.clone() for k, v in o.items()}
                eq = all(torch.equal(o[k], ref[k]) for k in KEYS)
                row[f'r{nrep}_{grouped}'] = time_fn(g, g.static_in)
                row[f'r{nrep}_{grouped}_equal'] = bool(eq)
                del g
            except Exception as e:
                row[f'r{nrep}_{grouped}'] = repr(e)[:100]
    line = json.dumps(row); print(line, flush=True); out.write(line + '\n'); out.flush()
