"""Fused tails (head.hip: tail_gemv_kernel): bit-equality vs the separate kernels and whole-step time, batch 1..10."""
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
def opt(n, v):
    ce.set_option(n, v); he.set_option(n, v)
def step_ms(pp, b, iters=200):
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
out = open(os.path.join(ROOT, 'gpurun_out', 'tail_check.jsonl'), 'a')
keys = ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'pred_shape', 'pred_cam', 'pred_pose_6d', 'cam_vfov', 'cam_pitch', 'cam_roll',
        'cam_f_pix', 'cam_rotmat', 'cam_intrinsics')
for b in (1, 2, 3, 5, 8, 10):
    ins = (x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
    res = {}
    for fuse in (0, 1):
        opt('tail_fuse', fuse)
        o = SpecPipeline(cc, hm)(*ins); torch.cuda.synchronize()
        res[fuse] = {k: o[k].clone() for k in keys}
    bad = [k for k in keys if not torch.equal(res[0][k], res[1][k])]
    row = {'batch': b, 'equal': not bad, 'differ': bad, 'finite': bool(all(torch.isfinite(v).all() for v in res[1].values()))}
    for fuse in (0, 1):
        opt('tail_fuse', fuse)
        row[f'fuse{fuse}_ms'] = step_ms(SpecPipeline(cc, hm), b)
    opt('tail_fuse', 0)
    line = json.dumps(row); print(line, flush=True); out.write(line + '\n'); out.flush()
