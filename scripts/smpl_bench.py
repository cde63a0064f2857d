#!/usr/bin/env python
"""GPU box: time the SMPL stage (pose chain + skinning + joints / projection) per batch size with the skinning variant forced
(option "smpl_skin_split": 0 = one wave per 32-vertex group, 1 = three) - the measurement behind SKIN_SPLIT_MAX_TILES."""
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spec_amd import synth
from spec_amd.engine import Engine
from spec_amd.pipeline import GraphedStep

DEV = torch.device('cuda:0')
e = Engine('hmr', DEV)
e.load(synth.hmr_state(1002, False), smpl=synth.smpl_model(1003), use_cam=0, use_cam_feats=0, img_res=224)
g = torch.Generator().manual_seed(0)
rows = []
for B in ([int(v) for v in sys.argv[1:]] or (1, 8, 32, 64, 96, 128, 256)):
    R = torch.linalg.qr(torch.randn(B * 24, 3, 3, generator=g))[0].view(B, 24, 3, 3).to(DEV)
    betas = torch.randn(B, 10, generator=g).to(DEV)
    cam = torch.tensor([[0.9, 0.1, 0.1]] * B).to(DEV)
    row = {'batch': B}
    for split in (0, 1):
        e.set_option('smpl_skin_split', split)
        step = GraphedStep(lambda: e.smpl(R, betas, cam))
        for _ in range(5): step()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        t0.record()
        for _ in range(n): step()
        t1.record(); torch.cuda.synchronize()
        row[f'split{split}_us'] = round(t0.elapsed_time(t1) / n * 1e3, 2)
    rows.append(row)
    print(json.dumps(row), flush=True)
