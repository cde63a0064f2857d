#!/usr/bin/env python
"""Eager two-stream step (no graph): CamCalib on a side stream taken from the pool vs one chosen by measurement, late in a
process that has created other streams (run on the GPU box)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spec_amd.pipeline import SpecPipeline
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
cc, hm, cs, hs = bench.build_models(dev)
ins = bench.make_inputs(256, dev, 20210001)


def t_step(p, n=6):
    for _ in range(2):
        p(*ins)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        p(*ins)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


one = SpecPipeline(cc, hm, overlap=False, grouped=False)
for rnd in range(6):
    junk = [torch.cuda.Stream(device=dev) for _ in range(rnd)]
    naive = SpecPipeline(cc, hm, overlap=True, grouped=False)
    naive._side[dev] = torch.cuda.Stream(device=dev)            # what the pool hands out next
    meas = SpecPipeline(cc, hm, overlap=True, grouped=False)   # measured at its first call
    print('round %d: one stream %.2f ms   two streams, next pool stream %.2f   two streams, measured stream %.2f'
          % (rnd, t_step(one), t_step(naive), t_step(meas)), flush=True)
