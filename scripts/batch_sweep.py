#!/usr/bin/env python
"""Whole-step latency / throughput over the batch size (2 streams + hipGraph replay), one JSON line per batch."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spec_amd.pipeline import SpecPipeline, GraphedPipeline

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cc, hm, _, _ = bench.build_models(dev)
pipe = SpecPipeline(cc, hm, overlap=True)
for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    x, sc, ce, iw, ih = bench.make_inputs(B, dev, 7)
    g = GraphedPipeline(pipe, x, sc, ce, iw, ih)
    ins = g.static_in
    for _ in range(5):
        g(*ins)
    torch.cuda.synchronize()
    n = 50 if B <= 64 else 15
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g(*ins)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(json.dumps({'batch': B, 'ms_per_step': round(ms, 3), 'images_per_s': round(B * 1e3 / ms, 1)}), flush=True)
    del g
