#!/usr/bin/env python
"""One small-batch step under hipGraph replay, N times - the command to run under `rocprofv3 --kernel-trace` for the
per-kernel timeline of the latency plan (scripts/rocprof_summary.py timeline)."""
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spec_amd.pipeline import SpecPipeline, GraphedPipeline

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--plan', default='auto')
ap.add_argument('--grouped', default='auto', help="'auto' (what a caller gets), 1 or 0")
ap.add_argument('--opt', action='append', default=[], help='library option name=value for both models (e.g. persist=1, tail_fuse=1)')
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cc, hm, _, _ = bench.build_models(dev)
for m in (cc, hm):
    m.set_plan(args.plan)
for o in args.opt:
    k, v = o.split('=')
    for m in (cc, hm):
        m.engine(dev).set_option(k, int(v))
x, sc, ce, iw, ih = bench.make_inputs(args.batch, dev, 7)
grouped = 'auto' if args.grouped == 'auto' else bool(int(args.grouped))
g = GraphedPipeline(SpecPipeline(cc, hm, grouped=grouped), x, sc, ce, iw, ih)
for _ in range(args.reps):
    g(*g.static_in)
torch.cuda.synchronize()
