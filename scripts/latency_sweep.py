#!/usr/bin/env python
"""Small-batch step time (hipGraph replay, grouped launches) over the latency plan's knobs: slices per layer
(latency_target_wgs, latency_min_chunks), where Winograd stays (latency_wino_min_tiles), and the throughput plan beside it."""
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import argparse, itertools, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spec_amd.pipeline import SpecPipeline, GraphedPipeline

ap = argparse.ArgumentParser()
ap.add_argument('--batches', default='1,8')
ap.add_argument('--targets', default='128,256,384,512,768')
ap.add_argument('--minchunks', default='2,4,8')
ap.add_argument('--wino', default='0,128,100000')
ap.add_argument('--fill', default='200')
ap.add_argument('--reps', type=int, default=200)
ap.add_argument('--grouped', default='1', help="1 = grouped launches, 0 = two trunks on two streams, auto = SpecPipeline's own choice")
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cc, hm, _, _ = bench.build_models(dev)


def time_step(b, reps):
    x, sc, ce, iw, ih = bench.make_inputs(b, dev, 7)
    g = GraphedPipeline(SpecPipeline(cc, hm, grouped={'1': True, '0': False}.get(args.grouped, 'auto')), x, sc, ce, iw, ih)
    ins = g.static_in
    for _ in range(10):
        g(*ins)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g(*ins)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    del g
    return best


def setopt(name, v):
    for m in (cc, hm):
        m._engine.set_option(name, v)


rows = []
for b in [int(v) for v in args.batches.split(',')]:
    setopt('plan', 1)
    ms = time_step(b, args.reps)
    rows.append({'batch': b, 'plan': 'throughput', 'ms': round(ms, 4)})
    print(rows[-1], flush=True)
    setopt('plan', 2)
    for tg, mc, wn, fl in itertools.product([int(v) for v in args.targets.split(',')], [int(v) for v in args.minchunks.split(',')],
                                            [int(v) for v in args.wino.split(',')], [int(v) for v in args.fill.split(',')]):
        setopt('latency_target_wgs', tg); setopt('latency_min_chunks', mc); setopt('latency_wino_min_tiles', wn)
        setopt('latency_fill_wgs', fl)
        ms = time_step(b, args.reps)
        rows.append({'batch': b, 'plan': 'latency', 'target_wgs': tg, 'min_chunks': mc, 'wino_min_tiles': wn, 'fill_wgs': fl,
                     'ms': round(ms, 4)})
        print(rows[-1], flush=True)
out = os.path.join(ROOT, 'gpurun_out')
if os.path.isdir(out):
    with open(os.path.join(out, 'latency_sweep.jsonl'), 'w') as f:
        for r in rows:
            f.write(json.dumps(r) + '\n')
