#!/usr/bin/env python
"""Per-layer time (library HIP-event profiler, grouped launches of both trunks, eager) of every alternative the latency plan
can choose from, per batch size: throughput kernels, and the sliced kernel with a leaf / a group / the whole K per workgroup,
Winograd kept or not.  Prints one table per batch: rows = layers, columns = alternatives."""
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spec_amd.pipeline import SpecPipeline

ap = argparse.ArgumentParser()
ap.add_argument('--batches', default='1,2,4,8,16')
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cc, hm, _, _ = bench.build_models(dev)
pipe = SpecPipeline(cc, hm, overlap=False, grouped=True)
CONFIGS = [('thru', {'plan': 1}),
           ('leaf', {'plan': 2, 'latency_force_unit': 1, 'latency_wino_min_tiles': 100000}),
           ('group', {'plan': 2, 'latency_force_unit': 2, 'latency_wino_min_tiles': 100000}),
           ('all', {'plan': 2, 'latency_force_unit': 3, 'latency_wino_min_tiles': 100000}),
           ('auto', {'plan': 2, 'latency_force_unit': 0, 'latency_wino_min_tiles': 128})]


def setopts(d):
    for m in (cc, hm):
        for k, v in d.items():
            m._engine.set_option(k, v)


for b in [int(v) for v in args.batches.split(',')]:
    x, sc, ce, iw, ih = bench.make_inputs(b, dev, 1)
    table, order, totals = {}, [], {}
    for tag, opts in CONFIGS:
        setopts(opts)
        for _ in range(3):
            pipe(x, sc, ce, iw, ih)
        torch.cuda.synchronize()
        cc._engine.profile(True)
        for _ in range(args.iters):
            pipe(x, sc, ce, iw, ih)
        torch.cuda.synchronize()
        rows = cc._engine.profile_read()
        cc._engine.profile(False)
        tot = 0.0
        for r in rows:
            if not r['label'].startswith('backbone.'):
                continue
            lab = r['label'][9:]
            if lab not in table:
                table[lab] = {}
                order.append(lab)
            table[lab][tag] = r['ms'] / args.iters * 1e3
            tot += r['ms'] / args.iters * 1e3
        totals[tag] = tot
    print(f'=== batch {b}: trunk-pair kernel time per step (us): ' + ' '.join(f'{t}={v:.0f}' for t, v in totals.items()))
    print(f'{"layer":28s}' + ''.join(f'{t:>9s}' for t, _ in CONFIGS) + '   best')
    for lab in order:
        r = table[lab]
        best = min((v, k) for k, v in r.items() if k != 'auto')
        print(f'{lab:28s}' + ''.join(f'{r.get(t, float("nan")):9.1f}' for t, _ in CONFIGS) + f'   {best[1]}')
