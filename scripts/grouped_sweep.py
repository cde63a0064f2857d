#!/usr/bin/env python
"""Whole step at several batch sizes: two streams + hipGraph (the default) against grouped launches + hipGraph.
usage (GPU box): python scripts/grouped_sweep.py [batch ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spec_amd.pipeline import SpecPipeline, GraphedPipeline  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
cc, hm, _, _ = bench.build_models(dev)
PLAN = os.environ.get('PLAN', 'auto')       # pin the execution plan of both networks: auto | throughput | latency
for m in (cc, hm):
    m.set_plan(PLAN)
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256]:
    x, sc, ce, iw, ih = bench.make_inputs(B, dev, 5)
    row = {'batch': B, 'plan': PLAN}
    for tag, kw in (('two_streams', dict(overlap=True, grouped=False)), ('grouped', dict(grouped=True)), ('one_stream', dict(overlap=False, grouped=False))):
        run = GraphedPipeline(SpecPipeline(cc, hm, **kw), x, sc, ce, iw, ih)
        n = 200 if B <= 16 else 40 if B <= 64 else 20
        for _ in range(5):
            run(x, sc, ce, iw, ih)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            run(x, sc, ce, iw, ih)
        torch.cuda.synchronize()
        row[tag + '_ms'] = round((time.perf_counter() - t0) / n * 1e3, 4)
    print(json.dumps(row), flush=True)
