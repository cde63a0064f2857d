#!/usr/bin/env python
"""Per-kernel time of one whole step at a small batch (library HIP-event profiler, one stream, eager)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cc, hm, _, _ = bench.build_models(dev)
from spec_amd.pipeline import SpecPipeline
from spec_amd import cam_utils
pipe = SpecPipeline(cc, hm, overlap=False, grouped=(os.environ.get('GROUPED', '1') == '1'))
x, sc, ce, iw, ih = bench.make_inputs(args.batch, dev, 1)
for _ in range(3):
    pipe(x, sc, ce, iw, ih)
torch.cuda.synchronize()
engs = [('camcalib', cc._engine), ('spec', hm._engine), ('decode', cam_utils._engine(dev))]
for _, e in engs:
    e.profile(True)
n = 20
for _ in range(n):
    pipe(x, sc, ce, iw, ih)
torch.cuda.synchronize()
rows = []
for tag, e in engs:
    for r in e.profile_read():
        rows.append((r['ms'] / n * 1e3, r['launches'] // n, tag, r['label'], r['kernel']))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'B={args.batch}: {tot:.1f} us of kernel time per step over {sum(r[1] for r in rows)} launches')
for us, nl, tag, lab, k in rows:
    print(f'{us:8.1f} us x{nl} {tag:8s} {lab:40s} {k}')
