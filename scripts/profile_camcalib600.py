#!/usr/bin/env python
"""CamCalib on ONE full frame at short side 600 (1 x 3 x 600 x 1066: what scripts/camcalib_demo.py runs per image) - per-layer time
in both execution plans (library HIP-event profiler, eager) and the graph-replayed total."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cc, hm, _, _ = bench.build_models(dev)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(F, 3, 600, 1066, device=dev)
tabs = {}
for plan in ('latency', 'throughput'):
    cc.set_plan(plan)
    for _ in range(3):
        cc(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = cc(x)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    cc._engine.profile(True)
    n = 10
    for _ in range(n):
        cc(x)
    torch.cuda.synchronize()
    rows = cc._engine.profile_read()
    cc._engine.profile(False)
    tabs[plan] = {r['label']: (r['ms'] / n * 1e3, r['kernel'], r['flops'] / n) for r in rows}
    print(f'{plan}: graph replay {ms:.3f} ms per call of {F} frame(s); event-timed kernel sum {sum(v[0] for v in tabs[plan].values()):.0f} us')
print(f'{"layer":34s} {"latency us":>11s} {"thru us":>9s}  TF/s(best)  kernels')
for lab in tabs['latency']:
    a, b = tabs['latency'][lab], tabs['throughput'].get(lab, (float('nan'), '', 0))
    best = min(a[0], b[0])
    print(f'{lab:34s} {a[0]:11.1f} {b[0]:9.1f}  {a[2] / best / 1e6:9.1f}  {a[1]} | {b[1]}')
