#!/usr/bin/env python
"""Which earlier activity lowers the overlap of FrameStream's uploads with the captured step?  (run on the GPU box)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spec_amd.pipeline import SpecPipeline, GraphedPipeline
from spec_amd.frames import FrameStream
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
cc, hm, cs, hs = bench.build_models(dev)
B = 256
x, scale, center, img_w, img_h = bench.make_inputs(B, dev, 20210001)
run = GraphedPipeline(SpecPipeline(cc, hm, overlap=True), x, scale, center, img_w, img_h)
Hf, Wf, K = 1080, 1920, 8
F = B // K


def hosts_for(fs):
    g = torch.Generator().manual_seed(77)
    out = []
    for _ in range(2):
        hf, hb, hi = fs.host_buffers()
        hf.copy_(torch.randint(0, 256, hf.shape, dtype=torch.uint8, generator=g))
        hb.copy_(torch.stack([torch.rand(B, generator=g) * Wf, torch.rand(B, generator=g) * Hf, 150 + torch.rand(B, generator=g) * 250,
                              300 + torch.rand(B, generator=g) * 500], 1))
        hi.copy_((torch.arange(B) // K).to(torch.int32))
        out.append((hf, hb, hi))
    return out


def plain(n=10):
    for _ in range(3):
        run(x, scale, center, img_w, img_h)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        run(x, scale, center, img_w, img_h)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def e2e(fs, hosts, n=10):
    for s in range(3):
        fs.submit(*hosts[s % 2])
    fs.drain(); t = time.perf_counter()
    for s in range(n):
        fs.submit(*hosts[s % 2])
    fs.drain()
    return (time.perf_counter() - t) / n * 1e3


fs0 = FrameStream(run, dev, (Hf, Wf), F, B)
h0 = hosts_for(fs0)
print('fresh process:            plain %.2f ms   e2e(old stream) %.2f ms' % (plain(), e2e(fs0, h0)), flush=True)


def report(tag):
    fs1 = FrameStream(run, dev, (Hf, Wf), F, B)
    fs2 = FrameStream(run, dev, (Hf, Wf), F, B, copy_stream=torch.cuda.Stream(device=dev))
    print('%-26s plain %.2f ms   e2e(first FrameStream) %.2f   e2e(new FrameStream, measured stream) %.2f %s   e2e(new FrameStream, next pool stream) %.2f'
          % (tag, plain(), e2e(fs0, h0), e2e(fs1, h0), getattr(fs1, 'copy_probe', None), e2e(fs2, h0)), flush=True)


def small(grouped, b, graph=True, n=20):
    pp = SpecPipeline(cc, hm, grouped=True) if grouped else SpecPipeline(cc, hm, overlap=True, grouped=False)
    a = [t[:b].contiguous() for t in (x, scale, center, img_w, img_h)]
    g = GraphedPipeline(pp, *a) if graph else pp
    ins = g.static_in if graph else a
    for _ in range(n):
        g(*ins)
    torch.cuda.synchronize()
    del g


which = sys.argv[1] if len(sys.argv) > 1 else 'all'
if which in ('all', 'eager2'):
    small(False, 8, graph=False); report('eager two-stream B=8:')
if which in ('all', 'eagerg'):
    small(True, 8, graph=False); report('eager grouped B=8:')
if which in ('all', 'graph2'):
    small(False, 8); report('graph two-stream B=8:')
if which in ('all', 'graphg'):
    small(True, 8); report('graph grouped B=8:')
if which in ('all', 'graphg1'):
    small(True, 1); report('graph grouped B=1:')
