#!/usr/bin/env python
"""Does the two-branch hipGraph of the step keep its overlap when it is captured late in a process (many streams already
created)?  And does the stream it is LAUNCHED on matter?  (run on the GPU box)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spec_amd.pipeline import SpecPipeline, GraphedPipeline
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
cc, hm, cs, hs = bench.build_models(dev)
B = 256
ins = bench.make_inputs(B, dev, 20210001)


def t_replay(g, stream=None, n=8):
    cur = torch.cuda.current_stream()
    st = stream or cur
    def once():
        if stream is not None:
            st.wait_stream(cur)
        with torch.cuda.stream(st):
            g.graph.replay()
        if stream is not None:
            cur.wait_stream(st)
    for _ in range(2):
        once()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def make(overlap):
    return GraphedPipeline(SpecPipeline(cc, hm, overlap=overlap, grouped=False), *ins)


g2, g1 = make(True), make(False)
print('fresh process:   two-branch graph %.2f ms   one-stream graph %.2f ms' % (t_replay(g2), t_replay(g1)), flush=True)
for rnd in range(4):
    junk = [torch.cuda.Stream(device=dev) for _ in range(1 + rnd)]          # process state drifts: more streams exist
    a = [t[:8].contiguous() for t in ins]
    gs = GraphedPipeline(SpecPipeline(cc, hm, overlap=True, grouped=False), *a)
    for _ in range(5):
        gs(*gs.static_in)
    torch.cuda.synchronize()
    del gs
    gn = make(True)
    alts = [torch.cuda.Stream(device=dev) for _ in range(4)]
    print('round %d: old two-branch graph %.2f   NEW two-branch graph on the default stream %.2f   on 4 other launch streams %s   one-stream graph %.2f'
          % (rnd, t_replay(g2), t_replay(gn), ['%.2f' % t_replay(gn, s) for s in alts], t_replay(g1)), flush=True)
    del gn
