"""Round-5 GPU check of the persistent multi-layer walker (spec_amd/csrc/conv_persist.hip).

1. bit-equality: the persistent launches must give exactly the bits of the per-layer launches of the same plan (same tile
   body, same canonical k-sum tree) - trunk features of the pair and of a single trunk, and the whole step;
2. hand-off state: specmi_sync_status == 0 after every run;
3. timing: hipGraph replay of the whole step, batch 1..16, persistent on / off, grid size and L2 prefetch variants.

Writes JSON lines to gpurun_out/persist_check.jsonl.   python scripts/gpu_persist_check.py [--quick]
"""
import argparse
import json
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from spec_amd import synth, assets                                   # noqa: E402
from spec_amd.modules import HMR, CameraRegressorNetwork             # noqa: E402
from spec_amd.pipeline import SpecPipeline, GraphedPipeline          # noqa: E402


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'persist_check.jsonl'))
    ap.add_argument('--batches', default='1,2,3,5,8,10')
    ap.add_argument('--time-batches', default='1,2,4,8,10,16')
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    fout = open(args.out, 'a')

    def emit(**kw):
        line = json.dumps(kw)
        print(line, flush=True)
        fout.write(line + '\n')
        fout.flush()

    torch.set_grad_enabled(False)
    dev = 'cuda:0'
    cs, hs = synth.camcalib_state(1001), synth.hmr_state(1002, True)
    assets.use_synthetic_assets(1003)
    cc = CameraRegressorNetwork(); cc.load_state_dict({k: t(v) for k, v in cs.items()})
    hm = HMR(use_cam=True, use_cam_feats=True); hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
    cc = cc.to(dev).eval(); hm = hm.to(dev).eval()
    cc.commit(dev, freeze=True); hm.commit(dev, freeze=True)
    ce, he = cc.engine(dev), hm.engine(dev)
    BMAX = 16
    x = t(synth.images(9, BMAX)).to(dev)
    sc, cen, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(9, BMAX, 640., 480.)]

    def opt(name, v):
        ce.set_option(name, v); he.set_option(name, v)

    def status():
        return ce.sync_status(), he.sync_status()

    # ---- 1. bit-equality -----------------------------------------------------------------------------------
    ok_all = True
    for plan in ('latency', 'single'):
        cc.set_plan(plan); hm.set_plan(plan)
        for b in [int(v) for v in args.batches.split(',')]:
            xb = x[:b].contiguous()
            res = {}
            for persist in (0, 1):
                opt('persist', persist)
                fa, fb = ce.trunk_pair(he, xb, xb)
                f1 = ce.trunk(xb)
                f2 = he.trunk(xb)
                pipe = SpecPipeline(cc, hm, grouped=True)
                out = pipe(xb, sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
                torch.cuda.synchronize()
                res[persist] = (fa.clone(), fb.clone(), f1.clone(), f2.clone(), out['smpl_vertices'].clone(), out['smpl_joints2d'].clone())
            st = status()
            names = ('pair_cam', 'pair_spec', 'single_cam', 'single_spec', 'vertices', 'joints2d')
            eq = {n: bool(torch.equal(a, c)) for n, a, c in zip(names, res[0], res[1])}
            md = {n: float((a - c).abs().max()) for n, a, c in zip(names, res[0], res[1])}
            fin = bool(all(torch.isfinite(v).all() for v in res[1]))
            # the pair and the single-trunk launches of one plan must agree too
            cross = bool(torch.equal(res[1][0], res[1][2]) and torch.equal(res[1][1], res[1][3]))
            good = all(eq.values()) and st == (0, 0) and fin and cross
            ok_all &= good
            emit(test='bit_equal', plan=plan, batch=b, equal=eq, maxdiff=md, sync_status=st, finite=fin, pair_equals_single=cross, ok=good)
    opt('persist', 1)

    # repeated replays of a captured graph stay bit-identical (counters self-clean)
    for plan, b in (('single', 1), ('latency', 8)):
        cc.set_plan(plan); hm.set_plan(plan)
        pp = SpecPipeline(cc, hm, grouped=True)
        g = GraphedPipeline(pp, x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
        ref = {k: v.clone() for k, v in g(*g.static_in).items() if isinstance(v, torch.Tensor)}
        same = True
        for _ in range(50):
            o = g(*g.static_in)
            torch.cuda.synchronize()
            same &= all(torch.equal(o[k], ref[k]) for k in ('smpl_vertices', 'smpl_joints2d', 'cam_vfov'))
        st = status()
        ok_all &= same and st == (0, 0)
        emit(test='replay_stable', plan=plan, batch=b, same=bool(same), sync_status=st)
        del g
    emit(test='summary_correctness', ok=bool(ok_all))

    # ---- 2. timing ---------------------------------------------------------------------------------------------
    def time_step(pp, b, iters=200):
        g = GraphedPipeline(pp, x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
        ins = g.static_in
        for _ in range(10):
            g(*ins)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                g(*ins)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
        del g
        return round(best, 4)

    tb = [int(v) for v in args.time_batches.split(',')]
    if args.quick:
        tb = [1, 8]
    for b in tb:
        row = {'test': 'timing', 'batch': b}
        for plan in (('single', 'latency') if b <= 4 else ('latency',)):
            cc.set_plan(plan); hm.set_plan(plan)
            opt('persist', 0)
            row[f'{plan}_perlayer_grouped'] = time_step(SpecPipeline(cc, hm, grouped=True), b)
            row[f'{plan}_perlayer_2streams'] = time_step(SpecPipeline(cc, hm, overlap=True, grouped=False), b)
            opt('persist', 1)
            for nwg in (256, 512, 768):
                opt('persist_wgs', nwg)
                for pf in (0, 1):
                    opt('persist_l2_prefetch', pf)
                    try:
                        row[f'{plan}_persist_w{nwg}_pf{pf}'] = time_step(SpecPipeline(cc, hm, grouped=True), b)
                    except Exception as e:          # e.g. 768 refused: more than half of the resident slots
                        row[f'{plan}_persist_w{nwg}_pf{pf}'] = repr(e)[:80]
            opt('persist_wgs', 512); opt('persist_l2_prefetch', 0)
            # two trunks on two streams, each its own persistent launch of 256 workgroups
            opt('persist_wgs', 256)
            row[f'{plan}_persist_2streams_w256'] = time_step(SpecPipeline(cc, hm, overlap=True, grouped=False), b)
            opt('persist_wgs', 512)
        row['sync_status'] = status()
        emit(**row)
    cc.set_plan('auto'); hm.set_plan('auto')
    emit(test='done', t=time.time())


if __name__ == '__main__':
    main()
