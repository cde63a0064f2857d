"""Round-5 GPU check of the wave-split unit (spec_amd/csrc/conv_wsplit.hip).

1. bit-equality with the 64x64 sliced kernel (option wsplit = 0) for every unit choice (wsplit = 1 auto, 2 group, 3 all), both
   plans, pair and single trunk, batch 1..10;
2. whole-step time (hipGraph replay) per batch for wsplit 0 / 1 / 2 / 3, grouped and two streams;
3. per-layer table (library HIP-event profiler, grouped eager launches): 64x64 auto unit vs wave-split group vs wave-split all.
Output: gpurun_out/wsplit_check.jsonl, gpurun_out/wsplit_layers.txt"""
import argparse
import json
import os
os.environ.setdefault('SPECMI_EXPERIMENTAL', '1')   # this script sets options of the experimental list (include/specmi.h)
import sys

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
    ap.add_argument('--batches', default='1,2,3,5,8,10')
    ap.add_argument('--time-batches', default='1,2,4,8,10')
    ap.add_argument('--layer-batches', default='1,2,4,8')
    ap.add_argument('--skip-layers', action='store_true')
    args = ap.parse_args()
    outdir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(outdir, exist_ok=True)
    fout = open(os.path.join(outdir, 'wsplit_check.jsonl'), 'a')

    def emit(**kw):
        line = json.dumps(kw)
        print(line, flush=True)
        fout.write(line + '\n'); fout.flush()

    torch.set_grad_enabled(False)
    dev = 'cuda:0'
    cs, hs = synth.camcalib_state(1001), synth.hmr_state(1002, True)
    assets.use_synthetic_assets(1003)
    cc = CameraRegressorNetwork(); cc.load_state_dict({k: t(v) for k, v in cs.items()})
    hm = HMR(use_cam=True, use_cam_feats=True); hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
    cc = cc.to(dev).eval(); hm = hm.to(dev).eval()
    cc.commit(dev, freeze=True); hm.commit(dev, freeze=True)
    ce, he = cc.engine(dev), hm.engine(dev)
    x = t(synth.images(9, 16)).to(dev)
    sc, cen, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(9, 16, 640., 480.)]

    def opt(name, v):
        ce.set_option(name, v); he.set_option(name, v)

    ok_all = True
    for plan in ('latency', 'single'):
        cc.set_plan(plan); hm.set_plan(plan)
        for b in [int(v) for v in args.batches.split(',')]:
            xb = x[:b].contiguous()
            res = {}
            CFG = {'k64': (0, -1), 'auto': (1, -1), 'group': (2, 0), 'all': (3, 0)}
            for tag, (ws, alds) in CFG.items():
                opt('wsplit', ws); opt('wsplit_alds', alds)
                fa, fb = ce.trunk_pair(he, xb, xb)
                f1 = ce.trunk(xb)
                out = SpecPipeline(cc, hm, grouped=True)(xb, sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
                torch.cuda.synchronize()
                res[tag] = (fa.clone(), fb.clone(), f1.clone(), out['smpl_vertices'].clone(), out['smpl_joints2d'].clone())
            names = ('pair_cam', 'pair_spec', 'single_cam', 'vertices', 'joints2d')
            row = {'test': 'bit_equal', 'plan': plan, 'batch': b}
            good = True
            for tag in CFG:
                if tag == 'k64':
                    continue
                eq = {n: bool(torch.equal(a, c)) for n, a, c in zip(names, res['k64'], res[tag])}
                md = max(float((a - c).abs().max()) for a, c in zip(res['k64'], res[tag]))
                row[tag] = {'equal': all(eq.values()), 'maxdiff': md, 'which': [n for n, e in eq.items() if not e]}
                good &= all(eq.values())
            good &= bool(all(torch.isfinite(v).all() for v in res['auto']))
            row['ok'] = good
            ok_all &= good
            emit(**row)
    emit(test='summary_correctness', ok=bool(ok_all))
    opt('wsplit', 1); opt('wsplit_alds', -1)

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

    for b in [int(v) for v in args.time_batches.split(',')]:
        row = {'test': 'timing', 'batch': b}
        for plan in (('single', 'latency') if b <= 4 else ('latency',)):
            cc.set_plan(plan); hm.set_plan(plan)
            for tag, (ws, alds) in {'k64': (0, -1), 'auto': (1, -1), 'group': (2, 0), 'all': (3, 0)}.items():
                opt('wsplit', ws); opt('wsplit_alds', alds)
                row[f'{plan}_{tag}_grouped'] = time_step(SpecPipeline(cc, hm, grouped=True), b)
                row[f'{plan}_{tag}_2streams'] = time_step(SpecPipeline(cc, hm, overlap=True, grouped=False), b)
        opt('wsplit', 1); opt('wsplit_alds', -1)
        emit(**row)

    if not args.skip_layers:
        pipe = SpecPipeline(cc, hm, overlap=False, grouped=True)
        CONFIGS = [('k64', {'wsplit': 0}), ('ws_group', {'wsplit': 2}), ('ws_all', {'wsplit': 3}), ('ws_auto', {'wsplit': 1})]
        with open(os.path.join(outdir, 'wsplit_layers.txt'), 'a') as fl:
            for plan in ('single', 'latency'):
                cc.set_plan(plan); hm.set_plan(plan)
                for b in [int(v) for v in args.layer_batches.split(',')]:
                    if plan == 'single' and b > 4:
                        continue
                    ins = (x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
                    table, order, totals = {}, [], {}
                    for tag, opts in CONFIGS:
                        for k, v in opts.items():
                            opt(k, v)
                        for _ in range(3):
                            pipe(*ins)
                        torch.cuda.synchronize()
                        ce.profile(True)
                        for _ in range(20):
                            pipe(*ins)
                        torch.cuda.synchronize()
                        rows = ce.profile_read()
                        ce.profile(False)
                        tot = 0.0
                        for r in rows:
                            if not r['label'].startswith('backbone.'):
                                continue
                            lab = r['label'][9:]
                            if lab not in table:
                                table[lab] = {}; order.append(lab)
                            table[lab][tag] = r['ms'] / 20 * 1e3
                            tot += r['ms'] / 20 * 1e3
                        totals[tag] = tot
                    hdr = f'=== plan {plan} batch {b}: trunk-pair kernel time per step (us, HIP events): ' + ' '.join(f'{k}={v:.0f}' for k, v in totals.items())
                    lines = [hdr, f'{"layer":28s}' + ''.join(f'{k:>10s}' for k, _ in CONFIGS) + '   best']
                    for lab in order:
                        r = table[lab]
                        best = min((v, k) for k, v in r.items() if k != 'ws_auto')
                        lines.append(f'{lab:28s}' + ''.join(f'{r.get(k, float("nan")):10.1f}' for k, _ in CONFIGS) + f'   {best[1]}')
                    print('\n'.join(lines), flush=True)
                    fl.write('\n'.join(lines) + '\n'); fl.flush()
        opt('wsplit', 1); opt('wsplit_alds', -1)
    cc.set_plan('auto'); hm.set_plan('auto')
    emit(test='done')


if __name__ == '__main__':
    main()
