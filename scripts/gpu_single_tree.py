"""Plan 'single' with / without its own four-leaf trees (option single_tree): parity vs the CPU oracle, batch invariance, per-layer and
whole-step times at batch 1-2.  Output: gpurun_out/single_tree.jsonl / single_tree_layers.txt"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spec_amd import synth, assets
from spec_amd.modules import HMR, CameraRegressorNetwork
from spec_amd.pipeline import SpecPipeline, GraphedPipeline
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
torch.set_grad_enabled(False)
dev = 'cuda:0'
cs, hs = synth.camcalib_state(1001), synth.hmr_state(1002, True)
smpl = assets.use_synthetic_assets(1003)
cc = CameraRegressorNetwork(); cc.load_state_dict({k: t(v) for k, v in cs.items()})
hm = HMR(use_cam=True, use_cam_feats=True); hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
cc = cc.to(dev).eval(); hm = hm.to(dev).eval()
cc.commit(dev, freeze=True); hm.commit(dev, freeze=True)
ce, he = cc.engine(dev), hm.engine(dev)
x = t(synth.images(9, 4)).to(dev)
sc, cen, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(9, 4, 640., 480.)]
def opt(n, v):
    ce.set_option(n, v); he.set_option(n, v)
out = open(os.path.join(ROOT, 'gpurun_out', 'single_tree.jsonl'), 'a')
def emit(**kw):
    line = json.dumps(kw); print(line, flush=True); out.write(line + '\n'); out.flush()
# oracle reference
from oracle import heads
from oracle.models import CamCalibOracle, HMROracle, load_numpy_state, full_pipeline
heads.set_assets(smpl_model=smpl)
occ = load_numpy_state(CamCalibOracle().eval(), cs)
ohm = load_numpy_state(HMROracle(use_cam=True, use_cam_feats=True).eval(), hs)
ref = full_pipeline(occ, ohm, x[:2].cpu(), sc[:2].cpu(), cen[:2].cpu(), iw[:2].cpu(), ih[:2].cpu())
cc.set_plan('single'); hm.set_plan('single')
KEYS = ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t')
for tree in (0, 1):
    opt('single_tree', tree)
    pipe = SpecPipeline(cc, hm, grouped=True)
    o2 = {k: v.clone() for k, v in pipe(x[:2], sc[:2], cen[:2], iw[:2], ih[:2]).items()}
    o1 = pipe(x[1:2], sc[1:2], cen[1:2], iw[1:2], ih[1:2])
    o4 = pipe(x, sc, cen, iw, ih)
    errs = {k: float(np.abs(o2[k].cpu().numpy().astype(np.float64) - ref[k].numpy().astype(np.float64)).max() / np.abs(ref[k].numpy()).max()) for k in KEYS}
    inv = all(torch.equal(o1[k][0], o2[k][1]) and torch.equal(o4[k][:2], o2[k]) for k in KEYS)
    two = SpecPipeline(cc, hm, overlap=True, grouped=False)(x[:2], sc[:2], cen[:2], iw[:2], ih[:2])
    same_struct = all(torch.equal(two[k], o2[k]) for k in KEYS)
    emit(test='parity', single_tree=tree, rel_err_vs_oracle=errs, batch_invariant=bool(inv), grouped_equals_two_streams=bool(same_struct))
def step_ms(pp, b, iters=200):
    g = GraphedPipeline(pp, x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
    ins = g.static_in
    for _ in range(10): g(*ins)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): g(*ins)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    del g
    return round(best, 4)
for b in (1, 2):
    row = {'test': 'timing', 'batch': b}
    for tree in (0, 1, 0, 1):
        opt('single_tree', tree)
        row.setdefault(f'tree{tree}_ms', []).append(step_ms(SpecPipeline(cc, hm, grouped=True), b))
    emit(**row)
pipe = SpecPipeline(cc, hm, overlap=False, grouped=True)
with open(os.path.join(ROOT, 'gpurun_out', 'single_tree_layers.txt'), 'a') as fl:
    for b in (1, 2):
        ins = (x[:b].contiguous(), sc[:b].contiguous(), cen[:b].contiguous(), iw[:b].contiguous(), ih[:b].contiguous())
        table, order = {}, []
        for tree in (0, 1):
            opt('single_tree', tree)
            for _ in range(3): pipe(*ins)
            torch.cuda.synchronize()
            ce.profile(True)
            for _ in range(20): pipe(*ins)
            torch.cuda.synchronize()
            rows = ce.profile_read(); ce.profile(False)
            for r in rows:
                if not r['label'].startswith('backbone.'): continue
                lab = r['label'][9:]
                if lab not in table: table[lab] = {}; order.append(lab)
                table[lab][tree] = (r['ms'] / 20 * 1e3, r['kernel'])
        lines = [f'=== plan single batch {b}: per-layer (us, HIP events, trunk pair): latency-plan trees vs single-plan trees', f'{"layer":28s}{"tree0":>8s}{"tree1":>8s}  kernel(tree1)']
        tot = [0.0, 0.0]
        for lab in order:
            a0, a1 = table[lab].get(0, (float("nan"), ''))[0], table[lab].get(1, (float("nan"), ''))[0]
            tot[0] += a0; tot[1] += a1
            mark = ' *' if abs(a0 - a1) > 0.7 else ''
            lines.append(f'{lab:28s}{a0:8.1f}{a1:8.1f}  {table[lab].get(1, (0, ""))[1]}{mark}')
        lines.append(f'{"total":28s}{tot[0]:8.1f}{tot[1]:8.1f}')
        print('\n'.join(lines), flush=True); fl.write('\n'.join(lines) + '\n')
opt('single_tree', 1)
