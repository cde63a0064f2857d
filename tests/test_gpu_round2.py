"""GPU parity tests of the round-2 paths: arg-max ('kl' / 'ce' / legacy) CamCalib decode against vectors the
reference's own camcalib/cam_utils.py produced, the packed per-image record written by the kernels in place, the
IEF regressor collapsed into one affine GEMM against the nine-GEMM reference loop and the reference-composed
fixtures, smplx-style native SMPL outputs, and the real RCCL 2-rank gather when two devices are visible."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import golden, gpu_models, rel_err, smpl_model, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def models():
    return gpu_models(True, True, DEV)


# ---- arg-max decode ---------------------------------------------------------------------------------------
def test_bins_argmax_decode_vs_reference_fixture():
    from spec_amd import cam_utils as CU
    g = golden('cam_bins.npz')
    lv, lp, lr = (t(g[k]).to(DEV) for k in ('logits_vfov', 'logits_pitch', 'logits_roll'))
    kv, kp, kr = CU.convert_preds_to_angles(lv, lp, lr)                       # reference default: 'kl', torch
    for a, k in ((kv, 'kl_vfov'), (kp, 'kl_pitch'), (kr, 'kl_roll')):
        assert isinstance(a, torch.Tensor) and a.dtype == torch.float64 and a.device.type == 'cpu'
        assert np.array_equal(a.numpy(), g[k]), k                             # index work + table gather: bit-exact
    nv, npi, nr = CU.convert_preds_to_angles(lv.cpu(), lp.cpu(), lr.cpu(), loss_type='ce', return_type='np')
    assert isinstance(nv, np.ndarray) and np.array_equal(nv, g['kl_vfov']) and np.array_equal(nr, g['kl_roll'])
    assert np.array_equal(CU.bins2horizon(lv), g['horizon'])
    x3 = t(g['logits_vfov'].reshape(4, 16, 256)).to(DEV)                      # argmax over the last axis of a 3-D tensor
    assert np.array_equal(CU.bins2pitch(x3), g['bins3d_pitch'])
    # legacy soft-arg-max branch: vfov / pitch soft, roll through the arg-max table (cam_utils.py:127-133)
    lg = CU.convert_preds_to_angles(lv, lp, lr, loss_type='softargmax_l2', legacy=True)
    # (the +inf row makes the softmax NaN in the reference too: NaN must meet NaN)
    np.testing.assert_allclose(lg[0].cpu().numpy(), g['legacy_vfov'], rtol=0, atol=2e-6, equal_nan=True)
    np.testing.assert_allclose(lg[1].cpu().numpy(), g['legacy_pitch'], rtol=0, atol=2e-6, equal_nan=True)
    assert np.array_equal(np.asarray(lg[2]), g['legacy_roll'])
    sa = CU.get_softargmax(lv)
    np.testing.assert_allclose(sa.cpu().numpy(), g['softargmax'], rtol=0, atol=2e-6, equal_nan=True)
    assert np.isnan(g['softargmax']).sum() == 1


def test_bins_argmax_numpy_semantics_ragged():
    """first maximum, NaN counts as the maximum, any row count / bin count (one wave per row, 4 rows per workgroup)."""
    from spec_amd.cam_utils import _engine
    eng = _engine(torch.device(DEV))
    rng = np.random.default_rng(3)
    for rows, nb in ((1, 2), (3, 5), (7, 64), (9, 65), (130, 256), (5, 1000)):
        x = np.round(rng.normal(size=(rows, nb)) * 2).astype(np.float32)
        if rows > 2:
            x[1, nb // 2] = np.nan
            x[2, :] = -np.inf
        idx, soft = eng.camcalib_bins(t(x).to(DEV), argmax=True, soft=False)
        assert soft is None and idx.dtype == torch.int32
        assert np.array_equal(idx.cpu().numpy(), np.argmax(x, axis=-1)), (rows, nb)


# ---- packed record ----------------------------------------------------------------------------------------
@pytest.mark.parametrize('overlap', [False, True])
def test_packed_record_written_in_place(models, overlap):
    """SpecPipeline(packed=True): every output tensor is a view of ONE (B, 21294) record the kernels wrote directly;
    bit-identical to the dense outputs and to torch.cat of them (the old packing copy)."""
    from spec_amd.pipeline import SpecPipeline, PACKED_KEYS, unpack_outputs, pack_outputs
    cc, hm = models
    B = 5
    x = t(synth.images(31, B)).to(DEV)
    sc, ce, iw, ih = [t(a).to(DEV) for a in synth.bbox_inputs(31, B, 640., 480.)]
    dense = SpecPipeline(cc, hm, overlap=overlap, packed=False)(x, sc, ce, iw, ih)
    out = SpecPipeline(cc, hm, overlap=overlap, packed=True)(x, sc, ce, iw, ih)
    torch.cuda.synchronize()
    rec = out['record']
    assert rec.shape == (B, 21294) and rec.is_contiguous()
    assert pack_outputs(out).data_ptr() == rec.data_ptr()                    # no copy on the gather path
    ref = torch.cat([dense[k].reshape(B, -1) for k, _ in PACKED_KEYS], dim=1)
    assert torch.equal(rec, ref)
    for k, _ in PACKED_KEYS:
        assert out[k].shape == dense[k].shape, k
        assert torch.equal(out[k], dense[k]), k
        lo = rec.data_ptr()
        assert lo <= out[k].data_ptr() < lo + rec.numel() * 4, k              # a view, not a copy
    back = unpack_outputs(rec, 6890)
    assert torch.equal(back['smpl_joints2d'], dense['smpl_joints2d'])
    # a caller-owned record with a larger row stride (e.g. a slice of a bigger buffer)
    big = torch.full((B, 21294 + 10), -7.0, device=DEV)
    out2 = SpecPipeline(cc, hm, overlap=overlap)(x, sc, ce, iw, ih, record=big[:, :21294])
    torch.cuda.synchronize()
    assert torch.equal(big[:, :21294], ref) and bool((big[:, 21294:] == -7.0).all())
    assert torch.equal(out2['smpl_vertices'], dense['smpl_vertices'])


def test_graphed_pipeline_two_buffers(models):
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline
    cc, hm = models
    B = 4
    x = t(synth.images(32, B)).to(DEV)
    sc, ce, iw, ih = [t(a).to(DEV) for a in synth.bbox_inputs(32, B, 640., 480.)]
    pipe = SpecPipeline(cc, hm, overlap=True)
    ref = pipe(x, sc, ce, iw, ih)['record'].clone()
    gp = GraphedPipeline(pipe, x, sc, ce, iw, ih, buffers=2)
    a = gp(x, sc, ce, iw, ih)['record']
    b = gp(x, sc, ce, iw, ih)['record']
    torch.cuda.synchronize()
    assert a.data_ptr() != b.data_ptr()
    assert torch.equal(a, ref) and torch.equal(b, ref)
    x2 = t(synth.images(33, B)).to(DEV)
    c = gp(x2, sc, ce, iw, ih)['record']
    torch.cuda.synchronize()
    assert c.data_ptr() == a.data_ptr() and not torch.equal(c, ref)
    assert torch.equal(c, pipe(x2, sc, ce, iw, ih)['record'])


def test_c4_shape_shards_equal_unsharded(models):
    """BASELINE.json config 4 at its full size on one GPU: a global batch of 2048 cut into the 8 contiguous rank-major
    shards of ``shard_range`` (256 images each, what rank r of an 8-GPU job runs), processed one after the other and
    concatenated like the all-gather does, equals the unsharded forward of all 2048 images bit for bit."""
    from spec_amd.pipeline import SpecPipeline, shard_range
    cc, hm = models
    total, world = 2048, 8
    x16 = t(synth.images(71, 16)).to(DEV)
    b16 = [t(a).to(DEV) for a in synth.bbox_inputs(71, 16, 640., 480.)]
    rep = lambda a: a.repeat(*([total // 16] + [1] * (a.dim() - 1))).contiguous()
    x, (sc, ce, iw, ih) = rep(x16), [rep(a) for a in b16]
    pipe = SpecPipeline(cc, hm, overlap=True)
    shards = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        assert hi - lo == 256
        shards.append(pipe(x[lo:hi], sc[lo:hi], ce[lo:hi], iw[lo:hi], ih[lo:hi])['record'].clone())
    gathered = torch.cat(shards, 0)
    full = pipe(x, sc, ce, iw, ih)['record']
    torch.cuda.synchronize()
    assert gathered.shape == (2048, 21294) and torch.isfinite(full).all()
    assert torch.equal(gathered, full)
    assert torch.equal(full[16:32], full[0:16])                  # the 16 distinct crops repeat
    del shards, gathered, full
    torch.cuda.empty_cache()


# ---- collapsed IEF regressor ------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,use_cam,ucf', [('camfeats', True, True), ('cam', True, False), ('nocam', False, False)])
def test_collapsed_regressor_vs_iterative_and_fixture(tag, use_cam, ucf):
    """The 3 IEF iterations composed in float64 into one affine map (default) against the nine-GEMM loop of the
    reference (head_collapse = 0) and against the fixture the reference's own hmr.py produced."""
    _, hm = gpu_models(use_cam, ucf, DEV)
    g = golden(f'hmr_e2e_{tag}.npz')
    B = int(g['batch'])
    x = t(synth.images(int(g['seed_images']), B)).to(DEV)
    kw = {}
    if use_cam:
        kw = dict(cam_rotmat=t(g['cam_rotmat']).to(DEV), cam_intrinsics=t(g['cam_intrinsics']).to(DEV),
                  bbox_scale=t(g['bbox_scale']).to(DEV), bbox_center=t(g['bbox_center']).to(DEV),
                  img_w=t(g['img_w']).to(DEV), img_h=t(g['img_h']).to(DEV))
    eng = hm.engine(torch.device(DEV))
    eng.profile(True)
    out_c = {k: v.clone() for k, v in hm(x, **kw).items()}
    labels_c = [e['label'] for e in eng.profile_read()]
    eng.set_option('head_collapse', 0)
    out_i = {k: v.clone() for k, v in hm(x, **kw).items()}
    labels_i = [e['label'] for e in eng.profile_read()]
    eng.set_option('head_collapse', 1)
    eng.profile(False)
    assert 'head.ief_collapsed' in labels_c and 'head.fc1' not in labels_c
    assert 'head.fc1' in labels_i and 'head.ief_collapsed' not in labels_i
    for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'pred_cam', 'pred_shape',
              'pred_pose_6d'):
        e_ci = rel_err(out_c[k].cpu().numpy(), out_i[k].cpu().numpy())
        e_cf = rel_err(out_c[k].cpu().numpy(), g['out_' + k])
        assert e_ci < 2e-5, (k, e_ci)
        assert e_cf < 1e-4, (k, e_cf)


def test_collapsed_regressor_trained_like_weights():
    """Random-init decoders are tiny (gain 0.01); with O(1) decoder weights the state feedback (I + Q)^3 matters:
    compare against the oracle's iterative head on the same weights."""
    from oracle import heads
    from oracle.models import HMROracle, load_numpy_state
    from spec_amd import assets
    from spec_amd.modules import HMR
    hs = dict(synth.hmr_state(1002, True))
    rng = np.random.default_rng(11)
    for k in ('head.decpose.weight', 'head.decshape.weight', 'head.deccam.weight'):
        hs[k] = (rng.standard_normal(hs[k].shape) * 0.02).astype(np.float32)
    for k in ('head.fc1.weight', 'head.fc2.weight'):
        hs[k] = (hs[k] * 1.5).astype(np.float32)
    assets.use_synthetic_assets(1003)
    heads.set_assets(smpl_model=smpl_model())
    hm = HMR(use_cam=True, use_cam_feats=True)
    hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
    hm = hm.to(DEV).eval()
    ohm = load_numpy_state(HMROracle(use_cam=True, use_cam_feats=True).eval(), hs)
    B = 3
    x = t(synth.images(41, B))
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(41, B, 640., 480.)]
    from spec_amd.cam_utils import cam_params_from_angles
    R, K = cam_params_from_angles(np.array([0.1, -0.2, 0.3], np.float32), np.array([0.05, 0.1, -0.1], np.float32),
                                  np.array([500., 700., 900.], np.float32), iw, ih)
    ref = ohm(x, cam_rotmat=R.cpu(), cam_intrinsics=K.cpu(), bbox_scale=sc, bbox_center=ce, img_w=iw, img_h=ih)
    out = hm(x.to(DEV), cam_rotmat=R, cam_intrinsics=K, bbox_scale=sc.to(DEV), bbox_center=ce.to(DEV),
             img_w=iw.to(DEV), img_h=ih.to(DEV))
    # the state moved away from init by a visible amount, so the feedback term is exercised
    assert float((ref['pred_pose_6d'] - t(hs['head.init_pose'])).abs().max()) > 0.05
    for k in ('pred_pose_6d', 'pred_shape', 'pred_cam', 'pred_pose', 'smpl_vertices', 'smpl_joints2d'):
        err = rel_err(out[k].cpu().numpy(), ref[k].numpy())
        assert err < 1e-4, (k, err)


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(1, 7, 7, 512, 512), (3, 9, 5, 64, 128), (2, 14, 14, 256, 256), (5, 7, 7, 512, 128)])
def test_winograd_wave_layouts_bit_identical(B, H, W, Cin, Cout):
    """The launcher picks the Winograd wave layout (8 or 16 frequencies per wave) by grid size, i.e. by batch: the two
    layouts must therefore agree bit for bit (same k order, same output-transform association)."""
    from spec_amd.engine import Engine
    eng = Engine('camcalib', torch.device(DEV))
    g = torch.Generator().manual_seed(B * 1000 + Cin)
    x = torch.randn(B, H, W, Cin, generator=g).relu().to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).numpy()
    sc = (1.0 + 0.1 * torch.randn(Cout, generator=g)).numpy()
    sh = (0.1 * torch.randn(Cout, generator=g)).numpy()
    outs = {}
    for v in (8, 16):
        eng.set_option('force_wino_variant', v)
        outs[v] = eng.conv2d(x, w, sc, sh, 1, 1, relu=True).clone()
    eng.set_option('force_wino_variant', 0)
    auto = eng.conv2d(x, w, sc, sh, 1, 1, relu=True)
    assert torch.equal(outs[8], outs[16]) and torch.equal(auto, outs[8])
    ref = torch.nn.functional.conv2d(x.cpu().permute(0, 3, 1, 2), torch.from_numpy(w), padding=1)
    ref = (ref * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1)).relu().permute(0, 2, 3, 1)
    assert rel_err(outs[8].cpu().numpy(), ref.numpy()) < 2e-5


def test_empty_batch_returns_empty_outputs(models):
    """len(dets) == 0 style inputs: empty tensors out, like the reference's torch modules (no launch, no error)."""
    cc, hm = models
    lg = cc(torch.empty(0, 3, 224, 224, device=DEV))
    assert len(lg) == 3 and all(l.shape == (0, 256) for l in lg)
    e = lambda *s: torch.empty(*s, device=DEV)
    out = hm(e(0, 3, 224, 224), cam_rotmat=e(0, 3, 3), cam_intrinsics=e(0, 3, 3), bbox_scale=e(0), bbox_center=e(0, 2), img_w=e(0), img_h=e(0))
    assert out['smpl_vertices'].shape == (0, 6890, 3) and out['pred_pose'].shape == (0, 24, 3, 3)


def test_batch1_latency_path(models):
    """B = 1 through the collapsed head: same results as the row of a larger batch (batch invariance)."""
    from spec_amd.pipeline import SpecPipeline
    cc, hm = models
    x = t(synth.images(35, 4)).to(DEV)
    sc, ce, iw, ih = [t(a).to(DEV) for a in synth.bbox_inputs(35, 4, 640., 480.)]
    pipe = SpecPipeline(cc, hm, overlap=False)
    from tests.util import pinned_plan
    for plan in ('latency', 'single', 'throughput'):      # bit-identity holds WITHIN a plan ('auto' switches at 2 and at 10 images)
        with pinned_plan(plan, cc, hm):
            full = pipe(x, sc, ce, iw, ih)['record'].clone()
            one = pipe(x[2:3], sc[2:3], ce[2:3], iw[2:3], ih[2:3])['record']
            assert torch.equal(one[0], full[2]), plan
    # default plan: rows of different batch sizes may come from different plans - equal to fp32 rounding
    full = pipe(x, sc, ce, iw, ih)['record'].clone()
    one = pipe(x[2:3], sc[2:3], ce[2:3], iw[2:3], ih[2:3])['record']
    assert float((one[0] - full[2]).abs().max()) <= 2e-5 * float(full[2].abs().max())


# ---- second device in one process (per-device kernel attributes) + real RCCL path -------------------------
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs')
def test_second_device_same_process():
    cc0, hm0 = gpu_models(True, True, 'cuda:0')
    cc1, hm1 = gpu_models(True, True, 'cuda:1')
    x = t(synth.images(36, 2))
    a = cc0(x.to('cuda:0'))
    b = cc1(x.to('cuda:1'))
    for u, v in zip(a, b):
        assert torch.equal(u.cpu(), v.cpu())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs')
@pytest.mark.timeout(900)
def test_two_rank_rccl_gather_matches_unsharded():
    """bench-style launch of 2 real ranks over RCCL: the gathered records equal the unsharded forward."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_gather_check.py'), '--gpus', '2'], env=env,
                       capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'RCCL_GATHER_OK' in r.stdout


@pytest.mark.timeout(900)
def test_single_rank_rccl_path():
    """The N > 1 machinery with one rank (always runnable on a 1-GPU box): torch.distributed.run launch, RCCL process
    group, in-place records, graph replay with two record buffers, asynchronous all-gather."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_gather_check.py'), '--gpus', '1'], env=env,
                       capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'RCCL_GATHER_OK world=1' in r.stdout


@pytest.mark.timeout(900)
@pytest.mark.parametrize('payload', ['full', 'joints'])
def test_bench_dist_path_single_rank(payload):
    """bench.py --gpus 1 --dist: the multi-GPU bench flow (self-launch, RCCL, AsyncGather with in-place / joints-only sends,
    persistent receive buffers, comm block with the with / without-collective step times) on one rank."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--dist', '--steps', '3', '--warmup', '1',
                        '--batch', '32', '--no-cpu-baseline', '--no-profile', '--no-c2', '--sustained-seconds', '0.2', '--gather', payload],
                       capture_output=True, text=True, timeout=850, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    comm = line['comm']
    assert line['n_gpus'] == 1 and comm['backend'] == 'nccl' and comm['world_size'] == 1 and comm['payload'] == payload
    assert comm['record_bytes_per_image'] == 85176 and comm['sent_bytes_per_image'] == (85176 if payload == 'full' else 2496)
    assert comm['ms_per_step_without_gather'] > 0 and 0.3 < comm['overlap_efficiency'] < 1.5 and line['value'] > 0
    assert line['config']['launch'] == 'hipGraph replay'


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'refusing' in r.stderr and not r.stdout.strip()
