"""Round 5: the wave-split unit of the canonical k-sum tree (conv_wsplit.hip), the 'single' plan (batch 1-2), the persistent
multi-layer walker (conv_persist.hip, opt-in) and the hand-off state hardening.  Reference operating point:
spec/tester.py:109-151 (batch = #detections of a frame), scripts/camcalib_demo.py:95-102 (batch 1)."""
import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import gpu_models, oracle_models, pinned_plan, rel_err, t

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _layer(cin, cout, k, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    return w, sc, sh, g


def _ref(x_nhwc, w, sc, sh, stride, pad, res, relu):
    y = torch.nn.functional.conv2d(x_nhwc.permute(0, 3, 1, 2).double().cpu(), w.double(), stride=stride, padding=pad)
    y = y * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.double().cpu()
    return (y.relu() if relu else y).float()


@pytest.fixture(scope='module')
def eng():
    from spec_amd.engine import Engine
    e = Engine('camcalib', torch.device(DEV))
    e.set_option('winograd', 0)
    return e


# (cin, cout, k, stride, (h, w), B, leaves): ResNet-50 shapes of the plans + ragged rows / partial column tiles / groups of 2 and 3 leaves
WS_SHAPES = [(512, 512, 3, 1, (7, 7), 1, -1), (512, 512, 3, 1, (7, 7), 5, -1), (1024, 256, 1, 1, (14, 14), 2, -1),
             (128, 128, 3, 2, (56, 56), 1, -1), (2048, 512, 1, 1, (7, 7), 3, -1), (512, 2048, 1, 1, (7, 7), 2, -1),
             (64, 64, 3, 1, (56, 56), 1, -1), (128, 128, 3, 1, (28, 28), 2, -1), (64, 96, 3, 1, (9, 7), 2, 6), (64, 96, 3, 1, (9, 7), 2, 2),
             (256, 160, 1, 1, (5, 3), 3, 4), (512, 64, 1, 2, (13, 11), 2, 8)]


@pytest.mark.parametrize('cin,cout,k,stride,hw,B,leaves', WS_SHAPES)
def test_wave_split_unit_is_bit_identical(eng, cin, cout, k, stride, hw, B, leaves):
    """A 32x32 tile per workgroup with a group's leaves on its four waves - all groups in one workgroup, or one group per workgroup
    with 4 KB group slabs - gives exactly the bits of the 64x64 sliced kernel (same leaves, same fold order), with residual + ReLU,
    ragged M, Cout that leaves a partial 32-column tile, stride 2 and groups of 2, 3 and 4 leaves."""
    w, sc, sh, g = _layer(cin, cout, k, cin * 5 + cout + k + B)
    pad = k // 2
    x = torch.randn(B, hw[0], hw[1], cin, generator=g).relu().to(DEV)
    oh, ow = (hw[0] + 2 * pad - k) // stride + 1, (hw[1] + 2 * pad - k) // stride + 1
    res = torch.randn(B, oh, ow, cout, generator=g).to(DEV) if k == 1 else None
    eng.set_option('conv2d_sk', leaves)
    eng.set_option('conv2d_wsplit', 0)
    base = eng.conv2d(x, w, sc, sh, stride, pad, residual=res, relu=True).clone()
    outs = {}
    for unit in (2, 3):
        eng.set_option('conv2d_wsplit', unit)
        eng.profile(True)
        outs[unit] = eng.conv2d(x, w, sc, sh, stride, pad, residual=res, relu=True).clone()
        prof = eng.profile_read()
        eng.profile(False)
        assert any('wsplit' in e['kernel'] for e in prof), prof     # (the shape list only holds trees the unit takes)
    eng.set_option('conv2d_wsplit', 0)
    eng.set_option('conv2d_sk', 0)
    for unit, o in outs.items():
        assert torch.equal(o, base), (unit, float((o - base).abs().max()))
    assert rel_err(base.cpu().numpy(), _ref(x, w, sc, sh, stride, pad, res, True).numpy()) < 2e-5


def test_wave_split_group_slabs_are_race_free(eng):
    """layer4.conv2 at batch 1: 32 tiles x 4 group slabs; the last group to arrive differs from launch to launch, the sum must not."""
    w, sc, sh, g = _layer(512, 512, 3, 77)
    x = torch.randn(1, 7, 7, 512, generator=g).relu().to(DEV)
    eng.set_option('conv2d_sk', 16)
    eng.set_option('conv2d_wsplit', 2)
    first = eng.conv2d(x, w, sc, sh, 1, 1, relu=True).clone()
    bad = sum(int(not torch.equal(eng.conv2d(x, w, sc, sh, 1, 1, relu=True), first)) for _ in range(200))
    eng.set_option('conv2d_wsplit', 0)
    eng.set_option('conv2d_sk', 0)
    assert bad == 0, bad


@pytest.fixture(scope='module')
def models():
    return gpu_models(True, True, DEV)


def _inputs(seed, B):
    x = t(synth.images(seed, B)).to(DEV)
    sc, ce, iw, ih = [t(a).to(DEV) for a in synth.bbox_inputs(seed, B, 640., 480.)]
    return x, sc, ce, iw, ih


KEYS = ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose_6d', 'cam_vfov', 'cam_pitch')


def test_single_plan_meets_oracle_and_is_batch_invariant(models):
    """'single' (auto at batch 1-2): no Winograd, every sliced layer on the wave-split unit.  <= 1e-4 vs the CPU oracle; an image's
    bits do not depend on the batch within the plan (1, 2, forced 5), on grouped vs separate launches or on the graph replay;
    auto == single up to 2 images and == latency from 3."""
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline
    from oracle.models import full_pipeline
    cc, hm = models
    B = 5
    ins = _inputs(53, B)
    occ, ohm = oracle_models(True, True)
    ref = full_pipeline(occ, ohm, *[a.cpu() for a in ins])
    with pinned_plan('single', cc, hm):
        grp = SpecPipeline(cc, hm, grouped=True)
        full = {k: v.clone() for k, v in grp(*ins).items() if k in KEYS}
        for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t'):
            assert rel_err(full[k].cpu().numpy(), ref[k].numpy()) < 1e-4, k
        two = SpecPipeline(cc, hm, overlap=True, grouped=False)(*ins)
        for k in KEYS:
            assert torch.equal(two[k], full[k]), k
        for lo, n in ((0, 1), (1, 2), (4, 1)):
            sl = slice(lo, lo + n)
            out = grp(*[a[sl].contiguous() for a in ins])
            for k in KEYS:
                assert torch.equal(out[k], full[k][sl]), (lo, n, k)
        one = [a[:1].contiguous() for a in ins]
        gp = GraphedPipeline(grp, *one)
        for _ in range(10):
            out = gp(*one)
            for k in KEYS:
                assert torch.equal(out[k], full[k][:1]), k
    with pinned_plan('latency', cc, hm):
        lat = {k: v.clone() for k, v in SpecPipeline(cc, hm, grouped=True)(*ins).items() if k in KEYS}
    with pinned_plan('auto', cc, hm):
        auto = SpecPipeline(cc, hm)
        for n, want in ((1, full), (2, full), (3, lat), (5, lat)):
            out = auto(*[a[:n].contiguous() for a in ins])
            for k in KEYS:
                assert torch.equal(out[k], want[k][:n]), (n, k)


@pytest.mark.parametrize('plan,B', [('single', 1), ('single', 2), ('latency', 3), ('latency', 8)])
def test_persistent_walker_is_bit_identical(models, plan, B):
    """Option persist = 1: every run of implicit-GEMM layers as ONE launch of resident workgroups (completion counters between
    layers, write-through hand-offs).  Same tile body, same tree: the bits of the per-layer launches - pair and single trunk,
    eager and replayed - and a clean control block afterwards."""
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline
    cc, hm = models
    ce, he = cc.engine(torch.device(DEV)), hm.engine(torch.device(DEV))
    ins = _inputs(61, B)
    with pinned_plan(plan, cc, hm):
        try:
            res = {}
            for persist in (0, 1):
                for e in (ce, he):
                    e.set_option('persist', persist)
                    e.set_option('wsplit', 0 if persist else 1)     # the walker runs the 64x64 body; its reference may use either unit
                fa, fb = ce.trunk_pair(he, ins[0], ins[0])
                f1 = he.trunk(ins[0])
                out = SpecPipeline(cc, hm, grouped=True)(*ins)
                torch.cuda.synchronize()
                res[persist] = [fa.clone(), fb.clone(), f1.clone()] + [out[k].clone() for k in KEYS]
            for a, b in zip(res[0], res[1]):
                assert torch.equal(a, b)
            assert torch.equal(res[1][1], res[1][2])                 # pair == single trunk
            assert ce.sync_status() == 0 and he.sync_status() == 0
            gp = GraphedPipeline(SpecPipeline(cc, hm, grouped=True), *ins)
            for _ in range(20):
                out = gp(*ins)
                for i, k in enumerate(KEYS):
                    assert torch.equal(out[k], res[0][3 + i]), k
            assert ce.sync_status() == 0 and he.sync_status() == 0
        finally:
            for e in (ce, he):
                e.set_option('persist', 0)
                e.set_option('wsplit', 1)


def test_poisoned_hand_off_counters_are_reset(models):
    """The split-K tickets and the walker's completion counters must be zero between launches.  A launch that died mid-flight
    leaves them dirty: specmi_sync_reset, specmi_commit and the error path of a forward all zero them - the next forward gives
    the same bits as before."""
    from spec_amd.pipeline import SpecPipeline
    from spec_amd._lib import SpecmiError
    cc, hm = models
    ce, he = cc.engine(torch.device(DEV)), hm.engine(torch.device(DEV))
    ins = _inputs(67, 2)
    with pinned_plan('latency', cc, hm):
        pipe = SpecPipeline(cc, hm, grouped=True)
        for persist in (0, 1):
            for e in (ce, he):
                e.set_option('persist', persist)
            try:
                ref = {k: v.clone() for k, v in pipe(*ins).items() if k in KEYS}
                # 1. explicit reset
                for e in (ce, he):
                    e.debug_poison_sync(0xDEADBEEF)
                    e.sync_reset()
                out = pipe(*ins)
                for k in KEYS:
                    assert torch.equal(out[k], ref[k]), ('reset', persist, k)
                # 2. a forward that fails (image too small: refused before any launch) cleans up behind itself
                for e in (ce, he):
                    e.debug_poison_sync(7)
                    with pytest.raises(SpecmiError):
                        e.trunk(torch.zeros(1, 3, 16, 16, device=DEV))
                out = pipe(*ins)
                for k in KEYS:
                    assert torch.equal(out[k], ref[k]), ('error path', persist, k)
                assert ce.sync_status() == 0 and he.sync_status() == 0
            finally:
                for e in (ce, he):
                    e.set_option('persist', 0)
    # 3. re-commit
    he.debug_poison_sync(3)
    ce.debug_poison_sync(3)
    hm._invalidate(); cc._invalidate()
    hm.commit(torch.device(DEV), freeze=False); cc.commit(torch.device(DEV), freeze=False)
    with pinned_plan('latency', cc, hm):
        out = SpecPipeline(cc, hm, grouped=True)(*ins)
        for k in KEYS:
            assert torch.equal(out[k], ref[k]), ('commit', k)


@pytest.mark.parametrize('B', [1, 2, 3, 7, 10])
def test_fused_tails_are_bit_identical(models, B):
    """Option tail_fuse (opt-in): at small batches each network's tail is ONE launch - CamCalib avg-pool -> three heads -> decode,
    HMR avg-pool + state init -> regressor map -> pose chains - with an in-launch completion counter and a last-arriver epilogue
    (head.hip: tail_gemv_kernel).  Same code as the separate kernels: every output bit, eager and replayed, grouped and two streams."""
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline
    cc, hm = models
    ce, he = cc.engine(torch.device(DEV)), hm.engine(torch.device(DEV))
    ins = _inputs(71 + B, B)
    keys = KEYS + ('pred_pose', 'pred_shape', 'pred_cam', 'cam_roll', 'cam_f_pix', 'cam_rotmat', 'cam_intrinsics')
    try:
        res = {}
        for fuse in (0, 1):
            for e in (ce, he):
                e.set_option('tail_fuse', fuse)
            res[fuse] = {}
            for tag, pp in (('grouped', SpecPipeline(cc, hm, grouped=True)), ('two_streams', SpecPipeline(cc, hm, overlap=True, grouped=False))):
                out = pp(*ins)
                torch.cuda.synchronize()
                res[fuse][tag] = {k: out[k].clone() for k in keys}
        for tag in ('grouped', 'two_streams'):
            for k in keys:
                assert torch.equal(res[0][tag][k], res[1][tag][k]), (tag, k)
        gp = GraphedPipeline(SpecPipeline(cc, hm), *ins)
        for _ in range(20):
            out = gp(*ins)
            for k in keys:
                assert torch.equal(out[k], res[0]['grouped'][k]), k
    finally:
        for e in (ce, he):
            e.set_option('tail_fuse', 0)


def test_bins_lookup_on_the_device_matches_the_numpy_api():
    """bins2centers_device: the bins2* table look-up without a host synchronisation - same indices, same float64 centres."""
    from spec_amd import cam_utils
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(37, 256, generator=g).to(DEV)
    logits[3, 10] = logits[3, 200] = logits[3].max() + 1.0          # a tie: the FIRST maximum wins (np.argmax)
    for centers, fn in ((cam_utils.vfov_bins_centers, cam_utils.bins2vfov), (cam_utils.pitch_bins_centers, cam_utils.bins2pitch),
                        (cam_utils.roll_bins_centers, cam_utils.bins2roll)):
        dev_out = cam_utils.bins2centers_device(logits, centers)
        assert dev_out.dtype == torch.float64 and dev_out.device.type == 'cuda'
        assert np.array_equal(dev_out.cpu().numpy(), fn(logits))


def test_trunk_plan_reports_what_a_forward_takes(models):
    """specmi_trunk_plan: the plan of a (B, H, W) call under the handle's options - auto thresholds (2 / 10 pair / 16 single trunk images'
    worth of pixels), a pinned plan, a full frame counted by its pixels."""
    cc, hm = models
    e = cc.engine(torch.device(DEV))
    with pinned_plan('auto', cc, hm):
        assert [e.trunk_plan(b) for b in (1, 2, 3, 10, 16, 17)] == ['single', 'single', 'latency', 'latency', 'latency', 'throughput']
        assert [e.trunk_plan(b, pair=True) for b in (2, 3, 10, 11)] == ['single', 'latency', 'latency', 'throughput']
        assert e.trunk_plan(1, 600, 1066) == 'latency'           # 12.7 crops' worth of pixels
        assert e.trunk_plan(2, 600, 1066) == 'throughput'
    for plan in ('throughput', 'latency', 'single'):
        with pinned_plan(plan, cc, hm):
            assert e.trunk_plan(1) == plan and e.trunk_plan(256, pair=True) == plan
    from spec_amd.pipeline import SpecPipeline
    with pinned_plan('auto', cc, hm):
        st = SpecPipeline(cc, hm).launch_structure((8, 3, 224, 224))
        assert st == {'grouped': False, 'structure': 'two trunks on two streams', 'plan': 'latency'}
        st = SpecPipeline(cc, hm).launch_structure((1, 3, 224, 224))
        assert st['grouped'] and st['plan'] == 'single'


def test_camcalib_head_decode_equals_the_two_calls(models):
    """specmi_camcalib_head_decode (separate kernels and the fused tail) == specmi_camcalib_head_forward + specmi_camcalib_decode,
    dense and with the angles as strided columns of a record, optional outputs absent."""
    cc, _ = models
    e = cc.engine(torch.device(DEV))
    for B in (1, 5):
        x = t(synth.images(81, B)).to(DEV)
        ih = torch.full((B,), 480., device=DEV); iw = torch.full((B,), 640., device=DEV)
        with pinned_plan('latency', cc):
            feat = e.trunk(x)
            logits = e.camcalib_head(feat)
            ref = e.camcalib_decode(logits[0], logits[1], logits[2], ih, iw)
            for fuse in (0, 1):
                e.set_option('tail_fuse', fuse)
                try:
                    lg, cam = e.camcalib_head_decode(feat, ih, iw)
                    for a, b in zip(lg, logits):
                        assert torch.equal(a, b)
                    for k in ('vfov', 'pitch', 'roll', 'f_pix', 'cam_rotmat', 'cam_intrinsics'):
                        assert torch.equal(cam[k], ref[k]), (B, fuse, k)
                    rec = torch.zeros(B, 7, device=DEV)
                    lg2, cam2 = e.camcalib_head_decode(feat, None, None, angles_out=(rec[:, 1], rec[:, 3], rec[:, 5]))
                    assert cam2['f_pix'] is None and cam2['cam_intrinsics'] is None
                    assert torch.equal(rec[:, 1], ref['vfov']) and torch.equal(rec[:, 3], ref['pitch']) and torch.equal(rec[:, 5], ref['roll'])
                    assert torch.equal(cam2['cam_rotmat'], ref['cam_rotmat']) and float(rec[:, [0, 2, 4, 6]].abs().max()) == 0.0
                finally:
                    e.set_option('tail_fuse', 0)


@pytest.mark.parametrize('num_fc_layers,B', [(2, 1), (2, 5), (3, 2), (2, 40)])
def test_head_decode_with_linear_chains_equals_head_then_decode(num_fc_layers, B):
    """specmi_camcalib_head_decode on a CamCalib model whose heads are Linear CHAINS (num_fc_layers > 1, camcalib/model.py:59-70):
    the decode reads the LAST layer's width (the bins), not the first layer's (num_fc_channels) - ADVICE r05.  Bit-identical to
    head_forward followed by decode, and equal to the oracle's decode of the same logits."""
    from oracle.models import decode_angles
    from spec_amd.modules import CameraRegressorNetwork
    sd = synth.camcalib_state(1500 + num_fc_layers, num_fc_layers=num_fc_layers, num_fc_channels=1024)
    m = CameraRegressorNetwork(num_fc_layers=num_fc_layers, num_fc_channels=1024)
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    eng = m.engine(torch.device(DEV))
    feat = eng.trunk(t(synth.images(600 + B, B)).to(DEV))
    ih = torch.full((B,), 480., device=DEV)
    iw = torch.full((B,), 640., device=DEV)
    lg_a = [l.clone() for l in eng.camcalib_head(feat)]
    dec_a = eng.camcalib_decode(*lg_a, img_h=ih, img_w=iw)
    lg_b, dec_b = eng.camcalib_head_decode(feat, img_h=ih, img_w=iw)
    assert lg_b[0].shape == (B, 256)
    for a, b in zip(lg_a, lg_b):
        assert torch.equal(a, b)
    for k in ('vfov', 'pitch', 'roll', 'f_pix', 'cam_rotmat', 'cam_intrinsics'):
        assert torch.equal(dec_a[k], dec_b[k]), k
    ref = decode_angles(*[l.cpu() for l in lg_a])
    for a, k in zip(ref, ('vfov', 'pitch', 'roll')):
        assert np.abs(dec_b[k].cpu().numpy() - a.numpy()).max() < 2e-6, k
