"""Parity where released checkpoints stress it: trunks with BatchNorm variances over six decades, zero /
negative / loud gammas, dead filters, Student-t(3) filters and calibrated running statistics, O(1)-gain decoders, inputs with
saturated regions (``spec_amd.synth``, ``stats='pretrained_like'``).  The reference's own fp32 arithmetic is 2e-5 away from the
float64 result at these statistics (``profiles/r06_*_parity_report.txt``), so next to BASELINE.json's 1e-4 against the CPU oracle the
GPU path is measured against a FLOAT64 oracle and must not be less accurate than the CPU fp32 oracle is - channel by channel of the
layer-4 map (a channel whose folded scale is 10^3 below its neighbours' is invisible to a tensor max-norm), for every execution
plan and every trunk structure (Winograd on / off, downsample branch folded into conv3 or launched separately).

Reference call sites that load such checkpoints: spec/tester.py:63-71, scripts/camcalib_demo.py:74-81, spec/models/hmr.py:124-136."""
import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import (PL_SEED_IMG, cpu_threads, golden, per_channel_errors, pinned_plan, pl_gpu_models, pl_oracle_models, rel_err, smpl_model, t)

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'
TOL = 1e-4            # BASELINE.json: within 1e-4 relative fp32 of the reference CPU path
# Channel by channel, "the CPU fp32 oracle's error" e_cpu is the larger of TWO runs of the same oracle that differ only in summation
# order (NCHW on all cores / channels_last on one thread).  A channel's max error over 98 positions is ONE draw of a heavy-tailed
# random variable: the second CPU run alone exceeds 2 x the first on 13-17 of the 2048 channels (median ratio 1.03, p99 1.95,
# max 3.0-3.2; header of profiles/r06_*_parity_report.txt), so "<= 2 x e_cpu on every channel" is a bar the reference's own
# arithmetic does not clear against itself.  What is asserted instead, per plan and structure:
CHANNEL_MEDIAN = 1.25  # median over channels of e_gpu / e_cpu: no systematic loss of accuracy (measured 0.83-1.07)
CHANNEL_P99 = 2.5      # 99th percentile (measured 1.6-2.1; CPU against CPU 1.95)
CHANNEL_HARD = 8.0     # EVERY channel: e_gpu <= 8 x max(e_cpu, the CPU's median relative error x the channel's maximum) (measured max
                       # 3.8 x e_cpu on ResNet-50; a mis-folded scale or a dropped term is 10^2-10^4) ...
CHANNEL_FLOOR = 4.0    # ... or <= 4 fp32 ulps of the channel's largest value (channels the CPU happens to get exactly right)
PLANS = ['single', 'latency', 'throughput']
STRUCTURES = [(1, 1), (0, 1), (1, 0), (0, 0)]     # (winograd, fuse_downsample)


def build_pl():
    """GPU modules, CPU fp32 oracles, and the float64 / fp32 oracle trunk maps of two saturated crops (NHWC)."""
    cc, hm = pl_gpu_models(DEV)
    occ, ohm = pl_oracle_models()
    _, ohm64 = pl_oracle_models(double=True)
    x = t(synth.images(PL_SEED_IMG, 2, saturate=True))
    with cpu_threads():
        f64 = ohm64.backbone(x.double()).permute(0, 2, 3, 1).contiguous()
    f32 = ohm.backbone(x).permute(0, 2, 3, 1).contiguous()
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        f32b = ohm.backbone(x.contiguous(memory_format=torch.channels_last)).permute(0, 2, 3, 1).contiguous()
    finally:
        torch.set_num_threads(nt)
    return {'cc': cc, 'hm': hm, 'occ': occ, 'ohm': ohm, 'ohm64': ohm64, 'x': x, 'f64': f64.numpy(), 'f32': f32.numpy(), 'f32b': f32b.numpy()}


@pytest.fixture(scope='module')
def pl():
    return build_pl()


def channel_report(feat, f32s, f64):
    """Per-channel comparison of a GPU layer-4 map with the float64 oracle, next to the CPU fp32 oracle's own error (``f32s``:
    the oracle's runs).  -> dict(e_gpu, e_cpu, bound, worst = max e_gpu / bound, n_over)."""
    e_gpu = per_channel_errors(feat, f64)
    e_cpu = np.max([per_channel_errors(f, f64) for f in f32s], axis=0)
    cmax = np.abs(f64).reshape(-1, f64.shape[-1]).max(axis=0)
    # a channel the CPU happens to get right to half an ulp (its value rides on the identity path, or every term is exact) is not
    # held against the GPU: the CPU error of a channel counts for at least the CPU's MEDIAN relative error x the channel's maximum
    live = cmax > 0
    med_rel = float(np.median(e_cpu[live] / cmax[live])) if live.any() else 0.0
    bound = np.maximum(CHANNEL_HARD * np.maximum(e_cpu, med_rel * cmax), CHANNEL_FLOOR * np.spacing(cmax.astype(np.float32)).astype(np.float64))
    ratio = e_gpu / np.maximum(bound, 1e-300)
    ratio[(e_gpu == 0)] = 0.0
    ok = e_cpu > 0
    q = e_gpu[ok] / e_cpu[ok]
    return {'e_gpu': e_gpu, 'e_cpu': e_cpu, 'cmax': cmax, 'bound': bound, 'worst': float(ratio.max()), 'n_over': int((ratio > 1).sum()),
            'argworst': int(ratio.argmax()), 'median': float(np.median(q)), 'p99': float(np.quantile(q, 0.99)), 'max': float(q.max()),
            'n_over2': int((q > 2).sum())}


def gpu_trunk(hm, x, plan, wino, fuse):
    eng = hm.engine(torch.device(DEV))
    eng.set_option('winograd', wino)
    eng.set_option('fuse_downsample', fuse)
    try:
        with pinned_plan(plan, hm):
            return eng.trunk(x.to(DEV)).cpu().numpy()
    finally:
        eng.set_option('winograd', 1)
        eng.set_option('fuse_downsample', 1)


def test_statistics_are_pretrained_like(pl):
    """The stand-in checkpoint has what the test is about: BN variances over >= 5 decades, exact-zero and negative gammas, dead
    filters, a layer-4 map whose channel maxima span decades, all-zero channels, activations beyond 10."""
    sd = {k: v for k, v in pl['hm'].state_dict().items() if k.startswith('backbone.')}
    var = torch.cat([v.flatten() for k, v in sd.items() if k.endswith('running_var')])
    gam = torch.cat([v.flatten() for k, v in sd.items() if k.endswith('.weight') and v.dim() == 1])
    assert var.max() / var.min() > 1e5
    assert 0.03 < float((gam == 0).float().mean()) < 0.08 and 0.05 < float((gam < 0).float().mean()) < 0.2
    assert any(bool((v.flatten(1).abs().sum(1) == 0).any()) for k, v in sd.items() if v.dim() == 4)
    cmax = np.abs(pl['f64']).reshape(-1, 2048).max(axis=0)
    assert cmax.max() > 10 and (cmax == 0).sum() >= 1 and cmax[cmax > 0].min() < 1e-2 * cmax.max()


@pytest.mark.parametrize('wino,fuse', STRUCTURES)
@pytest.mark.parametrize('plan', PLANS)
def test_trunk_per_channel_vs_float64(pl, plan, wino, fuse):
    feat = gpu_trunk(pl['hm'], pl['x'], plan, wino, fuse)
    assert np.isfinite(feat).all()
    rep = channel_report(feat, (pl['f32'], pl['f32b']), pl['f64'])
    c = rep['argworst']
    assert rep['n_over'] == 0, (plan, wino, fuse, rep['n_over'], rep['worst'], c, rep['e_gpu'][c], rep['e_cpu'][c], rep['cmax'][c])
    assert rep['median'] <= CHANNEL_MEDIAN and rep['p99'] <= CHANNEL_P99, (plan, wino, fuse, rep['median'], rep['p99'], rep['max'])
    # and the tensor-wise reading against the CPU oracle
    assert rel_err(feat, pl['f32']) < TOL


def _wmpjpe_mm(verts_a, verts_b):
    """spec/utils/compute_error.py:33-49,184: J_regressor @ vertices, pelvis aligned, mean L2 (mm)."""
    J = smpl_model()['J_regressor'].astype(np.float64)
    ja, jb = np.einsum('jv,bvc->bjc', J, verts_a), np.einsum('jv,bvc->bjc', J, verts_b)
    return float(np.sqrt((((ja - ja[:, :1]) - (jb - jb[:, :1])) ** 2).sum(-1)).mean() * 1000.0)


@pytest.mark.parametrize('B', [1, 8])
def test_whole_path_vs_oracle(pl, B):
    """CamCalib -> decode -> SPEC -> SMPL -> projection on the stand-in checkpoints against the CPU oracle: every output within
    1e-4 (max-norm), delta W-MPJPE <= 0.1 mm, under the plan 'auto' picks and under both pinned batch plans."""
    from oracle.models import full_pipeline
    from spec_amd.pipeline import SpecPipeline
    cc, hm = pl['cc'], pl['hm']
    x = t(synth.images(PL_SEED_IMG + B, B, saturate=True))
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(PL_SEED_IMG + B, B, 640., 480.)]
    ref = full_pipeline(pl['occ'], pl['ohm'], x, sc, ce, iw, ih)
    for plan in ('auto', 'latency', 'throughput'):
        with pinned_plan(plan, cc, hm):
            out = SpecPipeline(cc, hm)(x.to(DEV), sc.to(DEV), ce.to(DEV), iw.to(DEV), ih.to(DEV))
        for k in ('cam_vfov', 'cam_pitch', 'cam_roll'):
            assert np.abs(out[k].cpu().numpy() - ref[k].numpy()).max() < 2e-5, (plan, k)
        for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'pred_shape', 'pred_cam', 'pred_pose_6d'):
            err = rel_err(out[k].cpu().numpy(), ref[k].numpy())
            assert err < TOL, (plan, k, err)
        d = _wmpjpe_mm(out['smpl_vertices'].cpu().numpy().astype(np.float64), ref['smpl_vertices'].numpy().astype(np.float64))
        assert d < 0.1, f'{plan}: delta W-MPJPE {d} mm'
    if B == 8:      # the reference's nine-GEMM regressor loop instead of the float64-composed map (option head_collapse = 0)
        eng = hm.engine(torch.device(DEV))
        eng.set_option('head_collapse', 0)
        try:
            out = SpecPipeline(cc, hm)(x.to(DEV), sc.to(DEV), ce.to(DEV), iw.to(DEV), ih.to(DEV))
        finally:
            eng.set_option('head_collapse', 1)
        for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_pose', 'pred_shape', 'pred_cam'):
            assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < TOL, ('loop', k)


@pytest.mark.parametrize('plan', PLANS)
def test_mesh_vs_float64(pl, plan):
    """HMR.forward on the two crops of the trunk test against a float64 regressor head + SMPL run from the float64 trunk map.
    The head's outputs are linear in the features, so the RMS error over the 288 regressed pose numbers is a stable statistic:
    the GPU's is at most 1.5 x the CPU fp32 oracle's (3 x over the 20 shape numbers: few samples).  Mesh, joints and rotations (one ill-conditioned joint decides their
    max-norm: a single draw) stay within 4 x the CPU's max-norm distance, and within 1e-4 of the CPU oracle itself."""
    from oracle.models import cam_params
    hm, ohm, ohm64, x = pl['hm'], pl['ohm'], pl['ohm64'], pl['x']
    B = x.shape[0]
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(PL_SEED_IMG, B, 640., 480.)]
    R, K = cam_params(t(np.array([-0.4, 0.25], np.float32)), t(np.array([0.15, -0.2], np.float32)), np.array([520., 610.]), iw, ih)
    ref32 = ohm(x, R, K, sc, ce, iw, ih)
    vfov64 = 2 * torch.atan(ih.double() / (2 * K[:, 0, 0].double()))
    h64 = ohm64.head(t(pl['f64']).permute(0, 3, 1, 2), cam_rotmat=R.double(), cam_vfov=vfov64)      # (the float64 trunk map of the fixture)
    v64, j64 = ohm64.smpl.smpl(h64['pred_shape'], h64['pred_pose'])
    with pinned_plan(plan, hm):
        out = hm(x.to(DEV), R.to(DEV), K.to(DEV), sc.to(DEV), ce.to(DEV), iw.to(DEV), ih.to(DEV))
    rms = lambda a, b: float((a.double() - b).pow(2).mean().sqrt())
    for k, factor in (('pred_pose_6d', 1.5), ('pred_shape', 3.0)):
        e_gpu, e_cpu = rms(out[k].cpu(), h64[k]), rms(ref32[k], h64[k])
        assert e_gpu <= factor * e_cpu, (plan, k, e_gpu, e_cpu)
    for k, r64 in (('smpl_vertices', v64), ('smpl_joints3d', j64), ('pred_pose', h64['pred_pose'])):
        e_gpu = float((out[k].cpu().double() - r64).abs().max())
        e_cpu = float((ref32[k].double() - r64).abs().max())
        assert e_gpu <= 4.0 * e_cpu, (plan, k, e_gpu, e_cpu)
        assert rel_err(out[k].cpu().numpy(), ref32[k].numpy()) < TOL, (plan, k)


@pytest.mark.parametrize('plan', PLANS)
def test_reference_composed_fixture(pl, plan):
    """The same stand-in checkpoints through the reference's OWN spec/models/hmr.py and camcalib/model.py
    (tests/golden/make_fixtures.py, B = 2): the composition pin covers these statistics too."""
    cc, hm = pl['cc'], pl['hm']
    g = golden('camcalib_e2e_pl.npz')
    x = t(synth.images(int(g['seed_images']), int(g['batch']), saturate=True)).to(DEV)
    with pinned_plan(plan, cc):
        lg = cc(x)
    for l, k in zip(lg, ('logits_vfov', 'logits_pitch', 'logits_roll')):
        assert rel_err(l.cpu().numpy(), g[k]) < TOL, (plan, k)
    from spec_amd.cam_utils import convert_preds_to_angles
    for a, k in zip(convert_preds_to_angles(*lg, loss_type='softargmax_biased_l2'), ('vfov', 'pitch', 'roll')):
        assert np.abs(a.cpu().numpy() - g[k]).max() < 2e-5, (plan, k)
    g = golden('hmr_e2e_pl.npz')
    x = t(synth.images(int(g['seed_images']), int(g['batch']), saturate=True)).to(DEV)
    with pinned_plan(plan, hm):
        out = hm(x, t(g['cam_rotmat']).to(DEV), t(g['cam_intrinsics']).to(DEV), t(g['bbox_scale']).to(DEV),
                 t(g['bbox_center']).to(DEV), t(g['img_w']).to(DEV), t(g['img_h']).to(DEV))
    assert sorted(out.keys()) == sorted(g['out_keys'])
    for k in out:
        err = rel_err(out[k].cpu().numpy(), g[f'out_{k}'])
        assert err < TOL, (plan, k, err)


# ---- one layer at a time: folded BatchNorm scales over five decades, zero / negative scales, loud channels --------------------------
# (Cin, Cout, k, stride, H): the distinct convolution shapes of a ResNet-50 trunk at 224 x 224 (tests/test_gpu_kernels.py)
from tests.test_gpu_kernels import RESNET_SHAPES  # noqa: E402

LAYER_PATHS = {'throughput': dict(winograd=1, conv2d_sk=0, conv2d_wsplit=0),      # Winograd on the 3x3 stride-1 shapes, 64x64 / 128x128 tiles
               'direct': dict(winograd=0, conv2d_sk=0, conv2d_wsplit=0),
               'latency': dict(winograd=0, conv2d_sk=-1, conv2d_wsplit=0),         # the sliced 64x64 kernel with the plan's own tree
               'wave_split': dict(winograd=0, conv2d_sk=-1, conv2d_wsplit=2)}      # the 32x32 wave-split unit of the same tree


@pytest.fixture(scope='module')
def conv_eng():
    from spec_amd.engine import Engine
    e = Engine('camcalib', torch.device(DEV))
    yield e
    e.close()


# the second 3x3 convolution of a BasicBlock (ResNet-18 / 34, camcalib/config.py:81): stride 1 with the block input as residual
BASIC_RES_SHAPES = [(64, 64, 3, 1, 56), (128, 128, 3, 1, 28), (256, 256, 3, 1, 14), (512, 512, 3, 1, 7)]


@pytest.mark.parametrize('path', list(LAYER_PATHS))
@pytest.mark.parametrize('shape', RESNET_SHAPES + [s_ + ('res',) for s_ in BASIC_RES_SHAPES], ids=lambda s: 'c%d_%d_k%d_s%d_h%d' % s[:5] + ('_res' if len(s) > 5 else ''))
def test_conv_layer_per_channel_with_wide_scales(conv_eng, shape, path):
    """A single fused conv + BN (+ residual) launch with released-checkpoint-like folds: scale = gamma / sigma over five decades with
    exact zeros and negatives, shifts that cancel a large pre-BN mean, a few all-zero filters, heavy-tailed filters and post-ReLU-like
    inputs with outliers.  Every OUTPUT CHANNEL against a float64 reference, next to a CPU fp32 evaluation of the same formula: a channel
    whose folded scale is 10^3 below its neighbours' cannot hide behind the tensor's max-norm."""
    force_res = len(shape) > 5
    cin, cout, k, stride, H = shape[:5]
    g = torch.Generator().manual_seed(cin * 11 + cout * 3 + k + H + (7 if force_res else 0))
    B = 2 if H >= 28 else 3
    x = torch.relu(torch.randn(B, H, H, cin, generator=g) * 1.5 + 0.5)
    x = x * torch.where(torch.rand(cin, generator=g) < 0.02, 10.0, 1.0)                      # a few loud input channels
    tdist = torch.randn(cout, cin, k, k, generator=g) / torch.sqrt((torch.randn(3, cout, cin, k, k, generator=g) ** 2).mean(0).clamp_min(1e-3))
    w = tdist.clamp(-12, 12) * (1.0 / (3 * cin * k * k)) ** 0.5
    w[torch.rand(cout, generator=g) < 0.01] = 0.0                                            # dead filters
    sigma = torch.exp2(torch.randint(-8, 9, (cout,), generator=g).float()) * (1 + torch.rand(cout, generator=g))    # 2^-8 .. 2^9
    gamma = 0.5 + 0.4 * torch.randn(cout, generator=g)
    gamma[torch.rand(cout, generator=g) < 0.05] = 0.0
    sc = gamma / sigma
    sh = 0.5 * torch.randn(cout, generator=g) - (torch.randn(cout, generator=g) * 2.0) * sc
    pad = 1 if k == 3 else 0
    oh = (H + 2 * pad - k) // stride + 1
    use_res = force_res or (k == 1 and cout >= 2 * cin)
    res = torch.relu(torch.randn(B, oh, oh, cout, generator=g) * 2.0) if use_res else None
    relu = True if force_res else bool((cin + cout + k) % 2)

    def ref(dtype):
        y = torch.nn.functional.conv2d(x.to(dtype).permute(0, 3, 1, 2), w.to(dtype), stride=stride, padding=pad)
        y = (y * sc.to(dtype).view(1, -1, 1, 1) + sh.to(dtype).view(1, -1, 1, 1)).permute(0, 2, 3, 1)
        if res is not None:
            y = y + res.to(dtype)
        return (torch.relu(y) if relu else y).contiguous()

    r64, r32 = ref(torch.float64).numpy(), ref(torch.float32).numpy()
    opts = LAYER_PATHS[path]
    for n_, v_ in opts.items():
        conv_eng.set_option(n_, v_)
    try:
        y = conv_eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), stride, pad, residual=None if res is None else res.to(DEV),
                            relu=relu).cpu().numpy()
    finally:
        for n_, v_ in (('winograd', 1), ('conv2d_sk', 0), ('conv2d_wsplit', 0)):
            conv_eng.set_option(n_, v_)
    assert y.shape == r64.shape and np.isfinite(y).all()
    e_gpu, e_cpu = per_channel_errors(y, r64), per_channel_errors(r32, r64)
    # the magnitude a channel's rounding errors scale with: its largest |term sum| before the ReLU clips it
    pre = np.abs(r64).reshape(-1, cout).max(axis=0) + np.abs(sh.numpy().astype(np.float64))
    bound = np.maximum(8.0 * e_cpu, 64 * np.spacing(pre.astype(np.float32)).astype(np.float64))
    over = e_gpu > bound
    c = int(np.argmax(e_gpu / np.maximum(bound, 1e-300)))
    assert not over.any(), (path, shape, int(over.sum()), c, e_gpu[c], e_cpu[c], pre[c], float(sc[c]))
    ok = e_cpu > 0
    assert np.median(e_gpu[ok] / e_cpu[ok]) <= 2.0, (path, shape, float(np.median(e_gpu[ok] / e_cpu[ok])))


@pytest.fixture(scope='module')
def pl34():
    """CameraRegressorNetwork(backbone='resnet34') - the default of camcalib/config.py:81 - with released-checkpoint-like statistics:
    BasicBlock trunks take other fused paths (Winograd on both 3x3 convolutions of a block, the second through its residual epilogue)."""
    from oracle.models import CamCalibOracle, load_numpy_state
    from spec_amd.modules import CameraRegressorNetwork
    sd = synth.camcalib_state(2134, backbone='resnet34', stats='pretrained_like')
    ref = load_numpy_state(CamCalibOracle('resnet34').eval(), sd)
    ref64 = load_numpy_state(CamCalibOracle('resnet34').eval(), sd).double()
    m = CameraRegressorNetwork(backbone='resnet34')
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=True)
    x = t(synth.images(PL_SEED_IMG + 34, 2, saturate=True))
    with cpu_threads():
        f64 = ref64.backbone(x.double()).permute(0, 2, 3, 1).contiguous().numpy()
    f32 = ref.backbone(x).permute(0, 2, 3, 1).contiguous().numpy()
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        f32b = ref.backbone(x.contiguous(memory_format=torch.channels_last)).permute(0, 2, 3, 1).contiguous().numpy()
    finally:
        torch.set_num_threads(nt)
    return {'m': m.to(DEV).eval(), 'ref': ref, 'x': x, 'f64': f64, 'f32': f32, 'f32b': f32b}


@pytest.mark.parametrize('wino', [1, 0])
@pytest.mark.parametrize('plan', PLANS)
def test_resnet34_trunk_per_channel_vs_float64(pl34, plan, wino):
    m, x = pl34['m'], pl34['x']
    eng = m.engine(torch.device(DEV))
    eng.set_option('winograd', wino)
    try:
        with pinned_plan(plan, m):
            feat = eng.trunk(x.to(DEV)).cpu().numpy()
            logits = [l.cpu().numpy() for l in m(x.to(DEV))]
    finally:
        eng.set_option('winograd', 1)
    assert feat.shape == pl34['f64'].shape == (2, 7, 7, 512) and np.isfinite(feat).all()
    rep = channel_report(feat, (pl34['f32'], pl34['f32b']), pl34['f64'])
    c = rep['argworst']
    assert rep['n_over'] == 0, (plan, wino, rep['n_over'], rep['worst'], c, rep['e_gpu'][c], rep['e_cpu'][c], rep['cmax'][c])
    assert rep['median'] <= CHANNEL_MEDIAN and rep['p99'] <= CHANNEL_P99 * 1.2, (plan, wino, rep['median'], rep['p99'], rep['max'])   # (512 channels: p99 = 5 channels)
    for a, b in zip(logits, pl34['ref'](x)):
        assert rel_err(a, b.numpy()) < TOL, (plan, wino)


def test_camcalib_full_frame_vs_oracle(pl):
    """The demo's operating point for CamCalib (scripts/camcalib_demo.py:95-129: one full frame at short side 600, batch 1) with the
    stand-in checkpoint - a resolution the running statistics were NOT calibrated at: logits within 1e-4 of the CPU oracle, decoded
    angles within 2e-5 rad, under the plan 'auto' picks and under the throughput plan."""
    from spec_amd.cam_utils import convert_preds_to_angles
    from oracle.models import decode_angles
    cc, occ = pl['cc'], pl['occ']
    x = t(synth.images(PL_SEED_IMG + 600, 1, 600, 1066, saturate=True))
    want = occ(x)
    wang = decode_angles(*want)
    for plan in ('auto', 'throughput'):
        with pinned_plan(plan, cc):
            got = cc(x.to(DEV))
        for a, b in zip(got, want):
            assert rel_err(a.cpu().numpy(), b.numpy()) < TOL, plan
        for a, b in zip(convert_preds_to_angles(*got, loss_type='softargmax_l2'), wang):
            assert np.abs(a.cpu().numpy() - b.numpy()).max() < 2e-5, plan


@pytest.mark.parametrize('use_cam,ucf', [(True, False), (False, False)], ids=['cam', 'nocam'])
def test_other_head_variants_vs_oracle(use_cam, ucf):
    """The two other constructor variants of HMR (spec/models/hmr.py:66-74: SMPLCamHead without camera features, and the non-camera
    SMPLHead) on the same stand-in trunk: every output within 1e-4 of the CPU oracle, three plans, B = 2."""
    from oracle import heads
    from oracle.models import HMROracle, cam_params, load_numpy_state
    from spec_amd import assets
    from spec_amd.modules import HMR
    from tests.util import PL_CAM_GAIN, PL_DEC_GAIN, PL_SEED_HM
    assets.use_synthetic_assets(1003)
    heads.set_assets(smpl_model=smpl_model())
    sd = synth.hmr_state(PL_SEED_HM, ucf, dec_gain=PL_DEC_GAIN, cam_gain=PL_CAM_GAIN, stats='pretrained_like')
    ref = load_numpy_state(HMROracle(use_cam=use_cam, use_cam_feats=ucf).eval(), sd)
    m = HMR(use_cam=use_cam, use_cam_feats=ucf)
    missing, unexpected = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith('smpl.') for k in missing)
    m = m.to(DEV).eval()
    B = 2
    x = t(synth.images(PL_SEED_IMG + 50, B, saturate=True))
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(PL_SEED_IMG + 50, B, 640., 480.)]
    R, K = cam_params(t(np.array([-0.3, 0.2], np.float32)), t(np.array([0.1, -0.15], np.float32)), np.array([480., 640.]), iw, ih)
    want = ref(x, R, K, sc, ce, iw, ih) if use_cam else ref(x)
    for plan in PLANS:
        with pinned_plan(plan, m):
            out = m(x.to(DEV), R.to(DEV), K.to(DEV), sc.to(DEV), ce.to(DEV), iw.to(DEV), ih.to(DEV)) if use_cam else m(x.to(DEV))
        assert sorted(out.keys()) == sorted(want.keys())
        for k in want:
            assert rel_err(out[k].cpu().numpy(), want[k].numpy()) < TOL, (plan, k)


@pytest.mark.parametrize('backbone', ['hrnet_w48-interp'])       # (W48: the padded 48-wide branch; 'hrnet_w32-conv' passes too - 15 s more)
def test_hrnet_trunk_per_channel_vs_float64(backbone):
    """HRNet-W32 / W48 (spec/models/hmr.py:44-51) with released-checkpoint-like statistics (hook-calibrated on the oracle trunk,
    tests/util.py): exchange-unit sums of branches whose scales differ by decades, the padded 48-wide branch, both head modes.  The
    480 / 720-channel map per channel against a float64 oracle like the ResNet trunks, and HMR end to end within 1e-4 of the oracle."""
    from oracle import heads
    from oracle.models import HMROracle, cam_params, load_numpy_state
    from spec_amd import assets
    from spec_amd.modules import HMR
    from tests.util import PL_CAM_GAIN, PL_DEC_GAIN, pretrained_like_by_hooks
    assets.use_synthetic_assets(1003)
    heads.set_assets(smpl_model=smpl_model())
    sd = synth.hmr_state(2148, True, dec_gain=PL_DEC_GAIN, cam_gain=PL_CAM_GAIN, backbone=backbone, with_trunk=False)      # head only
    ref = HMROracle(backbone=backbone, use_cam=True, use_cam_feats=True).eval()
    trunk_sd = pretrained_like_by_hooks(ref.backbone, 2148, t(synth.images(2148 + 7919, 4, saturate=True)))
    sd.update({'backbone.' + k: v for k, v in trunk_sd.items()})
    sd['head.init_pose'] = synth._orthonormal_rot6d(2148, 'head.init_pose', 24).reshape(1, 144)
    load_numpy_state(ref, sd)
    ref64 = load_numpy_state(HMROracle(backbone=backbone, use_cam=True, use_cam_feats=True).eval(), sd).double()
    m = HMR(backbone=backbone, use_cam=True, use_cam_feats=True)
    missing, unexpected = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith('smpl.') for k in missing), (missing, unexpected)
    m = m.to(DEV).eval()
    B = 2
    x = t(synth.images(PL_SEED_IMG + 32, B, saturate=True))
    with cpu_threads():
        f64 = ref64.backbone(x.double()).permute(0, 2, 3, 1).contiguous().numpy()
    f32 = ref.backbone(x).permute(0, 2, 3, 1).contiguous().numpy()
    f32b = ref.backbone(x.contiguous(memory_format=torch.channels_last)).permute(0, 2, 3, 1).contiguous().numpy()
    var = np.concatenate([v.ravel() for k, v in trunk_sd.items() if k.endswith('running_var')])
    assert var.max() / var.min() > 1e5 and np.abs(f64).max() > 5
    feat = m.engine(torch.device(DEV)).trunk(x.to(DEV)).cpu().numpy()
    assert feat.shape == f64.shape and np.isfinite(feat).all()
    rep = channel_report(feat, (f32, f32b), f64)
    c = rep['argworst']
    assert rep['n_over'] == 0, (backbone, rep['n_over'], rep['worst'], c, rep['e_gpu'][c], rep['e_cpu'][c], rep['cmax'][c])
    assert rep['median'] <= CHANNEL_MEDIAN * 1.2 and rep['p99'] <= CHANNEL_P99 * 1.2, (backbone, rep['median'], rep['p99'], rep['max'])
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(PL_SEED_IMG + 32, B, 640., 480.)]
    R, K = cam_params(t(np.array([-0.3, 0.2], np.float32)), t(np.array([0.1, -0.15], np.float32)), np.array([480., 640.]), iw, ih)
    want = ref(x, R, K, sc, ce, iw, ih)
    out = m(x.to(DEV), R.to(DEV), K.to(DEV), sc.to(DEV), ce.to(DEV), iw.to(DEV), ih.to(DEV))
    for k in want:
        assert rel_err(out[k].cpu().numpy(), want[k].numpy()) < TOL, (backbone, k)
