"""The latency plan (batch <= 16, the reference's own operating point: spec/tester.py:109-151 runs the path at batch =
#detections of a frame, scripts/camcalib_demo.py:95-102 at batch 1): every convolution cut into K slices that run as ONE launch
(gridDim.y = slices), the last slice of a tile to arrive adds the partial tiles in slice order and applies BN / residual / ReLU.

Checked here: each sliced layer against the CPU fp32 convolution and against the throughput kernel; that the in-kernel reduction
is race-free (repeated launches bit-identical, whatever the arrival order); that an image's result does not depend on the batch
within the plan (1 ... 16) nor on grouped / separate launches; that the plan meets the oracle and the throughput plan."""
import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import gpu_models, oracle_models, pinned_plan, rel_err, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'

# (cin, cout, k, stride, hw): the conv shapes of a ResNet-50 trunk at 224^2 that the plan slices
SHAPES = [(256, 64, 1, 1, 56), (64, 64, 3, 1, 56), (256, 128, 1, 1, 56), (128, 128, 3, 2, 56), (512, 128, 1, 1, 28),
          (128, 128, 3, 1, 28), (512, 256, 1, 1, 28), (256, 256, 3, 2, 28), (1024, 256, 1, 1, 14), (256, 256, 3, 1, 14),
          (256, 1024, 1, 1, 14), (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14), (2048, 512, 1, 1, 7), (512, 512, 3, 1, 7),
          (512, 2048, 1, 1, 7)]


def _layer(cin, cout, k, seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5).numpy()
    sc = (1.0 + 0.1 * torch.randn(cout, generator=g)).numpy()
    sh = (0.1 * torch.randn(cout, generator=g)).numpy()
    return w, sc, sh, g


def _ref(x, w, sc, sh, stride, pad, res, relu):
    y = torch.nn.functional.conv2d(x.cpu().permute(0, 3, 1, 2), torch.from_numpy(w), stride=stride, padding=pad)
    y = y * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.cpu()
    return y.relu() if relu else y


@pytest.fixture(scope='module')
def eng():
    from spec_amd.engine import Engine
    e = Engine('camcalib', torch.device(DEV))
    e.set_option('winograd', 0)
    return e


@pytest.mark.parametrize('cin,cout,k,stride,hw', SHAPES)
@pytest.mark.parametrize('B', [1, 3])
def test_sliced_layer_vs_cpu_and_throughput_kernel(eng, cin, cout, k, stride, hw, B):
    w, sc, sh, g = _layer(cin, cout, k, cin * 7 + cout + k)
    pad = k // 2
    x = torch.randn(B, hw, hw, cin, generator=g).relu().to(DEV)
    oh = (hw + 2 * pad - k) // stride + 1
    res = torch.randn(B, oh, oh, cout, generator=g).to(DEV) if k == 1 and cout > cin else None
    ref = _ref(x, w, sc, sh, stride, pad, res, True)
    eng.set_option('conv2d_sk', 0)
    thr = eng.conv2d(x, w, sc, sh, stride, pad, residual=res, relu=True).clone()
    eng.profile(True)
    eng.set_option('conv2d_sk', -1)                 # the plan's own slice count for this shape
    lat = eng.conv2d(x, w, sc, sh, stride, pad, residual=res, relu=True).clone()
    prof = eng.profile_read()
    eng.profile(False)
    eng.set_option('conv2d_sk', 0)
    nch = k * k * cin // 32
    if nch >= 16:
        assert any('splitK' in e['kernel'] for e in prof), prof
    assert rel_err(lat.cpu().numpy(), ref.numpy()) < 2e-5
    assert rel_err(lat.cpu().numpy(), thr.cpu().numpy()) < 5e-6      # same products, another association of the k sum


@pytest.mark.parametrize('S', [2, 3, 6, 9, 18])
def test_every_slice_count_and_ragged_rows(eng, S):
    """M not a multiple of the 64-row tile, Cout not a multiple of the 64-column tile (padding columns), slices that cut
    through filter taps (Cin = 64: 2 chunks per tap, 18 chunks in all)."""
    w, sc, sh, g = _layer(64, 96, 3, 500 + S)
    x = torch.randn(2, 9, 7, 64, generator=g).to(DEV)
    ref = _ref(x, w, sc, sh, 1, 1, None, False)
    eng.set_option('conv2d_sk', S)
    out = eng.conv2d(x, w, sc, sh, 1, 1, relu=False)
    eng.set_option('conv2d_sk', 0)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize('cin,cout,k,stride,hw,B', [(512, 512, 3, 1, 7, 1), (512, 512, 3, 1, 7, 5), (1024, 256, 1, 1, 14, 2), (128, 128, 3, 2, 56, 1),
                                                    (2048, 512, 1, 1, 7, 16), (256, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 1), (128, 128, 3, 1, 28, 3), (512, 2048, 1, 1, 7, 2)])
def test_units_of_the_canonical_tree_are_bit_identical(eng, cin, cout, k, stride, hw, B):
    """How much of a layer's k-sum tree one workgroup computes - a leaf, a group of leaves, or the whole K - is chosen per batch
    size for speed; the association of the sum is the tree's, whoever adds: the three give the same bits."""
    w, sc, sh, g = _layer(cin, cout, k, cin + cout * 3 + k)
    x = torch.randn(B, hw, hw, cin, generator=g).relu().to(DEV)
    eng.set_option('conv2d_sk', -1)
    outs = []
    for unit in (1, 2, 3):
        eng.set_option('latency_force_unit', unit)
        outs.append(eng.conv2d(x, w, sc, sh, stride, k // 2, relu=True).clone())
    eng.set_option('latency_force_unit', 0)
    auto = eng.conv2d(x, w, sc, sh, stride, k // 2, relu=True).clone()
    eng.set_option('conv2d_sk', 0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], auto)
    assert rel_err(auto.cpu().numpy(), _ref(x, w, sc, sh, stride, k // 2, None, True).numpy()) < 2e-5


def test_in_kernel_reduction_is_race_free(eng):
    """layer4.conv2 at batch 1: 8 tiles x 16 slices.  The last slice to arrive differs from launch to launch; the sum must not:
    200 launches, bit-identical (a partial tile read before its writer's data had left the other XCD's L2 would show here)."""
    w, sc, sh, g = _layer(512, 512, 3, 99)
    x = torch.randn(1, 7, 7, 512, generator=g).relu().to(DEV)
    eng.set_option('conv2d_sk', 16)
    first = eng.conv2d(x, w, sc, sh, 1, 1, relu=True).clone()
    bad = 0
    for _ in range(200):
        bad += int(not torch.equal(eng.conv2d(x, w, sc, sh, 1, 1, relu=True), first))
    eng.set_option('conv2d_sk', 0)
    assert bad == 0, bad
    assert rel_err(first.cpu().numpy(), _ref(x, w, sc, sh, 1, 1, None, True).numpy()) < 2e-5


@pytest.fixture(scope='module')
def models():
    return gpu_models(True, True, DEV)


def test_plan_selection_and_launch_count(models):
    """auto: a single trunk takes the single plan up to 2 images (no Winograd anywhere), the latency plan up to 16, throughput
    beyond; one launch per layer in all three.  Sliced layers run on the 64x64 sliced kernel or on the wave-split unit of the same
    tree (round 5) - the small batches on the latter."""
    _, hm = models
    e = hm.engine(torch.device(DEV))
    for B, want in ((1, 'single'), (2, 'single'), (3, 'latency'), (16, 'latency'), (17, 'throughput')):
        x = t(synth.images(5, B)).to(DEV)
        e.profile(True)
        e.trunk(x)
        prof = e.profile_read()
        e.profile(False)
        assert sum(p['launches'] for p in prof) == 2 + 16 * 3
        kern = {p['label']: p['kernel'] for p in prof}
        sliced = [l for l, k in kern.items() if 'splitK' in k or 'wsplit' in k]
        assert (len(sliced) > 25) == (want != 'throughput'), (B, sliced)
        wino = [l for l, k in kern.items() if 'wino' in k]
        if want == 'single':
            assert not wino, (B, wino)
            assert 'wsplit' in kern['backbone.layer4.1.conv2'] and 'wsplit' in kern['backbone.layer3.2.conv1']
        if want == 'latency':   # layer3 / layer4 3x3 convolutions leave Winograd for the sliced direct kernels; layer1 / layer2 keep it
            assert all(s in kern['backbone.layer4.1.conv2'] or 'wsplit' in kern['backbone.layer4.1.conv2'] for s in ('splitK',))
            assert 'wino' in kern['backbone.layer1.1.conv2'] and 'wino' in kern['backbone.layer2.1.conv2']
            assert '2src' in kern['backbone.layer3.0.conv3+downsample'] and 'splitK' in kern['backbone.layer3.0.conv3+downsample']
        if B == 16:             # past the wave-split unit's range: the 64x64 sliced kernel
            assert 'splitK' in kern['backbone.layer4.1.conv2'] and 'splitK' in kern['backbone.layer3.2.conv2']


def test_latency_plan_is_batch_invariant_and_deterministic(models):
    """Within the plan an image's bits do not depend on the batch (1, 2, 5, 16), on grouped vs separate launches, on the
    replay (hipGraph) or on the run."""
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline
    cc, hm = models
    B = 16
    x = t(synth.images(41, B)).to(DEV)
    sc, ce, iw, ih = [t(a).to(DEV) for a in synth.bbox_inputs(41, B, 640., 480.)]
    keys = ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose_6d', 'cam_vfov', 'cam_pitch')
    with pinned_plan('latency', cc, hm):
        grp = SpecPipeline(cc, hm, grouped=True)
        full = {k: v.clone() for k, v in grp(x, sc, ce, iw, ih).items() if k in keys}
        for _ in range(5):
            again = grp(x, sc, ce, iw, ih)
            for k in keys:
                assert torch.equal(again[k], full[k]), k
        two = SpecPipeline(cc, hm, overlap=True, grouped=False)(x, sc, ce, iw, ih)
        for k in keys:
            assert torch.equal(two[k], full[k]), k
        for lo, n in ((0, 1), (3, 2), (7, 5), (15, 1)):
            sl = slice(lo, lo + n)
            out = grp(x[sl].contiguous(), sc[sl].contiguous(), ce[sl].contiguous(), iw[sl].contiguous(), ih[sl].contiguous())
            for k in keys:
                assert torch.equal(out[k], full[k][sl]), (lo, n, k)
        gp = GraphedPipeline(grp, x[:1].contiguous(), sc[:1].contiguous(), ce[:1].contiguous(), iw[:1].contiguous(), ih[:1].contiguous())
        for _ in range(20):
            out = gp(x[:1].contiguous(), sc[:1].contiguous(), ce[:1].contiguous(), iw[:1].contiguous(), ih[:1].contiguous())
            for k in keys:
                assert torch.equal(out[k], full[k][:1]), k
        # the plan may be forced beyond 16 images: still the same bits per image
        big = grp(x.repeat(2, 1, 1, 1), sc.repeat(2), ce.repeat(2, 1), iw.repeat(2), ih.repeat(2))
        for k in keys:
            assert torch.equal(big[k][:B], full[k]) and torch.equal(big[k][B:], full[k]), k


def test_latency_plan_vs_throughput_plan_and_oracle(models):
    from oracle.models import full_pipeline
    from spec_amd.pipeline import SpecPipeline
    cc, hm = models
    occ, ohm = oracle_models(True, True)
    B = 8
    x = t(synth.images(43, B))
    sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(43, B, 640., 480.)]
    ref = full_pipeline(occ, ohm, x, sc, ce, iw, ih)
    dev = [a.to(DEV) for a in (x, sc, ce, iw, ih)]
    outs = {}
    for plan in ('latency', 'throughput'):
        with pinned_plan(plan, cc, hm):
            outs[plan] = {k: v.clone() for k, v in SpecPipeline(cc, hm)(*dev).items()}
    for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'pred_shape', 'pred_cam'):
        for plan in outs:
            assert rel_err(outs[plan][k].cpu().numpy(), ref[k].numpy()) < 1e-4, (plan, k)
        assert rel_err(outs['latency'][k].cpu().numpy(), outs['throughput'][k].cpu().numpy()) < 2e-5, k
    for k in ('cam_vfov', 'cam_pitch', 'cam_roll'):
        assert np.abs(outs['latency'][k].cpu().numpy() - ref[k].numpy()).max() < 2e-5, k


def test_latency_plan_other_resolutions_and_resnet34(models):
    """CamCalib at a non-square size (the slice rule reads the layer's per-image shape) and a BasicBlock trunk."""
    cc, _ = models
    occ, _ = oracle_models(True, True)
    x = t(synth.images(57, 2, 288, 352))
    with pinned_plan('latency', cc):
        lg = cc(x.to(DEV))
        one = cc(x[1:].contiguous().to(DEV))
    for a, b in zip(lg, occ(x)):
        assert rel_err(a.cpu().numpy(), b.numpy()) < 5e-5
    for a, b in zip(lg, one):
        assert torch.equal(a[1:], b)
    from spec_amd.modules import CameraRegressorNetwork
    from spec_amd import assets
    assets.use_synthetic_assets(1003)
    m = CameraRegressorNetwork(backbone='resnet34').to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    for p_ in m.parameters():
        p_.data.copy_(torch.randn(p_.shape, generator=g) * 0.05)
    for n_, buf in m.named_buffers():
        if n_.endswith('running_var'):
            buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    m.commit(torch.device(DEV))
    xs = t(synth.images(3, 3)).to(DEV)
    outs = {}
    for plan in ('latency', 'throughput'):
        with pinned_plan(plan, m):
            outs[plan] = [l.clone() for l in m(xs)]
    for a, b in zip(outs['latency'], outs['throughput']):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-5


def test_fc_heads_small_batch_kernel(models):
    """Latency plan: the three CamCalib heads as ONE launch and the HMR regressor's composed map on the small-batch GEMV kernel
    (one wave per output column) - against the matrix-core GEMMs of the throughput plan, batch-invariant across the kernel's
    blocks of 8 images, and one launch where there were three."""
    cc, hm = models
    dev = torch.device(DEV)
    e = cc.engine(dev)
    feat = torch.randn(11, 7, 7, 2048, generator=torch.Generator().manual_seed(3)).relu().to(DEV)
    outs = {}
    for plan in ('latency', 'throughput'):
        with pinned_plan(plan, cc):
            e.profile(True)
            outs[plan] = [l.clone() for l in e.camcalib_head(feat)]
            prof = e.profile_read()
            e.profile(False)
            fc = [p_ for p_ in prof if p_['label'].startswith('fc_')]
            assert sum(p_['launches'] for p_ in fc) == (1 if plan == 'latency' else 3), prof
            assert ('gemv' in fc[0]['kernel']) == (plan == 'latency')
    for a, b in zip(outs['latency'], outs['throughput']):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 5e-6
    with pinned_plan('latency', cc):
        for lo, n in ((0, 1), (7, 2), (8, 3), (10, 1)):      # across the 8-image blocks of the kernel
            one = e.camcalib_head(feat[lo:lo + n].contiguous())
            for a, b in zip(one, outs['latency']):
                assert torch.equal(a, b[lo:lo + n]), (lo, n)
    # HMR: collapsed regressor and the reference's nine-GEMM loop (residual epilogue) through the same kernel
    eh = hm.engine(dev)
    g = torch.Generator().manual_seed(5)
    f2 = torch.randn(3, 7, 7, 2048, generator=g).relu().to(DEV)
    R = torch.eye(3).expand(3, 3, 3).contiguous().to(DEV)
    K = torch.tensor([[500., 0., 320.], [0., 500., 240.], [0., 0., 0.]]).expand(3, 3, 3).contiguous().to(DEV)
    ih = torch.full((3,), 480.0, device=DEV)
    res = {}
    for plan in ('latency', 'throughput'):
        for collapse in (1, 0):
            eh.set_option('head_collapse', collapse)
            with pinned_plan(plan, hm):
                res[(plan, collapse)] = {k: v.clone() for k, v in eh.hmr_head(f2, R, K, ih).items()}
    eh.set_option('head_collapse', 1)
    for k in ('pred_pose_6d', 'pred_shape', 'pred_cam'):
        ref = res[('throughput', 1)][k].cpu().numpy()
        for key, v in res.items():
            assert rel_err(v[k].cpu().numpy(), ref) < 2e-5, (key, k)


def test_graph_captured_before_workspace_growth_still_replays():
    """A hipGraph has the workspace addresses baked in.  Growing the workspaces afterwards (an eager call at a larger batch) must not
    free what the graph still writes to: outgrown buffers are retired, not released, until the handle is destroyed."""
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline
    cc, hm = gpu_models(True, True, DEV)          # fresh handles: their workspaces start at the size of the first call
    B = 2
    x = t(synth.images(61, 48)).to(DEV)
    sc, ce, iw, ih = [t(a).to(DEV) for a in synth.bbox_inputs(61, 48, 640., 480.)]
    small = [a[:B].contiguous() for a in (x, sc, ce, iw, ih)]
    pipe = SpecPipeline(cc, hm, grouped=True)
    g = GraphedPipeline(pipe, *small)
    ref = {k: v.clone() for k, v in g(*small).items() if k in ('smpl_vertices', 'smpl_joints2d', 'cam_vfov')}
    big = pipe(x, sc, ce, iw, ih)                 # activations, split-K slabs, head rows: everything grows 24-fold
    assert torch.isfinite(big['smpl_vertices']).all()
    junk = [torch.full((64 << 20,), float('nan'), device=DEV) for _ in range(8)]     # would land in freed blocks
    out = g(*small)
    for k, v in ref.items():
        assert torch.equal(out[k], v), k
    del junk


@pytest.mark.parametrize('use_cam,ucf', [(True, True), (False, False)])
def test_fused_head_nodes_bit_identical(use_cam, ucf):
    """Option "head_fuse" (bit 0 / bit 1, default 3): the IEF state columns are written by extra workgroups of the pooling
    launch, and head_final's work (rot6d -> rotmat, the pred_* gather) is done by the SMPL pose kernel - two graph nodes less
    on the small-batch path.  Same arithmetic, same bits as the separate kernels, on both plans and with a ragged batch."""
    from spec_amd import synth
    _, hm = gpu_models(use_cam, ucf, DEV)
    eng = hm.engine(torch.device(DEV))
    for B, plan in ((1, 'latency'), (5, 'latency'), (19, 'throughput')):
        hm.set_plan(plan)
        x = t(synth.images(70 + B, B)).to(DEV)
        args = [x]
        if use_cam:
            g = torch.Generator().manual_seed(B)
            R = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0].contiguous()
            K = torch.zeros(B, 3, 3); K[:, 0, 0] = K[:, 1, 1] = 500 + 200 * torch.rand(B, generator=g); K[:, 0, 2], K[:, 1, 2] = 320., 240.
            sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(5, B, 640., 480.)]
            args += [a.to(DEV) for a in (R, K, sc, ce, iw, ih)]
        outs = []
        for fuse in (0, 1, 2, 3):
            eng.set_option('head_fuse', fuse)
            outs.append({k: v.clone() for k, v in hm(*args).items()})
        eng.set_option('head_fuse', 3)
        for o in outs[1:]:
            assert outs[0].keys() == o.keys()
            for k in outs[0]:
                assert torch.equal(outs[0][k], o[k]), (B, plan, k)
