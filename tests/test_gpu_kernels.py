"""Kernel-level parity on the GPU: every HIP kernel against the CPU oracle / a plain PyTorch-CPU
fp32 reference of the same op, through the C ABI.  Index work is bit-exact, fp32 work within
the stated tolerance (sum order differs from the CPU library's)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import golden, rel_err, smpl_model, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def eng():
    from spec_amd.engine import Engine
    e = Engine('camcalib', torch.device(DEV))
    yield e
    e.set_option('force_conv_variant', 0)
    e.close()


def _conv_ref(x_nhwc, w, sc, sh, stride, pad, res, relu):
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w, stride=stride, padding=pad)
    y = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


# (Cin, Cout, k, stride, H) for every distinct conv shape of the ResNet-50 trunk at 224x224
RESNET_SHAPES = [
    (64, 64, 1, 1, 56), (64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56),
    (256, 128, 1, 1, 56), (128, 128, 3, 2, 56), (128, 512, 1, 1, 28), (256, 512, 1, 2, 56),
    (512, 128, 1, 1, 28), (128, 128, 3, 1, 28), (512, 256, 1, 1, 28), (256, 256, 3, 2, 28),
    (256, 1024, 1, 1, 14), (512, 1024, 1, 2, 28), (1024, 256, 1, 1, 14), (256, 256, 3, 1, 14),
    (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14), (512, 2048, 1, 1, 7), (1024, 2048, 1, 2, 14),
    (2048, 512, 1, 1, 7), (512, 512, 3, 1, 7),
]


# (Cin, Cout, k, stride, H, W, residual): the <= 32-output-channel layers of HRNet-W32's full-resolution branch run on the
# 128x32 tile (tile variant 13, picked automatically); the last two force it on wider layers (several 32-column tiles)
TILE_128x32_SHAPES = [(32, 32, 3, 1, 56, 56, True), (32, 32, 3, 1, 13, 9, False), (64, 32, 1, 1, 28, 28, False),
                      (256, 32, 3, 1, 10, 12, False), (32, 32, 3, 2, 15, 15, False), (64, 96, 1, 1, 14, 14, True),
                      (128, 256, 3, 2, 14, 14, False)]


@pytest.mark.parametrize('shape', TILE_128x32_SHAPES, ids=lambda s: 'c%d_%d_k%d_s%d_%dx%d_r%d' % s)
def test_conv_128x32_tile_parity(eng, shape):
    cin, cout, k, stride, H, W, use_res = shape
    g = torch.Generator().manual_seed(cin * 3 + cout + k + H)
    B = 3
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    pad = 1 if k == 3 else 0
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, oh, ow, cout, generator=g) if use_res else None
    eng.set_option('winograd', 0)
    eng.set_option('force_conv_variant', 13 if cout > 32 else 0)
    eng.profile(True)
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), stride, pad,
                   residual=None if res is None else res.to(DEV), relu=True).cpu()
    kernels = [e['kernel'] for e in eng.profile_read()]
    eng.profile(False)
    eng.set_option('force_conv_variant', 0)
    eng.set_option('winograd', 1)
    assert any('128x32' in kname for kname in kernels), kernels
    ref = _conv_ref(x, w, sc, sh, stride, pad, res, True)
    assert y.shape == ref.shape
    assert rel_err(y.numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize('variant', [0, 1, 2, 3])
@pytest.mark.parametrize('shape', RESNET_SHAPES, ids=lambda s: 'c%d_%d_k%d_s%d_h%d' % s)
def test_conv_layer_parity(eng, shape, variant):
    cin, cout, k, stride, H = shape
    if variant == 1 and cout % 128:
        pytest.skip('128x128 tile needs Cout % 128 == 0')
    g = torch.Generator().manual_seed(cin * 7 + cout + k + H)
    B = 2 if H >= 28 else 3
    x = torch.randn(B, H, H, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    pad = 1 if k == 3 else 0
    oh = (H + 2 * pad - k) // stride + 1
    use_res = (k == 1 and cout >= 4 * cin // 2)
    res = torch.randn(B, oh, oh, cout, generator=g) if use_res else None
    eng.set_option('force_conv_variant', variant)
    eng.set_option('winograd', 1 if variant == 0 else 0)   # variant 0 = what the trunk runs; 1-3 = direct tiles
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), stride, pad,
                   residual=None if res is None else res.to(DEV), relu=True).cpu()
    eng.set_option('force_conv_variant', 0)
    eng.set_option('winograd', 1)
    ref = _conv_ref(x, w, sc, sh, stride, pad, res, True)
    assert y.shape == ref.shape
    err = rel_err(y.numpy(), ref.numpy())
    assert err < 2e-5, f'rel err {err}'


@pytest.mark.parametrize('B,H,W', [(1, 7, 7), (1, 13, 9), (5, 17, 31), (2, 56, 40)])
def test_conv_ragged_sizes_and_tiles(eng, B, H, W):
    """M not a multiple of any tile, odd spatial sizes, no residual, no ReLU, stride 2 with padding."""
    g = torch.Generator().manual_seed(B * 100 + H)
    for (cin, cout, k, stride) in [(32, 64, 3, 2), (96, 192, 1, 1), (64, 128, 3, 1)]:
        x = torch.randn(B, H, W, cin, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) * 0.05
        sc, sh = torch.ones(cout), torch.zeros(cout)
        pad = 1 if k == 3 else 0
        for variant in (0, 2, 3):
            eng.set_option('force_conv_variant', variant)
            y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), stride, pad, relu=False).cpu()
            ref = _conv_ref(x, w, sc, sh, stride, pad, None, False)
            assert rel_err(y.numpy(), ref.numpy()) < 2e-5
    eng.set_option('force_conv_variant', 0)


@pytest.mark.parametrize('B,H,W', [(1, 7, 7), (3, 7, 9), (5, 5, 3), (2, 14, 14), (1, 56, 40), (7, 1, 1), (2, 2, 33)])
@pytest.mark.parametrize('cin,cout,nf', [(16, 128, 16), (48, 384, 8), (128, 128, 0), (32, 64, 0), (64, 192, 0), (512, 128, 0),
                                        (96, 96, 0), (48, 160, 0)])   # Cout % 64 == 32: half of the last co column is masked
def test_conv_winograd_ragged(eng, B, H, W, cin, cout, nf):
    """Fused Winograd F(2x2,3x3) path: odd sizes (half-empty edge tiles), tile count not a multiple of
    the 32-tile workgroup, one and several 16-channel stages, several co blocks, both wave layouts; against the fp32
    direct convolution of PyTorch-CPU AND against this library's own direct implicit GEMM."""
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W + cin)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    for relu in (False, True):
        eng.set_option('winograd', 1)
        eng.set_option('force_wino_variant', nf)   # 16 / 8 frequencies per wave, 0 = the launcher's choice
        y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 1, relu=relu).cpu()
        eng.set_option('force_wino_variant', 0)
        ref = _conv_ref(x, w, sc, sh, 1, 1, None, relu)
        assert y.shape == ref.shape
        assert rel_err(y.numpy(), ref.numpy()) < 2e-5
        if cin % 32 == 0:
            eng.set_option('winograd', 0)
            yd = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 1, relu=relu).cpu()
            eng.set_option('winograd', 1)
            assert rel_err(y.numpy(), yd.numpy()) < 2e-5


@pytest.mark.parametrize('B,H,W', [(1, 7, 7), (3, 9, 5), (2, 14, 14), (1, 56, 40), (5, 1, 3)])
@pytest.mark.parametrize('cin,cout', [(64, 64), (128, 128), (96, 96), (32, 192), (512, 512)])
def test_conv_winograd_residual(eng, B, H, W, cin, cout):
    """BasicBlock tail (ResNet-34, HRNet branches): out = ReLU(BN(conv3x3(x)) + identity) on the Winograd kernel's residual
    epilogue - against PyTorch-CPU fp32 and against this library's direct kernel; ragged tiles, masked co column (96)."""
    g = torch.Generator().manual_seed(B * 977 + H * 31 + W + cin + cout)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, H, W, cout, generator=g)
    for relu in (True, False):
        eng.profile(True)
        y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 1, residual=res.to(DEV), relu=relu).cpu()
        names = [e['kernel'] for e in eng.profile_read()]
        eng.profile(False)
        assert names == ['conv_wino_f32<32t x64,F(2x2,3x3),res>'], names
        ref = _conv_ref(x, w, sc, sh, 1, 1, res, relu)
        assert rel_err(y.numpy(), ref.numpy()) < 2e-5
        if cin % 32 == 0:
            eng.set_option('winograd', 0)
            yd = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 1, residual=res.to(DEV), relu=relu).cpu()
            eng.set_option('winograd', 1)
            assert rel_err(y.numpy(), yd.numpy()) < 2e-5


def test_conv_winograd_kernel_is_used(eng):
    """The profiler names the kernel each launch ran: 3x3/s1 with Cout % 128 == 0 must hit conv_wino."""
    x = torch.randn(1, 8, 8, 32)
    w = torch.randn(128, 32, 3, 3) * 0.05
    eng.profile(True)
    eng.conv2d(x.to(DEV), w.numpy(), np.ones(128, np.float32), np.zeros(128, np.float32), 1, 1, relu=False)
    eng.set_option('winograd', 0)
    eng.conv2d(x.to(DEV), w.numpy(), np.ones(128, np.float32), np.zeros(128, np.float32), 1, 1, relu=False)
    eng.set_option('winograd', 1)
    names = [e['kernel'] for e in eng.profile_read()]
    assert any('conv_wino' in n for n in names) and any('conv_igemm' in n for n in names), names
    # HRNet-W48's 96-channel branch: Winograd with a half-idle last column; 32 channels stay on the direct 128x32 tile
    for cout, want in ((96, 'conv_wino'), (32, 'conv_igemm')):
        w = torch.randn(cout, 32, 3, 3) * 0.05
        eng.conv2d(x.to(DEV), w.numpy(), np.ones(cout, np.float32), np.zeros(cout, np.float32), 1, 1, relu=False)
        names = [e['kernel'] for e in eng.profile_read()]
        assert len(names) == 1 and want in names[0], (cout, names)
    eng.profile(False)


def test_conv_winograd_input_alignment(eng):
    """The Winograd kernel fetches 16 bytes per lane when Cin % 32 == 0 (a pair of 16-channel stages per load) and 8 bytes
    otherwise: an input that is only 8-byte aligned is refused for the former (error, never a wrong read) and served for the latter."""
    from spec_amd._lib import SpecmiError
    g = torch.Generator().manual_seed(77)
    for cin, ok in ((48, True), (64, False)):
        B, H, W, cout = 2, 9, 7, 64
        buf = torch.randn(B * H * W * cin + 2, generator=g).to(DEV)
        x = buf[2:].view(B, H, W, cin)                       # contiguous, 8 bytes past a 256-byte aligned allocation
        assert x.data_ptr() % 16 == 8 and x.is_contiguous()
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        sc, sh = np.ones(cout, np.float32), np.zeros(cout, np.float32)
        if ok:
            y = eng.conv2d(x, w.numpy(), sc, sh, 1, 1, relu=False).cpu()
            ref = _conv_ref(x.cpu(), w, torch.from_numpy(sc), torch.from_numpy(sh), 1, 1, None, False)
            assert rel_err(y.numpy(), ref.numpy()) < 2e-5
        else:
            with pytest.raises(SpecmiError):
                eng.conv2d(x, w.numpy(), sc, sh, 1, 1, relu=False)
            y = eng.conv2d(x.clone(), w.numpy(), sc, sh, 1, 1, relu=False).cpu()     # the handle stays usable
            ref = _conv_ref(x.cpu(), w, torch.from_numpy(sc), torch.from_numpy(sh), 1, 1, None, False)
            assert rel_err(y.numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize('res', [False, True])
def test_conv1x1_wide_n_xcd_column_order(eng, res):
    """512 -> 2048 1x1 with >= 16 tile rows: the launch where every XCD owns four of the 32 tile columns (weight panel = 4 MiB
    = an XCD's whole L2).  Every tile must be visited exactly once: compared with PyTorch-CPU over the whole output."""
    g = torch.Generator().manual_seed(4242)
    B, H, W, cin, cout = 8, 14, 14, 512, 2048            # M = 1568 -> 25 tile rows of 64
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(B, H, W, cout, generator=g) if res else None
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 0, residual=None if r is None else r.to(DEV), relu=True).cpu()
    ref = _conv_ref(x, w, sc, sh, 1, 0, r, True)
    assert y.shape == ref.shape and not torch.isnan(y).any()
    assert rel_err(y.numpy(), ref.numpy()) < 2e-5


def test_conv_identity_asymmetric(eng):
    """A = I style check with an asymmetric weight: catches row/col transposes of the MFMA C layout."""
    cin = cout = 64
    x = torch.zeros(1, 8, 8, cin)
    for p in range(64):
        x[0, p // 8, p % 8, p] = 1.0                 # pixel p carries the unit vector e_p
    w = torch.arange(cout * cin, dtype=torch.float32).reshape(cout, cin, 1, 1) / 100.0
    y = eng.conv2d(x.to(DEV), w.numpy(), np.ones(cout, np.float32), np.zeros(cout, np.float32), 1, 0, relu=False).cpu()
    # y[pixel p][n] = w[n][p]
    assert torch.equal(y.reshape(64, cout), w.reshape(cout, cin).t().contiguous())


@pytest.mark.parametrize('B,H,W', [(2, 224, 224), (1, 97, 130), (1, 600, 450)])
def test_stem_conv_bn_relu(eng, B, H, W):
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    sc = torch.rand(64, generator=g) + 0.5
    sh = torch.randn(64, generator=g) * 0.1
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 2, 3, relu=True, nchw_input=True).cpu()
    ref = F.relu(F.conv2d(x, w, stride=2, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert rel_err(y.numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize('B,H,W,C', [(2, 112, 112, 64), (1, 49, 65, 64), (3, 7, 5, 8)])
def test_maxpool_bit_exact(eng, B, H, W, C):
    x = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(H))
    y = eng.maxpool(x.to(DEV)).cpu()
    ref = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(y, ref.contiguous())


def test_avgpool(eng):
    x = torch.randn(3, 7, 7, 2048, generator=torch.Generator().manual_seed(0))
    y = eng.avgpool(x.to(DEV)).cpu()
    ref = x.mean(dim=(1, 2))
    assert rel_err(y.numpy(), ref.numpy()) < 1e-6


def test_camcalib_decode_vs_reference_fixture(eng):
    from oracle.models import cam_params
    g = golden('camcalib_decode.npz')
    B = g['logits_vfov'].shape[0]
    img_h = torch.full((B,), 480.0)
    img_w = torch.full((B,), 640.0)
    d = eng.camcalib_decode(t(g['logits_vfov']).to(DEV), t(g['logits_pitch']).to(DEV), t(g['logits_roll']).to(DEV),
                            img_h.to(DEV), img_w.to(DEV))
    for k in ('vfov', 'pitch', 'roll'):
        assert np.abs(d[k].cpu().numpy() - g[k]).max() < 2e-6, k
    f_ref = 480.0 / 2.0 / np.tan(g['vfov'].astype(np.float64) / 2.0)
    assert rel_err(d['f_pix'].cpu().numpy(), f_ref) < 2e-6
    R, K = cam_params(g['pitch'], g['roll'], d['f_pix'].cpu().numpy(), img_w, img_h)
    assert np.abs(d['cam_rotmat'].cpu().numpy() - R.numpy()).max() < 2e-6
    Kg = d['cam_intrinsics'].cpu().numpy()
    assert np.array_equal(Kg[:, 2, :], np.zeros((B, 3), np.float32))          # K[2,2] == 0 exactly
    assert np.array_equal(Kg[:, 0, 2], np.full(B, 320.0, np.float32)) and np.array_equal(Kg[:, 1, 2], np.full(B, 240.0, np.float32))
    assert np.array_equal(Kg[:, 0, 0], d['f_pix'].cpu().numpy()) and np.array_equal(Kg[:, 1, 1], Kg[:, 0, 0])


@pytest.fixture(scope='module')
def hmr_engine():
    from spec_amd import synth
    from spec_amd.engine import Engine
    e = Engine('hmr', torch.device(DEV))
    e.load(synth.hmr_state(1002, True), smpl=smpl_model(), use_cam=1, use_cam_feats=1, img_res=224)
    yield e
    e.close()


def _rand_pose(B, seed):
    from oracle.geometry import rot6d_to_rotmat
    g = torch.Generator().manual_seed(seed)
    R = rot6d_to_rotmat(torch.randn(B * 24, 6, generator=g)).view(B, 24, 3, 3)
    betas = torch.randn(B, 10, generator=g)
    cam = torch.stack([0.7 + 0.4 * torch.rand(B, generator=g), 0.2 * torch.randn(B, generator=g),
                       0.2 * torch.randn(B, generator=g)], 1)
    return R, betas, cam


@pytest.mark.parametrize('B', [1, 5, 8, 19])
def test_smpl_lbs_joints_projection(hmr_engine, B):
    from oracle import heads
    from oracle.smpl import smpl_forward_f64
    heads.set_assets(smpl_model=smpl_model())
    head = heads.SMPLCamHead(224)
    R, betas, cam = _rand_pose(B, B)
    g = torch.Generator().manual_seed(B + 50)
    camR = _rand_pose(1, 99)[0][0, :1].expand(B, 3, 3).contiguous()
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 400 + 300 * torch.rand(B, generator=g)
    K[:, 0, 2], K[:, 1, 2] = 320.0, 240.0
    scale = 0.8 + 0.5 * torch.rand(B, generator=g)
    center = torch.stack([250 + 100 * torch.rand(B, generator=g), 200 + 80 * torch.rand(B, generator=g)], 1)
    iw, ih = torch.full((B,), 640.0), torch.full((B,), 480.0)
    ref = head(R, betas, cam, camR, K, scale, center, iw, ih)
    out = hmr_engine.smpl(R.to(DEV), betas.to(DEV), cam.to(DEV), camR.to(DEV), K.to(DEV), scale.to(DEV),
                          center.to(DEV), iw.to(DEV), ih.to(DEV))
    for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t'):
        err = rel_err(out[k].cpu().numpy(), ref[k].numpy())
        assert err < 1e-5, (k, err)
    v64, j64 = smpl_forward_f64(smpl_model(), betas.numpy(), R.numpy())
    assert np.abs(out['smpl_vertices'].cpu().numpy() - v64).max() < 5e-6      # as good as the fp32 oracle
    assert np.abs(out['smpl_joints3d'].cpu().numpy() - j64).max() < 5e-6


def test_smpl_index_paths_bit_exact(hmr_engine):
    """Vertex-picked joints are copies of vertices and joint_map duplicates are identical: exact."""
    from spec_amd import constants as C
    B = 3
    R, betas, cam = _rand_pose(B, 7)
    K = torch.zeros(B, 3, 3); K[:, 0, 0] = K[:, 1, 1] = 500.0
    out = hmr_engine.smpl(R.to(DEV), betas.to(DEV), cam.to(DEV), torch.eye(3).expand(B, 3, 3).contiguous().to(DEV),
                          K.to(DEV), torch.ones(B).to(DEV), torch.zeros(B, 2).to(DEV), torch.ones(B).to(DEV), torch.ones(B).to(DEV))
    v = out['smpl_vertices'].cpu().numpy()
    j = out['smpl_joints3d'].cpu().numpy()
    jm = np.array(C.JOINT_MAP49)
    ids = np.array(C.SMPL_EXTRA_VERTEX_IDS)
    for o, src in enumerate(jm):
        if 24 <= src < 45:
            assert np.array_equal(j[:, o], v[:, ids[src - 24]]), o
    for a in range(49):
        for b in range(a + 1, 49):
            if jm[a] == jm[b]:
                assert np.array_equal(j[:, a], j[:, b])


def test_smpl_zero_pose_kat(hmr_engine):
    m = smpl_model()
    B = 2
    R = torch.eye(3).expand(B, 24, 3, 3).contiguous()
    K = torch.zeros(B, 3, 3); K[:, 0, 0] = K[:, 1, 1] = 500.0
    out = hmr_engine.smpl(R.to(DEV), torch.zeros(B, 10).to(DEV), torch.tensor([[1., 0., 0.]] * B).to(DEV),
                          torch.eye(3).expand(B, 3, 3).contiguous().to(DEV), K.to(DEV), torch.ones(B).to(DEV),
                          torch.zeros(B, 2).to(DEV), torch.zeros(B).to(DEV), torch.zeros(B).to(DEV))
    assert np.abs(out['smpl_vertices'].cpu().numpy() - m['v_template'][None]).max() < 1e-6


@pytest.mark.parametrize('B', [1, 6])
def test_hmr_head_iterative_regressor(hmr_engine, B):
    from oracle import heads
    from oracle.models import load_numpy_state
    from spec_amd import synth
    heads.set_assets(smpl_model=smpl_model())
    hs = synth.hmr_state(1002, True)
    head = heads.HMRHead(2048, use_cam_feats=True)
    head.load_state_dict({k[5:]: t(v) for k, v in hs.items() if k.startswith('head.')})
    head.eval()
    g = torch.Generator().manual_seed(B)
    feat = torch.relu(torch.randn(B, 7, 7, 2048, generator=g))
    R = _rand_pose(B, 3)[0][:, 0].contiguous()
    K = torch.zeros(B, 3, 3); K[:, 0, 0] = K[:, 1, 1] = 300 + 400 * torch.rand(B, generator=g)
    ih = torch.full((B,), 480.0)
    vfov = 2 * torch.atan(ih / (2 * K[:, 0, 0]))
    ref = head(feat.permute(0, 3, 1, 2), cam_rotmat=R, cam_vfov=vfov)
    out = hmr_engine.hmr_head(feat.to(DEV), R.to(DEV), K.to(DEV), ih.to(DEV))
    for k in ('pred_pose_6d', 'pred_shape', 'pred_cam', 'pred_pose'):
        err = rel_err(out[k].cpu().numpy(), ref[k].numpy())
        assert err < 2e-5, (k, err)


def test_conv_batch_split_beyond_2gib(eng):
    """Buffer addressing is 32-bit: an activation tensor >= 2 GiB is processed in batch slices."""
    B, H, cin, cout = 172, 56, 1024, 64           # 172*56*56*1024*4 B = 2.06 GiB
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, H, cin, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    sc, sh = torch.ones(cout), torch.zeros(cout)
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 0, relu=False).cpu()
    idx = [0, 1, 85, 86, 170, 171]                # both sides of the slice boundary
    ref = _conv_ref(x[idx], w, sc, sh, 1, 0, None, False)
    assert rel_err(y[idx].numpy(), ref.numpy()) < 2e-5
    assert torch.isfinite(y).all()


def test_winograd_and_stem_batch_split_beyond_2gib(eng):
    """The Winograd kernel and the MFMA stem address both sides with 32-bit buffer offsets: an OUTPUT
    tensor >= 2 GiB must be processed in batch slices too."""
    # Winograd: 176 x 224 x 224 x 64 x 4 B = 2.10 GiB of output from a 0.53 GiB input
    B, H, cin, cout = 176, 224, 16, 64
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, H, H, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    sc, sh = torch.ones(cout), torch.zeros(cout)
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 1, relu=False)
    idx = [0, 1, 83, 84, 87, 88, 174, 175]
    got = y[idx].cpu()
    assert torch.isfinite(y).all()
    del y
    ref = _conv_ref(x[idx], w, sc, sh, 1, 1, None, False)
    assert rel_err(got.numpy(), ref.numpy()) < 2e-5
    del x
    # stem: 700 x 112 x 112 x 64 x 4 B = 2.09 GiB of output
    B = 700
    xs = torch.randn(B, 3, 224, 224, generator=g)
    ws = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    ys = eng.conv2d(xs.to(DEV), ws.numpy(), sc.numpy(), sh.numpy(), 2, 3, relu=True, nchw_input=True)
    idx = [0, 1, 333, 334, 350, 351, 698, 699]
    got = ys[idx].cpu()
    assert torch.isfinite(ys).all()
    del ys
    ref = F.relu(F.conv2d(xs[idx], ws, stride=2, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert rel_err(got.numpy(), ref.numpy()) < 2e-5


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        k = int(rng.choice([1, 3]))
        stride = int(rng.choice([1, 2]))
        cin = int(rng.choice([32, 64, 96, 160, 256]))
        cout = int(rng.choice([1, 3, 17, 32, 33, 48, 64, 100, 157, 192, 256]))
        H, W = int(rng.integers(1 if k == 1 else 2, 23)), int(rng.integers(1 if k == 1 else 2, 23))
        B = int(rng.integers(1, 5))
        cases.append((B, H, W, cin, cout, k, stride, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return cases


@pytest.mark.parametrize('case', _fuzz_cases(36, 20260926), ids=lambda c: 'b%d_%dx%d_c%d_%d_k%d_s%d_r%d_a%d' % c)
def test_conv_random_shapes(eng, case):
    """Seeded sweep over shapes no network here uses: odd spatial sizes down to 1x1, output channel counts that are not a
    multiple of 4 / 32 / 64 (partial tiles, scalar epilogue path), both strides, with and without residual / ReLU."""
    B, H, W, cin, cout, k, stride, use_res, relu = case
    g = torch.Generator().manual_seed(B * 131 + H * 17 + W + cin + cout)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    pad = 1 if k == 3 else 0
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, oh, ow, cout, generator=g) if use_res else None
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), stride, pad,
                   residual=None if res is None else res.to(DEV), relu=relu).cpu()
    ref = _conv_ref(x, w, sc, sh, stride, pad, res, relu)
    assert y.shape == ref.shape
    assert rel_err(y.numpy(), ref.numpy()) < 2e-5



def test_smpl_batch_variants_bit_identical(hmr_engine):
    """An image's vertices, joints and projection must not depend on the batch it travels in (partially filled 8-image skinning
    tiles: batches 1, 2, 3, 5, 8 against 17)."""
    Bmax = 17
    R, betas, cam = _rand_pose(Bmax, 123)
    g = torch.Generator().manual_seed(9)
    camR = _rand_pose(1, 98)[0][0, :1].expand(Bmax, 3, 3).contiguous()
    K = torch.zeros(Bmax, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 400 + 300 * torch.rand(Bmax, generator=g)
    K[:, 0, 2], K[:, 1, 2] = 320.0, 240.0
    scale = 0.8 + 0.5 * torch.rand(Bmax, generator=g)
    center = torch.stack([250 + 100 * torch.rand(Bmax, generator=g), 200 + 80 * torch.rand(Bmax, generator=g)], 1)
    iw, ih = torch.full((Bmax,), 640.0), torch.full((Bmax,), 480.0)
    args = [R, betas, cam, camR, K, scale, center, iw, ih]
    full = {k: v.clone() for k, v in hmr_engine.smpl(*[a.to(DEV) for a in args]).items()}
    for B in (1, 2, 3, 5, 8):
        out = hmr_engine.smpl(*[a[:B].contiguous().to(DEV) for a in args])
        for k in ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t'):
            assert torch.equal(out[k], full[k][:B]), (B, k)


def test_smpl_skin_split_bit_identical(hmr_engine):
    """The three-waves-per-vertex-group skinning variant (few image tiles) and the one-wave variant give the same bits, on
    both sides of the automatic switch (64 images), with ragged last tiles, and through smpl_native."""
    Bmax = 70
    R, betas, cam = _rand_pose(Bmax, 321)
    g = torch.Generator().manual_seed(10)
    camR = _rand_pose(1, 97)[0][0, :1].expand(Bmax, 3, 3).contiguous()
    K = torch.zeros(Bmax, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 400 + 300 * torch.rand(Bmax, generator=g)
    K[:, 0, 2], K[:, 1, 2] = 320.0, 240.0
    scale = 0.8 + 0.5 * torch.rand(Bmax, generator=g)
    center = torch.stack([250 + 100 * torch.rand(Bmax, generator=g), 200 + 80 * torch.rand(Bmax, generator=g)], 1)
    args = [R, betas, cam, camR, K, scale, center, torch.full((Bmax,), 640.0), torch.full((Bmax,), 480.0)]
    try:
        ref = {}
        for split in (0, 1, -1):
            hmr_engine.set_option('smpl_skin_split', split)
            for B in (1, 7, 33, 64, 70):
                out = hmr_engine.smpl(*[a[:B].contiguous().to(DEV) for a in args])
                v, j24 = hmr_engine.smpl_native(R[:B].to(DEV), betas[:B].to(DEV))
                got = (out['smpl_vertices'].clone(), out['smpl_joints3d'].clone(), v.clone(), j24.clone())
                if split == 0:
                    ref[B] = got
                    assert torch.equal(got[0], got[2])
                else:
                    for a, b in zip(got, ref[B]):
                        assert torch.equal(a, b), (split, B)
    finally:
        hmr_engine.set_option('smpl_skin_split', -1)
