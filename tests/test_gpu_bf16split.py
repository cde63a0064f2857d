"""OPTIONAL split-bf16 path (conv_bf16s.hip, option conv_precision = 6 / 3): the plain 1x1 / stride-1 convolutions as sums of
bf16 x bf16 piece products on the bf16 matrix cores.  Never the default and never the benchmark's `value`; what is checked
here is that it keeps the 1e-4 contract on the reference-composed fixtures, and how far it is from the exact fp32 path."""
import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import golden, gpu_models, rel_err, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def eng():
    from spec_amd.engine import Engine
    e = Engine('camcalib', torch.device(DEV))
    yield e
    e.set_option('conv_precision', 0)
    e.close()


# (Cin, Cout, H, W, B, residual, relu): every plain 1x1 shape of the ResNet-50 trunk + ragged rows / masked columns
SHAPES = [(64, 64, 56, 56, 2, False, True), (256, 64, 56, 56, 1, False, True), (64, 256, 56, 56, 1, True, True),
          (256, 128, 28, 28, 2, False, True), (512, 128, 28, 28, 2, False, True), (128, 512, 28, 28, 2, True, True),
          (512, 256, 14, 14, 3, False, True), (1024, 256, 14, 14, 3, False, True), (256, 1024, 14, 14, 3, True, True),
          (1024, 512, 7, 7, 5, False, True), (2048, 512, 7, 7, 5, False, True), (512, 2048, 7, 7, 5, True, True),
          (32, 96, 9, 7, 3, True, False), (96, 160, 5, 11, 2, False, False), (64, 128, 1, 1, 130, False, True)]


@pytest.mark.parametrize('terms,tol', [(6, 3e-6), (3, 1e-4)])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'c%d_%d_%dx%d_b%d_r%d_relu%d' % s)
def test_conv1x1_split_bf16_vs_fp64(eng, shape, terms, tol):
    cin, cout, H, W, B, use_res, relu = shape
    g = torch.Generator().manual_seed(cin + 7 * cout + H)
    x = torch.relu(torch.randn(B, H, W, cin, generator=g)) * 1.3                 # post-ReLU activations, like the trunk's
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, H, W, cout, generator=g) if use_res else None
    ref = torch.einsum('bhwc,oc->bhwo', x.double(), w.double().view(cout, cin)) * sc.double() + sh.double()
    if res is not None:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    eng.set_option('conv_precision', terms)
    eng.profile(True)
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 0, residual=None if res is None else res.to(DEV), relu=relu).cpu()
    kernels = [e['kernel'] for e in eng.profile_read()]
    eng.profile(False)
    eng.set_option('conv_precision', 0)
    y32 = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), 1, 0, residual=None if res is None else res.to(DEV), relu=relu).cpu()
    assert any('bf16split' in k and f'{terms} terms' in k for k in kernels), kernels
    e_split = float((y.double() - ref).abs().max() / ref.abs().max())
    e_fp32 = float((y32.double() - ref).abs().max() / ref.abs().max())
    assert e_split < tol, (e_split, e_fp32)
    if terms == 6:
        assert e_split < 4 * e_fp32 + 1e-7, (e_split, e_fp32)                  # the same class as fp32 accumulation itself


def _models(terms, use_cam=True, use_cam_feats=True):
    cc, hm = gpu_models(use_cam, use_cam_feats, DEV)
    cc.set_conv_precision(terms)
    hm.set_conv_precision(terms)
    return cc, hm


@pytest.mark.parametrize('terms', [6, 3])
@pytest.mark.parametrize('tag,uc,ucf', [('camfeats', True, True), ('cam', True, False), ('nocam', False, False)])
def test_hmr_fixtures_through_split_bf16(tag, uc, ucf, terms):
    """The three fixtures produced by the reference's own spec/models/hmr.py, at the contract's 1e-4, through the bf16 path."""
    g = golden(f'hmr_e2e_{tag}.npz')
    _, hm = _models(terms, uc, ucf)
    B = int(g['batch'])
    x = t(synth.images(int(g['seed_images']), B)).to(DEV)
    if uc:
        out = hm(x, t(g['cam_rotmat']).to(DEV), t(g['cam_intrinsics']).to(DEV), t(g['bbox_scale']).to(DEV),
                 t(g['bbox_center']).to(DEV), t(g['img_w']).to(DEV), t(g['img_h']).to(DEV))
    else:
        out = hm(x)
    eng = hm.engine(torch.device(DEV))
    eng.profile(True)
    hm(x, t(g['cam_rotmat']).to(DEV), t(g['cam_intrinsics']).to(DEV), t(g['bbox_scale']).to(DEV),
       t(g['bbox_center']).to(DEV), t(g['img_w']).to(DEV), t(g['img_h']).to(DEV)) if uc else hm(x)
    n_split = sum(e['launches'] for e in eng.profile_read() if 'bf16split' in e['kernel'])
    eng.profile(False)
    # 16 x conv1 + 16 x conv3 (4 of them with the downsample branch folded in) + the 3x3 layers: with three terms all 16, with six
    # only the 3 stride-2 ones (the fp32 Winograd kernel is faster than six bf16 products on the stride-1 layers)
    assert n_split == (48 if terms == 3 else 35), n_split
    for k in out:
        assert rel_err(out[k].cpu().numpy(), g['out_' + k]) < 1e-4, (k, rel_err(out[k].cpu().numpy(), g['out_' + k]))


@pytest.mark.parametrize('terms', [6, 3])
def test_camcalib_fixture_through_split_bf16(terms):
    g = golden('camcalib_e2e.npz')
    cc, _ = _models(terms)
    lg = cc(t(synth.images(int(g['seed_images']), int(g['batch']))).to(DEV))
    for l, k in zip(lg, ('logits_vfov', 'logits_pitch', 'logits_roll')):
        assert rel_err(l.cpu().numpy(), g[k]) < 1e-4, (k, rel_err(l.cpu().numpy(), g[k]))


def test_default_is_exact_fp32():
    cc, hm = gpu_models(True, True, DEV)
    assert cc.conv_precision == 0 and hm.conv_precision == 0
    x = t(synth.images(5, 2)).to(DEV)
    eng = cc.engine(torch.device(DEV))
    eng.profile(True)
    cc(x)
    ks = [e['kernel'] for e in eng.profile_read()]
    eng.profile(False)
    assert not any('bf16' in k for k in ks)
    with pytest.raises(ValueError):
        cc.set_conv_precision(4)


@pytest.mark.parametrize('terms,tol', [(6, 5e-6), (3, 1e-4)])
def test_trunk_features_split_vs_exact(terms, tol):
    """The whole ResNet-50 trunk (incl. the four conv3 + downsample layers that read a second, strided A source) through the
    bf16 path against the exact fp32 path on the same weights and images."""
    cc, _ = gpu_models(True, True, DEV)
    x = t(synth.images(77, 3)).to(DEV)
    ref = cc.engine(torch.device(DEV)).trunk(x).clone()
    cc2, _ = _models(terms)
    eng = cc2.engine(torch.device(DEV))
    eng.profile(True)
    got = eng.trunk(x)
    ents = eng.profile_read()
    eng.profile(False)
    assert sum(e['launches'] for e in ents if '2src' in e['kernel'] and 'bf16split' in e['kernel']) == 4, ents
    assert not any('conv_igemm' in e['kernel'] and '2src' in e['kernel'] for e in ents), ents
    err = float((got.double() - ref.double()).abs().max() / ref.double().abs().max())
    assert err < tol, err


# (Cin, Cout, k, stride, H, W, B, residual): 3x3 layers (stride 1 and 2, ragged sizes) through the bf16 implicit GEMM
KXK_SHAPES = [(64, 64, 3, 1, 56, 56, 1, False), (128, 128, 3, 2, 56, 56, 1, False), (256, 256, 3, 1, 14, 14, 3, False),
              (512, 512, 3, 2, 14, 14, 3, False), (32, 96, 3, 1, 9, 7, 2, True), (64, 128, 3, 2, 11, 13, 2, False),
              (64, 64, 5, 1, 8, 8, 2, False)]


@pytest.mark.parametrize('terms,tol', [(6, 3e-6), (3, 1e-4)])
@pytest.mark.parametrize('shape', KXK_SHAPES, ids=lambda s: 'c%d_%d_k%d_s%d_%dx%d_b%d_r%d' % s)
def test_conv_kxk_split_bf16_vs_fp64(eng, shape, terms, tol):
    import torch.nn.functional as F
    cin, cout, k, stride, H, W, B, use_res = shape
    pad = k // 2
    g = torch.Generator().manual_seed(cin + 7 * cout + H + k)
    x = torch.relu(torch.randn(B, H, W, cin, generator=g)) * 1.3
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=stride, padding=pad).permute(0, 2, 3, 1) * sc.double() + sh.double()
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res.double()
    ref = torch.relu(ref)
    eng.set_option('conv_precision', terms)
    eng.profile(True)
    y = eng.conv2d(x.to(DEV), w.numpy(), sc.numpy(), sh.numpy(), stride, pad, residual=None if res is None else res.to(DEV), relu=True).cpu()
    kernels = [e['kernel'] for e in eng.profile_read()]
    eng.profile(False)
    eng.set_option('conv_precision', 0)
    assert any('kxk_bf16split' in kk and f'{terms} terms' in kk for kk in kernels), kernels
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err < tol, err
