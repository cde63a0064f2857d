"""The oracle's un-vendored leaf arithmetic against INDEPENDENT published implementations that happen to be installed
(no weights or network needed): the ResNet-50 / ResNet-34 block wiring against Hugging Face ``transformers``' port of the
torchvision v1.5 network (stride on the 3x3, downsample shortcut, BN eps, ReLU placement, max-pool), ``batch_rodrigues``
and ``batch_euler2matrix`` against ``scipy.spatial.transform.Rotation``, the OpenCV crop / resize restatements against exact
float64 bilinear resampling in torch.  This narrows "parity unpinned" for those leaves:
the restatement agrees with a second upstream, not only with itself."""
import numpy as np
import pytest
import torch

from spec_amd import synth

torch.set_grad_enabled(False)


def _hf_name(k):
    """torchvision trunk key -> transformers ResNetModel key."""
    if k.startswith('conv1.'):
        return 'embedder.embedder.convolution.' + k[len('conv1.'):]
    if k.startswith('bn1.'):
        return 'embedder.embedder.normalization.' + k[len('bn1.'):]
    p = k.split('.')                                  # layerL.b.(convK|bnK|downsample.I).param
    stage, blk = int(p[0][5:]) - 1, p[1]
    base = f'encoder.stages.{stage}.layers.{blk}.'
    if p[2] == 'downsample':
        return base + 'shortcut.' + ('convolution.' if p[3] == '0' else 'normalization.') + p[4]
    idx = int(p[2][-1]) - 1
    return base + f'layer.{idx}.' + ('convolution.' if p[2].startswith('conv') else 'normalization.') + p[3]


@pytest.mark.parametrize('depth', [50, 34])
def test_resnet_trunk_vs_hf_transformers(depth):
    tr = pytest.importorskip('transformers')
    from oracle.models import load_numpy_state
    from oracle.resnet import ResNet34Trunk, ResNet50Trunk
    if depth == 50:
        cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                              layer_type='bottleneck', hidden_act='relu', downsample_in_first_stage=False,
                              downsample_in_bottleneck=False)
        sd, trunk = synth.resnet50_state(1001), ResNet50Trunk().eval()
    else:
        cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[3, 4, 6, 3],
                              layer_type='basic', hidden_act='relu', downsample_in_first_stage=False)
        sd, trunk = synth.resnet34_state(1001), ResNet34Trunk().eval()
    hf = tr.ResNetModel(cfg).eval()
    mapped = {_hf_name(k): torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    assert set(mapped) == set(hf.state_dict().keys())              # 318 / 216 tensors, one to one
    hf.load_state_dict(mapped, strict=True)
    load_numpy_state(trunk, sd)
    x = torch.from_numpy(synth.images(21, 2))[:, :, :160, :128].contiguous()
    ref = hf(x).last_hidden_state
    out = trunk(x)
    assert out.shape == ref.shape == (2, 2048 if depth == 50 else 512, 5, 4)
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < 1e-6, err                                          # same torch ops in the same order: ~0


def test_batch_rodrigues_vs_scipy():
    from scipy.spatial.transform import Rotation
    from oracle.smpl import batch_rodrigues
    rng = np.random.default_rng(0)
    rv = rng.standard_normal((200, 3)) * rng.uniform(0.01, 3.0, (200, 1))
    R = batch_rodrigues(torch.from_numpy(rv)).numpy()              # float64 in, float64 out
    ref = Rotation.from_rotvec(rv).as_matrix()
    assert np.abs(R - ref).max() < 1e-7                             # the +1e-8 guard moves the axis by ~1e-8
    R32 = batch_rodrigues(torch.from_numpy(rv.astype(np.float32))).numpy()
    assert np.abs(R32 - ref).max() < 5e-6


def test_batch_euler2matrix_vs_scipy():
    from scipy.spatial.transform import Rotation
    from oracle.geometry import batch_euler2matrix
    rng = np.random.default_rng(1)
    ang = rng.uniform(-1.2, 1.2, (100, 3)).astype(np.float32)
    R = batch_euler2matrix(torch.from_numpy(ang)).numpy()
    ref = Rotation.from_euler('XYZ', ang.astype(np.float64)).as_matrix()     # intrinsic X-Y-Z = Rx(x) Ry(y) Rz(z)
    assert np.abs(R - ref).max() < 2e-6
    # the hand-off uses (pitch, 0, roll): Rx(pitch) Rz(roll)  (spec/utils/cam_params.py:37)
    pr = np.stack([ang[:, 0], np.zeros(100, np.float32), ang[:, 2]], 1)
    R2 = batch_euler2matrix(torch.from_numpy(pr)).numpy()
    ref2 = Rotation.from_euler('x', pr[:, 0].astype(np.float64)).as_matrix() @ Rotation.from_euler('z', pr[:, 2].astype(np.float64)).as_matrix()
    assert np.abs(R2 - ref2).max() < 2e-6


def test_procrustes_vs_scipy():
    """reconstruction_error's similarity alignment against scipy.linalg.orthogonal_procrustes (no reflection cases)."""
    from scipy.linalg import orthogonal_procrustes
    from oracle.metrics import compute_similarity_transform
    rng = np.random.default_rng(2)
    for _ in range(10):
        S2 = rng.standard_normal((14, 3))
        Rm = np.linalg.qr(rng.standard_normal((3, 3)))[0]
        if np.linalg.det(Rm) < 0:
            Rm[:, 0] *= -1
        S1 = (S2 @ Rm.T) * 1.3 + rng.standard_normal(3) + 0.01 * rng.standard_normal((14, 3))
        hat = compute_similarity_transform(S1, S2)
        A, Bm = S1 - S1.mean(0), S2 - S2.mean(0)
        Rp, sc = orthogonal_procrustes(A, Bm)
        ref = (A @ Rp) * (sc / (A ** 2).sum()) + S2.mean(0)
        assert np.abs(hat - ref).max() < 1e-9


def _smooth_image(h, w, seed):
    """A band-limited uint8 test image (a few low-frequency waves per channel): bilinear resampling of it is insensitive to the
    1/32-pixel coordinate grid of OpenCV's fixed-point warp, so an exact float warp must agree within a grey level or two."""
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        for _ in range(4):
            fx, fy = g.uniform(-0.04, 0.04, 2)
            img[:, :, c] += g.uniform(0.3, 1.0) * np.sin(2 * np.pi * (fx * xx + fy * yy) + g.uniform(0, 6.28))
    img -= img.min()
    return np.rint(img / img.max() * 255).astype(np.uint8)


def test_warp_affine_restatement_vs_exact_bilinear():
    """The oracle's restatement of cv2.warpAffine (fixed-point INTER_LINEAR, BORDER_CONSTANT) against an exact float64 bilinear
    warp through ``torch.nn.functional.grid_sample`` (zeros padding): same geometry (pixel centres at integers, inverse map,
    3-point transform of the crop box), same border rule.  The two differ only by OpenCV's quantisation - coordinates on a
    1/32 px grid, 15-bit weights, round-to-nearest output - i.e. a grey level or two on a smooth image."""
    from oracle.preprocess import gen_trans_from_patch, invert_affine, warp_affine_linear_u8
    H, W, S = 240, 320, 224
    img = _smooth_image(H, W, 5)
    g = np.random.default_rng(6)
    src = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    boxes = [(160.0, 120.0, 200.0, 200.0), (20.0, 30.0, 180.0, 260.0), (300.0, 200.0, 150.0, 150.0), (160.5, 119.25, 90.0, 333.0)]
    boxes += [(g.uniform(0, W), g.uniform(0, H), g.uniform(60, 400), g.uniform(60, 400)) for _ in range(4)]
    for cx, cy, bw, bh in boxes:
        M = gen_trans_from_patch(cx, cy, bw, bh, S, S, 1.0)
        got = warp_affine_linear_u8(img, M, S, S).astype(np.float64)
        Mi = invert_affine(M)
        ys, xs = np.mgrid[0:S, 0:S].astype(np.float64)
        sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]                  # source pixel coordinates (centres at integers)
        sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
        grid = torch.from_numpy(np.stack([(2 * sx + 1) / W - 1, (2 * sy + 1) / H - 1], -1))[None]   # align_corners=False
        ref = torch.nn.functional.grid_sample(src, grid, mode='bilinear', padding_mode='zeros', align_corners=False)[0]
        ref = ref.permute(1, 2, 0).numpy()
        d = np.abs(got - ref).max(-1)
        inside = (sx >= 0.05) & (sx <= W - 1.05) & (sy >= 0.05) & (sy <= H - 1.05)          # all four taps are image pixels
        assert inside.any()
        assert d[inside].max() <= 2.0, (cx, cy, bw, bh, d[inside].max())
        assert d[inside].mean() <= 0.5, d[inside].mean()      # rounding to uint8 alone: 0.25 per channel, 0.375 for the worst of three
        # across the image border the signal drops to the constant 0 within one pixel: 1/32 px of coordinate rounding is worth
        # up to 255 / 32 = 8 grey levels there
        assert d.max() <= 255 / 32 + 1.5, d.max()
        outside = ~((sx > -1) & (sx < W) & (sy > -1) & (sy < H))
        assert (got[outside] == 0).all() and (ref[outside] == 0).all()                   # constant border: nothing leaks outside


@pytest.mark.parametrize('src_hw,dst', [((37, 53), 224), ((224, 224), 224), ((300, 280), 224), ((613, 411), 224), ((50, 50), 64)])
def test_cv2_resize_restatement_vs_torch_interpolate(src_hw, dst):
    """The oracle's cv2.resize(INTER_LINEAR) restatement (half-pixel centres, replicated border, NO anti-aliasing when
    shrinking) against ``torch.nn.functional.interpolate(mode='bilinear', align_corners=False)`` in float64 - the same
    published definition implemented independently."""
    from oracle.preprocess import cv2_resize_linear_f64
    g = np.random.default_rng(src_hw[0] * 7 + dst)
    img = g.uniform(0, 255, (src_hw[0], src_hw[1], 3))
    got = cv2_resize_linear_f64(img, dst, dst)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(dst, dst), mode='bilinear',
                                          align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    assert got.shape == ref.shape
    # OpenCV computes the source coordinate (d + 0.5) * scale - 0.5 in float32: near column 600 that is an ulp of 6e-5 px, times
    # a gradient of up to 255 grey levels per pixel; a convention error (corner alignment, missing clamp) would be ~100 levels
    assert np.abs(got - ref).max() < 255 * 1.2e-7 * max(src_hw) + 1e-4
