"""Crop + normalise: oracle KATs on CPU, bit-exact GPU parity through the C ABI."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import preprocess as P
from tests.util import t


# (seed, H, W, min_size) of the CamCalib Resize cases: down-scale by 1.6-5.4x (wide filter), up-scale, odd sizes, portrait
RESIZE_CASES = [(21, 97, 130, 48), (22, 120, 80, 64), (23, 40, 60, 64), (24, 161, 87, 30), (25, 64, 64, 64), (26, 75, 203, 33)]


def _frame(seed, H=180, W=260):
    rng = np.random.default_rng(seed)
    img = rng.random((H, W, 3)) * 255
    return ndimage.uniform_filter(img, size=(5, 5, 1)).astype(np.uint8)


def test_oracle_identity_crop_is_exact_copy():
    img = _frame(1, 300, 300)
    # a 224x224 box centred so that dst pixel u maps to src pixel u + 20: cx - 112 = 20
    norm, raw = P.get_single_image_crop_demo(img, [132.0, 142.0, 224.0, 224.0], 1.0, 224)
    assert np.array_equal(raw, img[30:254, 20:244])
    ref = ((raw.astype(np.float32) / np.float32(255) - P.MEAN) / P.STD).transpose(2, 0, 1)
    assert np.array_equal(norm, ref)


def test_oracle_close_to_float_bilinear_and_zero_border():
    img = _frame(2)
    bbox = [40.3, 35.7, 150.0, 150.0]              # sticks out of the frame on the left/top
    M = P.gen_trans_from_patch(*bbox, 224, 224, 1.0)
    Mi = P.invert_affine(M)
    raw = P.warp_affine_linear_u8(img, M, 224, 224)
    ys, xs = np.mgrid[0:224, 0:224]
    sx, sy = Mi[0, 0] * xs + Mi[0, 2], Mi[1, 1] * ys + Mi[1, 2]
    ref = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64), [sy, sx], order=1, mode='constant')
                    for c in range(3)], -1)
    inside = (sx >= 0) & (sx <= img.shape[1] - 1) & (sy >= 0) & (sy <= img.shape[0] - 1)
    assert inside.sum() > 20000
    assert np.abs(ref - raw)[inside].max() < 1.6   # cv2's 1/32-px, 15-bit fixed point vs exact bilinear
    assert np.all(raw[(sy < -1.1) | (sx < -1.1)] == 0)   # BORDER_CONSTANT
    assert P.BILINEAR_TAB.sum(1).min() == 32768 and P.BILINEAR_TAB.sum(1).max() == 32768


def test_oracle_detection_loop_contract():
    img = _frame(3)
    dets = np.array([[130.0, 90.0, 120.0, 120.0], [60.5, 70.25, 80.0, 80.0]], np.float32)
    imgs, raws, sc, ce = P.crop_detections(img, dets)
    assert imgs.shape == (2, 3, 224, 224) and raws.shape == (2, 224, 224, 3) and imgs.dtype == np.float32
    assert np.array_equal(sc, dets[:, 2] / np.float32(200)) and np.array_equal(ce, dets[:, :2])


@pytest.mark.gpu
@pytest.mark.parametrize('H,W', [(180, 260), (480, 640), (97, 131)])
def test_gpu_crop_bit_exact(H, W):
    from spec_amd.preprocess import crop_detections
    img = _frame(H, H, W)
    rng = np.random.default_rng(W)
    n = 7
    dets = np.stack([rng.uniform(0.1 * W, 0.9 * W, n), rng.uniform(0.1 * H, 0.9 * H, n),
                     rng.uniform(40, 1.2 * H, n), np.zeros(n)], 1).astype(np.float32)
    dets[:, 3] = dets[:, 2]                        # square boxes as the tracker emits
    dets[0] = [W / 2, H / 2, 2.5 * W, 2.5 * W]     # far larger than the frame
    dets[1] = [3.0, 2.0, 60.0, 60.0]               # mostly outside
    for scale in (1.0, 1.1):
        ref_imgs, ref_raw, ref_sc, ref_ce = P.crop_detections(img, dets, scale)
        out = crop_detections(t(img).to('cuda:0'), t(dets).to('cuda:0'), scale=scale, return_raw=True)
        assert np.array_equal(out['raw'].cpu().numpy(), ref_raw)
        assert np.array_equal(out['inp_images'].cpu().numpy(), ref_imgs)
        assert np.array_equal(out['bbox_scale'].cpu().numpy(), ref_sc)
        assert np.array_equal(out['bbox_center'].cpu().numpy(), ref_ce)


@pytest.mark.gpu
def test_gpu_crop_feeds_hmr():
    """The cropped batch goes straight into the trunk (same dtype/layout the tester builds)."""
    from spec_amd.preprocess import crop_detections
    out = crop_detections(t(_frame(9)).to('cuda:0'), torch.tensor([[130., 90., 120., 120.]]))
    x = out['inp_images']
    assert x.shape == (1, 3, 224, 224) and x.dtype == torch.float32 and x.is_contiguous()
    assert float(x.max()) <= (1 - 0.406) / 0.225 + 1e-5 and float(x.min()) >= -0.485 / 0.229 - 1e-5


# ---- CamCalib frame transform: Pillow's antialiased bilinear resize + ToTensor + Normalize -----------------

def _resize_golden():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'camcalib_transform.npz'))
    return [g[f'case{i}'] for i in range(len(RESIZE_CASES))]


def test_oracle_resize_matches_pillow_fixture():
    """The NumPy restatement of Pillow's ImagingResample is bit-exact against vectors produced by the real
    Pillow binary (tests/golden/make_pillow_fixture.py)."""
    for (seed, H, W, ms), ref in zip(RESIZE_CASES, _resize_golden()):
        ow, oh = P.resize_output_size(W, H, ms)
        got = P.pil_resize_bilinear_u8(_frame(seed, H, W), ow, oh)
        assert got.shape == ref.shape and np.array_equal(got, ref), (H, W, ms)


def test_oracle_resize_matches_installed_pillow():
    """Same check against whatever Pillow is importable (skipped without it), on sizes the demo sees."""
    Image = pytest.importorskip('PIL.Image')
    for seed, (H, W) in enumerate([(480, 640), (720, 1280), (600, 450), (97, 130)]):
        img = _frame(40 + seed, H, W)
        ow, oh = P.resize_output_size(W, H, 600)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(P.pil_resize_bilinear_u8(img, ow, oh), ref)


def test_resize_geometry_and_identity():
    assert P.resize_output_size(1920, 1080) == (1066, 600)      # int(600 * 1920 / 1080)
    assert P.resize_output_size(450, 600) == (600, 800)
    img = _frame(3, 64, 64)
    assert np.array_equal(P.pil_resize_bilinear_u8(img, 64, 64), img)
    x = P.camcalib_transform(img, 64)
    assert x.shape == (3, 64, 64) and x.dtype == np.float32
    np.testing.assert_array_equal(x, P.to_tensor_normalize(img))


@pytest.mark.gpu
@pytest.mark.parametrize('case', range(len(RESIZE_CASES)))
def test_gpu_resize_bit_exact(case):
    """specmi_resize_normalize vs the Pillow fixture (uint8 image) and vs the oracle (normalised tensor): bit-exact."""
    from spec_amd.preprocess import camcalib_transform
    seed, H, W, ms = RESIZE_CASES[case]
    img = _frame(seed, H, W)
    out, raw = camcalib_transform(t(img).to('cuda:0'), ms, return_raw=True)
    ref_u8 = _resize_golden()[case]
    assert np.array_equal(raw.cpu().numpy(), ref_u8)
    ref = P.camcalib_transform(img, ms)
    assert out.shape == (1,) + ref.shape
    np.testing.assert_array_equal(out[0].cpu().numpy(), ref)


@pytest.mark.gpu
def test_gpu_resize_demo_sizes_and_camcalib():
    """Full-size frames (the demo's Resize(600)), geometry changes between calls (the coefficient tables are
    rebuilt), and the result feeds the CamCalib network at its native variable resolution."""
    from spec_amd.preprocess import camcalib_transform
    from tests.util import gpu_models
    cc, _ = gpu_models(True, True, 'cuda:0')
    for seed, (H, W) in enumerate([(480, 640), (720, 1280), (450, 600), (480, 640)]):
        img = _frame(50 + seed, H, W)
        x = camcalib_transform(t(img).to('cuda:0'), 600)
        np.testing.assert_array_equal(x[0].cpu().numpy(), P.camcalib_transform(img, 600))
        lg = cc(x)
        assert all(torch.isfinite(l).all() and l.shape == (1, 256) for l in lg)


# ---- evaluation-dataset crop (pare `crop` + cv2.resize + rgb_processing + Normalize) ---------------------------------------
def test_dataset_crop_oracle_closed_form():
    """cv2.resize-style bilinear (half-pixel centres) reproduces a linear ramp exactly where no border clamp applies, the box
    arithmetic equals the reference's transform(), and a box that leaves the frame is zero padded."""
    from oracle import preprocess as OP
    from spec_amd.preprocess import pare_crop_boxes
    ramp = np.tile((np.arange(90, dtype=np.float64) * 2.0 + 5.0)[None, :, None], (60, 1, 3))
    out = OP.cv2_resize_linear_f64(ramp, 224, 224)
    d = np.arange(224)
    expect = 2.0 * ((d + 0.5) * (90 / 224) - 0.5) + 5.0
    inner = (expect >= 5.0) & (expect <= 2.0 * 89 + 5.0)
    assert np.abs(out[10, inner, 0] - expect[inner]).max() < 1e-4          # float coefficients, double accumulation
    rng = np.random.default_rng(0)
    centers, scales = rng.uniform(-20, 300, (50, 2)), rng.uniform(0.2, 2.5, 50)
    boxes = pare_crop_boxes(centers, scales, 224)
    for c, sc, b in zip(centers, scales, boxes):
        ul = np.array(OP.transform([1, 1], c, sc, [224, 224], invert=1)) - 1
        br = np.array(OP.transform([225, 225], c, sc, [224, 224], invert=1)) - 1
        assert list(b) == [ul[0], ul[1], br[0], br[1]]
        assert abs((b[2] - b[0]) - 200 * sc) <= 1.5 and abs((b[0] + b[2]) / 2 - c[0]) <= 1.0
    img = np.full((40, 50, 3), 200, np.uint8)
    x = OP.dataset_crop(img, [0.0, 0.0], 0.3, 224)                          # box [-30, 30)^2: three quarters outside
    assert x.shape == (3, 224, 224) and abs(float(x[0, 10, 10]) - (0 - 0.485) / 0.229) < 1e-6
    assert abs(float(x[0, 200, 200]) - (200 / 255 - 0.485) / 0.229) < 1e-6


@pytest.mark.gpu
def test_dataset_crop_gpu_bit_exact_vs_oracle():
    import torch
    from oracle import preprocess as OP
    from spec_amd.preprocess import dataset_crops
    rng = np.random.default_rng(3)
    for (H, W) in ((360, 480), (97, 131), (720, 1280)):
        img = (rng.random((H, W, 3)) * 255).astype(np.uint8)
        centers = np.array([[W / 2, H / 2], [5.5, 7.25], [W - 3.0, H - 10.0], [W * 0.3, H * 0.8], [-40.0, H / 2]])
        scales = np.array([min(H, W) / 200.0, 0.35, 1.7, 224 / 200.0, 0.9])
        out = dataset_crops(torch.from_numpy(img).to('cuda:0'), centers, scales, 224).cpu().numpy()
        for i in range(len(scales)):
            ref = OP.dataset_crop(img, centers[i], scales[i], 224)
            assert np.array_equal(out[i], ref), (H, W, i, np.abs(out[i] - ref).max())
