"""Oracle (test infrastructure): the crop + normalise step that feeds the hot path
(``spec/tester.py:116-128``: ``get_single_image_crop_demo(img, bbox, kp_2d=None, scale=1.0,
crop_size=224)`` per detection, then ``bbox_scale = bbox[2]/200``, ``bbox_center = bbox[:2]``).

``get_single_image_crop_demo`` lives in the un-vendored ``pare.utils.vibe_image_utils`` and is
a thin wrapper over OpenCV (``cv2.getAffineTransform`` + ``cv2.warpAffine(INTER_LINEAR,
BORDER_CONSTANT)``) followed by torchvision ``ToTensor`` + ``Normalize``.  Neither ``pare`` nor
``cv2`` is installed here, so this is a NumPy restatement of the published algorithms -
**parity unpinned** against the real OpenCV binary:

* ``gen_trans_from_patch`` - the 3-point affine of VIBE/PARE for rot = 0 (axis aligned);
* ``warp_affine_linear_u8`` - OpenCV's fixed-point bilinear warp: inverse map in float64,
  coordinates in 1/1024 px (AB_BITS = 10) rounded to 1/32 px (INTER_BITS = 5), 15-bit weight
  table whose four entries sum to 32768, result (sum + 2^14) >> 15;
* ToTensor / Normalize in float32: ``(u8 / 255 - mean) / std``.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
REMAP_COEF_BITS = 15
REMAP_COEF_SCALE = 1 << REMAP_COEF_BITS

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)   # spec/constants.py:20
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)    # spec/constants.py:21


def gen_trans_from_patch(c_x, c_y, src_w, src_h, dst_w, dst_h, scale):
    """2x3 forward affine (src -> dst) for rot = 0: the three point pairs are centre, centre +
    (0, h/2), centre + (w/2, 0), with the half extents rounded to float32 as upstream does."""
    sw = np.float32(np.float32(src_w * scale) * np.float32(0.5))
    sh = np.float32(np.float32(src_h * scale) * np.float32(0.5))
    cx, cy = np.float32(c_x), np.float32(c_y)
    ax = np.float64(np.float32(dst_w * 0.5)) / np.float64(sw)
    ay = np.float64(np.float32(dst_h * 0.5)) / np.float64(sh)
    return np.array([[ax, 0.0, np.float64(np.float32(dst_w * 0.5)) - ax * np.float64(cx)],
                     [0.0, ay, np.float64(np.float32(dst_h * 0.5)) - ay * np.float64(cy)]], dtype=np.float64)


def _bilinear_tab():
    """OpenCV's INTER_LINEAR fixed-point table: [32*32][4] int weights summing to 32768."""
    tab = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 4), dtype=np.int32)
    for iy in range(INTER_TAB_SIZE):
        fy = np.float32(iy) / np.float32(INTER_TAB_SIZE)
        for ix in range(INTER_TAB_SIZE):
            fx = np.float32(ix) / np.float32(INTER_TAB_SIZE)
            w = np.array([(np.float32(1) - fx) * (np.float32(1) - fy), fx * (np.float32(1) - fy),
                          (np.float32(1) - fx) * fy, fx * fy], dtype=np.float32)
            iw = np.rint(w * np.float32(REMAP_COEF_SCALE)).astype(np.int32)      # saturate_cast<short> = round half even
            diff = int(iw.sum()) - REMAP_COEF_SCALE
            if diff != 0:   # OpenCV: a deficit goes to the largest entry, an excess comes off the smallest
                k = int(np.argmax(iw)) if diff < 0 else int(np.argmin(iw))   # (never taken for 5-bit bilinear:
                iw[k] -= diff                                                 #  the products are exact)
            tab[iy * INTER_TAB_SIZE + ix] = iw
    return tab


BILINEAR_TAB = _bilinear_tab()


def invert_affine(M):
    """cv::warpAffine's in-place inversion of the 2x3 matrix (float64)."""
    M = np.array(M, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = A11, M[0, 1] * -D, M[1, 0] * -D, A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def _sat_int(x):
    return np.rint(x).astype(np.int64)           # saturate_cast<int>(double) = lrint (round half even)


def warp_affine_linear_u8(img, M, dst_w, dst_h):
    """cv2.warpAffine(img, M, (dst_w, dst_h), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT (0))."""
    H, W, C = img.shape
    Mi = invert_affine(M)
    x = np.arange(dst_w)
    adelta = _sat_int(Mi[0, 0] * x * AB_SCALE)
    bdelta = _sat_int(Mi[1, 0] * x * AB_SCALE)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    out = np.zeros((dst_h, dst_w, C), dtype=np.uint8)
    src = img.astype(np.int64)
    for y in range(dst_h):
        X0 = _sat_int((Mi[0, 1] * y + Mi[0, 2]) * AB_SCALE) + round_delta
        Y0 = _sat_int((Mi[1, 1] * y + Mi[1, 2]) * AB_SCALE) + round_delta
        X = (X0 + adelta) >> (AB_BITS - INTER_BITS)
        Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS)
        sx, sy = X >> INTER_BITS, Y >> INTER_BITS
        w = BILINEAR_TAB[(Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))]   # (dst_w,4)

        def tap(yy, xx):
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
            return np.where(ok[:, None], v, 0)
        acc = (tap(sy, sx) * w[:, 0:1] + tap(sy, sx + 1) * w[:, 1:2] + tap(sy + 1, sx) * w[:, 2:3]
               + tap(sy + 1, sx + 1) * w[:, 3:4])
        out[y] = np.clip((acc + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS, 0, 255).astype(np.uint8)
    return out


def to_tensor_normalize(img_u8):
    x = img_u8.astype(np.float32) / np.float32(255.0)
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32)


def get_single_image_crop_demo(image, bbox, scale=1.0, crop_size=224):
    """-> (norm_img (3,S,S) float32, raw_img (S,S,3) uint8)."""
    M = gen_trans_from_patch(bbox[0], bbox[1], bbox[2], bbox[3], crop_size, crop_size, scale)
    raw = warp_affine_linear_u8(image, M, crop_size, crop_size)
    return to_tensor_normalize(raw), raw


def crop_detections(image, dets, scale=1.0, crop_size=224):
    """The per-detection loop of spec/tester.py:116-128 -> inp_images, raw, bbox_scale, bbox_center."""
    imgs, raws, sc, ce = [], [], [], []
    for bbox in dets:
        n, r = get_single_image_crop_demo(image, bbox, scale, crop_size)
        imgs.append(n); raws.append(r)
        sc.append(np.float32(bbox[2] / 200.))
        ce.append([np.float32(bbox[0]), np.float32(bbox[1])])
    return np.stack(imgs), np.stack(raws), np.array(sc, np.float32), np.array(ce, np.float32)


# ------------------------------------------------------------------------------------------------
# CamCalib frame transform (camcalib/pano_dataset.py:156-162: torchvision Resize(600) on a PIL image
# -> ToTensor -> Normalize).  torchvision's Resize(int) on a PIL image calls
# ``img.resize((ow, oh), Image.BILINEAR)`` with the shorter side set to ``size`` and the longer to
# ``int(size * long / short)``; Pillow's resize is a separable two-pass convolution (horizontal, then
# vertical, uint8 in between) with a triangle filter whose support grows with the down-scale factor
# and 22-bit fixed-point coefficients.  Restated from Pillow's published algorithm
# (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
# Vertical_8bpc) and PINNED against the installed Pillow binary (tests/test_preprocess.py, bit-exact).
# ------------------------------------------------------------------------------------------------
PIL_PRECISION_BITS = 32 - 8 - 2


def resize_output_size(w, h, min_size=600):
    """torchvision.transforms.Resize(int) -> (ow, oh): shorter side = min_size, longer = int(min_size*long/short)."""
    if w <= h:
        return min_size, int(min_size * h / w)
    return int(min_size * w / h), min_size


def pil_bilinear_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the triangle filter over the box [0, in_size):
    -> bounds (out_size, 2) int32 [xmin, count], kk (out_size, ksize) int32 fixed-point weights."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] = k[:xmax] / ww
        for x in range(ksize):   # (int)(+-0.5 + k * 2^22): C truncation toward zero
            v = k[x] * (1 << PIL_PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pil_pass(img, bounds, kk, axis):
    """One 8bpc resample pass along ``axis`` (1 = horizontal, 0 = vertical) of an (H, W, C) uint8 image."""
    src = img.astype(np.int64)
    n = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n
    out = np.empty(shape, np.uint8)
    for i in range(n):
        lo, cnt = int(bounds[i, 0]), int(bounds[i, 1])
        w = kk[i, :cnt].astype(np.int64)
        if axis == 1:
            acc = (src[:, lo:lo + cnt, :] * w[None, :, None]).sum(axis=1)
        else:
            acc = (src[lo:lo + cnt, :, :] * w[:, None, None]).sum(axis=0)
        acc = (acc + (1 << (PIL_PRECISION_BITS - 1))) >> PIL_PRECISION_BITS
        v = np.clip(acc, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, i, :] = v
        else:
            out[i, :, :] = v
    return out


def pil_resize_bilinear_u8(img, ow, oh):
    """Image.resize((ow, oh), BILINEAR) of an (H, W, 3) uint8 array: horizontal pass (if the width changes),
    then vertical pass (if the height changes), uint8 intermediate."""
    H, W = img.shape[:2]
    out = img
    if ow != W:
        b, k = pil_bilinear_coeffs(W, ow)
        out = _pil_pass(out, b, k, axis=1)
    if oh != H:
        b, k = pil_bilinear_coeffs(H, oh)
        out = _pil_pass(out, b, k, axis=0)
    return np.ascontiguousarray(out)


def camcalib_transform(frame_u8, min_size=600):
    """ImageFolder.__getitem__ of camcalib/pano_dataset.py:156-179 -> (3, oh, ow) float32."""
    H, W = frame_u8.shape[:2]
    ow, oh = resize_output_size(W, H, min_size)
    return to_tensor_normalize(pil_resize_bilinear_u8(frame_u8, ow, oh))


# ------------------------------------------------------------------------------------------------------------------
# evaluation-dataset crop: pare / SPIN `crop` + cv2.resize + rgb_processing + Normalize
# (spec/dataset/cam_dataset.py:253-287,367-377; `crop` / `transform` / `get_transform` restated from the published SPIN
# image utilities that pare copies; cv2.resize INTER_LINEAR on float64 restated: parity unpinned vs the OpenCV binary)
# ------------------------------------------------------------------------------------------------------------------
def get_transform(center, scale, res, rot=0):
    h = 200 * scale
    t = np.zeros((3, 3))
    t[0, 0] = float(res[1]) / h
    t[1, 1] = float(res[0]) / h
    t[0, 2] = res[1] * (-float(center[0]) / h + .5)
    t[1, 2] = res[0] * (-float(center[1]) / h + .5)
    t[2, 2] = 1
    assert rot == 0
    return t


def transform(pt, center, scale, res, invert=0, rot=0):
    t = get_transform(center, scale, res, rot=rot)
    if invert:
        t = np.linalg.inv(t)
    new_pt = np.array([pt[0] - 1, pt[1] - 1, 1.]).T
    new_pt = np.dot(t, new_pt)
    return new_pt[:2].astype(int) + 1


def cv2_resize_linear_f64(src, dst_w, dst_h):
    """cv2.resize(src (h,w,c) float64, (dst_w, dst_h)) with INTER_LINEAR: half-pixel centres, replicated border, float
    coefficients, double accumulation (horizontal pass then vertical pass)."""
    h, w = src.shape[:2]

    def coeffs(ssize, dsize):
        scale = float(ssize) / dsize
        idx, a = np.zeros(dsize, np.int64), np.zeros(dsize, np.float32)
        for d in range(dsize):
            f = np.float32((d + 0.5) * scale - 0.5)
            s_ = int(np.floor(f))
            f = np.float32(f - np.float32(s_))
            if s_ < 0:
                f, s_ = np.float32(0), 0
            if s_ >= ssize - 1:
                f, s_ = np.float32(0), ssize - 1
            idx[d], a[d] = s_, f
        return idx, a
    xi, xa = coeffs(w, dst_w)
    yi, ya = coeffs(h, dst_h)
    xi1 = np.minimum(xi + 1, w - 1)
    yi1 = np.minimum(yi + 1, h - 1)
    a1 = xa.astype(np.float64)[None, :, None]
    a0 = (np.float32(1) - xa).astype(np.float64)[None, :, None]
    rows = src[:, xi] * a0 + src[:, xi1] * a1                          # (h, dst_w, c)
    b1 = ya.astype(np.float64)[:, None, None]
    b0 = (np.float32(1) - ya).astype(np.float64)[:, None, None]
    return rows[yi] * b0 + rows[yi1] * b1


def pare_crop(img, center, scale, res):
    """SPIN / pare ``crop(img, center, scale, res, rot=0)``: integer box copy (zero padded) + cv2.resize."""
    ul = np.array(transform([1, 1], center, scale, res, invert=1)) - 1
    br = np.array(transform([res[0] + 1, res[1] + 1], center, scale, res, invert=1)) - 1
    new_shape = [br[1] - ul[1], br[0] - ul[0]]
    if len(img.shape) > 2:
        new_shape += [img.shape[2]]
    new_img = np.zeros(new_shape)
    new_x = max(0, -ul[0]), min(br[0], len(img[0])) - ul[0]
    new_y = max(0, -ul[1]), min(br[1], len(img)) - ul[1]
    old_x = max(0, ul[0]), min(len(img[0]), br[0])
    old_y = max(0, ul[1]), min(len(img), br[1])
    new_img[new_y[0]:new_y[1], new_x[0]:new_x[1]] = img[old_y[0]:old_y[1], old_x[0]:old_x[1]]
    return cv2_resize_linear_f64(new_img, res[0], res[1])


def dataset_crop(img_u8, center, scale, res=224):
    """rgb_processing (flip 0, rot 0, pn = 1) + Normalize: (H,W,3) uint8 -> (3,res,res) fp32."""
    rgb = pare_crop(img_u8, center, scale, [res, res])
    for c in range(3):
        rgb[:, :, c] = np.minimum(255.0, np.maximum(0.0, rgb[:, :, c] * 1.0))
    x = np.transpose(rgb.astype('float32'), (2, 0, 1)) / 255.0
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)
    return ((x - mean) / std).astype(np.float32)
