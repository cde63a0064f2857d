// preprocess.hip - crop + normalise on the device.
//
// Replaces the per-detection host loop that feeds the hot path (spec/tester.py:116-128):
// get_single_image_crop_demo(img, bbox, scale=1.0, crop_size=224) = VIBE/PARE's 3-point affine
// (rot = 0) + cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) + ToTensor + Normalize, followed by
// bbox_scale = bbox[2]/200 and bbox_center = bbox[:2].  One launch handles all detections of a
// frame: uint8 RGB HWC frame in HBM -> (n,3,S,S) fp32 NCHW crops, already where the trunk reads
// them (no host crop, no H2D of 602 KB per detection).
//
// The warp reproduces OpenCV's fixed-point bilinear path exactly (integer arithmetic): inverse
// affine in fp64, source coordinates in 1/1024 px rounded to 1/32 px, weights
// (32-fx)(32-fy)*32 ... summing to 2^15, result (sum + 2^14) >> 15; then (u8/255 - mean)/std
// in fp32 like ToTensor + Normalize.  Lanes map to consecutive x of one output row, so the
// NCHW stores are coalesced; the 4 taps x 3 channels come through L2 (a frame is a few MB).
#include <cmath>

#include "specmi_internal.h"

namespace specmi {

__global__ void __launch_bounds__(256) crop_normalize_kernel(const unsigned char* __restrict__ frame, int H, int W,
                                                              const float* __restrict__ bboxes, float scale, int S,
                                                              float* __restrict__ out, unsigned char* __restrict__ raw,
                                                              float* __restrict__ bbox_scale,
                                                              float* __restrict__ bbox_center) {
    const int d = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const float cx = bboxes[d * 4 + 0], cy = bboxes[d * 4 + 1], bw = bboxes[d * 4 + 2], bh = bboxes[d * 4 + 3];
    if (idx == 0) {
        if (bbox_scale) bbox_scale[d] = bw / 200.0f;
        if (bbox_center) { bbox_center[d * 2 + 0] = cx; bbox_center[d * 2 + 1] = cy; }
    }
    if (idx >= S * S) return;
    const int y = idx / S, x = idx - y * S;
    // forward affine (src -> dst), then cv::warpAffine's inversion, all in fp64
    const float sw = (bw * scale) * 0.5f, sh = (bh * scale) * 0.5f;
    const double half = (double)((float)S * 0.5f);
    const double ax = half / (double)sw, ay = half / (double)sh;
    double M0 = ax, M1 = 0.0, M2 = half - ax * (double)cx, M3 = 0.0, M4 = ay, M5 = half - ay * (double)cy;
    double D = M0 * M4 - M1 * M3;
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M4 * D, A22 = M0 * D;
    M0 = A11; M1 *= -D; M3 *= -D; M4 = A22;
    const double b1 = -M0 * M2 - M1 * M5, b2 = -M3 * M2 - M4 * M5;
    M2 = b1; M5 = b2;
    const long long adelta = __double2ll_rn(M0 * x * 1024.0), bdelta = __double2ll_rn(M3 * x * 1024.0);
    const long long X0 = __double2ll_rn((M1 * y + M2) * 1024.0) + 16, Y0 = __double2ll_rn((M4 * y + M5) * 1024.0) + 16;
    const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const long long sx = X >> 5, sy = Y >> 5;
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int p00 = (y0 && x0) ? frame[((size_t)sy * W + sx) * 3 + c] : 0;
        const int p01 = (y0 && x1) ? frame[((size_t)sy * W + sx + 1) * 3 + c] : 0;
        const int p10 = (y1 && x0) ? frame[((size_t)(sy + 1) * W + sx) * 3 + c] : 0;
        const int p11 = (y1 && x1) ? frame[((size_t)(sy + 1) * W + sx + 1) * 3 + c] : 0;
        int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        if (raw) raw[((size_t)d * S * S + idx) * 3 + c] = (unsigned char)v;
        out[((size_t)(d * 3 + c) * S + y) * S + x] = ((float)v / 255.0f - mean[c]) / stdv[c];
    }
}

int launch_crop_normalize(const unsigned char* frame, int H, int W, const float* bboxes, int n, float scale, int S,
                          float* out, unsigned char* raw, float* bbox_scale, float* bbox_center, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "crop_normalize", 0.0, (double)n * S * S * (12.0 + 12.0 + (raw ? 3.0 : 0.0)));
    hipLaunchKernelGGL(crop_normalize_kernel, dim3((S * S + 255) / 256, n), dim3(256), 0, ctx.stream, frame, H, W, bboxes,
                       scale, S, out, raw, bbox_scale, bbox_center);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Evaluation-dataset crop (spec/dataset/cam_dataset.py:253-287,367-377: rgb_processing -> pare `crop` -> ToTensor ->
// Normalize).  PARE / SPIN `crop(img, center, scale, res)` copies the integer box [ul, br) (200*scale pixels around the
// centre, zero outside the frame) and scales it to res x res with cv2.resize (INTER_LINEAR on a float64 array: half-pixel
// centres, replicated border, float coefficients, double accumulation); rgb_processing clips to [0, 255], converts to
// float32 / 255 and the dataset normalises with the ImageNet mean / std.  The integer boxes come from the host (the
// reference computes them with a 3x3 float64 inverse; spec_amd/preprocess.py restates that), everything per pixel runs here.
__global__ void __launch_bounds__(256) crop_resize_normalize_kernel(const unsigned char* __restrict__ frame, int H, int W,
                                                                     const int* __restrict__ boxes, int S,
                                                                     float* __restrict__ out) {
    const int d = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    const int ulx = boxes[d * 4 + 0], uly = boxes[d * 4 + 1], brx = boxes[d * 4 + 2], bry = boxes[d * 4 + 3];
    const int bw = brx - ulx, bh = bry - uly;
    const int dy = idx / S, dx = idx - dy * S;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    if (bw <= 0 || bh <= 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((size_t)(d * 3 + c) * S + dy) * S + dx] = (0.0f - mean[c]) / stdv[c];
        return;
    }
    // cv::resize INTER_LINEAR: fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx, clamped to the box
    const double scale_x = (double)bw / (double)S, scale_y = (double)bh / (double)S;
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5), fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx; fy -= (float)sy;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= bw - 1) { fx = 0.f; sx = bw - 1; }
    if (sy < 0) { fy = 0.f; sy = 0; }
    if (sy >= bh - 1) { fy = 0.f; sy = bh - 1; }
    const int sx1 = sx + 1 < bw ? sx + 1 : sx, sy1 = sy + 1 < bh ? sy + 1 : sy;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    auto px = [&](int by, int bx, int c) -> double {   // box pixel (zero where the box leaves the frame)
        const int iy = uly + by, ix = ulx + bx;
        return ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? (double)frame[((size_t)iy * W + ix) * 3 + c] : 0.0;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // separate multiplies and adds (no fma contraction): the reference rounds each product
        const double r0 = __dadd_rn(__dmul_rn(px(sy, sx, c), (double)a0), __dmul_rn(px(sy, sx1, c), (double)a1));     // horizontal pass
        const double r1 = __dadd_rn(__dmul_rn(px(sy1, sx, c), (double)a0), __dmul_rn(px(sy1, sx1, c), (double)a1));
        double v = __dadd_rn(__dmul_rn(r0, (double)b0), __dmul_rn(r1, (double)b1));                                     // vertical pass
        v = fmin(255.0, fmax(0.0, v));                                                  // rgb_processing: pn = 1, clip
        const float t = (float)v / 255.0f;                                              // astype('float32') / 255.0
        out[((size_t)(d * 3 + c) * S + dy) * S + dx] = (t - mean[c]) / stdv[c];
    }
}

int launch_crop_resize_normalize(const unsigned char* frame, int H, int W, const int* boxes, int n, int S, float* out,
                                 const LaunchCtx& ctx) {
    ProfScope ps(ctx, "crop_resize_normalize", 0.0, (double)n * S * S * 24.0);
    hipLaunchKernelGGL(crop_resize_normalize_kernel, dim3((S * S + 255) / 256, n), dim3(256), 0, ctx.stream, frame, H, W, boxes, S, out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// CamCalib frame transform (camcalib/pano_dataset.py:156-162): torchvision Resize(600) on a PIL image =
// Pillow's separable triangle-filter resample (support grows with the down-scale factor, 22-bit fixed-point
// coefficients, horizontal pass -> uint8 -> vertical pass -> uint8), then ToTensor + Normalize.
// The coefficient tables are built on the host exactly like Pillow's precompute_coeffs /
// normalize_coeffs_8bpc (double precision, C truncation) - pillow_coeffs() - and the kernel evaluates
// both passes per output pixel in integer arithmetic, so the result is bit-identical to Pillow's.
// ------------------------------------------------------------------------------------------------
constexpr int PIL_BITS = 32 - 8 - 2;

// -> ksize; bounds[2*i] = first tap, bounds[2*i+1] = tap count, kk[i*ksize + t] = fixed-point weight
int pillow_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < ksize; ++x) k[x] = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < ksize; ++x)
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PIL_BITS)) : (int)(0.5 + k[x] * (1 << PIL_BITS));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

__global__ void __launch_bounds__(256) resize_normalize_kernel(const unsigned char* __restrict__ frame, int H, int W, int OH,
                                                                int OW, const int* __restrict__ hb, const int* __restrict__ hk,
                                                                int ksh, const int* __restrict__ vb,
                                                                const int* __restrict__ vk, int ksv,
                                                                float* __restrict__ out, unsigned char* __restrict__ raw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OH * OW) return;
    const int oy = idx / OW, ox = idx - oy * OW;
    const int xmin = hb[2 * ox], xcnt = hb[2 * ox + 1], ymin = vb[2 * oy], ycnt = vb[2 * oy + 1];
    int acc[3] = {1 << (PIL_BITS - 1), 1 << (PIL_BITS - 1), 1 << (PIL_BITS - 1)};
    for (int y = 0; y < ycnt; ++y) {
        const unsigned char* row = frame + ((size_t)(ymin + y) * W + xmin) * 3;
        int h[3] = {1 << (PIL_BITS - 1), 1 << (PIL_BITS - 1), 1 << (PIL_BITS - 1)};
        for (int x = 0; x < xcnt; ++x) {
            const int w = hk[(size_t)ox * ksh + x];
            h[0] += row[3 * x + 0] * w; h[1] += row[3 * x + 1] * w; h[2] += row[3 * x + 2] * w;
        }
        const int wv = vk[(size_t)oy * ksv + y];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v = h[c] >> PIL_BITS;                     // the uint8 image between the two passes
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            acc[c] += v * wv;
        }
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v = acc[c] >> PIL_BITS;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        if (raw) raw[(size_t)idx * 3 + c] = (unsigned char)v;
        out[((size_t)c * OH + oy) * OW + ox] = ((float)v / 255.0f - mean[c]) / stdv[c];
    }
}

int launch_resize_normalize(const unsigned char* frame, int H, int W, int OH, int OW, const int* hb, const int* hk, int ksh,
                            const int* vb, const int* vk, int ksv, float* out, unsigned char* raw, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "resize_normalize", 0.0, (double)H * W * 3 + (double)OH * OW * (12.0 + (raw ? 3.0 : 0.0)));
    hipLaunchKernelGGL(resize_normalize_kernel, dim3((OH * OW + 255) / 256), dim3(256), 0, ctx.stream, frame, H, W, OH, OW, hb,
                       hk, ksh, vb, vk, ksv, out, raw);
    return (int)hipGetLastError();
}

}  // namespace specmi
