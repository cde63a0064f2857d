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

// Work mapping shared by the two crop kernels: a workgroup covers 256 * kPX consecutive output pixels of one crop as
// kPX chunks of 256; lane t handles pixel chunk * 256 + t of each chunk.  Neighbouring lanes therefore read neighbouring
// source pixels (a wave's gather touches 2-5 cache lines, not the 10-20 a "4 consecutive pixels per lane" mapping does:
// the L1 tag rate, not HBM, bounds a gather) and write 256 contiguous bytes per colour plane.  All taps of the kPX
// pixels are loaded back to back, unconditionally, from clamped (always valid) addresses and zero-weighted afterwards
// where they leave the frame - predicated loads would each sit in their own exec-masked branch and serialise on the
// memory latency.  The two horizontally adjacent taps of a row (6 contiguous bytes) come from ONE unaligned 8-byte load.
constexpr int kPX = 4;

// RGB of the pixels c0 and c1 (0 <= c0 <= c1 <= c0 + 1 <= W - 1) of frame row cy.  WIDE needs W >= 2 and total >= 8 bytes.
template <bool WIDE>
__device__ __forceinline__ void load_tap_pair(const unsigned char* __restrict__ frame, size_t total, int W, int cy, int c0,
                                              int c1, int (&t0)[3], int (&t1)[3]) {
    // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): the launchers guarantee H, W < 2^24 and H*W*3 < 2^32
    if (WIDE) {
        const int bx = min(c0, W - 2);
        const unsigned addr = (__umul24((unsigned)cy, (unsigned)W) + (unsigned)bx) * 3u;
        const unsigned base = min(addr, (unsigned)total - 8u);          // never read past the end of the frame: shift instead
        unsigned long long w;
        __builtin_memcpy(&w, frame + base, 8);
        const unsigned sh = 8u * (addr - base);                          // <= 2 bytes, so 6 valid bytes (two pixels) remain
        const unsigned a = (unsigned)(w >> (sh + 24u * (unsigned)(c0 - bx))) & 0xFFFFFFu;
        const unsigned b = (unsigned)(w >> (sh + 24u * (unsigned)(c1 - bx))) & 0xFFFFFFu;
#pragma unroll
        for (int c = 0; c < 3; ++c) { t0[c] = (a >> (8 * c)) & 255; t1[c] = (b >> (8 * c)) & 255; }
    } else {
        const unsigned row = __umul24((unsigned)cy, (unsigned)W);
        const unsigned char* q0 = frame + (row + (unsigned)c0) * 3u;
        const unsigned char* q1 = frame + (row + (unsigned)c1) * 3u;
#pragma unroll
        for (int c = 0; c < 3; ++c) { t0[c] = q0[c]; t1[c] = q1[c]; }
    }
}

// cv::warpAffine's inverse map of get_single_image_crop_demo's affine (rot = 0: M1 = M3 = +-0, or NaN for a degenerate
// box - in both cases X does not depend on y nor Y on x, so the source coordinate is separable)
struct CropAffine {
    double M0, M1, M2, M3, M4, M5;
    __device__ CropAffine(float cx, float cy, float bw, float bh, float scale, int S) {
        // forward affine (src -> dst), then cv::warpAffine's inversion, all in fp64
        const float sw = (bw * scale) * 0.5f, sh = (bh * scale) * 0.5f;
        const double half = (double)((float)S * 0.5f);
        const double ax = half / (double)sw, ay = half / (double)sh;
        M0 = ax; M1 = 0.0; M2 = half - ax * (double)cx; M3 = 0.0; M4 = ay; M5 = half - ay * (double)cy;
        double D = M0 * M4 - M1 * M3;
        D = D != 0.0 ? 1.0 / D : 0.0;
        const double A11 = M4 * D, A22 = M0 * D;
        M0 = A11; M1 *= -D; M3 *= -D; M4 = A22;
        const double b1 = -M0 * M2 - M1 * M5, b2 = -M3 * M2 - M4 * M5;
        M2 = b1; M5 = b2;
    }
    // (integer source coordinate clamped to [-2, n] - enough to keep the border flags and the clamped taps -, 1/32 fraction)
    __device__ int2 col(int x, int W) const {     // X = (round((M1*y + M2)*1024) + 16 + round(M0*x*1024)) >> 5 at y = 0
        const long long X = (__double2ll_rn((M1 * 0 + M2) * 1024.0) + 16 + __double2ll_rn(M0 * x * 1024.0)) >> 5;
        return make_int2((int)min(max(X >> 5, -2LL), (long long)W), (int)(X & 31));
    }
    __device__ int2 row(int y, int H) const {     // Y = (round((M4*y + M5)*1024) + 16 + round(M3*x*1024)) >> 5 at x = 0
        const long long Y = (__double2ll_rn((M4 * y + M5) * 1024.0) + 16 + __double2ll_rn(M3 * 0 * 1024.0)) >> 5;
        return make_int2((int)min(max(Y >> 5, -2LL), (long long)H), (int)(Y & 31));
    }
};

constexpr int kTabX = 2048;                    // crops up to 2048 x 2048 use the per-workgroup coordinate tables
constexpr int kNB = 2;                        // batches of kPX pixels per lane: a workgroup covers 256 * kPX * kNB pixels
constexpr int kTabY = 256 * kPX * kNB + 2;     // rows a workgroup's pixels can touch (S = 1)

// ToTensor + Normalize of a uint8 value takes two IEEE divisions; the 3 x 256 possible results are tabulated in LDS
// once per workgroup with exactly those divisions.  The separable source coordinates (fp64 -> 1/1024 px fixed point) are
// tabulated too: S columns + the few rows of this workgroup instead of 4 fp64->int64 conversions per pixel.  The
// per-pixel path is integer arithmetic + three LDS reads + two 8-byte loads.
template <bool WIDE, bool TABLE>
__global__ void __launch_bounds__(256) crop_normalize_kernel(const unsigned char* __restrict__ frame, int H, int W,
                                                              const float* __restrict__ bboxes, float scale, int S,
                                                              float* __restrict__ out, unsigned char* __restrict__ raw,
                                                              float* __restrict__ bbox_scale,
                                                              float* __restrict__ bbox_center,
                                                              const int* __restrict__ frame_of, size_t frame_stride, int nframes) {
    __shared__ float lut[3][256];
    __shared__ int2 tabx[TABLE ? kTabX : 1], taby[TABLE ? kTabY : 1];
    const int d = blockIdx.y, t = threadIdx.x;
    // batched: crop d is cut from frame frame_of[d] of a slab of equal-sized frames.  The index comes from caller memory the host
    // side cannot check without a synchronisation: a stale / negative / too large value is clamped into the slab (a wrong crop,
    // never an out-of-bounds read); spec_amd.preprocess.crop_detections_batch rejects it while the index is still on the host
    if (frame_of) frame += (size_t)min((unsigned)max(frame_of[d], 0), (unsigned)(nframes - 1)) * frame_stride;
    {
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
        for (int c = 0; c < 3; ++c) lut[c][t] = ((float)t / 255.0f - mean[c]) / stdv[c];
    }
    const float cx = bboxes[d * 4 + 0], cy = bboxes[d * 4 + 1], bw = bboxes[d * 4 + 2], bh = bboxes[d * 4 + 3];
    if (blockIdx.x == 0 && t == 0) {
        if (bbox_scale) bbox_scale[d] = bw / 200.0f;
        if (bbox_center) { bbox_center[d * 2 + 0] = cx; bbox_center[d * 2 + 1] = cy; }
    }
    const CropAffine A(cx, cy, bw, bh, scale, S);
    const int npix = S * S, pix0 = blockIdx.x * (256 * kPX * kNB);
    const int yfirst = pix0 / S;
    if (TABLE) {
        const int ylast = min(pix0 + 256 * kPX * kNB - 1, npix - 1) / S;
        for (int i = t; i < S; i += 256) tabx[i] = A.col(i, W);
        for (int i = t; i <= ylast - yfirst; i += 256) taby[i] = A.row(yfirst + i, H);
    }
    __syncthreads();
    const size_t total = (size_t)H * W * 3;
#pragma unroll 1
    for (int kb = 0; kb < kNB; ++kb) {
        const int idx0 = pix0 + kb * (256 * kPX) + t;
        if (idx0 - t >= npix) break;
        // phase 1: taps
        int tap[kPX][4][3], wgt[kPX][4];
#pragma unroll
        for (int k = 0; k < kPX; ++k) {
            const int idx = min(idx0 + k * 256, npix - 1);     // a tail lane recomputes the last pixel and never stores it
            const int y = idx / S, x = idx - y * S;
            const int2 cxf = TABLE ? tabx[x] : A.col(x, W), cyf = TABLE ? taby[y - yfirst] : A.row(y, H);
            const int sx = cxf.x, fx = cxf.y, sy = cyf.x, fy = cyf.y;
            const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
            const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
            wgt[k][0] = (y0 && x0) ? __mul24(32 - fx, 32 - fy) * 32 : 0;      // a zero weight == a zero (border) pixel
            wgt[k][1] = (y0 && x1) ? __mul24(fx, 32 - fy) * 32 : 0;
            wgt[k][2] = (y1 && x0) ? __mul24(32 - fx, fy) * 32 : 0;
            wgt[k][3] = (y1 && x1) ? __mul24(fx, fy) * 32 : 0;
            const int cy0 = min(max(sy, 0), H - 1), cy1 = min(max(sy + 1, 0), H - 1);
            const int cx0 = min(max(sx, 0), W - 1), cx1 = min(max(sx + 1, 0), W - 1);
            load_tap_pair<WIDE>(frame, total, W, cy0, cx0, cx1, tap[k][0], tap[k][1]);
            load_tap_pair<WIDE>(frame, total, W, cy1, cx0, cx1, tap[k][2], tap[k][3]);
        }
        // phase 2: fixed-point blend, table look-up, stores (256 contiguous bytes per wave and colour plane)
#pragma unroll
        for (int k = 0; k < kPX; ++k) {
            const int idx = idx0 + k * 256;
            if (idx < npix) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    int v = (__mul24(tap[k][0][c], wgt[k][0]) + __mul24(tap[k][1][c], wgt[k][1]) + __mul24(tap[k][2][c], wgt[k][2]) +
                             __mul24(tap[k][3][c], wgt[k][3]) + (1 << 14)) >> 15;   // taps < 2^8, weights <= 2^15
                    v = v < 0 ? 0 : (v > 255 ? 255 : v);
                    if (raw) raw[((size_t)d * npix + idx) * 3 + c] = (unsigned char)v;
                    out[(size_t)(d * 3 + c) * npix + idx] = lut[c][v];
                }
            }
        }
    }
}

template <bool WIDE, bool TABLE>
static void crop_normalize_go(dim3 grid, hipStream_t st, const unsigned char* frame, int H, int W, const float* bboxes, float scale,
                              int S, float* out, unsigned char* raw, float* bbox_scale, float* bbox_center, const int* frame_of,
                              size_t frame_stride, int nframes) {
    hipLaunchKernelGGL((crop_normalize_kernel<WIDE, TABLE>), grid, dim3(256), 0, st, frame, H, W, bboxes, scale, S, out, raw, bbox_scale,
                       bbox_center, frame_of, frame_stride, nframes);
}

int launch_crop_normalize(const unsigned char* frame, int H, int W, const float* bboxes, int n, float scale, int S,
                          float* out, unsigned char* raw, float* bbox_scale, float* bbox_center, const LaunchCtx& ctx,
                          const int* frame_of, int nframes) {
    // algorithmic HBM bytes: every frame once + every output once (the 4 taps x 3 channels of a pixel come through L2)
    ProfScope ps(ctx, "crop_normalize", 0.0, (double)H * W * 3 * (frame_of ? nframes : 1) + (double)n * S * S * (12.0 + (raw ? 3.0 : 0.0)));
    if (H >= (1 << 24) || W >= (1 << 24) || (double)H * W * 3 >= 4294967296.0) return (int)hipErrorInvalidValue;   // 32-bit offsets
    const dim3 grid((S * S + 256 * kPX * kNB - 1) / (256 * kPX * kNB), n);
    const bool wide = W >= 2 && (size_t)H * W * 3 >= 8, table = S <= kTabX;
    auto go = wide ? (table ? crop_normalize_go<true, true> : crop_normalize_go<true, false>)
                   : (table ? crop_normalize_go<false, true> : crop_normalize_go<false, false>);
    if (frame_of && nframes < 1) return (int)hipErrorInvalidValue;
    go(grid, ctx.stream, frame, H, W, bboxes, scale, S, out, raw, bbox_scale, bbox_center, frame_of, (size_t)H * W * 3, nframes);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Evaluation-dataset crop (spec/dataset/cam_dataset.py:253-287,367-377: rgb_processing -> pare `crop` -> ToTensor ->
// Normalize).  PARE / SPIN `crop(img, center, scale, res)` copies the integer box [ul, br) (200*scale pixels around the
// centre, zero outside the frame) and scales it to res x res with cv2.resize (INTER_LINEAR on a float64 array: half-pixel
// centres, replicated border, float coefficients, double accumulation); rgb_processing clips to [0, 255], converts to
// float32 / 255 and the dataset normalises with the ImageNet mean / std.  The integer boxes come from the host (the
// reference computes them with a 3x3 float64 inverse; spec_amd/preprocess.py restates that), everything per pixel runs here.
// cv::resize's source index / fraction of destination index i along an axis of n source pixels (clamped to the box)
__device__ __forceinline__ void resize_coord(int i, double scale, int n, int& s0, float& f) {
    f = (float)(((double)i + 0.5) * scale - 0.5);
    s0 = (int)floorf(f);
    f -= (float)s0;
    if (s0 < 0) { f = 0.f; s0 = 0; }
    if (s0 >= n - 1) { f = 0.f; s0 = n - 1; }
}

template <bool WIDE, bool TABLE>
__global__ void __launch_bounds__(256) crop_resize_normalize_kernel(const unsigned char* __restrict__ frame, int H, int W,
                                                                     const int* __restrict__ boxes, int S,
                                                                     float* __restrict__ out) {
    __shared__ int txs[TABLE ? kTabX : 1], tys[TABLE ? kTabY : 1];
    __shared__ float txf[TABLE ? kTabX : 1], tyf[TABLE ? kTabY : 1];
    const int d = blockIdx.y, t = threadIdx.x;
    const int npix = S * S, pix0 = blockIdx.x * (256 * kPX * kNB);     // same mapping as crop_normalize_kernel
    const int ulx = boxes[d * 4 + 0], uly = boxes[d * 4 + 1], brx = boxes[d * 4 + 2], bry = boxes[d * 4 + 3];
    const int bw = brx - ulx, bh = bry - uly;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    if (bw <= 0 || bh <= 0) {
        for (int i = pix0 + t; i < min(pix0 + 256 * kPX * kNB, npix); i += 256) {
#pragma unroll
            for (int c = 0; c < 3; ++c) out[(size_t)(d * 3 + c) * npix + i] = (0.0f - mean[c]) / stdv[c];
        }
        return;
    }
    // cv::resize INTER_LINEAR: fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx, clamped to the box
    const double scale_x = (double)bw / (double)S, scale_y = (double)bh / (double)S;
    const int yfirst = pix0 / S;
    if (TABLE) {       // the coordinates are separable: S columns + this workgroup's few rows, once
        const int ylast = min(pix0 + 256 * kPX * kNB - 1, npix - 1) / S;
        for (int i = t; i < S; i += 256) resize_coord(i, scale_x, bw, txs[i], txf[i]);
        for (int i = t; i <= ylast - yfirst; i += 256) resize_coord(yfirst + i, scale_y, bh, tys[i], tyf[i]);
        __syncthreads();
    }
    const size_t total = (size_t)H * W * 3;
#pragma unroll 1
    for (int kb = 0; kb < kNB; ++kb) {
        const int idx0 = pix0 + kb * (256 * kPX) + t;
        if (idx0 - t >= npix) break;
        // phase 1: the 4 taps x 3 channels of kPX pixels
        int tap[kPX][4][3];
        bool in[kPX][4];
        float cf[kPX][2];     // fx, fy
#pragma unroll
        for (int k = 0; k < kPX; ++k) {
            const int idx = min(idx0 + k * 256, npix - 1);
            const int dy = idx / S, dx = idx - dy * S;
            int sx, sy;
            float fx, fy;
            if (TABLE) { sx = txs[dx]; fx = txf[dx]; sy = tys[dy - yfirst]; fy = tyf[dy - yfirst]; }
            else { resize_coord(dx, scale_x, bw, sx, fx); resize_coord(dy, scale_y, bh, sy, fy); }
            const int sx1 = sx + 1 < bw ? sx + 1 : sx, sy1 = sy + 1 < bh ? sy + 1 : sy;
            cf[k][0] = fx; cf[k][1] = fy;
            // box pixel -> frame pixel (zero where the box leaves the frame)
            const int iy0 = uly + sy, iy1 = uly + sy1, ix0 = ulx + sx, ix1 = ulx + sx1;
            const bool vy0 = (unsigned)iy0 < (unsigned)H, vy1 = (unsigned)iy1 < (unsigned)H;
            const bool vx0 = (unsigned)ix0 < (unsigned)W, vx1 = (unsigned)ix1 < (unsigned)W;
            in[k][0] = vy0 && vx0; in[k][1] = vy0 && vx1; in[k][2] = vy1 && vx0; in[k][3] = vy1 && vx1;
            const int cx0 = min(max(ix0, 0), W - 1), cx1 = min(max(ix1, 0), W - 1);
            load_tap_pair<WIDE>(frame, total, W, min(max(iy0, 0), H - 1), cx0, cx1, tap[k][0], tap[k][1]);
            load_tap_pair<WIDE>(frame, total, W, min(max(iy1, 0), H - 1), cx0, cx1, tap[k][2], tap[k][3]);
        }
        // phase 2: float64 blend exactly as cv::resize orders it, clip, / 255, Normalize, store
#pragma unroll
        for (int k = 0; k < kPX; ++k) {
            const int idx = idx0 + k * 256;
            if (idx < npix) {
                const double a0 = (double)(1.f - cf[k][0]), a1 = (double)cf[k][0], b0 = (double)(1.f - cf[k][1]), b1 = (double)cf[k][1];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double p00 = in[k][0] ? (double)tap[k][0][c] : 0.0, p01 = in[k][1] ? (double)tap[k][1][c] : 0.0;
                    const double p10 = in[k][2] ? (double)tap[k][2][c] : 0.0, p11 = in[k][3] ? (double)tap[k][3][c] : 0.0;
                    // separate multiplies and adds (no fma contraction): the reference rounds each product
                    const double r0 = __dadd_rn(__dmul_rn(p00, a0), __dmul_rn(p01, a1));     // horizontal pass
                    const double r1 = __dadd_rn(__dmul_rn(p10, a0), __dmul_rn(p11, a1));
                    double v = __dadd_rn(__dmul_rn(r0, b0), __dmul_rn(r1, b1));             // vertical pass
                    v = fmin(255.0, fmax(0.0, v));                                          // rgb_processing: pn = 1, clip
                    const float tt = (float)v / 255.0f;                                     // astype('float32') / 255.0
                    out[(size_t)(d * 3 + c) * npix + idx] = (tt - mean[c]) / stdv[c];
                }
            }
        }
    }
}

template <bool WIDE, bool TABLE>
static void crop_resize_go(dim3 grid, hipStream_t st, const unsigned char* frame, int H, int W, const int* boxes, int S, float* out) {
    hipLaunchKernelGGL((crop_resize_normalize_kernel<WIDE, TABLE>), grid, dim3(256), 0, st, frame, H, W, boxes, S, out);
}

int launch_crop_resize_normalize(const unsigned char* frame, int H, int W, const int* boxes, int n, int S, float* out,
                                 const LaunchCtx& ctx) {
    ProfScope ps(ctx, "crop_resize_normalize", 0.0, (double)H * W * 3 + (double)n * S * S * 12.0);
    if (H >= (1 << 24) || W >= (1 << 24) || (double)H * W * 3 >= 4294967296.0) return (int)hipErrorInvalidValue;   // 32-bit offsets
    const dim3 grid((S * S + 256 * kPX * kNB - 1) / (256 * kPX * kNB), n);
    const bool wide = W >= 2 && (size_t)H * W * 3 >= 8, table = S <= kTabX;
    auto go = wide ? (table ? crop_resize_go<true, true> : crop_resize_go<true, false>)
                   : (table ? crop_resize_go<false, true> : crop_resize_go<false, false>);
    go(grid, ctx.stream, frame, H, W, boxes, S, out);
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
