// stem.hip - the non-GEMM spatial kernels of the ResNet-50 trunk (gfx950).
//
//   * stem_conv7x7_kernel : Conv2d(3,64,7,s2,p3) + BatchNorm(eval) + ReLU, NCHW fp32 image in,
//     NHWC out.  (pare resnet50 conv1/bn1/relu; reference call sites spec/models/hmr.py:92,
//     camcalib/model.py:73.)  Implicit GEMM on the fp32 matrix cores with K = 147 (+1 zero), the
//     input patch and the filter bank in LDS - see the comment at the kernel.
//   * maxpool3x3s2_kernel : MaxPool2d(3,2,1) on NHWC, float4 per lane, HBM-bound.
//   * avgpool_kernel      : AdaptiveAvgPool2d(1) on NHWC -> row-strided (B, ldo) output so the
//     pooled features land directly inside the regressor's concatenated input row.
#include "specmi_internal.h"

namespace specmi {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- stem as an implicit GEMM on the fp32 matrix cores -------------------------------------------
// out[px][co] = sum_k patch[px][k] * w[k][co], k = (c, ky, kx) = 49 c + 7 ky + kx, K = 147 padded to 148.
// A persistent workgroup (4 waves) walks 8x16 output tiles; a wave owns two tile rows = 32 pixels x 64
// channels = two 32x32 accumulator tiles.  Per tile the 3 x 21 x 37 input patch is staged in LDS
// (register-staged one tile ahead, zero padding = out-of-range buffer offset); the filter bank lives in
// LDS for the workgroup's lifetime as [k pair][k half][32][co, co+32] so a lane's two B operands are one
// conflict-free ds_read_b64.  The A operand of MFMA lane (pixel m, k half h) is patch[c][2 oy + ky][2 ox + kx]:
// a ds_read_b32 at (per-lane base) + (compile-time offset of the even k); the odd k of the pair sits 1, 32
// (next filter row) or 564 (next channel) floats further, i.e. one of three per-lane base registers.
constexpr int SM_TH = 8, SM_TW = 16;          // output tile
constexpr int SM_PH = 2 * SM_TH + 5;           // 21 patch rows
constexpr int SM_PW = 38;                      // 2*16+5 = 37 patch columns, padded to an even count
constexpr int SM_CS = SM_PH * SM_PW;           // floats per patch channel
constexpr int SM_PATCH = 3 * SM_CS;            // 2394
constexpr int SM_KK = 74;                      // k pairs
constexpr int SM_WF = SM_KK * 128;             // filter floats in LDS
constexpr int SM_NLD = (SM_PATCH + 255) / 256; // patch loads per thread (10)
constexpr unsigned kStemOOB = 0x80000000u;

struct StemArgs {
    const float* x; const float* w; const float* scale; const float* shift; float* out;
    unsigned x_bytes, out_bytes;
    int B, H, W, OH, OW, relu;
    int tiles_x, tiles_y, ntiles;
    // grouped launch (gridDim.y = 2): blockIdx.y = 1 runs the stem of a second network on these tensors
    struct { const float *x, *w, *scale, *shift; float* out; } g1;
};

__global__ void __launch_bounds__(256) stem_conv7x7_kernel(const StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;              // [74][2][32][2]
    float* patch = smem + SM_WF;   // [3][21][38]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const bool grp = blockIdx.y != 0;          // wave-uniform
    const float* const px = grp ? p.g1.x : p.x;
    const float* const pw = grp ? p.g1.w : p.w;
    const float* const pscale = grp ? p.g1.scale : p.scale;
    const float* const pshift = grp ? p.g1.shift : p.shift;
    float* const pout = grp ? p.g1.out : p.out;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(pout, 0, p.out_bytes, 0x00020000);

    for (int i = tid; i < SM_WF / 4; i += 256)
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(pw)[i];

    // loader role: element idx = tid + 256 i of the patch -> (channel, row, col), fixed for the kernel's life
    int ld_rel[SM_NLD];    // element offset relative to the patch origin in the image
    int ld_rc[SM_NLD];     // row << 8 | col   (row = 255: element beyond the patch)
#pragma unroll
    for (int i = 0; i < SM_NLD; ++i) {
        const int idx = tid + 256 * i;
        const int c = idx / SM_CS, rem = idx - c * SM_CS;
        const int rr = rem / SM_PW, cc = rem - rr * SM_PW;
        ld_rel[i] = (c * p.H + rr) * p.W + cc;
        ld_rc[i] = idx < SM_PATCH ? (rr << 8 | cc) : (0x7FFF << 8);   // beyond the patch: never inside an image
    }
    float stg[SM_NLD];
    auto tile_coords = [&](int t, int& b, int& oy0, int& ox0) {
        b = t / (p.tiles_x * p.tiles_y);
        const int r = t - b * (p.tiles_x * p.tiles_y);
        const int ty = r / p.tiles_x;
        oy0 = ty * SM_TH;
        ox0 = (r - ty * p.tiles_x) * SM_TW;
    };
    auto load_patch = [&](int t) {
        int b, oy0, ox0;
        tile_coords(t, b, oy0, ox0);
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        const int base = (b * 3 * p.H + iy0) * p.W + ix0;
#pragma unroll
        for (int i = 0; i < SM_NLD; ++i) {
            const int rr = ld_rc[i] >> 8, cc = ld_rc[i] & 255;
            const bool in = (unsigned)(iy0 + rr) < (unsigned)p.H && (unsigned)(ix0 + cc) < (unsigned)p.W;
            const unsigned off = in ? (unsigned)((base + ld_rel[i]) * 4) : kStemOOB;
            stg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < SM_NLD; ++i)
            if (i < SM_NLD - 1 || tid + 256 * i < SM_PATCH) patch[tid + 256 * i] = stg[i];
    };

    // consumer role
    const int oyl = 2 * wave + (l31 >> 4), oxl = l31 & 15;
    const char* a0 = reinterpret_cast<const char*>(patch) + ((2 * oyl) * SM_PW + 2 * oxl) * 4;
    const char* ab[3] = {a0 + hh * 4, a0 + hh * (SM_PW - 6) * 4, a0 + hh * (SM_CS - 6 * SM_PW - 6) * 4};
    const char* bb = reinterpret_cast<const char*>(wl) + (hh * 32 + l31) * 8;
    const float sc0 = pscale[l31], sc1 = pscale[32 + l31], sh0 = pshift[l31], sh1 = pshift[32 + l31];
    const unsigned o_lane = (unsigned)((4 * hh) * 256 + l31 * 4);

    int t = blockIdx.x;
    if (t < p.ntiles) load_patch(t);
    for (; t < p.ntiles; t += gridDim.x) {
        __syncthreads();          // every wave is done with the previous tile's patch (and, first time, wl is written)
        store_patch();
        __syncthreads();
        const int tn = t + gridDim.x;
        if (tn < p.ntiles) load_patch(tn);   // in flight under this tile's MFMAs

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < SM_KK; ++kk) {
            const int k0 = 2 * kk;
            const int c = k0 / 49, ky = (k0 % 49) / 7, kx = k0 % 7;
            const int off0 = (c * SM_CS + ky * SM_PW + kx) * 4;
            // k0 = 146 pairs with the zero-weight k = 147: any finite in-patch value will do (delta 1)
            const int sel = (kx < 6 || kk == SM_KK - 1) ? 0 : (ky < 6 ? 1 : 2);
            const float a = *reinterpret_cast<const float*>(ab[sel] + off0);
            const f32x2 bw = *reinterpret_cast<const f32x2*>(bb + kk * 512);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[1], acc1, 0, 0, 0);
        }

        // epilogue: accumulator r of a lane = pixel (row r >> 3, col (r & 3) + 8 ((r >> 2) & 1) + 4 hh) of the wave's
        // two tile rows, channels l31 and 32 + l31
        int b, oy0, ox0;
        tile_coords(t, b, oy0, ox0);
        const int oy = oy0 + 2 * wave;
        const bool edge = ox0 + SM_TW > p.OW;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r >> 3, col = (r & 3) + 8 * ((r >> 2) & 1);
            if (oy + row >= p.OH) continue;   // wave-uniform
            const unsigned s_off = (unsigned)((((b * p.OH + oy + row) * p.OW + ox0 + col)) * 256);
            unsigned v_off = o_lane;
            if (edge && ox0 + col + 4 * hh >= p.OW) v_off = kStemOOB;
            float v0 = fmaf(acc0[r], sc0, sh0), v1 = fmaf(acc1[r], sc1, sh1);
            if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), ors, v_off, s_off, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), ors, v_off + 128, s_off, 0);
        }
    }
}

// OIHW (64, 3, 7, 7) -> [k pair kk][k half h][co & 31][co >> 5], k = 2 kk + h = 49 c + 7 ky + kx (k = 147: zero)
void pack_stem_weights(const float* w, std::vector<float>& out) {
    out.assign(SM_WF, 0.f);
    for (int n = 0; n < 64; ++n)
        for (int k = 0; k < 147; ++k) out[((k >> 1) * 2 + (k & 1)) * 64 + (n & 31) * 2 + (n >> 5)] = w[n * 147 + k];
}

int launch_stem(const float* x, const float* w, const float* scale, const float* shift, float* out, int B, int H,
                int W, int OH, int OW, int relu, const LaunchCtx& ctx, const StemPair* pair) {
    constexpr size_t smem = (SM_WF + SM_PATCH) * sizeof(float);
    static DevOnce once;
    if (int e = set_dyn_lds_once(once, reinterpret_cast<const void*>(&stem_conv7x7_kernel), (int)smem)) return e;
    // 32-bit buffer offsets on both sides: split the batch if a side reaches 2 GiB
    const size_t in_img = (size_t)3 * H * W * 4, out_img = (size_t)OH * OW * 64 * 4;
    const size_t limit = (size_t)1 << 31;
    const size_t big = in_img > out_img ? in_img : out_img;
    if (big >= limit) return (int)hipErrorInvalidValue;
    const int max_b = (int)((limit - 1) / big);
    if (pair && B > max_b) return (int)hipErrorInvalidValue;
    const int groups = pair ? 2 : 1;
    const double flops = 2.0 * B * OH * OW * 64.0 * 147.0 * groups;
    const double bytes = 4.0 * ((double)B * 3 * H * W + (double)B * OH * OW * 64 + 147.0 * 64) * groups;
    ProfScope ps(ctx, "stem_conv7x7_f32", flops, bytes);
    for (int b0 = 0; b0 < B; b0 += max_b) {
        StemArgs a;
        a.B = (B - b0 < max_b) ? B - b0 : max_b;
        a.x = x + (size_t)b0 * 3 * H * W; a.w = w; a.scale = scale; a.shift = shift;
        a.out = out + (size_t)b0 * OH * OW * 64;
        a.x_bytes = (unsigned)(in_img * a.B); a.out_bytes = (unsigned)(out_img * a.B);
        a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.relu = relu;
        a.tiles_x = (OW + SM_TW - 1) / SM_TW; a.tiles_y = (OH + SM_TH - 1) / SM_TH;
        a.ntiles = a.B * a.tiles_x * a.tiles_y;
        a.g1.x = pair ? pair->x : nullptr; a.g1.w = pair ? pair->w : nullptr; a.g1.scale = pair ? pair->scale : nullptr;
        a.g1.shift = pair ? pair->shift : nullptr; a.g1.out = pair ? pair->out : nullptr;
        const int slots = 768 / groups;                     // 3 workgroups per CU (47 KB LDS each), persistent
        const int grid = a.ntiles < slots ? a.ntiles : slots;
        hipLaunchKernelGGL(stem_conv7x7_kernel, dim3(grid, groups), dim3(256), smem, ctx.stream, a);
        const int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// One lane per (output pixel, 4 channels).  The 9 taps are loaded UNCONDITIONALLY from coordinates clamped into the image
// (a clamped tap repeats a tap that is already in the window, so the maximum is unchanged): `if (outside) continue;`
// compiles to one exec-masked branch and one s_waitcnt per load, i.e. nine memory latencies in a row.  Index arithmetic
// is 32-bit (the launcher checks the sizes).  Workgroups are renumbered so that each XCD (workgroup id mod 8, the
// dispatch order) owns one contiguous eighth of the output: the 3x3 / stride-2 windows of neighbouring output rows share
// input rows, and that re-use then hits the XCD's own L2 instead of fetching the row again through another one.
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const float* __restrict__ x0, float* __restrict__ out0,
                                                            int H, int W, int C4, int OH, int OW, unsigned total,
                                                            const float* __restrict__ x1, float* __restrict__ out1) {
    const float* const x = blockIdx.y ? x1 : x0;       // grouped launch: blockIdx.y = 1 pools a second network's map
    float* const out = blockIdx.y ? out1 : out0;
    const unsigned nb = gridDim.x, per = nb >> 3;
    const unsigned blk = blockIdx.x < per * 8 ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned i = blk * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned c4 = i % (unsigned)C4;
    unsigned pix = i / (unsigned)C4;
    const int ox = (int)(pix % (unsigned)OW); pix /= (unsigned)OW;
    const int oy = (int)(pix % (unsigned)OH);
    const unsigned b = pix / (unsigned)OH;
    const f32x4* img = reinterpret_cast<const f32x4*>(x) + (size_t)b * H * W * C4 + c4;
    f32x4 v[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = min(max(oy * 2 - 1 + ky, 0), H - 1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = min(max(ox * 2 - 1 + kx, 0), W - 1);
            v[ky * 3 + kx] = img[(unsigned)(iy * W + ix) * (unsigned)C4];
        }
    }
    f32x4 m = v[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) { m[0] = fmaxf(m[0], v[k][0]); m[1] = fmaxf(m[1], v[k][1]); m[2] = fmaxf(m[2], v[k][2]); m[3] = fmaxf(m[3], v[k][3]); }
    reinterpret_cast<f32x4*>(out)[i] = m;
}

int launch_maxpool3x3s2(const float* x, float* out, int B, int H, int W, int C, int OH, int OW, const LaunchCtx& ctx,
                        const float* x1, float* out1) {
    if (C % 4) return (int)hipErrorInvalidValue;
    const long total = (long)B * OH * OW * (C / 4);
    if (total >= (1L << 31) - 256 || (long)H * W * (C / 4) >= (1L << 31)) return (int)hipErrorInvalidValue;   // 32-bit indices
    const int groups = x1 ? 2 : 1;
    const double bytes = 4.0 * ((double)B * H * W * C + (double)B * OH * OW * C) * groups;
    ProfScope ps(ctx, "maxpool3x3s2_f32", 0.0, bytes);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)((total + 255) / 256), groups), dim3(256), 0, ctx.stream, x, out, H, W, C / 4, OH, OW,
                       (unsigned)total, x1, out1);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// workgroups past the pooling ones (nblk_pool .. nblk_pool + B - 1, only launched with an IEF head behind the pool) write the
// state columns of one row each: head_init's work without its graph node
__global__ void __launch_bounds__(256) avgpool_kernel(const float* __restrict__ x, float* __restrict__ out, int HW,
                                                       int C4, int ldo, int total, int nblk_pool, const HeadInit init) {
    if ((int)blockIdx.x >= nblk_pool) {
        head_init_row(init, blockIdx.x - nblk_pool, threadIdx.x, blockDim.x);
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = i % C4, b = i / C4;
    const f32x4* src = reinterpret_cast<const f32x4*>(x) + (size_t)b * HW * C4 + c4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 7   // independent loads, one accumulation order: several rows in flight (latency-bound at small batch)
    for (int p = 0; p < HW; ++p) {
        const f32x4 v = src[(size_t)p * C4];
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const float inv = 1.0f / (float)HW;
    float* o = out + (size_t)b * ldo + c4 * 4;
    // PyTorch's mean is sum / count
    o[0] = s[0] / (float)HW; o[1] = s[1] / (float)HW; o[2] = s[2] / (float)HW; o[3] = s[3] / (float)HW;
    (void)inv;
}

// Large maps at small batch (CamCalib on a full frame: 19 x 34 = 646 positions, one image): one thread per (image, channel quad)
// walks all positions serially - 37 us for 5 MB.  Here a workgroup = 64 channel quads x P parts of the map (P a function of HW ALONE,
// so the summation order - part sums left to right, parts folded in order - never depends on the batch); P = 1 is the kernel above
// (224^2 crops: HW = 49, bits unchanged).
__global__ void __launch_bounds__(1024) avgpool_parts_kernel(const float* __restrict__ x, float* __restrict__ out, int HW,
                                                             int C4, int ldo, int parts, int chunk) {
    __shared__ f32x4 red[16][64];
    const int lane = threadIdx.x, part = threadIdx.y;
    const int c4 = blockIdx.x * 64 + lane, b = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(x) + (size_t)b * HW * C4 + c4;
        const int p0 = part * chunk, p1 = min(HW, p0 + chunk);
#pragma unroll 8
        for (int p = p0; p < p1; ++p) {
            const f32x4 v = src[(size_t)p * C4];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
    red[part][lane] = s;
    __syncthreads();
    if (part == 0 && c4 < C4) {
        f32x4 t = red[0][lane];
        for (int q = 1; q < parts; ++q) {
            const f32x4 v = red[q][lane];
            t[0] += v[0]; t[1] += v[1]; t[2] += v[2]; t[3] += v[3];
        }
        float* o = out + (size_t)b * ldo + c4 * 4;
        o[0] = t[0] / (float)HW; o[1] = t[1] / (float)HW; o[2] = t[2] / (float)HW; o[3] = t[3] / (float)HW;
    }
}

int launch_avgpool(const float* x, float* out, int B, int HW, int C, int ldo, const LaunchCtx& ctx, const HeadInit* init, bool* init_done) {
    if (C % 4) return (int)hipErrorInvalidValue;
    if (init_done) *init_done = false;
    const int total = B * (C / 4);
    ProfScope ps(ctx, "avgpool_f32", 0.0, 4.0 * ((double)B * HW * C + (double)B * C));
    int parts = HW / 32;                       // a function of the map size only
    parts = parts < 1 ? 1 : (parts > 16 ? 16 : parts);
    if (parts == 1) {
        const int nblk = (total + 255) / 256;
        const bool fuse = init && init_done;
        hipLaunchKernelGGL(avgpool_kernel, dim3(nblk + (fuse ? B : 0)), dim3(256), 0, ctx.stream, x, out, HW, C / 4, ldo, total, nblk,
                           fuse ? *init : HeadInit{});
        if (fuse) *init_done = true;
    } else {
        const int C4 = C / 4, chunk = (HW + parts - 1) / parts;
        hipLaunchKernelGGL(avgpool_parts_kernel, dim3((C4 + 63) / 64, B), dim3(64, parts), 0, ctx.stream, x, out, HW, C4, ldo, parts, chunk);
    }
    return (int)hipGetLastError();
}

}  // namespace specmi
