// stem.hip - the non-GEMM spatial kernels of the ResNet-50 trunk (gfx950).
//
//   * stem_conv7x7_kernel : Conv2d(3,64,7,s2,p3) + BatchNorm(eval) + ReLU, NCHW fp32 image in,
//     NHWC out.  (pare resnet50 conv1/bn1/relu; reference call sites spec/models/hmr.py:92,
//     camcalib/model.py:73.)  K = 147 is a poor MFMA fit and fp32 MFMA runs at the VALU rate
//     anyway, so this is a direct VALU convolution: the 21x69x3 input patch of an 8x32
//     output tile and the whole 147x64 filter bank are staged in LDS (55 KB, 2 blocks/CU);
//     a lane owns 4 horizontally adjacent output pixels x 16 output channels, so the 13 input
//     values of a (channel, filter-row) are read from LDS once and reused for 7 taps x 4
//     pixels, and the 16 filter values of a tap are a wave-uniform (broadcast) LDS read.
//   * maxpool3x3s2_kernel : MaxPool2d(3,2,1) on NHWC, float4 per lane, HBM-bound.
//   * avgpool_kernel      : AdaptiveAvgPool2d(1) on NHWC -> row-strided (B, ldo) output so the
//     pooled features land directly inside the regressor's concatenated input row.
#include "specmi_internal.h"

namespace specmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ST_TH = 8, ST_TW = 32;               // output tile (rows x cols)
constexpr int ST_PH = 2 * ST_TH + 5;               // 21 input rows
constexpr int ST_PW = 72;                          // 2*32+5 = 69 input cols, padded to 72
constexpr int ST_K = 147;
constexpr int ST_SMEM_FLOATS = ST_K * 64 + 3 * ST_PH * ST_PW;  // 9408 + 4536

__global__ void __launch_bounds__(256) stem_conv7x7_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ out,
                                                            int H, int W, int OH, int OW, int relu) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                // [147][64]
    float* patch = smem + ST_K * 64; // [3][21][72]
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * ST_TH, ox0 = blockIdx.x * ST_TW;

    for (int i = tid; i < ST_K * 64 / 4; i += 256)
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(w)[i];
    const int iy_base = oy0 * 2 - 3, ix_base = ox0 * 2 - 3;
    for (int i = tid; i < 3 * ST_PH * ST_PW; i += 256) {
        const int c = i / (ST_PH * ST_PW);
        const int rem = i - c * (ST_PH * ST_PW);
        const int rr = rem / ST_PW, cc = rem - rr * ST_PW;
        const int iy = iy_base + rr, ix = ix_base + cc;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = x[((size_t)(b * 3 + c) * H + iy) * W + ix];
        patch[i] = v;
    }
    __syncthreads();

    const int lane = tid & 63, cg = tid >> 6;  // wave = channel group of 16
    const int r = lane >> 3, q = lane & 7;     // output row in tile, pixel quad
    float acc[4][16];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[p][c] = 0.f;

    for (int c = 0; c < 3; ++c) {
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky) {
            const float* prow = patch + (c * ST_PH + 2 * r + ky) * ST_PW + 8 * q;
            float in[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(prow + 4 * j);
                in[4 * j + 0] = v[0]; in[4 * j + 1] = v[1]; in[4 * j + 2] = v[2]; in[4 * j + 3] = v[3];
            }
            const float* wrow = wl + ((c * 7 + ky) * 7) * 64 + cg * 16;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                float wv[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + kx * 64 + 4 * j);
                    wv[4 * j + 0] = v[0]; wv[4 * j + 1] = v[1]; wv[4 * j + 2] = v[2]; wv[4 * j + 3] = v[3];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int ch = 0; ch < 16; ++ch) acc[p][ch] = fmaf(in[2 * p + kx], wv[ch], acc[p][ch]);
            }
        }
    }

    const int oy = oy0 + r;
    if (oy >= OH) return;
    float sc[16], sh[16];
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) { sc[ch] = scale[cg * 16 + ch]; sh[ch] = shift[cg * 16 + ch]; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int ox = ox0 + 4 * q + p;
        if (ox >= OW) continue;
        float* o = out + ((size_t)(b * OH + oy) * OW + ox) * 64 + cg * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = fmaf(acc[p][4 * j + e], sc[4 * j + e], sh[4 * j + e]);
                v[e] = relu ? fmaxf(t, 0.f) : t;
            }
            *reinterpret_cast<f32x4*>(o + 4 * j) = v;
        }
    }
}

int launch_stem(const float* x, const float* w, const float* scale, const float* shift, float* out, int B, int H,
                int W, int OH, int OW, int relu, const LaunchCtx& ctx) {
    constexpr size_t smem = ST_SMEM_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv7x7_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((OW + ST_TW - 1) / ST_TW, (OH + ST_TH - 1) / ST_TH, B);
    const double flops = 2.0 * B * OH * OW * 64.0 * 147.0;
    const double bytes = 4.0 * ((double)B * 3 * H * W + (double)B * OH * OW * 64 + 147.0 * 64);
    ProfScope ps(ctx, "stem_conv7x7_f32", flops, bytes);
    hipLaunchKernelGGL(stem_conv7x7_kernel, grid, dim3(256), smem, ctx.stream, x, w, scale, shift, out, H, W, OH, OW,
                       relu);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                            int H, int W, int C4, int OH, int OW, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long pix = i / C4;
        const int ox = (int)(pix % OW); pix /= OW;
        const int oy = (int)(pix % OH);
        const int b = (int)(pix / OH);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = reinterpret_cast<const f32x4*>(x)[((size_t)(b * H + iy) * W + ix) * C4 + c4];
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        reinterpret_cast<f32x4*>(out)[i] = m;
    }
}

int launch_maxpool3x3s2(const float* x, float* out, int B, int H, int W, int C, int OH, int OW, const LaunchCtx& ctx) {
    if (C % 4) return (int)hipErrorInvalidValue;
    const long total = (long)B * OH * OW * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    const double bytes = 4.0 * ((double)B * H * W * C + (double)B * OH * OW * C);
    ProfScope ps(ctx, "maxpool3x3s2_f32", 0.0, bytes);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid), dim3(256), 0, ctx.stream, x, out, H, W, C / 4, OH, OW, total);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) avgpool_kernel(const float* __restrict__ x, float* __restrict__ out, int HW,
                                                       int C4, int ldo, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = i % C4, b = i / C4;
    const f32x4* src = reinterpret_cast<const f32x4*>(x) + (size_t)b * HW * C4 + c4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
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

int launch_avgpool(const float* x, float* out, int B, int HW, int C, int ldo, const LaunchCtx& ctx) {
    if (C % 4) return (int)hipErrorInvalidValue;
    const int total = B * (C / 4);
    ProfScope ps(ctx, "avgpool_f32", 0.0, 4.0 * ((double)B * HW * C + (double)B * C));
    hipLaunchKernelGGL(avgpool_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx.stream, x, out, HW, C / 4, ldo,
                       total);
    return (int)hipGetLastError();
}

}  // namespace specmi
