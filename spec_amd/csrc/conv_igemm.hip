// conv_igemm.hip - implicit-GEMM convolution / linear layer on the gfx950 fp32 matrix cores.
//
// One kernel family serves every contraction of the SPEC hot path except the 7x7 stem:
//   * the 36 1x1 and 16 3x3 convolutions of each ResNet-50 trunk (reference call sites
//     spec/models/hmr.py:92, camcalib/model.py:73) with BatchNorm (eval) folded into a
//     per-channel scale/shift epilogue, optional residual add and ReLU fused,
//   * the FC layers of the CamCalib heads (camcalib/model.py:77-79) and of the HMR iterative
//     regressor (spec/models/hmr.py:96), as H=W=1 "convolutions".
//
// GEMM view: out[M = B*OH*OW][N = Cout] = A[M][K = KH*KW*Cin] * Wp[K][N], activations NHWC so
// that for a fixed filter tap the Cin slice of a pixel is contiguous.
//
// MI355X mapping (CDNA4, wave64):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cycles, 16 acc VGPRs.
//     A workgroup = 4 waves in a 2x2 grid; a wave owns (BM/2)x(BN/2) = TMxTN tiles of 32x32.
//   * K is consumed in chunks of 32 (one filter tap x 32 channels).  Inside a chunk the k
//     order is permuted so that each lane's 4 consecutive k values are one 16-byte LDS read:
//     sub-chunk q (8 k's), lane half h, step s  ->  k = 8q + 4h + s.  A and B use the same
//     permutation, so the sum over k is unchanged.
//   * LDS: A tile [BM][32+4] fp32 (row pad 4 floats => ds_read_b128 of 16 rows hits 64
//     distinct banks, ds_write_b128 of one row's 8 quads hits 32 distinct banks);
//     B tile [8][BN][4] fp32 - exactly the HBM layout of the packed weights [K/4][Npad][4],
//     so the copy is linear and the fragment read (consecutive n per lane) is conflict-free.
//   * double-buffered LDS, register-staged prefetch of chunk c+1 issued before the MFMAs of
//     chunk c (one barrier per chunk); 2 workgroups / CU (<= 69.6 KB LDS, <= 256 VGPR).
//   * blockIdx -> tile map is XCD-aware: the 8 XCDs get contiguous runs of tiles ordered
//     n-fastest, so tiles sharing an A row-panel hit the same 4 MiB L2.
#include "specmi_internal.h"

namespace specmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct KArgs {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int H, W, ldx;
    int OW, OHW, Cout, Npad, ldo;
    int KW, stride, pad;
    int M, nbn, nchunks, cpc;  // cpc = chunks per filter tap = Cin / 32
    int relu;
    int vec_ok;  // out/res rows are 16-byte aligned: float4 epilogue traffic allowed
};

template <int BM, int BN>
__global__ void __launch_bounds__(256, 2) conv_igemm_f32_kernel(const KArgs p) {
    constexpr int BK = 32;
    constexpr int LDA = BK + 4;
    constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave
    constexpr int AI = BM / 32, BI = BN / 32;  // float4 loads per thread per chunk
    constexpr int A_STAGE = BM * LDA, B_STAGE = 8 * BN * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;

    // ---- XCD-aware tile order (bijective for any grid size) ------------------------------
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = L / p.nbn, tile_n = L - tile_m * p.nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread im2col rows ------------------------------------------------------------
    const int a_kq = tid & 7, a_r = tid >> 3;
    int a_pix[AI], a_iy[AI], a_ix[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + a_r + 32 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / p.OHW;
        const int rem = mm - b * p.OHW;
        const int oy = rem / p.OW;
        const int ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_iy[i] = ok ? iy0 : -(1 << 20);  // rows past M read as zeros
        a_ix[i] = ix0;
        a_pix[i] = (b * p.H + iy0) * p.W + ix0;
    }

    f32x4 ra[AI], rb[BI];
    unsigned a_okmask = 0;
    // Loads are unconditional (an out-of-image tap or a row past M reads pixel 0 and is zeroed
    // when it is written to LDS): no exec-masked branches, so the 2*AI/BI loads of a chunk are
    // issued back to back and stay in flight under the MFMAs.
    auto load_chunk = [&](int c) {
        const int tap = c / p.cpc;
        const int c0 = (c - tap * p.cpc) * BK;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        a_okmask = 0;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const long long pix = ok ? (long long)(a_pix[i] + ky * p.W + kx) : 0ll;
            ra[i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)pix * p.ldx + c0 + a_kq * 4);
            a_okmask |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int e = tid + 256 * i;
            const int kq = e / BN, n = e % BN;
            rb[i] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)(c * 8 + kq) * p.Npad + n0 + n) * 4);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            f32x4 v = ra[i];
            if (!((a_okmask >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&As[buf * A_STAGE + (a_r + 32 * i) * LDA + a_kq * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<f32x4*>(&Bs[buf * B_STAGE + (tid + 256 * i) * 4]) = rb[i];
    };

    // ---- wave / lane coordinates -------------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- epilogue coordinates (known up front so the residual can be prefetched) ------------
    constexpr int LDC = BN + 4;
    constexpr int QPR = BN / 4;         // float4 quads per tile row
    constexpr int RPP = 256 / QPR;      // rows per pass
    constexpr int NP = BM / RPP;        // passes over the tile rows
    const int cq = tid % QPR, r0 = tid / QPR;
    const int n = n0 + cq * 4;
    const bool full = (n + 3 < p.Cout) && p.vec_ok;
    f32x4 rr[NP];

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    auto compute = [&](int buf) {
        const float* Ab = As + buf * A_STAGE + (wm * (BM / 2) + l31) * LDA + hh * 4;
        const float* Bb = Bs + buf * B_STAGE + (hh * BN + wn * (BN / 2) + l31) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDA + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(Bb + (q * 2 * BN + j * 32) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    };

    const int last = p.nchunks - 1;
    for (int c = 0; c < last; ++c) {
        load_chunk(c + 1);          // global loads in flight under the MFMAs below
        __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the loads next to their use
        compute(c & 1);
        __builtin_amdgcn_sched_barrier(0);
        store_chunk((c + 1) & 1);
        __syncthreads();
    }
    // last chunk: nothing left to stage - fetch the residual rows of the epilogue under its MFMAs
    if (p.res && full) {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int m = m0 + r0 + ps * RPP;
            const size_t o = (m < p.M) ? (size_t)m * p.ldo + n : (size_t)n;
            rr[ps] = *reinterpret_cast<const f32x4*>(p.res + o);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(last & 1);
    __syncthreads();

    // ---- epilogue -------------------------------------------------------------------------
    // The accumulators go through LDS once so that the HBM side is whole-row traffic: the
    // MFMA C/D layout (col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)) would give
    // 4-byte stores at a row stride; after the transpose every lane moves 16 contiguous bytes
    // and a wave covers 2 (BN=128) or 4 (BN=64) full output rows per instruction, for the
    // store, the residual read and the scale/shift fetch alike.
    // All waves have passed the loop's last barrier, so the A/B stages are free to reuse.
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                Cs[row * LDC + wn * (BN / 2) + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();

    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
    if (full) {
        // the residual rows were fetched under the last chunk's MFMAs (rr)
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int row = r0 + ps * RPP;
            const int m = m0 + row;
            const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + row * LDC + cq * 4);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(a[e], sc[e], sh[e]);
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rr[ps][e];
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (m < p.M) *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.ldo + n) = v;
        }
    } else {
        for (int ps = 0; ps < NP; ++ps) {
            const int row = r0 + ps * RPP;
            const int m = m0 + row;
            if (m >= p.M) break;
            const size_t o = (size_t)m * p.ldo + n;
            for (int e = 0; e < 4; ++e) {
                if (n + e < p.Cout) {
                    float t = fmaf(Cs[row * LDC + cq * 4 + e], sc[e], sh[e]);
                    if (p.res) t += p.res[o + e];
                    if (p.relu) t = fmaxf(t, 0.f);
                    p.out[o + e] = t;
                }
            }
        }
    }
}

template <int BM, int BN>
static int launch_variant(const KArgs& k, int M, const LaunchCtx& ctx, const char* name, double flops,
                          double bytes) {
    constexpr size_t smem = (size_t)(2 * BM * 36 + 2 * 8 * BN * 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32_kernel<BM, BN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    KArgs kk = k;
    kk.nbn = k.Npad / BN;
    const int nbm = (M + BM - 1) / BM;
    const int grid = nbm * kk.nbn;
    ProfScope ps(ctx, name, flops, bytes);
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN>), dim3(grid), dim3(256), smem, ctx.stream, kk);
    return (int)hipGetLastError();
}

static int g_force_variant = 0;  // 0 auto, 1: 128x128, 2: 128x64, 3: 64x64
void conv_igemm_force_variant(int v) { g_force_variant = v; }

static int pick_variant(int M, int Npad) {
    if (g_force_variant == 1 && Npad % 128 == 0) return 1;
    if (g_force_variant == 2 || g_force_variant == 3) return g_force_variant;
    // Measured on MI355X at B=256 (profiles/): the 64x64 tile (4 workgroups = 16 waves per CU,
    // 4 waves per SIMD sharing the 64-cycle fp32 MFMA pipe) beats 128x64 and 128x128 on every
    // layer of the trunk: finer work quantisation over 256 CUs and better latency hiding
    // outweigh the larger tiles' lower L2 traffic.  The bigger tiles stay selectable.
    (void)M;
    return 3;
}

const char* conv_igemm_variant(const ConvArgs& a) {
    static const char* names[] = {"", "conv_igemm_f32<128x128>", "conv_igemm_f32<128x64>", "conv_igemm_f32<64x64>"};
    return names[pick_variant(a.B * a.OH * a.OW, a.Npad)];
}

int launch_conv_igemm(const ConvArgs& a, const LaunchCtx& ctx) {
    if (a.Cin % 32 != 0 || a.Npad % 64 != 0 || a.ldx % 4 != 0) return (int)hipErrorInvalidValue;
    KArgs k;
    k.x = a.x; k.w = a.w; k.scale = a.scale; k.shift = a.shift; k.res = a.res; k.out = a.out;
    k.H = a.H; k.W = a.W; k.ldx = a.ldx;
    k.OW = a.OW; k.OHW = a.OH * a.OW; k.Cout = a.Cout; k.Npad = a.Npad; k.ldo = a.ldo;
    k.KW = a.KW; k.stride = a.stride; k.pad = a.pad;
    const int M = a.B * a.OH * a.OW;
    k.M = M;
    k.cpc = a.Cin / 32;
    k.nchunks = a.KH * a.KW * k.cpc;
    k.nbn = 0;
    k.relu = a.relu;
    k.vec_ok = (a.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0) &&
               (!a.res || (reinterpret_cast<uintptr_t>(a.res) & 15) == 0);
    const double Kd = (double)a.KH * a.KW * a.Cin;
    const double flops = 2.0 * (double)M * a.Cout * Kd;
    const double bytes = 4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)M * a.Cout * (a.res ? 2.0 : 1.0) +
                                Kd * a.Cout);
    switch (pick_variant(M, a.Npad)) {
        case 1: return launch_variant<128, 128>(k, M, ctx, "conv_igemm_f32<128x128>", flops, bytes);
        case 2: return launch_variant<128, 64>(k, M, ctx, "conv_igemm_f32<128x64>", flops, bytes);
        default: return launch_variant<64, 64>(k, M, ctx, "conv_igemm_f32<64x64>", flops, bytes);
    }
}

}  // namespace specmi
