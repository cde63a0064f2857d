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
//     A workgroup = WGM x WGN waves; a wave owns TM x TN tiles of 32x32.
//   * K is consumed in chunks of 32 (one filter tap x 32 channels).  Inside a chunk the k
//     order is permuted so that each lane's 4 consecutive k values are one 16-byte LDS read:
//     sub-chunk q (8 k's), lane half h, step s  ->  k = 8q + 4h + s.  A and B use the same
//     permutation, so the sum over k is unchanged.
//   * LDS: A tile [BM][32+4] fp32 (row pad 4 floats => ds_read_b128 of 16 rows hits 64
//     distinct banks, ds_write_b128 of one row's 8 quads hits 32 distinct banks).
//   * B (weights), 64x64 kernel (BDIR): never staged.  The packed HBM layout [K/4][Npad][4] IS the
//     MFMA fragment layout (lane (n, k half) reads quad 2q + hh of column n), so every wave streams
//     its own [32 k][32 n] blocks from L2 with 1-KiB coalesced buffer loads into rolling registers,
//     one chunk ahead: no B traffic through LDS, A-only LDS = 18 KB -> 7 workgroups per CU (+6 %).
//     The 8-wave 128x128 kernel (HBM-bound expand convs) keeps B staged in LDS as [8][BN][4].
//   * Variants: two A sources (K = Cin + Cin2: a bottleneck's downsample conv folded into conv3),
//     split-K over blockIdx.y (SPLITK) for small M - the FC GEMMs of every plan and every convolution of the
//     LATENCY plan (batch <= 10 by default: the reference's own operating point, spec/tester.py:109-151 runs the path at
//     batch = #detections of a frame, scripts/camcalib_demo.py:95-102 at batch 1): a layer that offers 8-64
//     output tiles walks K = 1024-4608 on as many CUs while 200 idle; cut into S slices of whole 32-channel
//     chunks it fills the chip.  Slice z leaves its raw accumulators in a workspace, takes a ticket from the
//     tile's counter, and the LAST slice to arrive adds the partial tiles in a fixed order (whatever the arrival
//     order) and runs the usual epilogue - one launch, no second pass.
//     The k sum of a sliced layer has ONE canonical association, fixed by the layer's shape alone: K is cut into
//     LEAVES of L chunks (each an MFMA chain from +0), G consecutive leaves fold (left to right, from +0) into a
//     GROUP, the groups fold into the result.  How much of that tree one workgroup computes is a pure speed
//     choice made per batch size - a leaf (batch 1: most workgroups), a group, or the whole K (batch 16: no
//     slabs at all): a workgroup that owns several leaves keeps the leaf / group / result accumulators apart
//     in registers and adds them at the canonical boundaries (16 v_add per 32x32 block per leaf, < 1 %), the
//     last arriver folds whatever level the slabs hold.  An image's bits therefore do not depend on the batch.
//   * The fp32 MFMA holds a SIMD's matrix pipe for 64 cycles but the SIMD has only ~16 issue
//     slots in that time, shared by all its waves - so everything that is not an MFMA is kept
//     off the VALU: tile rows are addressed with buffer loads (32-bit per-row offset computed
//     once + a scalar per-chunk offset in SGPRs), im2col padding and the M tail are handled by
//     the buffer's hardware range check (an out-of-image tap gets an out-of-range offset and
//     reads as 0.0), 1x1 convolutions need no per-chunk VALU work at all.
//   * double-buffered LDS, register-staged prefetch of chunk c+1 issued before the MFMAs of
//     chunk c (one barrier per chunk); the last chunk prefetches the residual rows instead.
//   * blockIdx -> tile map is XCD-aware: the 8 XCDs get contiguous runs of tiles ordered
//     n-fastest, so tiles sharing an A row-panel hit the same 4 MiB L2.
//   * epilogue: accumulators are transposed through LDS so that HBM sees 16-byte,
//     row-contiguous stores / residual reads with scale/shift/ReLU fused.
#include "conv_igemm_tile.h"

namespace specmi {

// One workgroup = one (tile, K slice, network) of one layer: the body lives in conv_igemm_tile.h (shared with the persistent
// multi-layer walker of conv_persist.hip)
template <int BM, int BN, int WGM, int WGN, bool IS1X1, int BK, bool DUAL = false, bool SPLITK = false, bool BDIR = false>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_igemm_f32_kernel(const KArgs p) {
    constexpr bool PERSIST = false;
    TileCtx t;
    t.bid = blockIdx.x; t.nblk = gridDim.x;
    t.y = blockIdx.y; t.S = gridDim.y;
    t.z = blockIdx.z;
    t.ws = p.sk_ws; t.cnt = p.sk_cnt;
    t.tile = blockIdx.z * gridDim.x + blockIdx.x;
#include "conv_igemm_body.inc"
}

template <int BM, int BN, int WGM, int WGN, bool IS1X1, int BK = 32, bool DUAL = false, bool SPLITK = false, bool BDIR = false>
static int launch_variant(const KArgs& k, int M, const LaunchCtx& ctx, const char* name, double flops,
                          double bytes, int nsplit = 1) {
    const int groups = k.g1.x ? 2 : 1;
    constexpr bool b_lds = !BDIR;   // BDIR kernels stage only A: 18 KB -> 7 workgroups per CU instead of 4
    constexpr size_t ab = (size_t)(2 * BM * (BK + 4) + (b_lds ? 2 * (BK / 4) * BN * 4 : 0)) * sizeof(float);
    constexpr size_t cb = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t smem = ab > cb ? ab : cb;
    static DevOnce once;
    if (int e = set_dyn_lds_once(once, reinterpret_cast<const void*>(&conv_igemm_f32_kernel<BM, BN, WGM, WGN, IS1X1, BK, DUAL, SPLITK, BDIR>), (int)smem))
        return e;
    KArgs kk = k;
    kk.nbn = BN < 64 ? (k.Cout + BN - 1) / BN : k.Npad / BN;   // 32-wide tiles skip the all-padding half of a 64-padded weight panel
    const int nbm = (M + BM - 1) / BM;
    const int grid = nbm * kk.nbn;
    // tile order for weight panels beyond L2 (see the kernel): worth it when eight reads of A cost less than re-streaming B
    // for every other tile row, i.e. 8 * nbm * BM * K < K * N * nbm / 2  <=>  N > 16 * BM
    const size_t w_bytes = (size_t)k.nchunks * 32 * k.Npad * 4;
    kk.xcd_cols = 0;
#ifndef SPECMI_NO_XCD_COLS
    if (IS1X1 && !SPLITK && kk.nbn % 8 == 0 && w_bytes >= ((size_t)4 << 20) && k.Npad > 16 * BM && nbm >= 16)
        kk.xcd_cols = kk.nbn / 8;
#endif
    ProfScope ps(ctx, name, flops * groups, bytes * groups);
    kk.cpc = k.cpc * 32 / BK;
    kk.nchunks = k.nchunks * 32 / BK;
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WGM, WGN, IS1X1, BK, DUAL, SPLITK, BDIR>), dim3(grid, nsplit, groups), dim3(64 * WGM * WGN), smem,
                       ctx.stream, kk);
    return (int)hipGetLastError();
}

#ifdef SPECMI_TUNE
static int g_ablate = 0;
static unsigned long long* g_tprof = nullptr;
void conv_igemm_set_ablate(int v) { g_ablate = v; }
void conv_igemm_set_tprof(unsigned long long* p) { g_tprof = p; }
#endif
// forced tile (ConvArgs::force_variant, per handle): 0 auto, 1: 128x128/4 waves, 2: 128x64/4, 3: 64x64/4, 4: 128x128/8 waves

static const char* kVariantNames[] = {"", "conv_igemm_f32<128x128,2x2>", "conv_igemm_f32<128x64,2x2>",
                                      "conv_igemm_f32<64x64,2x2>", "conv_igemm_f32<128x128,4x2>",
                                      "conv_igemm_f32<128x64,4x2>", "conv_igemm_f32<64x128,2x4>", "conv_igemm_f32<64x64,2x2,bk16>",
                                      "conv_igemm_f32<64x64,2x2,ldsB>", "conv_igemm_f32<128x128,4x2,bdir>",
                                      "conv_igemm_f32<64x128,2x2,bdir>", "conv_igemm_f32<128x64,2x2,bdir>", "conv_igemm_f32<64x64,2x2,bdir,bk64>",
                                      "conv_igemm_f32<128x32,4x1>"};

static int pick_variant(int M, int Npad, bool is1x1, int K, int force, int Cout) {
    if (force >= 1 && force <= 13) {
        const bool needs128 = (force == 1 || force == 4 || force == 6 || force == 9 || force == 10);
        if (!needs128 || Npad % 128 == 0) return force;
    }
    // Measured on MI355X at B=256 (tools/igemm_bench, profiles/): the 64x64 tile with the B fragments
    // streamed straight from L2 (A alone in LDS: 18 KB, 7 workgroups per CU) wins everywhere except on
    // the large-M expand convs with K <= 128 (layer1/layer2 conv3 + residual) that sit on the HBM
    // roofline - there the 8-wave 128x128 tile with both operands staged moves half the L2 traffic.
    if (is1x1 && Npad % 128 == 0 && M >= 131072 && K <= 128) return 4;
    // <= 32 output channels (HRNet-W32's full-resolution branch): a 64-wide tile would spend half its MFMAs on the zero
    // padding of the weight panel; 128 rows x 32 columns keeps every matrix-core cycle on real outputs
    if (Cout <= 32) return 13;
    return 3;
}

const char* conv_igemm_variant(const ConvArgs& a) {
    return kVariantNames[pick_variant(a.B * a.OH * a.OW, a.Npad, a.KH == 1 && a.KW == 1 && a.pad == 0, a.Cin + a.Cin2, a.force_variant, a.Cout)];
}

static int dispatch_dual(int v, const KArgs& k, int M, const LaunchCtx& ctx, double flops, double bytes) {
    if (v == 4) return launch_variant<128, 128, 4, 2, true, 32, true>(k, M, ctx, "conv_igemm_f32<128x128,4x2,2src>", flops, bytes);
    return launch_variant<64, 64, 2, 2, true, 32, true, false, true>(k, M, ctx, "conv_igemm_f32<64x64,2x2,2src>", flops, bytes);
}

template <bool IS1X1>
static int dispatch(int v, const KArgs& k, int M, const LaunchCtx& ctx, double flops, double bytes) {
    switch (v) {
        case 1: return launch_variant<128, 128, 2, 2, IS1X1>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 2: return launch_variant<128, 64, 2, 2, IS1X1>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 4: return launch_variant<128, 128, 4, 2, IS1X1>(k, M, ctx, kVariantNames[v], flops, bytes);
#ifdef SPECMI_TUNE
        case 5: return launch_variant<128, 64, 4, 2, IS1X1>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 6: return launch_variant<64, 128, 2, 4, IS1X1>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 7: return launch_variant<64, 64, 2, 2, IS1X1, 16>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 8: return launch_variant<64, 64, 2, 2, IS1X1, 32, false, false, false>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 9: return launch_variant<128, 128, 4, 2, IS1X1, 32, false, false, true>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 10: return launch_variant<64, 128, 2, 2, IS1X1, 32, false, false, true>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 11: return launch_variant<128, 64, 2, 2, IS1X1, 32, false, false, true>(k, M, ctx, kVariantNames[v], flops, bytes);
        case 12: return launch_variant<64, 64, 2, 2, IS1X1, 64, false, false, true>(k, M, ctx, kVariantNames[v], flops, bytes);
#endif
        case 13: return launch_variant<128, 32, 4, 1, IS1X1, 32, false, false, true>(k, M, ctx, kVariantNames[v], flops, bytes);
        default: return launch_variant<64, 64, 2, 2, IS1X1, 32, false, false, true>(k, M, ctx, kVariantNames[3], flops, bytes);
    }
}

// floor(n / d) == umulhi(n, mg) >> sh for every n < 2^31 and 2 <= d < 2^31:
// L = 31 + ceil(log2 d), mg = floor(2^L / d) + 1 (< 2^32), sh = L - 32.  d == 1 is handled in the kernel.
static void magic_u32(unsigned d, unsigned* mg, unsigned* sh) {
    if (d < 2) { *mg = 0; *sh = 0; return; }
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    const unsigned L = 31 + s;
    *mg = (unsigned)((1ull << L) / d + 1ull);
    *sh = L - 32;
}

static void make_kargs(const ConvArgs& a, const ConvArgs* b, KArgs& k, double* flops, double* bytes) {
    k.x = a.x; k.w = a.w; k.scale = a.scale; k.shift = a.shift; k.res = a.res; k.out = a.out;
    k.g1.x = nullptr; k.g1.w = nullptr; k.g1.scale = nullptr; k.g1.shift = nullptr; k.g1.res = nullptr; k.g1.x2 = nullptr; k.g1.out = nullptr;
    if (b) { k.g1.x = b->x; k.g1.w = b->w; k.g1.scale = b->scale; k.g1.shift = b->shift; k.g1.res = b->res; k.g1.x2 = b->x2; k.g1.out = b->out; }
    k.H = a.H; k.W = a.W; k.ldx = a.ldx;
    k.OW = a.OW; k.OHW = a.OH * a.OW; k.Cout = a.Cout; k.Npad = a.Npad; k.ldo = a.ldo;
    k.KH = a.KH; k.KW = a.KW; k.stride = a.stride; k.pad = a.pad;
    const int M = a.B * a.OH * a.OW;
    k.M = M;
    const bool dual = a.x2 != nullptr;
    k.cpc = a.Cin / 32;
    k.nchunks = a.KH * a.KW * k.cpc + (dual ? a.Cin2 / 32 : 0);
    k.x2 = a.x2; k.H2 = a.H2; k.W2 = a.W2; k.ldx2 = a.ldx2; k.stride2 = a.stride2; k.cpc1 = k.cpc;
    k.x2_bytes = dual ? (unsigned)((size_t)a.B * a.H2 * a.W2 * a.ldx2 * 4) : 0u;
    k.nbn = 0;
    k.relu = a.relu;
    k.sk_ws = nullptr; k.sk_cnt = nullptr; k.sk_leaf = 0; k.sk_G = 1; k.sk_unit = 1;
    magic_u32((unsigned)k.OHW, &k.mg_ohw, &k.sh_ohw);
    magic_u32((unsigned)a.OW, &k.mg_ow, &k.sh_ow);
    k.x_bytes = (unsigned)((size_t)a.B * a.H * a.W * a.ldx * 4);
    k.w_bytes = (unsigned)(((size_t)a.KH * a.KW * a.Cin + (dual ? a.Cin2 : 0)) * a.Npad * 4);
#ifdef SPECMI_TUNE
    k.ablate = g_ablate;
    k.tprof = g_tprof;
#endif
    k.vec_ok = (a.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0) &&
               (!a.res || (reinterpret_cast<uintptr_t>(a.res) & 15) == 0) &&
               (!b || (((reinterpret_cast<uintptr_t>(b->out) & 15) == 0) && (!b->res || (reinterpret_cast<uintptr_t>(b->res) & 15) == 0)));
    const double Kd = (double)a.KH * a.KW * a.Cin + (dual ? a.Cin2 : 0);
    *flops = 2.0 * (double)M * a.Cout * Kd;
    *bytes = 4.0 * ((double)a.B * a.H * a.W * a.Cin + (dual ? (double)M * a.Cin2 : 0.0) +
                    (double)M * a.Cout * (a.res ? 2.0 : 1.0) + Kd * a.Cout);
}

static int launch_one(const ConvArgs& a, const LaunchCtx& ctx, const ConvArgs* b = nullptr) {
    KArgs k;
    double flops, bytes;
    make_kargs(a, b, k, &flops, &bytes);
    const int M = k.M;
    const bool dual = a.x2 != nullptr;
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.pad == 0);
    const int v = pick_variant(M, a.Npad, is1x1, a.Cin + (dual ? a.Cin2 : 0), a.force_variant, a.Cout);
    if (dual) return dispatch_dual(v, k, M, ctx, flops, bytes);
    return is1x1 ? dispatch<true>(v, k, M, ctx, flops, bytes) : dispatch<false>(v, k, M, ctx, flops, bytes);
}

// ---- split-K: the FC GEMMs of every plan, every convolution of the latency plan (see the file header) -----------------
// K slices of the FC GEMMs (CamCalib heads, HMR regressor; M = batch rows <= 1024): the rule every plan has used since
// round 1, so the headline's FC results keep their bits
int conv_igemm_splitk_plan(const ConvArgs& a) {
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.pad == 0 && a.stride == 1);
    const int M = a.B * a.OH * a.OW;
    if (!is1x1 || a.x2 || M > 1024 || a.force_variant) return 1;
    const int nch = a.Cin / 32;
    int best = 1;
    for (int s = 2; s <= 16; ++s)
        if (nch % s == 0 && nch / s >= 8) best = s;
    return best;
}

// K slices of a convolution in the latency plan.  A function of the layer's PER-IMAGE shape only - never of the batch -
// so that an image's summation order, and with it every bit of its result, is the same in a batch of 1 and of 16 (and in a
// grouped launch and a separate one): slices of whole 32-channel chunks, at least `min_chunks` each; the smallest slice
// count that gives the pair of trunks at batch 1 (2 x tiles) `target_wgs` workgroups, else the largest allowed.
int conv_igemm_sk_slices(const ConvArgs& a, int target_wgs, int min_chunks) {
    if (a.Npad % 64 != 0 || a.Cin % 32 != 0 || a.force_variant) return 1;
    const int nch = a.KH * a.KW * (a.Cin / 32) + (a.x2 ? a.Cin2 / 32 : 0);
    // K < 512: the unsplit kernel wins at every batch size (measured per layer, profiles/r04_b_latency_layers.txt: a second
    // slab round trip costs more than walking 8-12 chunks)
    if (nch < 16) return 1;
    const int tiles1 = ((a.OH * a.OW + 63) / 64) * (a.Npad / 64);
    int best = 1;
    for (int s = 1; s <= nch; ++s) {
        if (nch % s != 0 || nch / s < min_chunks) continue;
        best = s;
        if (2 * tiles1 * s >= target_wgs) break;
    }
    return best;
}

// The canonical tree of a layer (leaves, leaves per group) and how much of it one workgroup computes at THIS batch: the
// largest unit (fewest slabs to write and fold) that still gives `fill_wgs` workgroups.  The unit changes speed, never bits.
SkPlan conv_igemm_sk_plan(const ConvArgs& a, int groups, int target_wgs, int min_chunks, int fill_wgs) {
    SkPlan pl;
    pl.leaves = conv_igemm_sk_slices(a, target_wgs, min_chunks);
    pl.G = 1;
    for (int g = 2; g <= 4; ++g)
        if (pl.leaves % g == 0) pl.G = g;
    const long tiles = (long)conv_igemm_sk_tiles(a, groups);
    if (fill_wgs <= 0) {
        // fill_wgs <= 0: the round model instead of a threshold (round 5; slots = -fill_wgs, 0 = one per CU): a launch lasts
        // ceil(workgroups / slots) rounds of one workgroup's chunks (~0.5 us each on the shared matrix pipe) plus, when K is also cut
        // across workgroups, the slab hop (~3 us) and the last arriver's read of S slabs of 16 KB (~0.25 us each).  The candidates
        // are the three units; ties go to the larger one (fewer slabs).  It reproduces what the threshold sweeps found batch by
        // batch - group at 10 images, leaves at 12, where 240 / 400 workgroups were each right once (profiles/r05_n_*).
        const long slots = fill_wgs < 0 ? -fill_wgs : 256;
        const int nch = a.KH * a.KW * (a.Cin / 32) + (a.x2 ? a.Cin2 / 32 : 0);
        const int cand[3] = {pl.leaves, pl.G, 1};
        double best = 0.0;
        pl.unit = pl.leaves;
        for (int i = 0; i < 3; ++i) {
            const int u = cand[i];
            if (i && u == cand[i - 1]) continue;
            const long S = pl.leaves / u;
            const long wgs = tiles * S;
            const double chunks = (double)(nch / pl.leaves) * u;
            const double cost = (double)((wgs + slots - 1) / slots) * chunks * 0.5 + (S > 1 ? 3.0 + 0.25 * (double)S : 0.0);
            if (i == 0 || cost < best - 1e-9) { best = cost; pl.unit = u; }
        }
        return pl;
    }
    if (tiles >= fill_wgs) pl.unit = pl.leaves;
    else if (tiles * (pl.leaves / pl.G) >= fill_wgs) pl.unit = pl.G;
    else pl.unit = 1;
    return pl;
}

size_t conv_igemm_sk_ws_floats(const ConvArgs& a, int S, int groups) {
    const size_t tiles = (size_t)((a.B * a.OH * a.OW + 63) / 64) * (a.Npad / 64);
    return S > 1 ? tiles * S * groups * 64 * 64 : 0;
}
int conv_igemm_sk_tiles(const ConvArgs& a, int groups) { return ((a.B * a.OH * a.OW + 63) / 64) * (a.Npad / 64) * groups; }

int conv_igemm_sk_check(const ConvArgs& a, const SkPlan& pl, const ConvArgs* b) {
    const int nch = a.KH * a.KW * (a.Cin / 32) + (a.x2 ? a.Cin2 / 32 : 0);
    if (pl.leaves < 1 || nch % pl.leaves != 0 || pl.G < 1 || pl.leaves % pl.G != 0 || (pl.unit != 1 && pl.unit != pl.G && pl.unit != pl.leaves) ||
        a.Cin % 32 != 0 || a.Npad % 64 != 0 || a.ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15))
        return (int)hipErrorInvalidValue;
    if (a.x2 && (a.KH != 1 || a.KW != 1 || a.pad != 0 || a.Cin2 % 32 != 0 || a.ldx2 % 4 != 0 || (reinterpret_cast<uintptr_t>(a.x2) & 15)))
        return (int)hipErrorInvalidValue;
    if (b && (b->B != a.B || b->H != a.H || b->W != a.W || b->Cin != a.Cin || b->ldx != a.ldx || b->OH != a.OH || b->OW != a.OW ||
              b->Cout != a.Cout || b->Npad != a.Npad || b->ldo != a.ldo || b->KH != a.KH || b->KW != a.KW || b->stride != a.stride ||
              b->pad != a.pad || b->relu != a.relu || (b->res != nullptr) != (a.res != nullptr) || (b->x2 != nullptr) != (a.x2 != nullptr) ||
              b->H2 != a.H2 || b->W2 != a.W2 || b->ldx2 != a.ldx2 || b->Cin2 != a.Cin2 || b->stride2 != a.stride2 ||
              (reinterpret_cast<uintptr_t>(b->x) & 15) || (b->x2 && (reinterpret_cast<uintptr_t>(b->x2) & 15))))
        return (int)hipErrorInvalidValue;
    size_t img_bytes = (size_t)a.H * a.W * a.ldx * 4;
    if (a.x2 && (size_t)a.H2 * a.W2 * a.ldx2 * 4 > img_bytes) img_bytes = (size_t)a.H2 * a.W2 * a.ldx2 * 4;
    const size_t limit = (size_t)1 << 31;
    if (img_bytes * a.B >= limit || (size_t)nch * 32 * a.Npad * 4 >= limit) return SK_NEEDS_BATCH_SPLIT;   // small-M path: no batch splitting
    return 0;
}

void conv_igemm_make_sk_kargs(const ConvArgs& a, const SkPlan& pl, const ConvArgs* b, KArgs& k) {
    const int nch = a.KH * a.KW * (a.Cin / 32) + (a.x2 ? a.Cin2 / 32 : 0);
    double flops, bytes;
    make_kargs(a, b, k, &flops, &bytes);
    k.sk_leaf = nch / pl.leaves; k.sk_G = pl.G; k.sk_unit = pl.unit;
    k.nchunks = k.sk_leaf * pl.unit;
    k.nbn = a.Npad / 64;
    k.xcd_cols = 0;
}

// One launch, gridDim.y = pl.leaves / pl.unit slabs per tile; sk.ws holds conv_igemm_sk_ws_floats(a, slabs, groups) floats,
// sk.cnt one zeroed counter per (tile, group) - the kernel leaves them zeroed
int launch_conv_igemm_sk(const ConvArgs& a, const SkPlan& pl, const SkWs& sk, const LaunchCtx& ctx, const ConvArgs* b) {
    const int groups = b ? 2 : 1;
    if (int rc = conv_igemm_sk_check(a, pl, b)) return rc;
    const int S = pl.leaves / pl.unit;
    if (S > 1 && (!sk.ws || !sk.cnt || conv_igemm_sk_ws_floats(a, S, groups) > sk.floats || conv_igemm_sk_tiles(a, groups) > sk.ncnt))
        return (int)hipErrorInvalidValue;
    KArgs k;
    double flops, bytes;
    make_kargs(a, b, k, &flops, &bytes);
    const int nch = a.KH * a.KW * (a.Cin / 32) + (a.x2 ? a.Cin2 / 32 : 0);
    k.sk_leaf = nch / pl.leaves; k.sk_G = pl.G; k.sk_unit = pl.unit;
    k.nchunks = k.sk_leaf * pl.unit;
    k.sk_ws = sk.ws; k.sk_cnt = sk.cnt;
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.pad == 0);
    if (a.x2) return launch_variant<64, 64, 2, 2, true, 32, true, true, true>(k, k.M, ctx, "conv_igemm_f32<64x64,2x2,2src,splitK>", flops, bytes, S);
    if (is1x1) return launch_variant<64, 64, 2, 2, true, 32, false, true, true>(k, k.M, ctx, "conv_igemm_f32<64x64,2x2,splitK>", flops, bytes, S);
    return launch_variant<64, 64, 2, 2, false, 32, false, true, true>(k, k.M, ctx, "conv_igemm_f32<64x64,2x2,splitK>", flops, bytes, S);
}

// b != nullptr: the same layer shape of a second network (its own x / w / scale / shift / res / x2 / out) in the same launch
int launch_conv_igemm(const ConvArgs& a, const LaunchCtx& ctx, const ConvArgs* b) {
    if (a.Cin % 32 != 0 || a.Npad % 64 != 0 || a.ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15))
        return (int)hipErrorInvalidValue;
    if (b && (b->B != a.B || b->H != a.H || b->W != a.W || b->Cin != a.Cin || b->ldx != a.ldx || b->OH != a.OH || b->OW != a.OW ||
              b->Cout != a.Cout || b->Npad != a.Npad || b->ldo != a.ldo || b->KH != a.KH || b->KW != a.KW || b->stride != a.stride ||
              b->pad != a.pad || b->relu != a.relu || (b->res != nullptr) != (a.res != nullptr) || (b->x2 != nullptr) != (a.x2 != nullptr) ||
              b->H2 != a.H2 || b->W2 != a.W2 || b->ldx2 != a.ldx2 || b->Cin2 != a.Cin2 || b->stride2 != a.stride2 ||
              (reinterpret_cast<uintptr_t>(b->x) & 15) || (b->x2 && (reinterpret_cast<uintptr_t>(b->x2) & 15))))
        return (int)hipErrorInvalidValue;
    if (a.x2 && (a.KH != 1 || a.KW != 1 || a.pad != 0 || a.Cin2 % 32 != 0 || a.ldx2 % 4 != 0 ||
                 (reinterpret_cast<uintptr_t>(a.x2) & 15)))
        return (int)hipErrorInvalidValue;
    // buffer addressing is 32-bit: split the batch when the activation tensor reaches 2 GiB
    size_t img_bytes = (size_t)a.H * a.W * a.ldx * 4;
    if (a.x2 && (size_t)a.H2 * a.W2 * a.ldx2 * 4 > img_bytes) img_bytes = (size_t)a.H2 * a.W2 * a.ldx2 * 4;
    const size_t limit = (size_t)1 << 31;
    if (img_bytes >= limit || ((size_t)a.KH * a.KW * a.Cin + a.Cin2) * a.Npad * 4 >= limit) return (int)hipErrorInvalidValue;
    const int max_b = (int)((limit - 1) / img_bytes);
    if (a.B <= max_b) return launch_one(a, ctx, b);
    if (b) return (int)hipErrorInvalidValue;     // (the caller falls back to two launches: batches this large fill the chip anyway)
    for (int b0 = 0; b0 < a.B; b0 += max_b) {
        ConvArgs s = a;
        s.B = (a.B - b0 < max_b) ? a.B - b0 : max_b;
        s.x = a.x + (size_t)b0 * a.H * a.W * a.ldx;
        if (a.x2) s.x2 = a.x2 + (size_t)b0 * a.H2 * a.W2 * a.ldx2;
        const size_t orow = (size_t)b0 * a.OH * a.OW * a.ldo;
        s.out = a.out + orow;
        if (a.res) s.res = a.res + orow;
        const int rc = launch_one(s, ctx);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace specmi
