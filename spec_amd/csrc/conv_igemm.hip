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
#include <type_traits>

#include "specmi_internal.h"

namespace specmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct KArgs {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    unsigned x_bytes, w_bytes;  // buffer extents (< 2^31)
    // optional second A source of a 1x1 layer (K = Cin + Cin2): the block input of a fused downsample branch
    const float* x2;
    unsigned x2_bytes;
    int H2, W2, ldx2, stride2, cpc1;   // cpc1 = 32-channel chunks that come from x
    int H, W, ldx;
    int OW, OHW, Cout, Npad, ldo;
    int KH, KW, stride, pad;
    int M, nbn, nchunks, cpc;  // cpc = chunks per filter tap = Cin / 32
    int xcd_cols;              // > 0: XCD x owns tile columns [x * xcd_cols, (x + 1) * xcd_cols) and walks all tile rows (see the tile order)
    unsigned mg_ohw, sh_ohw, mg_ow, sh_ow;  // magic multipliers: n / OHW, n / OW for n < 2^31
    int relu;
    // SPLITK: blockIdx.y = K slice z of nchunks chunks; the raw accumulators of slice z of tile t (t = blockIdx.x + gridDim.x *
    // blockIdx.z) go to sk_ws[(t * S + z) * BM * BN ..] in accumulator order, sk_cnt[t] counts the slices that have arrived
    float* sk_ws;
    unsigned* sk_cnt;
    int sk_leaf, sk_G, sk_unit;   // chunks per leaf; leaves per group; leaves per workgroup (1, sk_G or all: nchunks = sk_unit * sk_leaf)
    int vec_ok;  // out/res rows are 16-byte aligned: float4 epilogue traffic allowed
    // grouped launch (gridDim.z = 2): blockIdx.z = 1 runs the SAME layer shape of a second network on its own tensors - the
    // two ResNet-50 trunks of the path (CamCalib + SPEC) as one launch per layer: half the launches, and the partially
    // filled last round of workgroups of one network is filled by the other
    struct { const float *x, *w, *scale, *shift, *res, *x2; float* out; } g1;
#ifdef SPECMI_TUNE
    int ablate;  // perf ablation bits (wrong results!): 1 no global loads in loop, 2 no LDS restage, 4 no epilogue stores
    unsigned long long* tprof;  // per-phase cycle counters (s_memtime)
#endif
};

#ifdef SPECMI_TUNE
#define TUNE_ABLATE(bit) (p.ablate & (bit))
#define TUNE_T(var) const long long var = __builtin_amdgcn_s_memtime()
#else
#define TUNE_ABLATE(bit) 0
#define TUNE_T(var)
#endif

constexpr unsigned kOutOfRange = 0x80000000u;  // >= any buffer extent: the load returns zeros

template <int BM, int BN, int WGM, int WGN, bool IS1X1, int BK, bool DUAL = false, bool SPLITK = false, bool BDIR = false>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_igemm_f32_kernel(const KArgs p) {
    static_assert(!SPLITK || BDIR, "split-K exists for the 64x64 kernel that streams its B fragments");
    static_assert(!(DUAL && !IS1X1), "");
    static_assert(!DUAL || IS1X1, "the second A source exists for 1x1 layers only");
    constexpr int LDA = BK + 4;
    constexpr int KQ = BK / 4;   // 16-byte k-quads per chunk row
    constexpr int NQ = BK / 8;   // 8-k sub-chunks per chunk
    constexpr int NT = 64 * WGM * WGN;                         // threads: WGM x WGN waves
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);  // 32x32 MFMA tiles per wave
    constexpr int AI = BM * KQ / NT, BI = BN * KQ / NT;        // float4 loads per thread per chunk
    constexpr int ARS = NT / KQ;                               // A rows covered per load pass
    constexpr int A_STAGE = BM * LDA, B_STAGE = KQ * BN * 4;
    static_assert(TM >= 1 && TN >= 1 && AI >= 1 && BI >= 1, "tile too small for the wave grid");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const bool grp = blockIdx.z != 0;          // wave-uniform: scalar selects
    const float* const px = grp ? p.g1.x : p.x;
    const float* const pw = grp ? p.g1.w : p.w;
    const float* const pscale = grp ? p.g1.scale : p.scale;
    const float* const pshift = grp ? p.g1.shift : p.shift;
    const float* const pres = grp ? p.g1.res : p.res;
    const float* const px2 = grp ? p.g1.x2 : p.x2;
    float* const pout = grp ? p.g1.out : p.out;
#ifdef SPECMI_TUNE
    const long long t_start = __builtin_amdgcn_s_memtime();
    long long tp[4] = {0, 0, 0, 0};
    (void)tp;
#endif

    // ---- XCD-aware tile order (bijective for any grid size) ------------------------------
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int tile_m = L / p.nbn, tile_n = L - tile_m * p.nbn;
    if (p.xcd_cols) {
        // Weight panel larger than an XCD's 4 MiB L2 and many tile columns (layer4 conv3: 512 / 3072 x 2048 = 4 / 12 MiB, 32
        // columns): with the row-major order every XCD streams the WHOLE panel again for each few tile rows (measured 0.45 / 2.0 GB
        // of L2 misses per launch against 0.24 / 0.19 GB algorithmic, profiles/r03_v_layer_traffic.txt).  Here an XCD owns a
        // fixed eighth of the columns - its slice of the panel stays in its L2 - and walks all tile rows; the (smaller) A
        // operand is then read by all eight XCDs instead.  nbn % 8 == 0, so the grid splits evenly.
        tile_n = xcd * p.xcd_cols + (bid >> 3) % p.xcd_cols;
        tile_m = (bid >> 3) / p.xcd_cols;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- buffer descriptors (wave-uniform) and per-thread row offsets -----------------------
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pw), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? px2 : px), 0, DUAL ? p.x2_bytes : p.x_bytes, 0x00020000);
    const int a_kq = tid % KQ, a_r = tid / KQ;
    unsigned a_voff[AI];   // byte offset of (row's tap-(0,0) pixel, quad a_kq); out-of-range when the row is past M (1x1)
    unsigned a_mask[AI];   // 3x3: bit t = filter tap t lies inside the image for this row
    unsigned a_voff2[DUAL ? AI : 1];   // DUAL: the row's pixel in the second source (its own size / stride / channel count)
    // The tile prologue sits on every workgroup's critical path, so the pixel decode avoids the
    // ~40-instruction integer divide: 1x1/stride-1 rows address the input with m itself, other
    // shapes divide by OH*OW and OW with host-computed magic multipliers.
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + a_r + ARS * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        if (DUAL) {
            if (p.stride2 == 1) {
                a_voff2[i] = ok ? (unsigned)(mm * p.ldx2 * 4 + a_kq * 16) : kOutOfRange;
            } else {
                const int b2 = p.OHW == 1 ? mm : (int)(__umulhi((unsigned)mm, p.mg_ohw) >> p.sh_ohw);
                const int rem2 = mm - b2 * p.OHW;
                const int oy2 = p.OW == 1 ? rem2 : (int)(__umulhi((unsigned)rem2, p.mg_ow) >> p.sh_ow);
                const int ox2 = rem2 - oy2 * p.OW;
                const int pix2 = (b2 * p.H2 + oy2 * p.stride2) * p.W2 + ox2 * p.stride2;
                a_voff2[i] = ok ? (unsigned)(pix2 * p.ldx2 * 4 + a_kq * 16) : kOutOfRange;
            }
        }
        if (IS1X1 && p.stride == 1) {
            a_voff[i] = ok ? (unsigned)(mm * p.ldx * 4 + a_kq * 16) : kOutOfRange;
            a_mask[i] = 0;
            continue;
        }
        const int b = p.OHW == 1 ? mm : (int)(__umulhi((unsigned)mm, p.mg_ohw) >> p.sh_ohw);
        const int rem = mm - b * p.OHW;
        const int oy = p.OW == 1 ? rem : (int)(__umulhi((unsigned)rem, p.mg_ow) >> p.sh_ow);
        const int ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        const int pix0 = (b * p.H + iy0) * p.W + ix0;
        const unsigned off = (unsigned)(pix0 * p.ldx * 4 + a_kq * 16);   // wraps for padded rows; only used on valid taps
        if (IS1X1) {
            a_voff[i] = ok ? off : kOutOfRange;
            a_mask[i] = 0;
        } else {
            a_voff[i] = off;
            unsigned colbits = 0, mk = 0;   // tap (ky,kx) is inside the image iff row ky and column kx are
            for (int kx = 0; kx < p.KW; ++kx) colbits |= ((unsigned)(ix0 + kx) < (unsigned)p.W ? 1u : 0u) << kx;
            for (int ky = 0; ky < p.KH; ++ky)
                if ((unsigned)(iy0 + ky) < (unsigned)p.H) mk |= colbits << (ky * p.KW);
            a_mask[i] = ok ? mk : 0u;
        }
    }
    unsigned b_voff[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int e = tid + NT * i;
        const int kq = e / BN, n = e % BN;
        b_voff[i] = (unsigned)(((kq * p.Npad) + n0 + n) * 16);
    }

    f32x4 ra[AI], rb[BI];
    // chunk c = (tap, 32-channel slice); the per-chunk part of every address is scalar
    const int cbase = SPLITK ? (int)blockIdx.y * p.nchunks : 0;   // first chunk of this workgroup's K slice
    auto load_chunk = [&](int cl) {
        const int c = cl + cbase;
        const int tap = IS1X1 ? 0 : c / p.cpc;
        const int c0 = IS1X1 ? c : c - tap * p.cpc;
        unsigned tap_bytes = 0;
        if (!IS1X1) {
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            tap_bytes = (unsigned)((ky * p.W + kx) * p.ldx * 4);
        }
        const bool second = DUAL && c >= p.cpc1;   // wave-uniform: chunks past cpc1 read the second source
        const unsigned s_a = (unsigned)((second ? c0 - p.cpc1 : c0) * BK * 4);
        const unsigned s_b = (unsigned)(c * KQ * p.Npad * 16);
        if (!TUNE_ABLATE(16)) {
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                unsigned voff = a_voff[i];
                if (!IS1X1) voff = ((a_mask[i] >> tap) & 1u) ? voff + tap_bytes : kOutOfRange;
                if (DUAL) {   // one load either way: descriptor picked with scalar selects, offset with one v_cndmask
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? x2rs : xrs, second ? a_voff2[i] : voff, s_a, 0));
                } else {
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, s_a, 0));
                }
            }
        }
        if (!BDIR && !TUNE_ABLATE(32)) {
#pragma unroll
            for (int i = 0; i < BI; ++i)
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff[i], s_b, 0));
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<f32x4*>(&As[buf * A_STAGE + (a_r + ARS * i) * LDA + a_kq * 4]) = ra[i];
        if (!BDIR) {
#pragma unroll
            for (int i = 0; i < BI; ++i)
                *reinterpret_cast<f32x4*>(&Bs[buf * B_STAGE + (tid + NT * i) * 4]) = rb[i];
        }
    };

    // ---- wave / lane coordinates -------------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, hh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // BDIR: the B (weight) fragments never touch LDS - every wave streams its own [32 k][32 n] blocks straight
    // from L2 into rolling registers (the packed layout [K/4][Npad][4] is already the fragment layout:
    // lane (n, k half) reads quad 2q + hh of column n), one chunk ahead
    f32x4 fbq[BDIR ? NQ : 1][TN];
    const unsigned fb_voff = (unsigned)((hh * p.Npad + n0 + wn * (BN / WGN) + l31) * 16);
    auto load_bfrag = [&](int c, int q) {
        const int ca = c + cbase;
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fbq[BDIR ? q : 0][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wrs, fb_voff + (unsigned)(j * 32 * 16), (unsigned)((ca * KQ + 2 * q) * p.Npad * 16), 0));
    };

    // ---- epilogue coordinates (known up front so the residual can be prefetched) ------------
    constexpr int LDC = BN + 4;
    constexpr int QPR = BN / 4;         // float4 quads per tile row
    constexpr int RPP = NT / QPR;       // rows per pass
    constexpr int NP = BM / RPP;        // passes over the tile rows
    const int cq = tid % QPR, r0 = tid / QPR;
    const int n = n0 + cq * 4;
    const bool full = (n + 3 < p.Cout) && p.vec_ok;
    f32x4 rr[NP];

    if constexpr (!SPLITK) {
        load_chunk(0);
        if (BDIR) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) load_bfrag(0, q);
        }
        store_chunk(0);
        __syncthreads();
    }

    // ---- one K chunk, hand-scheduled ---------------------------------------------------------
    // A wave issues in order, and an fp32 MFMA occupies the matrix pipe for 64 cycles while its
    // issue takes ~4: whatever is placed BETWEEN two MFMAs in program order executes for free
    // under the first one.  So the chunk is written as 16 steps (q, s) of TM*TN MFMAs with the
    // other work slotted between them - next chunk's buffer loads after step 0, the fragment
    // reads of sub-chunk q+1 inside sub-chunk q, the LDS writes of the staged next chunk in the
    // last four steps - and sched_barrier keeps hipcc from regrouping it.
    auto read_frags = [&](const float* Ab, const float* Bb, int q, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDA + q * 8);
        if (!BDIR) {
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + (q * 2 * BN + j * 32) * 4);
        }
    };
    auto chunk = [&](int c, auto prefetch) {
        constexpr bool PF = decltype(prefetch)::value;
        const int buf = c & 1;
        const float* Ab = As + buf * A_STAGE + (wm * (BM / WGM) + l31) * LDA + hh * 4;
        const float* Bb = Bs + buf * B_STAGE + (hh * BN + wn * (BN / WGN) + l31) * 4;
        float* Asn = As + (buf ^ 1) * A_STAGE;
        float* Bsn = Bs + (buf ^ 1) * B_STAGE;
        f32x4 fa[2][TM], fb[2][TN];
        read_frags(Ab, Bb, 0, fa[0], fb[0]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][i][s], BDIR ? fbq[BDIR ? q : 0][j][s] : fb[q & 1][j][s], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (BDIR && PF && s == 3) load_bfrag(c + 1, q);   // rolling: these registers are next read one chunk from now
                if (q == 0 && s == 0) {
                    if (PF) {
                        if (!TUNE_ABLATE(1)) load_chunk(c + 1);
                    } else if (!SPLITK && pres && full) {   // last chunk: fetch the residual rows of the epilogue
#pragma unroll
                        for (int ps = 0; ps < NP; ++ps) {
                            const int m = m0 + r0 + ps * RPP;
                            const size_t o = (m < p.M) ? (size_t)m * p.ldo + n : (size_t)n;
                            rr[ps] = *reinterpret_cast<const f32x4*>(pres + o);
                        }
                    }
                }
                if (s == 1 && q < NQ - 1) read_frags(Ab, Bb, q + 1, fa[(q + 1) & 1], fb[(q + 1) & 1]);
                if (PF && q == NQ - 1 && !TUNE_ABLATE(2)) {   // stage the next chunk: stores spread over the last 4 steps
#pragma unroll
                    for (int t = 0; t < AI + (BDIR ? 0 : BI); ++t) {
                        if ((t & 3) != s) continue;
                        if (t < AI)
                            *reinterpret_cast<f32x4*>(&Asn[(a_r + ARS * t) * LDA + a_kq * 4]) = ra[t];
                        else
                            *reinterpret_cast<f32x4*>(&Bsn[(tid + NT * (t - AI)) * 4]) = rb[t - AI];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    TUNE_T(t_loop);
    if constexpr (!SPLITK) {
        const int last = p.nchunks - 1;
        for (int c = 0; c < last; ++c) {
            chunk(c, std::true_type{});
            __syncthreads();
        }
        chunk(last, std::false_type{});
        __syncthreads();
    } else {
        // ---- split-K pipeline: a slice is 2-12 chunks on a CU that holds one or two workgroups - nothing hides a load
        // but distance.  Operands are fetched TWO chunks ahead (a chunk's MFMAs take ~0.4 us on a lone wave, the Infinity
        // Cache / HBM answer in 0.7-1 us): two register sets alternate by chunk parity (the loop is unrolled by two so that
        // the set is a compile-time index), A of chunk c+1 moves from registers into the other LDS stage during chunk c.
        // Loads past the end of the slice get an out-of-range offset: they return 0 without touching memory.
        f32x4 ra2[2][AI], fb2[2][NQ][TN];
        auto sk_load_a = [&](int cl, auto slot) {
            constexpr int SL = decltype(slot)::value;
            const bool oob = cl >= p.nchunks;
            const int c = cl + cbase;
            const int tap = IS1X1 ? 0 : c / p.cpc;
            const int c0 = IS1X1 ? c : c - tap * p.cpc;
            unsigned tap_bytes = 0;
            if (!IS1X1) {
                const int ky = tap / p.KW, kx = tap - ky * p.KW;
                tap_bytes = (unsigned)((ky * p.W + kx) * p.ldx * 4);
            }
            const bool second = DUAL && c >= p.cpc1;
            const unsigned s_a = oob ? 0u : (unsigned)((second ? c0 - p.cpc1 : c0) * BK * 4);
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                unsigned voff = a_voff[i];
                if (!IS1X1) voff = ((a_mask[i] >> (tap & 31)) & 1u) ? voff + tap_bytes : kOutOfRange;
                if (DUAL) voff = second ? a_voff2[i] : voff;
                if (oob) voff = kOutOfRange;
                if (DUAL) {
                    ra2[SL][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? x2rs : xrs, voff, s_a, 0));
                } else {
                    ra2[SL][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, s_a, 0));
                }
            }
        };
        auto sk_load_b = [&](int cl, int q, auto slot) {
            constexpr int SL = decltype(slot)::value;
            const bool oob = cl >= p.nchunks;
            const int ca = cl + cbase;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb2[SL][q][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    wrs, oob ? kOutOfRange : fb_voff + (unsigned)(j * 32 * 16), oob ? 0u : (unsigned)((ca * KQ + 2 * q) * p.Npad * 16), 0));
        };
        auto sk_chunk = [&](int c, auto par) {
            constexpr int P = decltype(par)::value;   // == c & 1
            const float* Ab = As + P * A_STAGE + (wm * (BM / WGM) + l31) * LDA + hh * 4;
            float* Asn = As + (P ^ 1) * A_STAGE;
            f32x4 fa[2][TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDA);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][i][s_], fb2[P][q][j][s_], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s_ == 3) sk_load_b(c + 2, q, par);          // this set's next use is two chunks from now
                    if (q == 0 && s_ == 0) sk_load_a(c + 2, par);   // (its previous content went to LDS during chunk c - 1)
                    if (s_ == 1 && q < NQ - 1) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[(q + 1) & 1][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDA + (q + 1) * 8);
                    }
                    if (q == NQ - 1) {   // chunk c + 1: registers -> the other stage, spread over the last four steps
#pragma unroll
                        for (int t_ = 0; t_ < AI; ++t_)
                            if ((t_ & 3) == s_) *reinterpret_cast<f32x4*>(&Asn[(a_r + ARS * t_) * LDA + a_kq * 4]) = ra2[P ^ 1][t_];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        const std::integral_constant<int, 0> even{};
        const std::integral_constant<int, 1> odd{};
        sk_load_a(0, even);
#pragma unroll
        for (int q = 0; q < NQ; ++q) sk_load_b(0, q, even);
        sk_load_a(1, odd);
#pragma unroll
        for (int q = 0; q < NQ; ++q) sk_load_b(1, q, odd);
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(a_r + ARS * i) * LDA + a_kq * 4]) = ra2[0][i];
        __syncthreads();
        // canonical boundaries (file header): leaf chain -> group fold -> result fold, every fold from +0, left to right
        f32x16 accG[TM][TN], accR[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { accG[i][j][r] = 0.f; accR[i][j][r] = 0.f; }
        int lc = 0, gl = 0;
        auto leaf_end = [&]() {
            if (++lc < p.sk_leaf) return;
            lc = 0;
            if (p.sk_unit == 1) return;          // the slab is the leaf itself
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { accG[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
            if (++gl < p.sk_G) return;
            gl = 0;
            if (p.sk_unit == p.sk_G) return;     // the slab is the group
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { accR[i][j][r] += accG[i][j][r]; accG[i][j][r] = 0.f; }
        };
        // The loop body is a whole (even, odd) pair and an odd chunk count ends in a peeled tail: with the odd chunk under an
        // `if` inside the loop the compiler's wait-count pass sees a path even -> even, assumes six fewer loads in flight and
        // waits for vmcnt(3) instead of vmcnt(9) at the top of every even chunk - the two-chunk prefetch distance collapses to
        // less than one.
        int c = 0;
        for (; c + 2 <= p.nchunks; c += 2) {
            sk_chunk(c, even);
            __syncthreads();
            leaf_end();
            sk_chunk(c + 1, odd);
            __syncthreads();
            leaf_end();
        }
        if (c < p.nchunks) {
            sk_chunk(c, even);
            __syncthreads();
            leaf_end();
        }
        if (p.sk_unit != 1) {
            const bool grp_level = p.sk_unit == p.sk_G;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = grp_level ? accG[i][j][r] : accR[i][j][r];
        }
    }
    TUNE_T(t_epi);

    // ---- epilogue -------------------------------------------------------------------------
    // The accumulators go through LDS once so that the HBM side is whole-row traffic: the
    // MFMA C/D layout (col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)) would give
    // 4-byte stores at a row stride; after the transpose every lane moves 16 contiguous bytes.
    // All waves have passed the barrier above, so the A/B stages are free to reuse.
    float* Cs = smem;
    float* const outp = pout;
    if (SPLITK && gridDim.y > 1) {
        // Partial tile -> workspace in accumulator order (16 bytes per lane, consecutive lanes consecutive: coalesced), then
        // the arrival ticket.  Per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's
        // stores, so the hand-off is the write-through form (MI355X_MICROARCH.md, inter-workgroup visibility): sc1 stores
        // leave the XCD's L2 for memory, every wave drains its stores (vmcnt 0), ONE lane takes the ticket with a relaxed
        // agent-scope atomic, and the last arriver reads all slabs with sc1 loads (no L1, fresh from the fabric) - no
        // cache-wide write-back / invalidate (a __threadfence() per workgroup measured 35 us per launch here).
        const unsigned S = gridDim.y;
        const size_t tile = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
        constexpr unsigned SLAB = BM * BN * 4;   // bytes
        constexpr int NQD = TM * TN * 4;          // 16-byte quads per lane
        float* const tile_ws = p.sk_ws + tile * S * (size_t)(BM * BN);
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(tile_ws, 0, S * SLAB, 0x00020000);
        const unsigned soff = (unsigned)blockIdx.y * SLAB;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][r4 * 4 + e];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs,
                                                           (unsigned)((((i * TN + j) * 4 + r4) * NT + tid) * 16), soff, /*sc1*/ 16);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slab stores have left for memory
        __syncthreads();
        int* const flag = reinterpret_cast<int*>(smem);     // (the one LDS array: the stages are free after the loop's last barrier)
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(p.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last_in = ticket == S - 1;
            // every slice has arrived: the counter is free again for the next launch / graph replay
            if (last_in) __hip_atomic_store(p.sk_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last_in;
        }
        __syncthreads();
        if (!*flag) return;
        __syncthreads();   // (the flag word is about to be overwritten by the transpose)
        // the rest of the canonical tree, whichever slice arrived last: leaf slabs fold G at a time into groups and the
        // groups into the result; group slabs fold into the result.  All slabs of a group (<= 4 x NQD loads) are in flight at once.
        const unsigned gsz = p.sk_unit == 1 ? (unsigned)p.sk_G : 1u;
        f32x4 tot[NQD];
#pragma unroll
        for (int u = 0; u < NQD; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[u][e] = 0.f;
        auto slab = [&](unsigned z, int u) {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)((u * NT + tid) * 16), z * SLAB, 16));
        };
        if (gsz == 4) {
            for (unsigned z = 0; z < S; z += 4) {
                f32x4 v[4][NQD];
#pragma unroll
                for (int zz = 0; zz < 4; ++zz)
#pragma unroll
                    for (int u = 0; u < NQD; ++u) v[zz][u] = slab(z + zz, u);
#pragma unroll
                for (int u = 0; u < NQD; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tg = 0.f;
#pragma unroll
                        for (int zz = 0; zz < 4; ++zz) tg += v[zz][u][e];
                        tot[u][e] += tg;
                    }
            }
        } else if (gsz > 1) {
            for (unsigned z = 0; z < S; z += gsz) {
                f32x4 tg[NQD];
#pragma unroll
                for (int u = 0; u < NQD; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) tg[u][e] = 0.f;
                for (unsigned zz = 0; zz < gsz; ++zz) {
                    f32x4 v[NQD];
#pragma unroll
                    for (int u = 0; u < NQD; ++u) v[u] = slab(z + zz, u);
#pragma unroll
                    for (int u = 0; u < NQD; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) tg[u][e] += v[u][e];
                }
#pragma unroll
                for (int u = 0; u < NQD; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) tot[u][e] += tg[u][e];
            }
        } else {
            unsigned z = 0;
            for (; z + 4 <= S; z += 4) {
                f32x4 v[4][NQD];
#pragma unroll
                for (int zz = 0; zz < 4; ++zz)
#pragma unroll
                    for (int u = 0; u < NQD; ++u) v[zz][u] = slab(z + zz, u);
#pragma unroll
                for (int zz = 0; zz < 4; ++zz)
#pragma unroll
                    for (int u = 0; u < NQD; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) tot[u][e] += v[zz][u][e];
            }
            for (; z < S; ++z) {
                f32x4 v[NQD];
#pragma unroll
                for (int u = 0; u < NQD; ++u) v[u] = slab(z, u);
#pragma unroll
                for (int u = 0; u < NQD; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) tot[u][e] += v[u][e];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][r4 * 4 + e] = tot[(i * TN + j) * 4 + r4][e];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / WGM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                Cs[row * LDC + wn * (BN / WGN) + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();

    const f32x4 sc = *reinterpret_cast<const f32x4*>(pscale + n);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(pshift + n);
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
            if (pres) {
                if (SPLITK) rr[ps] = *reinterpret_cast<const f32x4*>(pres + ((m < p.M) ? (size_t)m * p.ldo + n : (size_t)n));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rr[ps][e];
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (m < p.M && !TUNE_ABLATE(4)) *reinterpret_cast<f32x4*>(outp + (size_t)m * p.ldo + n) = v;
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
                    if (pres) t += pres[o + e];
                    if (p.relu) t = fmaxf(t, 0.f);
                    outp[o + e] = t;
                }
            }
        }
    }
#ifdef SPECMI_TUNE
    if (p.tprof && (tid & 63) == 0) {
        const long long t_end = __builtin_amdgcn_s_memtime();
        atomicAdd(p.tprof + 0, (unsigned long long)tp[0]);
        atomicAdd(p.tprof + 1, (unsigned long long)tp[1]);
        atomicAdd(p.tprof + 2, (unsigned long long)tp[2]);
        atomicAdd(p.tprof + 3, (unsigned long long)tp[3]);
        atomicAdd(p.tprof + 4, (unsigned long long)(t_loop - t_start));
        atomicAdd(p.tprof + 5, (unsigned long long)(t_epi - t_loop));
        atomicAdd(p.tprof + 6, (unsigned long long)(t_end - t_epi));
        atomicAdd(p.tprof + 7, 1ull);
    }
#endif
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

// One launch, gridDim.y = pl.leaves / pl.unit slabs per tile; sk.ws holds conv_igemm_sk_ws_floats(a, slabs, groups) floats,
// sk.cnt one zeroed counter per (tile, group) - the kernel leaves them zeroed
int launch_conv_igemm_sk(const ConvArgs& a, const SkPlan& pl, const SkWs& sk, const LaunchCtx& ctx, const ConvArgs* b) {
    const int nch = a.KH * a.KW * (a.Cin / 32) + (a.x2 ? a.Cin2 / 32 : 0);
    const int groups = b ? 2 : 1;
    if (pl.leaves < 1 || nch % pl.leaves != 0 || pl.G < 1 || pl.leaves % pl.G != 0 || (pl.unit != 1 && pl.unit != pl.G && pl.unit != pl.leaves) ||
        a.Cin % 32 != 0 || a.Npad % 64 != 0 || a.ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15))
        return (int)hipErrorInvalidValue;
    const int S = pl.leaves / pl.unit;
    if (S > 1 && (!sk.ws || !sk.cnt || conv_igemm_sk_ws_floats(a, S, groups) > sk.floats || conv_igemm_sk_tiles(a, groups) > sk.ncnt))
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
    if (img_bytes * a.B >= limit || (size_t)nch * 32 * a.Npad * 4 >= limit) return (int)hipErrorInvalidValue;   // small-M path: no batch splitting
    KArgs k;
    double flops, bytes;
    make_kargs(a, b, k, &flops, &bytes);
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
