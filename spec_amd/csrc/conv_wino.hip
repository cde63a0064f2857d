// conv_wino.hip - 3x3 / stride-1 / pad-1 convolution as Winograd F(2x2, 3x3) on the gfx950 fp32
// matrix cores, fused end to end (input transform -> 16 frequency GEMMs -> output transform ->
// BatchNorm scale/shift -> ReLU) in one launch.
//
// Serves the conv2 of every stride-1 ResNet-50 bottleneck of both trunks (reference call sites
// spec/models/hmr.py:92, camcalib/model.py:73; torchvision Bottleneck.conv2 + bn2 + relu).  The
// direct implicit GEMM (conv_igemm.hip) spends 9 MACs per (pixel, ci, co); F(2x2,3x3) spends
// 16 per 2x2 output tile = 4 per pixel, i.e. 2.25x fewer matrix-core cycles for the same result
// up to fp32 rounding (|err| ~ 1e-6 relative, the same class cuDNN picks for these layers).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      d: 4x4 input patch, g: 3x3 filter, Y: 2x2
//
// Work decomposition (CDNA4, wave64):
//   * GEMM view per frequency f in 0..15:  D_f[t][co] = sum_ci V_f[t][ci] * U_f[ci][co] with
//     t = 2x2 output tile index in [0, B*TH*TW).  Two wave layouts (template NF = frequencies per wave):
//       NF = 16: workgroup = 32 tiles x 128 co, wave w = co block w for ALL 16 frequencies: 16 accumulator
//                tiles of v_mfma_f32_32x32x2_f32 = 256 accumulator registers per lane, lane-local output
//                transform, one wave per SIMD, V triple buffered (picked from Cin = 512);
//       NF = 8 : workgroup = 32 tiles x 64 co, wave = (co block, frequency columns {2fh, 2fh+1}): 128
//                accumulators, two workgroups per CU cover each other's epilogues; the two halves of a co
//                block meet in the output transform through 32 floats per lane of LDS.
//   * A operand (transformed input V): the 256 threads each own one (tile, channel pair) per
//     16-channel stage: 16 buffer_load_b64 of the raw 4x4 patch (image border = hardware range
//     check -> 0.0), B^T d B in registers, 16 ds_write_b64 into the stage buffer laid out
//     [f][c2][tile][2] exactly as the MFMA A fragments are read back (ds_read_b64, conflict free).
//   * B operand (transformed filters U = G g G^T, computed in fp64 at commit): never staged -
//     every wave streams its own fragments straight from L2 with coalesced 16-byte loads
//     ([co block][ci/4][f/2][lane][(f&1)*2 + jj] in HBM: one load = a frequency pair), 1-2 micro-chunks
//     ahead, rolling registers.
//   * K loop: stage = 16 input channels = 4 micro-chunks of NF MFMA pairs, one barrier per stage; the
//     loads / transform / LDS writes of the next stage are slotted between the MFMAs of this one.
//   * Persistent workgroups: the (tile, stage) sequence of a workgroup is ONE software pipeline, so the
//     next tile's patch loads and transform run under the current tile's last stages.
//   * Epilogue: store addresses come from a 16-byte-per-tile LDS table written by the loader threads;
//     BatchNorm scale/shift + ReLU fused; buffer stores (out-of-range offset = masked).
#include "specmi_internal.h"

namespace specmi {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WArgs {
    const float* x;
    const float* u;
    const float* scale;
    const float* shift;
    float* out;
    const float* res;      // residual added after BN, same layout as out (two-wave layout only), or nullptr
    unsigned x_bytes, u_bytes, out_bytes;
    int H, W, ldx, Cout, ldo;
    int TH, TW, THW, Mt;   // 2x2 output tiles per column / row / image / launch
    int nbm;               // tile rows: ceil(Mt / 32)
    int nbn;               // co columns: Cout / (32 * co blocks per workgroup)
    int nstage;            // Cin / 16
    unsigned mg_thw, sh_thw, mg_tw, sh_tw;
    int relu;
    int cpx;               // co columns per XCD (see the workgroup -> (column, tile row) map in the kernel); 1 = one column per XCD
    // grouped launch: the second half of the (persistent) grid runs the SAME layer shape of a second network on these tensors
    // (the CamCalib and SPEC trunks as one launch per layer): twice the tile rows to deal out, half the rounding loss of the
    // last persistent round
    int groups;
    struct { const float *x, *u, *scale, *shift, *res; float* out; } g1;
#ifdef WINO_PROF
    unsigned long long* tprof;   // [prologue, loop, epilogue, count] summed s_memtime ticks (wave 0 of each workgroup)
#endif
};

#ifdef WINO_ABLATE   // compile-time perf ablation (wrong results): 1 no U loads, 2 no raw loads, 4 no transform, 8 no A reads, 16 no stage barrier, 32 no output stores, 64 no epilogue exchange
#define WABL(bit) ((WINO_ABLATE) & (bit))
#else
#define WABL(bit) 0
#endif
constexpr unsigned kOOB = 0x80000000u;
constexpr int WINO_C2_STRIDE = 272;                  // bytes: 32 tiles x 8 B + 16 B pad (conflict-free b64 writes)
constexpr int WINO_F_STRIDE = 8 * WINO_C2_STRIDE;    // 8 channel pairs per stage
constexpr int WINO_BUF = 16 * WINO_F_STRIDE;         // 34816 B per stage buffer
constexpr int WINO_TAB = 32 * 16;                    // per tile: byte offsets of its 2x2 outputs (or out of range)

// lane id recomputed where it is needed: volatile, so the compiler neither hoists it out of the tile loop nor keeps the first
// copy alive (with no register to spare across the K loop either would end in scratch)
__device__ __forceinline__ int wino_lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ int wino_div(int n, int d, unsigned mg, unsigned sh) {
    return d == 1 ? n : (int)(__umulhi((unsigned)n, mg) >> sh);
}

// NF = frequencies per wave.
//   NF == 16: workgroup = 32 tiles x 128 co, wave w = co block w, all frequencies (256 accumulator
//             registers, one wave per SIMD), V triple buffered.
//   NF == 8 : workgroup = 32 tiles x 64 co, wave = (co block w & 1, frequency columns j in {2fh, 2fh+1}
//             with fh = w >> 1): 128 accumulator registers, two workgroups per CU, V double buffered;
//             the two frequency halves of a co block meet in the output transform through LDS.
// Workgroups are persistent: workgroup (tile_n, j) walks the tile rows j, j + G, j + 2G ... of its co
// column, and the (tile, stage) sequence is one continuous software pipeline - the raw loads, the input
// transform and the U / A fragments of the next tile's first stages are issued under the MFMAs of the
// current tile's last stages, so only the first tile of a workgroup pays a prologue.
// RES: out = [ReLU](BN(conv) + residual) - the BasicBlock tail of ResNet-34 / HRNet (pare BasicBlock.forward: out += identity
// before the ReLU).  Two-wave layout only: its 32 residual values per lane are fetched between the two halves of the output
// transform, when the 128 accumulator registers have just died, and before the first store (loads and stores retire in
// order on one counter: a load issued after a store would wait for it).
// WIDE (Cin % 32 == 0): the patch loads are 16 bytes per lane and fetch a PAIR of stages at once - half the vector-memory
// instructions per stage.  A stage's 16 channels are then not 16 consecutive ones: of every 16-byte quad of a 128-byte group of
// 32 channels, the even stage takes the first 8 bytes and the odd stage the second 8 (channel 32 m + 4 q + 2 (st & 1) + e is
// local channel 2 q + e of stage 2 m + (st & 1)); pack_wino_weights() orders U's k axis the same way.  Everything after the
// load (thread = (tile, local channel pair), V layout in LDS, MFMA order) is unchanged.
template <int NF, bool RES = false, bool WIDE = false>
__global__ void __launch_bounds__(256, NF == 16 ? 1 : 2) conv_wino_f32_kernel(const WArgs p) {
    static_assert(!(RES && NF == 16), "the residual epilogue exists for the two-wave layout only");
    constexpr int NWN = NF == 16 ? 4 : 2;          // co blocks (of 32) per workgroup
    constexpr bool TRIPLE = NF == 16;
#ifdef WINO_LATE_BARRIER
    constexpr bool EARLYB = TRIPLE;                 // (measurement only) the old placement: the barrier closes the stage
#else
    constexpr bool EARLYB = true;
#endif
    constexpr int SPS = 16 / NF;                    // loader pieces per MFMA step
#ifndef WINO_UD
#define WINO_UD 2
#endif
#ifndef WINO_AD
#define WINO_AD 8
#endif
    // U prefetch distance in micro-chunks (NF MFMA pairs each) and depth of the A-fragment ring in steps.  Measured (round 3,
    // profiles/r03_n_wino_prefetch_distance.txt): UD = 4 (a whole stage ahead, registers taken from a 4-deep A ring: 251 VGPRs,
    // no spills) and UD = 2 run at the same speed with one and with two workgroups per CU - the stall the patch loads cause is
    // the wait for the patch data itself (3 micro-chunks after issue), not U fragments queued behind them
    constexpr int UD = NF == 16 ? 1 : WINO_UD;
    constexpr int TAB0 = (TRIPLE ? 3 : 2) * WINO_BUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef WINO_PROF
    const long long t_start = __builtin_amdgcn_s_memtime();
    long long t_epi_sum = 0, t_seg[5] = {0, 0, 0, 0, 0};   // epilogue: transform + send, barrier, finish + stores, barrier; stage barriers
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbw = wave % NWN, fh = wave / NWN;

    // co column = blockIdx % nbn: with nbn in {1, 2, 4, 8} an XCD (blockIdx % 8) keeps one column, so its
    // U slice stays in that XCD's L2
    const int nwg = p.groups == 2 ? (int)(gridDim.x >> 1) : (int)gridDim.x;   // workgroups of one network
    const bool grp = p.groups == 2 && (int)blockIdx.x >= nwg;                   // wave-uniform: scalar selects below
    const int bid = grp ? (int)blockIdx.x - nwg : (int)blockIdx.x;
    const float* const px = grp ? p.g1.x : p.x;
    const float* const pu = grp ? p.g1.u : p.u;
    const float* const pscale = grp ? p.g1.scale : p.scale;
    const float* const pshift = grp ? p.g1.shift : p.shift;
    const float* const pres = grp ? p.g1.res : p.res;
    float* const pout = grp ? p.g1.out : p.out;
    int tile_n = bid % p.nbn;
    const int G = nwg / p.nbn;                     // workgroups per co column (of one network)
    int tile_m = bid / p.nbn;
    if (p.cpx > 1) {
        // With one column per XCD every input patch is fetched by nbn XCDs (measured 2 - 3.3x the algorithmic bytes,
        // profiles/r03_v_layer_traffic.txt).  When the U slices of cpx columns fit an XCD's L2 together, the cpx workgroups that
        // walk the same tile rows are placed on ONE XCD (workgroup id % 8) instead, and the patches are fetched by nbn / cpx XCDs.
        const int xcd = bid & 7, j = bid >> 3;
        const int ngroups = p.nbn / p.cpx, xpg = 8 / ngroups;       // column groups; XCDs serving one group
        tile_n = (xcd % ngroups) * p.cpx + j % p.cpx;
        tile_m = (j / p.cpx) * xpg + xcd / ngroups;
    }

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pu), 0, p.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(pout, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? pres : px), 0, RES ? p.out_bytes : 0, 0x00020000);

    // ---- loader role: thread = (tile tl, channel pair c2l) ------------------------------------
    const int c2l = tid & 7, tl = tid >> 3;
    // patch element (dy, dx) lives at a_row[dy] + a_col[dx]: 8 registers instead of 16 offsets.  An invalid row is 0xC0000000, an
    // invalid column 0x80000000: any sum with an invalid part is >= 2^30 modulo 2^32, i.e. out of range for the <= 1 GiB input
    // the launcher passes (the hardware range check returns 0.0 = the zero padding)
    unsigned a_row[4], a_col[4];
    // decode tile row tm: the 16 patch offsets of this thread's tile and, for the epilogue, the byte
    // offsets of the tile's 2x2 outputs (table slot `slot`; out-of-range offset = no load / no store)
    auto decode_tile = [&](int tm, int slot) {
        // everything here is derived from a freshly computed thread id: values the compiler could recognise as loop invariants
        // (tid >> 3, tid & 7, ...) would be hoisted out of the tile loop and, with no register to spare, spilled to scratch
        const int tid_ = wave * 64 + wino_lane_now();
        const int c2l = tid_ & 7, tl = tid_ >> 3;
        const int t = tm * 32 + tl;
        const bool ok = t < p.Mt;
        const int tt = ok ? t : 0;
        const int b = wino_div(tt, p.THW, p.mg_thw, p.sh_thw);
        const int rem = tt - b * p.THW;
        const int ty = wino_div(rem, p.TW, p.mg_tw, p.sh_tw);
        const int tx = rem - ty * p.TW;
        const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
        // per-thread base + wave-uniform (dy, dx) term: the uniform part stays in SGPRs
        const unsigned base = (unsigned)(((b * p.H + iy0) * p.W + ix0) * p.ldx * 4 + c2l * (WIDE ? 16 : 8));
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            a_row[d] = (ok && (unsigned)(iy0 + d) < (unsigned)p.H) ? base + (unsigned)(d * p.W * p.ldx * 4) : 0xC0000000u;
            a_col[d] = (unsigned)(ix0 + d) < (unsigned)p.W ? (unsigned)(d * p.ldx * 4) : 0x80000000u;
        }
        if (c2l < 4) {
            const int oy = 2 * ty + (c2l >> 1), ox = 2 * tx + (c2l & 1);
            const bool in = ok && oy < p.H && ox < p.W;
            reinterpret_cast<unsigned*>(smem + TAB0 + slot * WINO_TAB)[tl * 4 + c2l] =
                in ? (unsigned)(((b * p.H + oy) * p.W + ox) * p.ldo * 4) : kOOB;
        }
    };
    f32x2 raw[WIDE ? 1 : 16];
    f32x4 raw4[WIDE ? 16 : 1];                      // WIDE: .xy = the even stage of the pair, .zw = the odd one
    auto load_raw1 = [&](int st, int q) {           // WIDE: st is the even stage of the pair
        if (WIDE) raw4[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, a_row[q >> 2] + a_col[q & 3], (unsigned)((st >> 1) * 128), 0));
        else raw[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, a_row[q >> 2] + a_col[q & 3], (unsigned)(st * 64), 0));
    };
    auto rawv = [&](int q, int half) {
        if (!WIDE) return raw[q];
        f32x2 r;
        r[0] = raw4[q][2 * half]; r[1] = raw4[q][2 * half + 1];
        return r;
    };
    // V = B^T d B on both channels of the pair at once (packed fp32 adds), cut into 12 pieces so that the
    // K loop can slot them under the MFMAs: pieces 0-3 = row pass of patch column j (the four raw values
    // of the column die, four T values are born: T and raw share registers), pieces 4-11 = column pass
    // of half a T row + the LDS writes of its two frequencies.
    f32x2 T[16];
    // this thread's slot in a stage buffer, recomputed where it is used (see decode_tile: no register to keep it in)
    auto vw_base_now = [&]() {
        const int t_ = wave * 64 + wino_lane_now();
        return smem + (t_ & 7) * WINO_C2_STRIDE + (t_ >> 3) * 8;
    };
    auto transform_piece = [&](int s, char* dst, int half) {
        if (s < 4) {
            const int j = s;
            const f32x2 r0 = rawv(0 + j, half), r1 = rawv(4 + j, half), r2 = rawv(8 + j, half), r3 = rawv(12 + j, half);
            T[0 + j] = r0 - r2;
            T[4 + j] = r1 + r2;
            T[8 + j] = r2 - r1;
            T[12 + j] = r1 - r3;
        } else {
            const int i = (s - 4) >> 1;
            if ((s & 1) == 0) {
                *reinterpret_cast<f32x2*>(dst + (4 * i + 0) * WINO_F_STRIDE) = T[4 * i + 0] - T[4 * i + 2];
                *reinterpret_cast<f32x2*>(dst + (4 * i + 1) * WINO_F_STRIDE) = T[4 * i + 1] + T[4 * i + 2];
            } else {
                *reinterpret_cast<f32x2*>(dst + (4 * i + 2) * WINO_F_STRIDE) = T[4 * i + 2] - T[4 * i + 1];
                *reinterpret_cast<f32x2*>(dst + (4 * i + 3) * WINO_F_STRIDE) = T[4 * i + 1] - T[4 * i + 3];
            }
        }
    };
    // pieces of step fl of the u == 1 micro-chunk (NF steps): NF == 16: piece fl (12 used); NF == 8: one row
    // piece per step for steps 0-3, two column pieces per step for steps 4-7
    auto transform_step = [&](int fl, char* dst, int half) {
        if (NF == 16) {
            if (fl < 12) transform_piece(fl, dst, half);
        } else if (fl < 4) {
            transform_piece(fl, dst, half);
        } else {
            transform_piece(4 + 2 * (fl - 4), dst, half);
            transform_piece(5 + 2 * (fl - 4), dst, half);
        }
    };

    // ---- consumer role: lane = (tile row l31 / co column l31, k half hh) ------------------------
    // local frequency fl -> f = 4i + j:  NF == 16: f = fl;  NF == 8: i = fl >> 1, j = 2 fh + (fl & 1)
    auto f_of = [](int fl) { return NF == 16 ? fl : 4 * (fl >> 1) + (fl & 1); };
    const int l31 = lane & 31, hh = lane >> 5;
    const int nb = tile_n * NWN + nbw;
    const int S = p.nstage, nmu = S * 4;
    // U fragments: one 16-byte load per lane covers the frequency pair (f, f+1) of a micro-chunk
    // ([co block][mu][f / 2][lane][(f & 1) * 2 + jj] in HBM) - vector-memory instructions, not bytes, are
    // what the CU's address unit runs out of next to 16 patch loads per stage
    const unsigned u_voff = (unsigned)(lane * 16);
    const unsigned u_block = (unsigned)(nb * nmu) * 8192u + (unsigned)(fh * 1024);   // 8 pairs x 1 KiB per micro-chunk
    f32x4 bq[UD][NF / 2];
    auto load_u = [&](int slot, int mu, int fp) {   // fp = local pair index: local frequencies 2 fp, 2 fp + 1
        bq[slot][fp] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u_voff, u_block + (unsigned)(mu * 8192 + (f_of(2 * fp) >> 1) * 1024), 0));
    };
    const char* const vr_base = smem + hh * WINO_C2_STRIDE + l31 * 8 + fh * 2 * WINO_F_STRIDE;
    auto read_a = [&](const char* vr, int u, int fl) {
        return *reinterpret_cast<const f32x2*>(vr + f_of(fl) * WINO_F_STRIDE + u * 2 * WINO_C2_STRIDE);
    };
    const int co = nb * 32 + l31;
    // the two-wave layout has no register to spare across the K loop: it re-reads these two in every epilogue
    float sc = 0.f, sh = 0.f;
    // channels past Cout (last co column of a Cout % 64 == 32 layer: zero U block, see pack_wino_weights) are never stored:
    // 2^30 added to any offset of a <= 2^30-byte output (launch_conv_wino guarantees that for such layers) is out of
    // range, and added to kOOB it stays out of range (no wrap to a valid address)
    unsigned co_b = 0;
    auto emit = [&](float v, unsigned off) {   // one-wave layout
        v = fmaf(v, sc, sh);
        if (p.relu) v = fmaxf(v, 0.f);
        if (!WABL(32)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, off + co_b, 0, 0);
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    // ---- prologue of the workgroup's first tile --------------------------------------------------
    decode_tile(tile_m, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) load_raw1(0, q);
#pragma unroll
    for (int d = 0; d < UD; ++d)
#pragma unroll
        for (int f = 0; f < NF / 2; ++f) load_u(d, d, f);   // nmu >= 4 > UD
    {
        char* const vw0 = vw_base_now();
#pragma unroll
        for (int s = 0; s < 12; ++s) transform_piece(s, vw0, 0);
    }
    if (!WIDE) {
#pragma unroll
        for (int q = 0; q < 16; ++q) load_raw1(S > 1 ? 1 : 0, q);
    }
    __syncthreads();
    constexpr int AD = NF == 16 ? 16 : (WIDE ? 4 : WINO_AD);     // A fragments (LDS, ~150 cycles away) are fetched AD steps = AD x 128 cycles ahead
    f32x2 af[AD];
#pragma unroll
    for (int f = 0; f < AD; ++f) af[f] = read_a(vr_base, 0, f);

    // ---- K loop, hand-scheduled -------------------------------------------------------------
    // A wave issues in order and an fp32 MFMA holds the matrix pipe for 64 cycles, so whatever sits
    // between two MFMA pairs in program order runs under them.  One stage = 4 micro-chunks x NF
    // steps; step (u, fl) = the two MFMAs of one frequency plus one slice of everything else:
    //   every step : stream U(mu+UD, f) and the A fragment of the next micro-chunk into the registers
    //                just consumed (rolling: they are needed NF steps later)
    //   u == 1     : pieces of the input transform of the next stage (writes V into the next buffer)
    //   u == 2     : raw patch loads of the stage after that
    // "next stage" runs on into the next tile of this workgroup: its patch offsets replace the current
    // ones before stage S-2 (whose loads are the first that need them).
    // The single barrier of a stage sits after u == 2: by then every wave has issued its last reads of this stage's
    // buffer (micro-chunk 3's fragments are fetched during u == 2) and finished the next stage's V (written during
    // u == 1), so micro-chunk 3 already prefetches the first fragments of the next stage and no wave waits on LDS after
    // a barrier.  Two buffers are enough for that (the writes of stage st + 1 go to the buffer whose last reads preceded
    // this barrier); the one-wave layout keeps a third so that a fast wave may run a whole stage ahead.
#ifdef WINO_PROF
    const long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    int o_cur = 0, o_nxt = WINO_BUF, o_nn = TRIPLE ? 2 * WINO_BUF : 0;
    for (int it = 0;; ++it) {
        const int next_m = tile_m + G;
        const bool has_next = next_m < p.nbm;
        const int nm = has_next ? next_m : tile_m;   // nothing follows: keep prefetching this tile (unused)
        f32x16 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

        // par = st & 1 as a literal (WIDE runs the stages in pairs: which half of raw4 a transform reads, and whether the
        // stage loads, must be known at compile time)
        auto stage_body = [&](const int st, const int par) {
            if (S > 1 && st == (S > 2 ? S - 2 : 0)) decode_tile(nm, (it + 1) & 1);
            const char* vr_cur = vr_base + o_cur;
            const char* vr_nxt = vr_base + o_nxt;
            char* vw_nxt = vw_base_now() + o_nxt;
            // stage after next; past the tile's end it is the next tile's stage 0 / 1 (S == 1 never has a next tile)
            const int st2 = st + 2 < S ? st + 2 : (st + 2 - S < S ? st + 2 - S : S - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int mu = st * 4 + u;
                const int mu1 = mu + UD < nmu ? mu + UD : mu + UD - nmu;
#pragma unroll
                for (int fl = 0; fl < NF; ++fl) {
                    acc[fl] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fl % AD][0], bq[u % UD][fl >> 1][(fl & 1) * 2 + 0], acc[fl], 0, 0, 0);
                    acc[fl] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fl % AD][1], bq[u % UD][fl >> 1][(fl & 1) * 2 + 1], acc[fl], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!WABL(1) && (fl & 1)) load_u(u % UD, mu1, fl >> 1);
                    if (!WABL(8)) {
                    {   // rolling: the fragment AD steps ahead goes into the register pair just consumed
                        const int fn = fl + AD;
                        if (fn < NF) af[fl % AD] = read_a(vr_cur, u, fn);
                        else if (u < 3) af[fl % AD] = read_a(vr_cur, u + 1, fn - NF);
                        else if (EARLYB) af[fl % AD] = read_a(vr_nxt, 0, fn - NF);
                    }
                    }
                    if (u == 1 && !WABL(4)) transform_step(fl, vw_nxt, WIDE ? 1 - par : 0);
                    if (u == 2 && !WABL(2) && (!WIDE || par == 0)) {
#pragma unroll
                        for (int k = 0; k < SPS; ++k) load_raw1(st2, fl * SPS + k);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (EARLYB && u == 2 && !WABL(16)) __syncthreads();
            }
            if (!EARLYB) {
#if defined(WINO_PROF) && WINO_PROF > 1
                const long long t_b0 = __builtin_amdgcn_s_memtime();
                if (!WABL(16)) __syncthreads();
                t_seg[4] += __builtin_amdgcn_s_memtime() - t_b0;
#else
                if (!WABL(16)) __syncthreads();
#endif
#pragma unroll
                for (int fl = 0; fl < AD; ++fl) af[fl] = read_a(vr_nxt, 0, fl);
            }
            const int t = o_cur; o_cur = o_nxt; o_nxt = TRIPLE ? o_nn : t; o_nn = t;
        };
        if (WIDE) {
            for (int st = 0; st < S; st += 2) {
                stage_body(st, 0);
                stage_body(st + 1, 1);
            }
        } else {
            for (int st = 0; st < S; ++st) stage_body(st, 0);
        }

        // ---- epilogue of this tile: lane-local output transform, BN scale/shift, ReLU, store ------
        // Accumulator register r of a lane belongs to tile (r & 3) + 8 (r >> 2) + 4 hh: its four output
        // addresses come from the table the loader threads left in LDS (16 bytes per tile), so an
        // output costs one add, one fma, one max and a buffer store.
#ifdef WINO_PROF
        const long long t_e0 = __builtin_amdgcn_s_memtime();
#endif
        if (NF == 16) {
            const int lane_ = wino_lane_now();              // see decode_tile
            const int co_ = nb * 32 + (lane_ & 31);
            sc = pscale[co_]; sh = pshift[co_];
            co_b = co_ < p.Cout ? (unsigned)(co_ * 4) : 0x40000000u;
            const char* tab = smem + TAB0 + (it & 1) * WINO_TAB + (lane_ >> 5) * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const u32x4 to = *reinterpret_cast<const u32x4*>(tab + ((r & 3) + 8 * (r >> 2)) * 16);
                float Sx[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Sx[0][j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
                    Sx[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // the SAME association as the two-wave layout below ((S0 + S1) + S2 and ((-S2) - S3) + S1), so both
                    // layouts give bit-identical outputs and the launcher may pick either one by grid size
                    emit(Sx[i][0] + Sx[i][1] + Sx[i][2], to[2 * i]);
                    emit((-Sx[i][2] - Sx[i][3]) + Sx[i][1], to[2 * i + 1]);
                }
            }
        } else {
            // This wave holds D[i][j] for j in {2fh, 2fh+1} (acc[2i + jl]).  Y[a][0] = S[a][0]+S[a][1]+S[a][2],
            // Y[a][1] = S[a][1]-S[a][2]-S[a][3] with S = row transform (lane-local): the fh = 0 wave finishes
            // output column 0 and needs S[a][2] from its partner; the fh = 1 wave finishes column 1 and
            // needs S[a][1].  32 floats per lane cross through the stage buffer that the last stage just
            // released (the other one already holds the next tile's first stage).
            const int lane = wino_lane_now();               // see decode_tile
            const int l31 = lane & 31, hh = lane >> 5;
            const int co = nb * 32 + l31;
            sc = pscale[co]; sh = pshift[co];
            const int coq = nb * 32 + (lane & 7) * 4;      // first of the four channels this lane stores
            const unsigned cq_b = coq < p.Cout ? (unsigned)(coq * 4) : 0x40000000u;
            char* const xw = smem + o_nxt + wave * 8192 + lane * 4;               // this wave's 8 KB: [r][a][lane]
            const char* const xr = smem + o_nxt + (wave ^ NWN) * 8192 + lane * 4;   // the partner's
            float mine[16][2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float Sx[2][2];
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) {
                    Sx[0][jl] = acc[0 + jl][r] + acc[2 + jl][r] + acc[4 + jl][r];
                    Sx[1][jl] = acc[2 + jl][r] - acc[4 + jl][r] - acc[6 + jl][r];
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    // fh == 0: keep S0+S1 (for Y[a][0]), send S1 (for Y[a][1]);  fh == 1: keep -S2-S3, send S2
                    mine[r][a] = fh == 0 ? Sx[a][0] + Sx[a][1] : -Sx[a][0] - Sx[a][1];
                    *reinterpret_cast<float*>(xw + (r * 2 + a) * 256) = fh == 0 ? Sx[a][1] : Sx[a][0];
                }
            }
#ifdef WINO_PROF
            const long long t_e1 = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef WINO_PROF
            const long long t_e2 = __builtin_amdgcn_s_memtime();
#endif
            // Finish this wave's 32 outputs per lane (BN scale / shift applied here, where a lane owns one channel), then turn
            // them by 90 degrees through LDS so that a lane holds FOUR consecutive channels of one pixel: 8 x 16-byte stores per
            // lane instead of 32 dword stores.  The vector-memory unit of a CU takes store instructions one after the other
            // whatever their width, and it is shared with the co-resident workgroup's K loop loads, so the dword version held
            // up both workgroups.  The turn goes through the PARTNER's 8 KB (this wave is its only reader and has just read it;
            // one wave's LDS instructions execute in order, so the reads above precede the writes below).
            char* const tw = smem + o_nxt + (wave ^ NWN) * 8192;   // [tile][a][32 channels]
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    mine[r][a] = fmaf(mine[r][a] + *reinterpret_cast<const float*>(xr + (r * 2 + a) * 256), sc, sh);
            // (all 32 reads of the region are issued before the first write into it)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    *reinterpret_cast<float*>(tw + ((((r & 3) + 8 * (r >> 2) + 4 * hh) * 2 + a) * 32 + l31) * 4) = mine[r][a];
            // lane -> (tile t = 4 i + lane / 16, pixel row a = (lane / 8) & 1, channel quad q = lane & 7); pixel column b = fh
            const char* const tabw = smem + TAB0 + (it & 1) * WINO_TAB + (lane >> 4) * 16 + ((lane >> 3) & 1) * 8 + fh * 4;
            f32x4 y[8];
            unsigned yo[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                y[i] = *reinterpret_cast<const f32x4*>(tw + (i * 64 + lane) * 16);
                yo[i] = *reinterpret_cast<const unsigned*>(tabw + i * 64) + cq_b;
            }
            if (RES) {
                f32x4 rv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)   // out-of-range tile / channel quad: the offset is out of range and reads 0
                    rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, yo[i], 0, 0));
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] += rv[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (p.relu)
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[i][c] = fmaxf(y[i][c], 0.f);
                if (!WABL(32)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[i]), ors, yo[i], 0, 0);
            }
#ifdef WINO_PROF
            const long long t_e3 = __builtin_amdgcn_s_memtime();
#endif
            if (has_next) __syncthreads();   // the next tile's stage 0 transforms into this buffer
#ifdef WINO_PROF
            t_seg[0] += t_e1 - t_e0; t_seg[1] += t_e2 - t_e1; t_seg[2] += t_e3 - t_e2; t_seg[3] += __builtin_amdgcn_s_memtime() - t_e3;
#endif
        }
        // a two-stage tile rewrites the store table before its first barrier
        if (NF == 16 && has_next && S <= 2) __syncthreads();
#ifdef WINO_PROF
        t_epi_sum += __builtin_amdgcn_s_memtime() - t_e0;
#endif
        if (!has_next) break;
        tile_m = next_m;
    }
#ifdef WINO_PROF
    if (p.tprof && tid == 0) {
        const long long t_end = __builtin_amdgcn_s_memtime();
        atomicAdd(p.tprof + 0, (unsigned long long)(t_loop - t_start));
        atomicAdd(p.tprof + 1, (unsigned long long)(t_end - t_loop - t_epi_sum));
        atomicAdd(p.tprof + 2, (unsigned long long)t_epi_sum);
        atomicAdd(p.tprof + 3, 1ull);
        for (int i = 0; i < 5; ++i) atomicAdd(p.tprof + 4 + i, (unsigned long long)t_seg[i]);
    }
#endif
}

static void wino_magic(unsigned d, unsigned* mg, unsigned* sh) {
    if (d < 2) { *mg = 0; *sh = 0; return; }
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    const unsigned L = 31 + s;
    *mg = (unsigned)((1ull << L) / d + 1ull);
    *sh = L - 32;
}

bool conv_wino_supported(const ConvArgs& a) {
    // Cout % 64 == 32 runs the 64-channel layout with half of its last co column idle: worth it from 96 channels
    // (<= 25 % idle MFMAs against the direct kernel's 2.25x flops), not for 32
    return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 16 == 0 &&
           (a.Cout % 64 == 0 || (a.Cout % 32 == 0 && a.Cout > 64)) && a.ldx % 2 == 0 && a.OH == a.H && a.OW == a.W &&
           (!a.res || (reinterpret_cast<uintptr_t>(a.res) & 3) == 0);
}

// OIHW (cout, cin, 3, 3) -> U = G g G^T packed [cout/32][cin/4][f/2][lane = h*32+n][(f&1)*2 + jj], f = 4i+j, ci = 4*mu + 2*h + jj
#ifndef WINO_NO_WIDE
static inline bool wino_wide_cin(int cin) { return cin % 32 == 0; }
#else
static inline bool wino_wide_cin(int) { return false; }
#endif

void pack_wino_weights(const float* w, int cout, int cin, std::vector<float>& out) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    out.assign((size_t)16 * cin * cout, 0.f);
    const int nmu = cin / 4;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            double Gg[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) Gg[i][b] = G[i][0] * g[0 * 3 + b] + G[i][1] * g[1 * 3 + b] + G[i][2] * g[2 * 3 + b];
            // k position of input channel ci: plain order, or (Cin % 32 == 0, the WIDE kernels) the stage-pair order
            const int r = ci % 32;
            const int kp = wino_wide_cin(cin) ? (ci / 32) * 32 + ((r % 4) / 2) * 16 + (r / 4) * 2 + (r % 2) : ci;
            const int nb = co / 32, n = co % 32, mu = kp / 4, h = (kp % 4) / 2, jj = kp % 2;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double uij = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
                    const int f = 4 * i + j;
                    out[((((size_t)nb * nmu + mu) * 8 + (f >> 1)) * 64 + h * 32 + n) * 4 + (f & 1) * 2 + jj] = (float)uij;
                }
        }
}

#ifdef WINO_PROF
static unsigned long long* g_wino_tprof = nullptr;

void conv_wino_set_tprof(unsigned long long* p) { g_wino_tprof = p; }
#endif
static int g_wino_persistent = 1;
void conv_wino_set_persistent(int v) { g_wino_persistent = v; }

static int wino_pick(const ConvArgs& a, int groups = 1) {
    // ConvArgs::wino_variant (per handle): 0 auto, 16 / 8 = force the frequencies-per-wave variant
    if (a.wino_variant == 8 || a.res) return 8;    // the residual epilogue exists in the two-wave layout only
    if (a.wino_variant == 16 && a.Cout % 128 == 0) return 16;
    // Measured on MI355X at B=256 (tools/wino_bench): two 128-accumulator waves per SIMD cover each other's
    // prologue / epilogue and win by ~10 % up to Cin = 256; from Cin = 512 (32 stages per tile) the
    // 128-channel workgroup (half the A traffic per MFMA) is ahead.
    if (!(a.Cout % 128 == 0 && a.Cin >= 512)) return 8;
    // Small batches: a 32t x 128 grid of a few workgroups walks its 32+ stages alone (layer4 at B = 1: 4 workgroups,
    // 132 us); the 64-channel layout has twice the workgroups and half the serial MFMA chain.  The two layouts are
    // bit-identical (same k order, same output-transform association), so the choice never changes a result.
    const long tiles = (long)a.B * ((a.H + 1) / 2) * ((a.W + 1) / 2);
    const long wgs16 = ((tiles + 31) / 32) * (a.Cout / 128) * groups;
    return wgs16 < 128 ? 8 : 16;
}

template <int NF, bool RES = false, bool WIDE = false>
static int wino_launch_variant(WArgs k, int Cout, const LaunchCtx& ctx, double flops, double bytes) {
    constexpr int NT = NF == 16 ? 128 : 64;
    constexpr int smem = (NF == 16 ? 3 : 2) * WINO_BUF + 2 * WINO_TAB;
    static DevOnce once;
    if (int e = set_dyn_lds_once(once, reinterpret_cast<const void*>(&conv_wino_f32_kernel<NF, RES, WIDE>), smem)) return e;
    k.nbn = (Cout + NT - 1) / NT;
    k.nbm = (k.Mt + 31) / 32;
    // persistent grid: as many workgroups as the chip holds at once (256 CUs x 1 or 2), split evenly over the
    // co columns; a single 16-channel stage cannot pipeline across tiles (the loads run two stages ahead)
    int G = (256 * (NF == 16 ? 1 : 2)) / k.nbn / k.groups;    // per network
#ifdef WINO_PROF
    if (const char* e = getenv("WINO_WG_PER_CU")) G = (256 * atoi(e)) / k.nbn / k.groups;
#endif
    if (G < 1) G = 1;
    if (G > k.nbm || k.nstage < 2 || g_wino_persistent == 0) G = k.nbm;
    // columns per XCD: as many as keep their U slices (16 frequencies x Cin x NT channels) within 2 MiB of the 4 MiB L2
    k.cpx = 1;
#ifndef WINO_NO_CPX
    {
        const size_t u_col = (size_t)16 * k.nstage * 16 * NT * 4;
        int c = 1;
        while (c * 2 <= k.nbn && k.nbn % (c * 2) == 0 && (size_t)(c * 2) * u_col <= ((size_t)2 << 20)) c *= 2;
        const int nwg = G * k.nbn;
        if (c > 1 && nwg % 8 == 0 && (nwg / 8) % c == 0 && 8 % (k.nbn / c) == 0) k.cpx = c;
    }
#endif
    ProfScope ps(ctx, NF == 16 ? "conv_wino_f32<32t x128,F(2x2,3x3)>" : RES ? "conv_wino_f32<32t x64,F(2x2,3x3),res>" : "conv_wino_f32<32t x64,F(2x2,3x3)>",
                 flops * k.groups, bytes * k.groups);
    hipLaunchKernelGGL((conv_wino_f32_kernel<NF, RES, WIDE>), dim3(G * k.nbn * k.groups), dim3(256), smem, ctx.stream, k);
    return (int)hipGetLastError();
}

static int wino_launch_one(const ConvArgs& a, const LaunchCtx& ctx, const ConvArgs* b = nullptr) {
    WArgs k;
    k.x = a.x; k.u = a.w; k.scale = a.scale; k.shift = a.shift; k.out = a.out; k.res = a.res;
    k.groups = b ? 2 : 1;
    k.g1.x = b ? b->x : nullptr; k.g1.u = b ? b->w : nullptr; k.g1.scale = b ? b->scale : nullptr; k.g1.shift = b ? b->shift : nullptr;
    k.g1.res = b ? b->res : nullptr; k.g1.out = b ? b->out : nullptr;
    k.H = a.H; k.W = a.W; k.ldx = a.ldx; k.Cout = a.Cout; k.ldo = a.ldo;
    k.TH = (a.H + 1) / 2; k.TW = (a.W + 1) / 2; k.THW = k.TH * k.TW;
    k.Mt = a.B * k.THW;
    k.nbn = 0;
    k.nstage = a.Cin / 16;
    k.relu = a.relu;
    wino_magic((unsigned)k.THW, &k.mg_thw, &k.sh_thw);
    wino_magic((unsigned)k.TW, &k.mg_tw, &k.sh_tw);
    k.x_bytes = (unsigned)((size_t)a.B * a.H * a.W * a.ldx * 4);
    k.u_bytes = (unsigned)((size_t)16 * a.Cin * a.Cout * 4);
    k.out_bytes = (unsigned)((size_t)a.B * a.H * a.W * a.ldo * 4);
#ifdef WINO_PROF
    k.tprof = g_wino_tprof;
#endif
    const double M = (double)a.B * a.H * a.W;
    const double flops = 2.0 * M * a.Cout * 9.0 * a.Cin;   // algorithmic (direct-convolution) flops
    const double bytes = 4.0 * (M * a.Cin + M * a.Cout * (a.res ? 2.0 : 1.0) + 9.0 * a.Cin * a.Cout);
    if (wino_wide_cin(a.Cin)) {   // the layout pack_wino_weights() chose for this Cin
        if (a.res) return wino_launch_variant<8, true, true>(k, a.Cout, ctx, flops, bytes);
        return wino_pick(a, k.groups) == 16 ? wino_launch_variant<16, false, true>(k, a.Cout, ctx, flops, bytes)
                                            : wino_launch_variant<8, false, true>(k, a.Cout, ctx, flops, bytes);
    }
    if (a.res) return wino_launch_variant<8, true>(k, a.Cout, ctx, flops, bytes);
    return wino_pick(a, k.groups) == 16 ? wino_launch_variant<16>(k, a.Cout, ctx, flops, bytes)
                              : wino_launch_variant<8>(k, a.Cout, ctx, flops, bytes);
}

int launch_conv_wino(const ConvArgs& a, const LaunchCtx& ctx, const ConvArgs* b) {
    // 8-byte patch loads; 16-byte ones (pixel rows 16-byte aligned too) in the WIDE layout
    const uintptr_t amask = wino_wide_cin(a.Cin) ? 15 : 7;
    if (!conv_wino_supported(a) || (reinterpret_cast<uintptr_t>(a.x) & amask) || (wino_wide_cin(a.Cin) && a.ldx % 4)) return (int)hipErrorInvalidValue;
    if (b && (!conv_wino_supported(*b) || (reinterpret_cast<uintptr_t>(b->x) & amask) || b->B != a.B || b->H != a.H || b->W != a.W ||
              b->Cin != a.Cin || b->ldx != a.ldx || b->Cout != a.Cout || b->ldo != a.ldo || b->relu != a.relu ||
              (b->res != nullptr) != (a.res != nullptr) || b->wino_variant != a.wino_variant))
        return (int)hipErrorInvalidValue;
    const size_t in_bytes = (size_t)a.H * a.W * a.ldx * 4, o_bytes = (size_t)a.H * a.W * a.ldo * 4;
    const size_t img_bytes = in_bytes > o_bytes ? in_bytes : o_bytes;   // both sides use 32-bit buffer offsets
    const size_t limit = (size_t)1 << 30;   // see a_row / a_col and co_b in the kernel
    if (img_bytes >= limit || (size_t)16 * a.Cin * a.Cout * 4 >= ((size_t)1 << 31)) return (int)hipErrorInvalidValue;
    const int max_b = (int)((limit - 1) / img_bytes);
    if (b) return a.B <= max_b ? wino_launch_one(a, ctx, b) : (int)hipErrorInvalidValue;
    for (int b0 = 0; b0 < a.B; b0 += max_b) {
        ConvArgs s = a;
        s.B = (a.B - b0 < max_b) ? a.B - b0 : max_b;
        s.x = a.x + (size_t)b0 * a.H * a.W * a.ldx;
        s.out = a.out + (size_t)b0 * a.OH * a.OW * a.ldo;
        if (a.res) s.res = a.res + (size_t)b0 * a.OH * a.OW * a.ldo;
        const int rc = wino_launch_one(s, ctx);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace specmi
