// conv_wsplit.hip - the small-M unit of the latency plan's canonical k-sum tree (round 5).
//
// The sliced 64x64 kernel (conv_igemm.hip) gives a workgroup a 64x64 output tile and one leaf / group / the whole K of the tree;
// at the reference's operating point (spec/tester.py:109-151: batch = #detections of a frame, scripts/camcalib_demo.py:95-102:
// batch 1) a layer3 / layer4 convolution offers 16-100 such tiles per network to 256 CUs (tile quantization: 200 tiles of the
// trunk pair at batch 8), and filling the chip by slicing K across workgroups costs 16 KB of slab per (tile, slice) written to
// memory and read back by ONE last arriver (8-16 slabs = 128-256 KB at batch 1).
//
// Here a workgroup owns a 32x32 output tile - four times as many tiles - and its four waves run four LEAVES of the same tree
// side by side (the K split stays inside the CU):
//   * a wave = one leaf = one chain of v_mfma_f32_32x32x2_f32 from +0 over the leaf's L chunks, in exactly the k order of the
//     64x64 kernel (chunk = 32 k's of one filter tap; sub-chunk q, lane half h, step s -> k = 8q + 4h + s).  Both operands
//     come straight from L2 / memory in MFMA fragment layout - A: lane (row, h) reads the 16 bytes k = 8q + 4h .. + 3 of its
//     pixel, B: quad 2q + h of column n of the packed [K/4][Npad][4] weights - with 16-byte buffer loads, two chunks ahead in two
//     register sets: no LDS staging, no barrier inside the K loop, the four waves never wait for each other;
//   * the G leaves of a group meet in LDS (4 KB each, accumulator order) and are folded left to right from +0 by all 256
//     threads (four consecutive columns of one row each), the groups fold into the result in order - the canonical
//     association of conv_igemm.hip's header, so every output BIT equals the 64x64 kernel's, whatever unit ran;
//   * unit `all` (gridDim.y = 1): the workgroup walks all groups, no slab at all; unit `group` (gridDim.y = leaves / G): one
//     group per workgroup, a 4 KB slab per group (write-through sc1 stores + ticket, the last arriver folds the groups and
//     runs the epilogue) - 16 KB instead of 256 KB for layer4's 3x3 layers at batch 1;
//   * BatchNorm scale / shift, residual and ReLU in the epilogue with 16-byte row-contiguous stores; im2col padding and the
//     M tail through the buffer range check; second A source (folded downsample branch) and grouped launches (blockIdx.z =
//     network) as in the 64x64 kernel.
// Operand traffic per MFMA is twice the 64x64 kernel's (no sharing between waves): fine while M is small (L2-resident
// activations, every weight byte still leaves HBM once per tile row); the throughput plan never uses this kernel.
#include "conv_igemm_tile.h"

namespace specmi {

// ALDS = false: A fragments straight from L2 (lane (row, h) reads its own 16 bytes: 32 rows x 32 bytes per instruction - 32 cache
//   lines touched for 1 KB; fine while a CU holds one workgroup, the batch 1-2 regime);
// ALDS = true:  A as whole 128-byte row segments (8 lanes per row, 8 rows per instruction - 8 full lines) into registers two chunks
//   ahead, transposed through a PRIVATE per-wave LDS stage (2 x 32 x 36 floats, the 64x64 kernel's layout) - still no workgroup
//   barrier in the K loop (a wave's LDS operations execute in order); what 3-4 co-resident workgroups per CU need (batch >= 4:
//   the scattered form saturates the vector L1's tag rate, measured at half the matrix pipe).
template <bool IS1X1, bool DUAL, bool ALDS>
__global__ void __launch_bounds__(256, ALDS ? 3 : 4) conv_wsplit_f32_kernel(const KArgs p) {
    static_assert(!DUAL || IS1X1, "the second A source exists for 1x1 layers only");
    __shared__ __attribute__((aligned(16))) float lds[4 * 1024];   // one 32x32 leaf tile per wave, [reg][lane]
    __shared__ __attribute__((aligned(16))) float lds_a[ALDS ? 4 * 2 * 32 * 36 : 4];   // ALDS: [wave][stage][row][32 k + 4 pad]
    __shared__ int flag;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform BY CONSTRUCTION: tell the compiler (scalar chunk offsets)
    const int l31 = lane & 31, hh = lane >> 5;
    const bool grp = blockIdx.z != 0;
    const float* const px = grp ? p.g1.x : p.x;
    const float* const pw = grp ? p.g1.w : p.w;
    const float* const pscale = grp ? p.g1.scale : p.scale;
    const float* const pshift = grp ? p.g1.shift : p.shift;
    const float* const pres = grp ? p.g1.res : p.res;
    const float* const px2 = grp ? p.g1.x2 : p.x2;
    float* const pout = grp ? p.g1.out : p.out;

    // ---- tile: XCD-aware order as in the 64x64 kernel (tiles sharing an A row panel on one XCD) -------------------------
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    const int Lt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = Lt / p.nbn, tile_n = Lt - tile_m * p.nbn;   // nbn = Npad / 32
    const int m0 = tile_m * 32, n0 = tile_n * 32;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pw), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? px2 : px), 0, DUAL ? p.x2_bytes : p.x_bytes, 0x00020000);

    // ---- this lane's A rows: byte offset of (tap (0,0) pixel, this lane's 16-byte quad).  Direct form: ONE row (l31), quad = the
    // lane half's 4 k's of sub-chunk 0; ALDS: FOUR rows (lane / 8 + 8 i), quad lane % 8 of the chunk's 32 k's ----------------------
    constexpr int NA = ALDS ? 4 : 1;
    unsigned a_voff[NA], a_voff2[NA], a_mask[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + (ALDS ? (lane >> 3) + 8 * i : l31);
        const int qb = (ALDS ? (lane & 7) : hh) * 16;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        a_voff2[i] = 0; a_mask[i] = 0;
        if (DUAL) {
            if (p.stride2 == 1) {
                a_voff2[i] = ok ? (unsigned)(mm * p.ldx2 * 4 + qb) : kOutOfRange;
            } else {
                const int b2 = p.OHW == 1 ? mm : (int)(__umulhi((unsigned)mm, p.mg_ohw) >> p.sh_ohw);
                const int rem2 = mm - b2 * p.OHW;
                const int oy2 = p.OW == 1 ? rem2 : (int)(__umulhi((unsigned)rem2, p.mg_ow) >> p.sh_ow);
                const int ox2 = rem2 - oy2 * p.OW;
                const int pix2 = (b2 * p.H2 + oy2 * p.stride2) * p.W2 + ox2 * p.stride2;
                a_voff2[i] = ok ? (unsigned)(pix2 * p.ldx2 * 4 + qb) : kOutOfRange;
            }
        }
        if (IS1X1 && p.stride == 1) {
            a_voff[i] = ok ? (unsigned)(mm * p.ldx * 4 + qb) : kOutOfRange;
        } else {
            const int b = p.OHW == 1 ? mm : (int)(__umulhi((unsigned)mm, p.mg_ohw) >> p.sh_ohw);
            const int rem = mm - b * p.OHW;
            const int oy = p.OW == 1 ? rem : (int)(__umulhi((unsigned)rem, p.mg_ow) >> p.sh_ow);
            const int ox = rem - oy * p.OW;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            const int pix0 = (b * p.H + iy0) * p.W + ix0;
            const unsigned off = (unsigned)(pix0 * p.ldx * 4 + qb);   // wraps for padded rows; only used on valid taps
            if (IS1X1) {
                a_voff[i] = ok ? off : kOutOfRange;
            } else {
                a_voff[i] = off;
                unsigned colbits = 0, mk = 0;
                for (int kx = 0; kx < p.KW; ++kx) colbits |= ((unsigned)(ix0 + kx) < (unsigned)p.W ? 1u : 0u) << kx;
                for (int ky = 0; ky < p.KH; ++ky)
                    if ((unsigned)(iy0 + ky) < (unsigned)p.H) mk |= colbits << (ky * p.KW);
                a_mask[i] = ok ? mk : 0u;
            }
        }
    }
    const unsigned b_voff = (unsigned)((hh * p.Npad + n0 + l31) * 16);

    // ---- this wave's K stream: the leaves (g0 + gi) * G + wave, gi = 0 .. ngroups_wg - 1, as ONE sequence of virtual chunks
    // v = gi * L + cl, so that the operands of the next leaf are already in flight when a leaf ends ------------------------------
    const int L = p.sk_leaf, G = p.sk_G;
    const int S = gridDim.y;
    const int ngroups_wg = p.sk_unit / G;          // groups this workgroup walks (leaves / G when S == 1, else 1)
    const int g0 = (int)blockIdx.y * ngroups_wg;
    const int total = ngroups_wg * L;
    const bool active = wave < G;                  // (a group of 2 or 3 leaves leaves waves idle)
    f32x4 fa[2][4], fb[2][4];   // fa: direct form: A fragments [set][q]; ALDS: the raw row quads [set][row i] on their way to LDS
    struct ChunkAddr { bool oob, second; unsigned s_a, tap_bytes, s_b; int tap; };
    auto chunk_addr = [&](int v) {
        ChunkAddr a;
        a.oob = v >= total;
        const int gi = v / L;                      // (scalar)
        const int c = ((g0 + gi) * G + wave) * L + (v - gi * L);
        a.tap = IS1X1 ? 0 : c / p.cpc;
        const int c0 = IS1X1 ? c : c - a.tap * p.cpc;
        a.tap_bytes = 0;
        if (!IS1X1) {
            const int ky = a.tap / p.KW, kx = a.tap - ky * p.KW;
            a.tap_bytes = (unsigned)((ky * p.W + kx) * p.ldx * 4);
        }
        a.second = DUAL && c >= p.cpc1;
        a.s_a = a.oob ? 0u : (unsigned)((a.second ? c0 - p.cpc1 : c0) * 128);
        a.s_b = a.oob ? 0u : (unsigned)(c * 8 * p.Npad * 16);
        return a;
    };
    auto load_a_one = [&](const ChunkAddr& ca, int i, unsigned extra) {
        unsigned voff = a_voff[i];
        if (!IS1X1) voff = ((a_mask[i] >> (ca.tap & 31)) & 1u) ? voff + ca.tap_bytes : kOutOfRange;
        if (DUAL) voff = ca.second ? a_voff2[i] : voff;
        if (ca.oob) voff = kOutOfRange;
        if (DUAL) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ca.second ? x2rs : xrs, voff, ca.s_a + extra, 0));
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, ca.s_a + extra, 0));
    };
    auto load_b_one = [&](const ChunkAddr& ca, int q) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            wrs, ca.oob ? kOutOfRange : b_voff, ca.oob ? 0u : ca.s_b + (unsigned)(2 * q * p.Npad * 16), 0));
    };
    // direct form: A and B fragments of sub-chunk q of virtual chunk v into set SL
    auto load_chunk_q = [&](int v, int q, auto slot) {
        constexpr int SL = decltype(slot)::value;
        const ChunkAddr ca = chunk_addr(v);
        if (!ALDS) fa[SL][q] = load_a_one(ca, 0, (unsigned)(q * 32));
        fb[SL][q] = load_b_one(ca, q);
    };
    // ALDS: the four row quads of virtual chunk v into set SL
    auto load_rows = [&](int v, auto slot) {
        constexpr int SL = decltype(slot)::value;
        const ChunkAddr ca = chunk_addr(v);
#pragma unroll
        for (int i = 0; i < NA; ++i) fa[SL][i] = load_a_one(ca, i, 0u);
    };
    float* const stage = lds_a + (ALDS ? wave * (2 * 32 * 36) : 0);
    auto stage_rows = [&](auto slot, int which) {   // set SL -> LDS stage `which` (row-major, 36-float rows)
        constexpr int SL = decltype(slot)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            *reinterpret_cast<f32x4*>(&stage[which * (32 * 36) + ((lane >> 3) + 8 * i) * 36 + (lane & 7) * 4]) = fa[SL][i];
    };
    auto wave_lds_fence = [&]() {   // a wave's LDS operations execute in order: only the COMPILER must not reorder across this point
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto chunk = [&](int v, auto par) {
        constexpr int P = decltype(par)::value;
        if constexpr (!ALDS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[P][q][s], fb[P][q][s], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                load_chunk_q(v + 2, q, par);   // this set's next use is two chunks from now
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // stage P holds chunk v (written during chunk v - 1 / the prologue); set P^1 holds chunk v + 1's rows
            const float* const st = stage + P * (32 * 36) + l31 * 36 + hh * 4;
            f32x4 fq[2];
            fq[0] = *reinterpret_cast<const f32x4*>(st);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[q & 1][s], fb[P][q][s], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                fb[P][q] = load_b_one(chunk_addr(v + 2), q);
                if (q == 0) load_rows(v + 2, par);      // (set P's rows went to LDS during chunk v - 1)
                if (q < 3) fq[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(st + (q + 1) * 8);
                if (q == 3) {
                    std::integral_constant<int, P ^ 1> other;
                    stage_rows(other, P ^ 1);           // chunk v + 1 -> the other stage
                    wave_lds_fence();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    const std::integral_constant<int, 0> even{};
    const std::integral_constant<int, 1> odd{};

    // ---- fold coordinates: thread -> accumulator register r, lanes 4 * (tid & 15) .. + 3 = one row, four consecutive columns ----
    const int fr = tid >> 4, fl = (tid & 15) * 4;
    const int frow = (fr & 3) + 8 * (fr >> 2) + 4 * (fl >> 5), fcol = fl & 31;
    f32x4 result = {0.f, 0.f, 0.f, 0.f};

    int lc = 0;
    auto leaf_end = [&]() {
        if (++lc < L) return;
        lc = 0;
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { lds[wave * 1024 + r * 64 + lane] = acc[r]; acc[r] = 0.f; }
        }
        __syncthreads();
        // the group: its leaves left to right from +0; then into the result (groups left to right from +0)
        f32x4 tg = {0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < G; ++w) {
            const f32x4 lv = *reinterpret_cast<const f32x4*>(&lds[w * 1024 + fr * 64 + fl]);
#pragma unroll
            for (int e = 0; e < 4; ++e) tg[e] += lv[e];
        }
        if (S > 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) result[e] = tg[e];   // the slab holds the group itself
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) result[e] += tg[e];
        }
        __syncthreads();   // the leaf tiles are free for the next round
    };
    // Waves without a leaf (groups of 2 or 3) only keep the barriers company.  Two separate loops, not `if (active)` around the
    // chunks: hipcc's wait-count pass is path-insensitive - with the chunk under an `if` it assumes the other register set's
    // eight loads are not in flight and waits for vmcnt(6) instead of vmcnt(14) at the top of every chunk (the two-chunk
    // distance collapses to one; seen in the disassembly).  Whole (even, odd) pairs plus a peeled tail for the same reason
    // (conv_igemm_body.inc).
    int v = 0;
    if (active) {
        if (ALDS) load_rows(0, even);
#pragma unroll
        for (int q = 0; q < 4; ++q) load_chunk_q(0, q, even);
        if (ALDS) load_rows(1, odd);
#pragma unroll
        for (int q = 0; q < 4; ++q) load_chunk_q(1, q, odd);
        if (ALDS) {
            stage_rows(even, 0);
            wave_lds_fence();
        }
        for (; v + 2 <= total; v += 2) {
            chunk(v, even);
            leaf_end();
            chunk(v + 1, odd);
            leaf_end();
        }
        if (v < total) {
            chunk(v, even);
            leaf_end();
        }
    } else {
        for (; v < total; ++v) leaf_end();
    }

    if (S > 1) {
        // group slab -> workspace (16 bytes per thread, consecutive threads consecutive: coalesced) with write-through stores,
        // drain, ticket; the last arriver folds the S groups in order (conv_igemm_body.inc: same hand-off)
        const size_t tile = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
        float* const tile_ws = p.sk_ws + tile * S * (size_t)1024;
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(tile_ws, 0, (unsigned)S * 4096u, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, result), srs, (unsigned)(tid * 16), (unsigned)blockIdx.y * 4096u, /*sc1*/ 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(p.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == (unsigned)S - 1;
            if (last) __hip_atomic_store(p.sk_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag = last;
        }
        __syncthreads();
        if (!flag) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) result[e] = 0.f;
        int z = 0;
        for (; z + 4 <= S; z += 4) {
            f32x4 sv[4];
#pragma unroll
            for (int zz = 0; zz < 4; ++zz)
                sv[zz] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)(tid * 16), (unsigned)(z + zz) * 4096u, 16));
#pragma unroll
            for (int zz = 0; zz < 4; ++zz)
#pragma unroll
                for (int e = 0; e < 4; ++e) result[e] += sv[zz][e];
        }
        for (; z < S; ++z) {
            const f32x4 sv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)(tid * 16), (unsigned)z * 4096u, 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) result[e] += sv[e];
        }
    }

    // ---- epilogue: BN scale / shift, residual, ReLU; one row x four columns per thread ------------------------------------------
    const int m = m0 + frow, n = n0 + fcol;
    if (m >= p.M || n >= p.Cout) return;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(pscale + n);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(pshift + n);
    const size_t o = (size_t)m * p.ldo + n;
    f32x4 o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] = fmaf(result[e], sc[e], sh[e]);
    if (n + 3 < p.Cout && p.vec_ok) {
        if (pres) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(pres + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] += rr[e];
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = fmaxf(o4[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(pout + o) = o4;
    } else {
        for (int e = 0; e < 4; ++e) {
            if (n + e >= p.Cout) break;
            float t = o4[e];
            if (pres) t += pres[o + e];
            if (p.relu) t = fmaxf(t, 0.f);
            pout[o + e] = t;
        }
    }
}

// workgroups of the launch: 32x32 tiles x slabs per tile x networks
int conv_wsplit_tiles(const ConvArgs& a, int groups) { return ((a.B * a.OH * a.OW + 31) / 32) * ((a.Cout + 31) / 32) * groups; }
size_t conv_wsplit_ws_floats(const ConvArgs& a, int S, int groups) { return S > 1 ? (size_t)conv_wsplit_tiles(a, groups) * S * 1024 : 0; }

// the canonical tree must have 2..4 leaves per group (one per wave); unit = leaves (no slabs) or G (one group per workgroup)
bool conv_wsplit_supported(const ConvArgs& a, const SkPlan& pl) {
    return pl.leaves >= 2 && pl.G >= 2 && pl.G <= 4 && pl.leaves % pl.G == 0 && (pl.unit == pl.leaves || pl.unit == pl.G) && !a.force_variant;
}

int launch_conv_wsplit(const ConvArgs& a, const SkPlan& pl, const SkWs& sk, const LaunchCtx& ctx, const ConvArgs* b, bool alds) {
    if (!conv_wsplit_supported(a, pl)) return (int)hipErrorInvalidValue;
    if (int rc = conv_igemm_sk_check(a, pl, b)) return rc;
    const int groups = b ? 2 : 1;
    const int S = pl.leaves / pl.unit;
    if (S > 1 && (!sk.ws || !sk.cnt || conv_wsplit_ws_floats(a, S, groups) > sk.floats || conv_wsplit_tiles(a, groups) > sk.ncnt))
        return (int)hipErrorInvalidValue;
    KArgs k;
    conv_igemm_make_sk_kargs(a, pl, b, k);
    k.nbn = (a.Cout + 31) / 32;   // 32-wide tiles skip the all-padding half of a 64-padded panel
    k.sk_ws = sk.ws; k.sk_cnt = sk.cnt;
    const int M = k.M;
    const int grid = ((M + 31) / 32) * k.nbn;
    const double Kd = (double)a.KH * a.KW * a.Cin + (a.x2 ? a.Cin2 : 0);
    const double flops = 2.0 * (double)M * a.Cout * Kd;
    const double bytes = 4.0 * ((double)a.B * a.H * a.W * a.Cin + (a.x2 ? (double)M * a.Cin2 : 0.0) + (double)M * a.Cout * (a.res ? 2.0 : 1.0) + Kd * a.Cout);
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.pad == 0);
    const char* name = alds ? (a.x2 ? "conv_wsplit_f32<32x32,4 leaves,ldsA,2src>" : "conv_wsplit_f32<32x32,4 leaves,ldsA>")
                            : (a.x2 ? "conv_wsplit_f32<32x32,4 leaves,2src>" : "conv_wsplit_f32<32x32,4 leaves>");
    ProfScope ps(ctx, name, flops * groups, bytes * groups);
    const dim3 g(grid, S, groups), blk(256);
    if (alds) {
        if (a.x2) hipLaunchKernelGGL((conv_wsplit_f32_kernel<true, true, true>), g, blk, 0, ctx.stream, k);
        else if (is1x1) hipLaunchKernelGGL((conv_wsplit_f32_kernel<true, false, true>), g, blk, 0, ctx.stream, k);
        else hipLaunchKernelGGL((conv_wsplit_f32_kernel<false, false, true>), g, blk, 0, ctx.stream, k);
    } else {
        if (a.x2) hipLaunchKernelGGL((conv_wsplit_f32_kernel<true, true, false>), g, blk, 0, ctx.stream, k);
        else if (is1x1) hipLaunchKernelGGL((conv_wsplit_f32_kernel<true, false, false>), g, blk, 0, ctx.stream, k);
        else hipLaunchKernelGGL((conv_wsplit_f32_kernel<false, false, false>), g, blk, 0, ctx.stream, k);
    }
    return (int)hipGetLastError();
}

}  // namespace specmi
