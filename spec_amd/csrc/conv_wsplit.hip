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
// Operand traffic per MFMA is 8 KB per wave-chunk against the 64x64 kernel's 6 (no sharing between waves), and an A fragment
// load touches 32 cache lines: fine while a CU holds one or two workgroups (batch 1-3; layer4 up to batch 4), slower than the
// 64x64 kernel beyond (measured per layer and batch: profiles/r05_c_wsplit_layers.txt; a variant that stages A as whole rows
// through a private per-wave LDS buffer was built and measured too - never ahead of the 64x64 kernel at batch >= 4, dropped).
// The K loop is ISSUE-sensitive (four waves per SIMD interleave their MFMA chains): the chunk addressing is incremental
// scalar state advanced once per chunk (no division in the steady state), every address is formed once per chunk, and the
// last two chunks are peeled so that no "past the end" select remains in the loop - 100 scalar instructions more per chunk
// pair cost 8-17 % (same profile).
#include "conv_igemm_tile.h"

namespace specmi {

template <bool IS1X1, bool DUAL>
__global__ void __launch_bounds__(256, 4) conv_wsplit_f32_kernel(const KArgs p) {
    static_assert(!DUAL || IS1X1, "the second A source exists for 1x1 layers only");
    __shared__ __attribute__((aligned(16))) float lds[4 * 1024];   // one 32x32 leaf tile per wave, [reg][lane]
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

    // ---- this lane's A row: byte offset of (tap (0,0) pixel, channel 4 * hh) -----------------------------------------------
    unsigned a_voff, a_voff2 = 0, a_mask = 0;
    {
#ifdef WS_ABLATE_A      // timing experiment (wrong results): every tile reads the SAME 32 activation rows
        const int m = l31;
#else
        const int m = m0 + l31;
#endif
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        if (DUAL) {
            if (p.stride2 == 1) {
                a_voff2 = ok ? (unsigned)(mm * p.ldx2 * 4 + hh * 16) : kOutOfRange;
            } else {
                const int b2 = p.OHW == 1 ? mm : (int)(__umulhi((unsigned)mm, p.mg_ohw) >> p.sh_ohw);
                const int rem2 = mm - b2 * p.OHW;
                const int oy2 = p.OW == 1 ? rem2 : (int)(__umulhi((unsigned)rem2, p.mg_ow) >> p.sh_ow);
                const int ox2 = rem2 - oy2 * p.OW;
                const int pix2 = (b2 * p.H2 + oy2 * p.stride2) * p.W2 + ox2 * p.stride2;
                a_voff2 = ok ? (unsigned)(pix2 * p.ldx2 * 4 + hh * 16) : kOutOfRange;
            }
        }
        if (IS1X1 && p.stride == 1) {
            a_voff = ok ? (unsigned)(mm * p.ldx * 4 + hh * 16) : kOutOfRange;
        } else {
            const int b = p.OHW == 1 ? mm : (int)(__umulhi((unsigned)mm, p.mg_ohw) >> p.sh_ohw);
            const int rem = mm - b * p.OHW;
            const int oy = p.OW == 1 ? rem : (int)(__umulhi((unsigned)rem, p.mg_ow) >> p.sh_ow);
            const int ox = rem - oy * p.OW;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            const int pix0 = (b * p.H + iy0) * p.W + ix0;
            const unsigned off = (unsigned)(pix0 * p.ldx * 4 + hh * 16);   // wraps for padded rows; only used on valid taps
            if (IS1X1) {
                a_voff = ok ? off : kOutOfRange;
            } else {
                a_voff = off;
                unsigned colbits = 0, mk = 0;
                for (int kx = 0; kx < p.KW; ++kx) colbits |= ((unsigned)(ix0 + kx) < (unsigned)p.W ? 1u : 0u) << kx;
                for (int ky = 0; ky < p.KH; ++ky)
                    if ((unsigned)(iy0 + ky) < (unsigned)p.H) mk |= colbits << (ky * p.KW);
                a_mask = ok ? mk : 0u;
            }
        }
    }
#ifdef WS_ABLATE_B      // timing experiment (wrong results): every tile reads the SAME weight columns -> B traffic becomes L1 / L2 hits
    const unsigned b_voff = (unsigned)((hh * p.Npad + l31) * 16);
#else
    const unsigned b_voff = (unsigned)((hh * p.Npad + n0 + l31) * 16);
#endif

    // ---- this wave's K stream: the leaves (g0 + gi) * G + wave, gi = 0 .. ngroups_wg - 1, as ONE sequence of virtual chunks
    // v = gi * L + cl, so that the operands of the next leaf are already in flight when a leaf ends ------------------------------
    const int L = p.sk_leaf, G = p.sk_G;
    const int S = gridDim.y;
    const int ngroups_wg = p.sk_unit / G;          // groups this workgroup walks (leaves / G when S == 1, else 1)
    const int g0 = (int)blockIdx.y * ngroups_wg;
    const int total = ngroups_wg * L;
    const bool active = wave < G;                  // (a group of 2 or 3 leaves leaves waves idle)
    f32x4 fa[2][4], fb[2][4];
    // ---- prefetch cursor: the chunk two ahead of the one being consumed, as scalar state advanced once per chunk ----------------
    const unsigned npad32 = (unsigned)p.Npad * 32u;   // bytes between the quad pairs of consecutive sub-chunks of B
    int pcl = 0, pgi = 0;         // chunk within the leaf, group within this workgroup's groups
    int pc = 0, pc0 = 0;          // absolute chunk; chunk within its filter tap (KxK)
    int pky = 0, pkx = 0;         // filter tap of pc (KxK)
    auto cursor_leaf = [&]() {    // first chunk of leaf (g0 + pgi) * G + wave: the only place that divides (once per leaf)
        pc = ((g0 + pgi) * G + wave) * L;
        if (!IS1X1) {
            const int tap = pc / p.cpc;
            pc0 = pc - tap * p.cpc;
            pky = tap / p.KW;
            pkx = tap - pky * p.KW;
        }
    };
    auto cursor_next = [&]() {
        if (++pcl == L) {
            pcl = 0; ++pgi;
            cursor_leaf();
            return;
        }
        ++pc;
        if (!IS1X1) {
            if (++pc0 == p.cpc) {
                pc0 = 0;
                if (++pkx == p.KW) { pkx = 0; ++pky; }
            }
        }
    };
    // addresses of the cursor's chunk, formed once per chunk: scalar byte offsets of A / B and this lane's A row offset
    unsigned cs_a = 0, cs_b = 0, cv_a = 0;
    bool c_second = false;
    auto cursor_addr = [&]() {
        c_second = DUAL && pc >= p.cpc1;
        cs_a = (unsigned)((IS1X1 ? (c_second ? pc - p.cpc1 : pc) : pc0) * 128);
        cs_b = (unsigned)pc * 8u * (unsigned)p.Npad * 16u;
        if (IS1X1) {
            cv_a = DUAL ? (c_second ? a_voff2 : a_voff) : a_voff;
        } else {
            const int tap = pky * p.KW + pkx;
            const unsigned tap_bytes = (unsigned)((pky * p.W + pkx) * p.ldx * 4);
            cv_a = ((a_mask >> tap) & 1u) ? a_voff + tap_bytes : kOutOfRange;
        }
    };
    auto load_q = [&](int q, auto slot) {   // sub-chunk q of the cursor's chunk into set SL
        constexpr int SL = decltype(slot)::value;
        if (DUAL) fa[SL][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_second ? x2rs : xrs, cv_a, cs_a + (unsigned)(q * 32), 0));
        else      fa[SL][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, cv_a, cs_a + (unsigned)(q * 32), 0));
        fb[SL][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff, cs_b + (unsigned)q * npad32, 0));
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // one chunk out of set P; PF: the chunk two ahead is fetched into the same set, sub-chunk by sub-chunk as the set is consumed
    auto chunk = [&](auto par, auto prefetch) {
        constexpr int P = decltype(par)::value;
        constexpr bool PF = decltype(prefetch)::value;
        if (PF) cursor_addr();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[P][q][s], fb[P][q][s], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (PF) load_q(q, par);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PF) cursor_next();
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
    // distance collapses to one; seen in the disassembly).  Whole (even, odd) pairs for the same reason (conv_igemm_body.inc);
    // the last two or three chunks are peeled: they fetch nothing (or one chunk), so the loop body has no "past the end" test.
    const std::true_type pf{};
    const std::false_type nopf{};
    if (active) {
        cursor_leaf();
        cursor_addr();
#pragma unroll
        for (int q = 0; q < 4; ++q) load_q(q, even);
        cursor_next();
        cursor_addr();
#pragma unroll
        for (int q = 0; q < 4; ++q) load_q(q, odd);
        cursor_next();
        int v = 0;
        for (; v + 4 <= total; v += 2) {     // chunks v, v + 1: their prefetches v + 2, v + 3 exist
            chunk(even, pf);
            leaf_end();
            chunk(odd, pf);
            leaf_end();
        }
        if (total - v == 3) {
            chunk(even, pf);
            leaf_end();
            chunk(odd, nopf);
            leaf_end();
            chunk(even, nopf);
            leaf_end();
        } else {                              // 2 left (total >= 4: a leaf has at least four chunks)
            chunk(even, nopf);
            leaf_end();
            chunk(odd, nopf);
            leaf_end();
        }
    } else {
        for (int v = 0; v < total; ++v) leaf_end();
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
    const int nch = a.KH * a.KW * (a.Cin / 32) + (a.x2 ? a.Cin2 / 32 : 0);
    return pl.leaves >= 2 && pl.G >= 2 && pl.G <= 4 && pl.leaves % pl.G == 0 && (pl.unit == pl.leaves || pl.unit == pl.G) && !a.force_variant &&
           a.KH * a.KW <= 32 &&                            // (the padding mask of a pixel holds one bit per filter tap)
           nch % pl.leaves == 0 && nch / pl.leaves >= 2;   // (the K loop keeps two chunks in flight)
}

int launch_conv_wsplit(const ConvArgs& a, const SkPlan& pl, const SkWs& sk, const LaunchCtx& ctx, const ConvArgs* b) {
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
    const char* name = a.x2 ? "conv_wsplit_f32<32x32,4 leaves,2src>" : "conv_wsplit_f32<32x32,4 leaves>";
    ProfScope ps(ctx, name, flops * groups, bytes * groups);
    const dim3 g(grid, S, groups), blk(256);
    if (a.x2) hipLaunchKernelGGL((conv_wsplit_f32_kernel<true, true>), g, blk, 0, ctx.stream, k);
    else if (is1x1) hipLaunchKernelGGL((conv_wsplit_f32_kernel<true, false>), g, blk, 0, ctx.stream, k);
    else hipLaunchKernelGGL((conv_wsplit_f32_kernel<false, false>), g, blk, 0, ctx.stream, k);
    return (int)hipGetLastError();
}

}  // namespace specmi
