// conv_persist.hip - a run of consecutive trunk convolutions as ONE launch (round 5).
//
// Why.  The reference runs the path at batch = #detections of one frame (spec/tester.py:109-151) and CamCalib at batch 1
// (scripts/camcalib_demo.py:95-102).  There a ResNet-50 layer is 5-20 us of work and every layer is its own graph node: kernel
// boundary, grid fill / drain, a prologue that cannot request a single byte before the previous layer has retired
// (profiles/r04_y_b1_timeline_rocprof.txt: 58 nodes, 0.66 ms, the smallest node 4.6 us whatever it does).  Here ONE grid of
// resident workgroups (two per CU for the trunk pair) walks the (tile, K slice, network) items of up to 64 consecutive layers:
//
//   * an item is exactly one workgroup of the per-layer launch (conv_igemm_tile.h: same body, same MFMA chain, same canonical
//     k-sum tree -> the same bits as the latency plan's launches); item i of layer l runs on workgroup (i + rot_l) mod grid with
//     rot_l alternating between 0 and grid / 2, so consecutive layers land on different halves of the grid: the workgroups that
//     idle during layer l already hold layer l + 1's first weight fragments in registers (and, option l2_prefetch, the rest of
//     their weight slice in their XCD's L2) when the layer's inputs appear;
//   * no grid barrier: the producing layer's completion is ONE counter per (layer, network) that the last arriver of each tile
//     bumps after its write-through output stores have drained; a consumer polls that word (relaxed sc1 load + s_sleep) before
//     its first activation load.  The two networks of the pair never wait for each other.  A chain of layers is a total order
//     per network, which also covers every write-after-read on the ping-pong activation buffers;
//   * everything one workgroup hands to another inside the launch - activations, residuals, split-K slabs - travels as 16-byte
//     sc1 (write-through) stores + s_waitcnt vmcnt(0) + relaxed agent-scope counter and is read back with sc1 loads
//     (MI355X_MICROARCH.md, "inter-workgroup visibility": per-XCD L2s are not coherent with each other, a CU's L1 is never
//     refreshed); weights and BN vectors are read-only for the whole launch and use the default policy;
//   * split-K slabs of layer l live in region (l & 1) of their network, tile counters are per layer: nothing of layer l + 1
//     can touch what a straggler of layer l still reads;
//   * every spin is bounded (spin_limit polls): a protocol error sets PCtl::err and the launch still ends;
//   * the last workgroup to leave zeroes the completion counters: the control block is clean for the next launch / replay.
//
// Residency: the grid (<= 512 workgroups of 256 threads, 128 VGPRs, 18.4 KB LDS) must be co-resident - two such launches may
// run side by side (1024 slots), which is what two handles on two streams need; specmi.h says so.
#include "conv_igemm_tile.h"

namespace specmi {

struct PLayer {
    KArgs k;
    float *ws0, *ws1;    // split-K slab region of each network for this layer
    unsigned* cnt;       // tile arrival counters of this layer: [z * nblk + bid]
    int nblk, nblk8;     // tiles per network; rounded up to a multiple of 8 (item index -> XCD keeps the tile order's XCD)
    int S, groups;       // slabs per tile; networks
    int body;            // 0: 1x1 (optionally two sources), 1: KH x KW
    int rot;             // workgroup of item 0
    int dep_target;      // tiles per network of the producing layer (the previous entry); 0: no wait
    int out_arg;         // 1: the output pointer is the launch argument out0 / out1 (a caller-owned feature buffer)
    int l2_prefetch;
    int pad_;
};
static_assert(sizeof(PLayer) % 8 == 0, "table entries hold pointers");

typedef const PLayer __attribute__((address_space(4))) CPLayer;   // read-only for the launch, uniform: scalar loads
typedef const KArgs __attribute__((address_space(4))) CKArgs;

__global__ void __launch_bounds__(256, 4) conv_persist_kernel(const PLayer* __restrict__ layers, int nl, PersistCtl* __restrict__ ctl,
                                                              float* out0, float* out1, unsigned spin_limit) {
    const int w = blockIdx.x, nwg = gridDim.x;
    for (int l = 0; l < nl; ++l) {
        CPLayer& L = *(CPLayer*)(uintptr_t)(layers + l);
        const int nblk = L.nblk, nblk8 = L.nblk8, S = L.S;
        const int nitems = nblk8 * S * L.groups;
        int i = w - L.rot;
        if (i < 0) i += nwg;
        int first_item = 1;
        for (; i < nitems; i += nwg) {
            const int r = i / nblk8, bid = i - r * nblk8;
            if (bid >= nblk) continue;
            const int z = r / S, y = r - z * S;
            TileCtx t;
            t.bid = bid; t.nblk = nblk;
            t.y = y; t.S = S;
            t.z = z;
            t.ws = z ? L.ws1 : L.ws0; t.cnt = L.cnt + z * nblk;
            t.tile = (unsigned)bid;
            const int dep_target = L.dep_target;
            t.dep = dep_target ? &ctl->done[(l - 1) * 2 + z] : nullptr;
            t.dep_target = (unsigned)dep_target;
            t.done = &ctl->done[l * 2 + z];
            t.err = &ctl->err;
            t.out = L.out_arg ? (z ? out1 : out0) : nullptr;
            t.spin_limit = spin_limit;
            t.l2_prefetch = L.l2_prefetch;
            t.walk_first = first_item;
            first_item = 0;
            if (L.body == 0) igemm_tile<64, 64, 2, 2, true, 32, true, true, true, true, CKArgs>(L.k, t);
            else             igemm_tile<64, 64, 2, 2, false, 32, false, true, true, true, CKArgs>(L.k, t);
        }
    }
    // leave the control block clean: the last workgroup out zeroes the completion counters (nobody polls any more)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = __hip_atomic_fetch_add(&ctl->exit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == (unsigned)nwg - 1) {
            for (int j = 0; j < 2 * nl; ++j) __hip_atomic_store(&ctl->done[j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->exit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

size_t persist_table_bytes(int nl) { return (size_t)nl * sizeof(PLayer); }

// Lay a run of layers out for one launch.  ws_floats_needed / cnt_needed: what sk must hold (call with sk = {} to size it).
int persist_fill_table(const PersistLayerHost* layers, int nl, const SkWs& sk, int nwg, int l2_prefetch, void* img,
                       size_t* ws_floats_needed, int* cnt_needed) {
    if (nl < 1 || nl > kPersistMaxLayers || nwg < 16 || nwg % 16 != 0) return (int)hipErrorInvalidValue;
    PLayer* tab = static_cast<PLayer*>(img);
    size_t region = 0;   // floats of one (parity, network) slab region
    int cnt_total = 0;
    for (int l = 0; l < nl; ++l) {
        const PersistLayerHost& h = layers[l];
        const ConvArgs& a = h.a;
        const ConvArgs* b = h.pair ? &h.b : nullptr;
        if (int rc = conv_igemm_sk_check(a, h.pl, b)) return rc;
        // the walker's epilogue moves whole 16-byte quads with buffer addressing: full 64-column panels, aligned rows, < 2 GiB
        const size_t obytes = (size_t)a.B * a.OH * a.OW * a.ldo * 4;
        auto misaligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
        if (a.Cout % 64 != 0 || a.Npad != a.Cout || a.ldo % 4 != 0 || obytes >= ((size_t)1 << 31) || misaligned(a.out) || (a.res && misaligned(a.res)) ||
            (b && (misaligned(b->out) || (b->res && misaligned(b->res)))))
            return (int)hipErrorInvalidValue;
        const int tiles = conv_igemm_sk_tiles(a, 1);
        const int S = h.pl.leaves / h.pl.unit;
        if (S > 1) {
            const size_t need = (size_t)tiles * S * 64 * 64;
            if (need > region) region = need;
        }
        cnt_total += tiles * (b ? 2 : 1);
    }
    region = (region + 1023) / 1024 * 1024;
    *ws_floats_needed = 4 * region;
    *cnt_needed = cnt_total;
    if (!img) return 0;
    if ((region && (!sk.ws || sk.floats < 4 * region)) || !sk.cnt || sk.ncnt < cnt_total) return (int)hipErrorInvalidValue;
    int cnt_off = 0;
    for (int l = 0; l < nl; ++l) {
        const PersistLayerHost& h = layers[l];
        const ConvArgs* b = h.pair ? &h.b : nullptr;
        PLayer& P = tab[l];
        memset(&P, 0, sizeof(P));
        conv_igemm_make_sk_kargs(h.a, h.pl, b, P.k);
        P.k.sk_ws = nullptr; P.k.sk_cnt = nullptr;
        const int tiles = conv_igemm_sk_tiles(h.a, 1);
        P.nblk = tiles;
        P.nblk8 = (tiles + 7) / 8 * 8;
        P.S = h.pl.leaves / h.pl.unit;
        P.groups = b ? 2 : 1;
        P.body = (h.a.KH == 1 && h.a.KW == 1 && h.a.pad == 0) ? 0 : 1;
        if (P.body == 0 && !h.a.x2) {
            // one source through the two-source body: no chunk ever comes from the second
            P.k.cpc1 = P.k.sk_leaf * h.pl.leaves + 1;
            P.k.stride2 = 1; P.k.ldx2 = 0; P.k.x2 = nullptr; P.k.x2_bytes = 0;
        }
        P.rot = (l & 1) ? nwg / 2 : 0;
        P.dep_target = l ? tab[l - 1].nblk : 0;
        P.out_arg = h.out_arg;
        P.l2_prefetch = l2_prefetch;
        P.ws0 = sk.ws ? sk.ws + (size_t)((l & 1) * 2 + 0) * region : nullptr;
        P.ws1 = sk.ws ? sk.ws + (size_t)((l & 1) * 2 + 1) * region : nullptr;
        P.cnt = sk.cnt + cnt_off;
        cnt_off += tiles * P.groups;
    }
    return 0;
}

int launch_persist(const void* dev_table, int nl, PersistCtl* ctl, float* out0, float* out1, int nwg, unsigned spin_limit,
                   const LaunchCtx& ctx, double flops, double bytes, bool allow_full) {
    constexpr size_t ab = (size_t)(2 * 64 * 36) * sizeof(float);
    constexpr size_t cb = (size_t)64 * 68 * sizeof(float);
    constexpr size_t smem = ab > cb ? ab : cb;
    static DevOnce once;
    if (int e = set_dyn_lds_once(once, reinterpret_cast<const void*>(&conv_persist_kernel), (int)smem)) return e;
    static int max_wgs[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (!max_wgs[dev]) {
        // co-residency of the whole grid is what the in-launch waits rely on
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_persist_kernel, 256, smem) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return (int)hipGetLastError();
        if (per_cu > 4) per_cu = 4;
        max_wgs[dev] = per_cu * prop.multiProcessorCount;
    }
    // two such launches may run side by side (two handles on two streams): each takes at most half of the slots
    // (allow_full: experiments with the whole chip's slots - the caller guarantees that nothing else persistent is in flight)
    if (nwg > (allow_full ? max_wgs[dev] : max_wgs[dev] / 2)) return (int)hipErrorLaunchOutOfResources;
    ProfScope ps(ctx, "conv_persist_f32<64x64,2x2>", flops, bytes);
    hipLaunchKernelGGL(conv_persist_kernel, dim3(nwg), dim3(256), smem, ctx.stream, static_cast<const PLayer*>(dev_table), nl, ctl,
                       out0, out1, spin_limit);
    return (int)hipGetLastError();
}

}  // namespace specmi
