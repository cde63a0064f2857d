// conv_igemm_tile.h - the tile body of the implicit-GEMM convolution (conv_igemm.hip header: GEMM view, MFMA mapping, LDS
// layout, split-K canonical tree), as ONE device function shared by
//   * conv_igemm.hip: one workgroup = one (tile, K slice) of one layer (a launch per layer), and
//   * conv_persist.hip: a persistent workgroup walks the (tile, K slice) items of MANY layers of a trunk inside one launch
//     (PERSIST = true) - same operand order, same MFMA chain, same canonical k-sum tree, hence the same bits; what differs is the
//     memory policy of everything another workgroup of the SAME launch has written or will read (activations, residuals,
//     outputs: sc1 write-through stores / sc1 loads, MI355X_MICROARCH.md "inter-workgroup visibility") and the order of the
//     prologue (weights are requested BEFORE the wait for the producing layer, activations after it).
// Reference call sites served: spec/models/hmr.py:92, camcalib/model.py:73 (the ResNet trunks), spec/models/hmr.py:96,
// camcalib/model.py:77-79 (FC layers).
#pragma once
#include <type_traits>

// The split-K / walker hand-offs (sc1 write-through stores, vmcnt(0), relaxed agent-scope ticket, sc1 loads: no release / acquire
// fence) are the form MI355X_MICROARCH.md documents for gfx950 and are stress-tested there (tests/test_gpu_latency.py::
// test_in_kernel_reduction_is_race_free, tests/test_gpu_round5.py); they are NOT the portable HIP memory-model form (release on the
// ticket + acquire in the last arriver = buffer_wbl2 + buffer_inv per workgroup, measured 35 us per launch here).  Refuse to build
// for anything else rather than run a protocol nobody validated there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libspecmi's in-launch hand-offs are validated on gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

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

// Which (tile, K slice, network) of the layer a workgroup computes - blockIdx / gridDim of the per-layer launch, an item of the
// persistent walker's list otherwise - and where its split-K hand-off lives.
struct TileCtx {
    int bid, nblk;     // tile index within the network's tile list / tiles per network   (blockIdx.x / gridDim.x)
    int y, S;          // K slice / slices (slabs per tile)                                 (blockIdx.y / gridDim.y)
    int z;             // network of a grouped launch (0 / 1)                               (blockIdx.z)
    float* ws;         // slab s of tile `tile` at ws + (tile * S + s) * BM * BN
    unsigned* cnt;     // arrival counter of tile `tile` at cnt[tile]
    unsigned tile;
    // PERSIST only
    const unsigned* dep = nullptr;   // != nullptr: activations may be touched once *dep >= dep_target (tiles of the producing layer)
    unsigned dep_target = 0;
    unsigned* done = nullptr;        // += 1 when this tile's outputs are visible device-wide
    unsigned* err = nullptr;         // set to 1 when a bounded spin gave up (results are garbage, the launch still ends)
    float* out = nullptr;            // != nullptr: overrides the layer's output pointer (caller-owned feature buffer)
    unsigned spin_limit = 0;
    int l2_prefetch = 0;             // touch the workgroup's weight slice (one load per 128-byte line) before waiting
    int walk_first = 1;              // 1: the first item this workgroup runs in this layer (SPECMI_WALK_ABLATE builds only)
};

constexpr unsigned kOutOfRange = 0x80000000u;  // >= any buffer extent: the load returns zeros

// The body is a textual fragment (conv_igemm_body.inc: statements over `p`, `t` and the template parameters BM ... PERSIST) so that
// the per-layer kernel of conv_igemm.hip contains it directly - its code generation is then exactly that of a hand-written
// kernel, whatever the inliner would have done with a call - while the persistent walker calls it as a function per item.
// KA: KArgs, or KArgs in the constant address space (a table entry in device memory: every field read becomes a scalar load at
// its use, exactly like a kernel argument - copying the struct first would park ~70 SGPRs)
template <int BM, int BN, int WGM, int WGN, bool IS1X1, int BK, bool DUAL = false, bool SPLITK = false, bool BDIR = false, bool PERSIST = false, typename KA = KArgs>
__device__ __forceinline__ void igemm_tile(const KA& p, const TileCtx& t) {
#include "conv_igemm_body.inc"
}


// fused-conv arguments -> kernel arguments of the sliced 64x64 body (conv_igemm.hip); pl: the layer's canonical tree + unit
void conv_igemm_make_sk_kargs(const ConvArgs& a, const SkPlan& pl, const ConvArgs* b, KArgs& k);
// the shapes the sliced body accepts (what launch_conv_igemm_sk checks): 0 or hipErrorInvalidValue
int conv_igemm_sk_check(const ConvArgs& a, const SkPlan& pl, const ConvArgs* b);

}  // namespace specmi
