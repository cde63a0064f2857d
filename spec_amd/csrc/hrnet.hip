// hrnet.hip - HRNet-W32 / W48 trunks for HMR (gfx950).
//
// The reference builds `eval(backbone)(pretrained=True, downsample=True, use_conv=(use_conv == 'conv'))` for
// backbone = 'hrnet_w32-conv' | 'hrnet_w32-interp' | 'hrnet_w48-...' (spec/models/hmr.py:44-51) from the un-vendored
// pare.models.backbone.hrnet (PoseHighResolutionNet of the published HRNet pose code + PARE's multi-scale head):
//
//   stem  conv 3x3/s2 (3->64) + BN + ReLU, conv 3x3/s2 (64->64) + BN + ReLU, layer1 = 4 Bottlenecks (64 -> 256)
//   stage2 (1 module, 2 branches), stage3 (4 modules, 3 branches), stage4 (3 modules, 4 branches); every branch = 4
//   BasicBlocks of width C * 2^l (C = 32 | 48); every module ends with the exchange unit
//       y_i = ReLU( sum_j f_ij(x_j) ),  f_ij = identity (j == i), 1x1 conv + BN + nearest upsample (j > i),
//                                        (i - j) x [3x3/s2 conv + BN (+ ReLU except the last)] (j < i)
//   transitions open a new branch with a 3x3/s2 conv + BN + ReLU of the previous lowest-resolution branch
//   head (downsample=True): every branch is brought to the 1/32 resolution - 'conv': 3 / 2 / 1 x [3x3/s2 conv + BN +
//   ReLU] (downsample_stage_1..3), 'interp': bilinear, align_corners=True - and concatenated: 480 | 720 channels.
//
// Everything dense runs on the kernels of the ResNet path (conv_igemm / conv_wino with BN, residual and ReLU fused);
// this file adds the graph walk, three small kernels (the 3-channel stem conv, the exchange-unit sum with nearest
// upsampling, the align_corners bilinear resize) and the activation pool.  Activations are NHWC; widths that are
// not a multiple of 32 (W48: 48) live in tensors padded to the next multiple (64) whose extra channels are zero.
#include <cstring>

#include "handle.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- 3x3 / stride 2 / pad 1 convolution from the 3-channel NCHW image, + BN + ReLU -> NHWC (B,OH,OW,64) -------------
// 256 threads = 64 output pixels x 4 groups of 16 output channels; the 27 x 64 filter bank sits in LDS.
__global__ void __launch_bounds__(256) hr_stem3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ out, int H, int W, int OH, int OW, long npix) {
    __shared__ float ws[27 * 64];
    for (int i = threadIdx.x; i < 27 * 64; i += 256) ws[i] = w[i];
    __syncthreads();
    const long pix = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
    const int g = threadIdx.x & 3;
    if (pix >= npix) return;
    const unsigned upix = (unsigned)pix;                 // 32-bit index arithmetic (the launcher checks npix < 2^31)
    const int ox = (int)(upix % (unsigned)OW);
    const unsigned r = upix / (unsigned)OW;
    const int oy = (int)(r % (unsigned)OH);
    const unsigned b = r / (unsigned)OH;
    // all 27 taps are loaded unconditionally from clamped coordinates and zeroed afterwards where they fall into the padding
    // (a predicated load compiles to its own exec-masked branch + s_waitcnt: 27 memory latencies in a row)
    const float* img = x + (size_t)b * 3 * H * W;
    float tap[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky, cy = min(max(iy, 0), H - 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx, cx = min(max(ix, 0), W - 1);
                tap[(c * 3 + ky) * 3 + kx] = img[(unsigned)((c * H + cy) * W + cx)];
            }
        }
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const float v = ok ? tap[(c * 3 + ky) * 3 + kx] : 0.f;
                const float* wk = ws + ((c * 3 + ky) * 3 + kx) * 64 + g * 16;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = fmaf(v, wk[e], acc[e]);
            }
        }
    float* o = out + pix * 64 + g * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(acc[q * 4 + e], scale[g * 16 + q * 4 + e], shift[g * 16 + q * 4 + e]), 0.f);
        *reinterpret_cast<f32x4*>(o + q * 4) = v;
    }
}

// ---- exchange-unit sum: out = [ReLU]( ((t0 + t1) + t2) + t3 ), term k nearest-upsampled by 2^sh[k] -------------------
struct FuseArgs {
    const float* p[4];
    int ld[4], sh[4];
    int n, C4, H, W, relu;
    float* out;
    int ldo;
    long total;   // B*H*W*C4
};

__global__ void __launch_bounds__(256) hr_fuse_sum_kernel(const FuseArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const unsigned ui = (unsigned)i;                     // 32-bit index arithmetic (the launcher checks the sizes)
    const unsigned c4 = ui % (unsigned)a.C4;
    unsigned pix = ui / (unsigned)a.C4;
    const unsigned x = pix % (unsigned)a.W;
    pix /= (unsigned)a.W;
    const unsigned y = pix % (unsigned)a.H;
    const unsigned b = pix / (unsigned)a.H;
    // the (up to) four terms are loaded together; absent terms re-read term 0 and are dropped by the select below
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool has = k < a.n;                        // constant indices only: the argument struct stays in SGPRs
        const int sh = has ? a.sh[k] : a.sh[0];
        const unsigned ld = (unsigned)(has ? a.ld[k] : a.ld[0]);
        const float* p = has ? a.p[k] : a.p[0];
        const size_t src = ((size_t)(b * (unsigned)(a.H >> sh) + (y >> sh)) * (unsigned)(a.W >> sh) + (x >> sh)) * ld + c4 * 4;
        v[k] = *reinterpret_cast<const f32x4*>(p + src);
    }
    f32x4 s = v[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.n) { s[0] += v[k][0]; s[1] += v[k][1]; s[2] += v[k][2]; s[3] += v[k][3]; }
    if (a.relu) { s[0] = fmaxf(s[0], 0.f); s[1] = fmaxf(s[1], 0.f); s[2] = fmaxf(s[2], 0.f); s[3] = fmaxf(s[3], 0.f); }
    *reinterpret_cast<f32x4*>(a.out + ((size_t)(b * (unsigned)a.H + y) * (unsigned)a.W + x) * (unsigned)a.ldo + c4 * 4) = s;
}

// ---- F.interpolate(mode='bilinear', align_corners=True) on NHWC ------------------------------------------------------
__global__ void __launch_bounds__(256) hr_resize_ac_kernel(const float* __restrict__ x, int H, int W, int ld, int C4,
                                                            float* __restrict__ out, int OH, int OW, int ldo, float sy,
                                                            float sx, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    long pix = i / C4;
    const int ox = (int)(pix % OW);
    pix /= OW;
    const int oy = (int)(pix % OH);
    const long b = pix / OH;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* base = x + b * (long)H * W * ld + c4 * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x0) * ld);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x1) * ld);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((long)y1 * W + x0) * ld);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((long)y1 * W + x1) * ld);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
    *reinterpret_cast<f32x4*>(out + ((b * OH + oy) * (long)OW + ox) * ldo + c4 * 4) = r;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// network description
// ------------------------------------------------------------------------------------------------------------------
struct HrBlock {
    ConvW c1, c2, c3, ds;
    bool bottleneck = false, has_ds = false;
};
struct HrModule {
    int nb = 0;
    std::vector<std::vector<HrBlock>> branches;          // [branch][block]
    std::vector<std::vector<std::vector<ConvW>>> fuse;   // [i][j] -> conv chain (empty for j == i)
};
constexpr int HR_SLOTS = 10;
struct HrNet {
    int width = 32, use_conv = 1;
    int C[4] = {0, 0, 0, 0}, Cp[4] = {0, 0, 0, 0};
    ConvW conv1, conv2;
    std::vector<HrBlock> layer1;
    ConvW trans1[2], trans2, trans3;
    std::vector<HrModule> stages[3];
    std::vector<ConvW> down[3];
    // activation pool: HR_SLOTS buffers per resolution level, handed out / returned explicitly
    std::vector<void*> allocs;
    float* slot[4][HR_SLOTS] = {};
    bool used[4][HR_SLOTS] = {};
    float* feat_ws = nullptr;
    int pool_B = 0, pool_H = 0, pool_W = 0;
};

const float* hrnet_feat_ws(specmi_handle* h) { return h->hrnet ? h->hrnet->feat_ws : nullptr; }

void hrnet_free(HrNet* n) {
    if (!n) return;
    for (void* p : n->allocs) (void)hipFree(p);
    delete n;
}

static void mk(ConvW& c, const std::string& conv, const std::string& bn, int cin, int cout, int k, int s, int cin_p, int cout_p) {
    c = ConvW();
    c.name = conv; c.bn_name = bn; c.cin = cin; c.cout = cout; c.k = k; c.stride = s; c.pad = k == 3 ? 1 : 0;
    c.cin_p = cin_p; c.cout_p = cout_p;
}

static void build_module(HrNet* n, HrModule& m, const std::string& p, int nb) {
    m.nb = nb;
    m.branches.assign(nb, {});
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < 4; ++k) {
            HrBlock blk;
            const std::string q = p + ".branches." + std::to_string(b) + "." + std::to_string(k);
            mk(blk.c1, q + ".conv1", q + ".bn1", n->C[b], n->C[b], 3, 1, n->Cp[b], n->Cp[b]);
            mk(blk.c2, q + ".conv2", q + ".bn2", n->C[b], n->C[b], 3, 1, n->Cp[b], n->Cp[b]);
            m.branches[b].push_back(blk);
        }
    m.fuse.assign(nb, std::vector<std::vector<ConvW>>(nb));
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) {
            const std::string q = p + ".fuse_layers." + std::to_string(i) + "." + std::to_string(j);
            if (j > i) {            // Sequential(conv1x1, bn, Upsample(nearest))
                ConvW c;
                mk(c, q + ".0", q + ".1", n->C[j], n->C[i], 1, 1, n->Cp[j], n->Cp[i]);
                m.fuse[i][j].push_back(c);
            } else if (j < i) {     // Sequential of (i-j) x Sequential(conv3x3 s2, bn[, relu])
                for (int k = 0; k < i - j; ++k) {
                    const bool last = (k == i - j - 1);
                    ConvW c;
                    const std::string r = q + "." + std::to_string(k);
                    mk(c, r + ".0", r + ".1", n->C[j], last ? n->C[i] : n->C[j], 3, 2, n->Cp[j], last ? n->Cp[i] : n->Cp[j]);
                    m.fuse[i][j].push_back(c);
                }
            }
        }
}

static int commit_module(specmi_handle* h, const std::string& prefix, HrModule& m) {
    int rc;
    for (auto& br : m.branches)
        for (auto& blk : br) {
            if ((rc = commit_conv(h, prefix, blk.c1))) return rc;
            if ((rc = commit_conv(h, prefix, blk.c2))) return rc;
        }
    for (auto& fi : m.fuse)
        for (auto& chain : fi)
            for (auto& c : chain)
                if ((rc = commit_conv(h, prefix, c))) return rc;
    return SPECMI_OK;
}

int hrnet_commit(specmi_handle* h, const std::string& prefix, int width, int use_conv) {
    if (h->hrnet) { hrnet_free(h->hrnet); h->hrnet = nullptr; }
    HrNet* n = new HrNet();
    h->hrnet = n;
    n->width = width; n->use_conv = use_conv;
    for (int l = 0; l < 4; ++l) { n->C[l] = width << l; n->Cp[l] = round_up(n->C[l], 32); }
    int rc;
    mk(n->conv1, "conv1", "bn1", 3, 64, 3, 2, 0, 0);
    mk(n->conv2, "conv2", "bn2", 64, 64, 3, 2, 0, 0);
    if ((rc = commit_conv(h, prefix, n->conv1)) || (rc = commit_conv(h, prefix, n->conv2))) return rc;
    for (int b = 0; b < 4; ++b) {       // layer1 = _make_layer(Bottleneck, 64, 4)
        HrBlock blk;
        blk.bottleneck = true;
        const std::string q = "layer1." + std::to_string(b);
        const int inpl = b == 0 ? 64 : 256;
        mk(blk.c1, q + ".conv1", q + ".bn1", inpl, 64, 1, 1, 0, 0);
        mk(blk.c2, q + ".conv2", q + ".bn2", 64, 64, 3, 1, 0, 0);
        mk(blk.c3, q + ".conv3", q + ".bn3", 64, 256, 1, 1, 0, 0);
        blk.has_ds = b == 0;
        if (blk.has_ds) mk(blk.ds, q + ".downsample.0", q + ".downsample.1", 64, 256, 1, 1, 0, 0);
        if ((rc = commit_conv(h, prefix, blk.c1)) || (rc = commit_conv(h, prefix, blk.c2)) || (rc = commit_conv(h, prefix, blk.c3)))
            return rc;
        if (blk.has_ds && (rc = commit_conv(h, prefix, blk.ds))) return rc;
        n->layer1.push_back(blk);
    }
    // transition1: [Sequential(conv3x3(256->C0), bn, relu), Sequential(Sequential(conv3x3 s2 (256->C1), bn, relu))]
    mk(n->trans1[0], "transition1.0.0", "transition1.0.1", 256, n->C[0], 3, 1, 0, n->Cp[0]);
    mk(n->trans1[1], "transition1.1.0.0", "transition1.1.0.1", 256, n->C[1], 3, 2, 0, n->Cp[1]);
    mk(n->trans2, "transition2.2.0.0", "transition2.2.0.1", n->C[1], n->C[2], 3, 2, n->Cp[1], n->Cp[2]);
    mk(n->trans3, "transition3.3.0.0", "transition3.3.0.1", n->C[2], n->C[3], 3, 2, n->Cp[2], n->Cp[3]);
    if ((rc = commit_conv(h, prefix, n->trans1[0])) || (rc = commit_conv(h, prefix, n->trans1[1])) ||
        (rc = commit_conv(h, prefix, n->trans2)) || (rc = commit_conv(h, prefix, n->trans3)))
        return rc;
    const int nmod[3] = {1, 4, 3};
    for (int s = 0; s < 3; ++s) {
        n->stages[s].assign(nmod[s], HrModule());
        for (int m = 0; m < nmod[s]; ++m) {
            build_module(n, n->stages[s][m], "stage" + std::to_string(s + 2) + "." + std::to_string(m), s + 2);
            if ((rc = commit_module(h, prefix, n->stages[s][m]))) return rc;
        }
    }
    if (use_conv) {   // downsample_stage_{1,2,3} = Sequential([conv3x3 s2, bn, relu] x {3,2,1}): conv at 3k, bn at 3k + 1
        for (int d = 0; d < 3; ++d) {
            const std::string q = "downsample_stage_" + std::to_string(d + 1) + ".";
            for (int k = 0; k < 3 - d; ++k) {
                ConvW c;
                mk(c, q + std::to_string(3 * k), q + std::to_string(3 * k + 1), n->C[d], n->C[d], 3, 2, n->Cp[d], n->Cp[d]);
                if ((rc = commit_conv(h, prefix, c))) return rc;
                n->down[d].push_back(c);
            }
        }
    }
    h->feat_ch = n->C[0] + n->C[1] + n->C[2] + n->C[3];
    return SPECMI_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct Ten {
    float* p = nullptr;
    int lvl = -1, slot = -1;   // slot < 0: not pool-owned
};

struct Runner {
    specmi_handle* h;
    HrNet* n;
    hipStream_t s;
    int B;
    int hh[4], ww[4];
    int err = 0;

    Ten alloc(int lvl) {
        for (int i = 0; i < HR_SLOTS; ++i)
            if (!n->used[lvl][i]) { n->used[lvl][i] = true; Ten t; t.p = n->slot[lvl][i]; t.lvl = lvl; t.slot = i; return t; }
        err = fail(h, SPECMI_ERR_STATE, "hrnet: activation pool of level %d exhausted", lvl);
        return Ten();
    }
    void release(Ten& t) {
        if (t.slot >= 0) n->used[t.lvl][t.slot] = false;
        t = Ten();
    }
    // fused conv + BN (+ residual)(+ ReLU) between NHWC tensors; in_ld / out_ld = channel strides
    int conv(const ConvW& c, const float* in, int ih, int iw, int in_ld, float* out, int out_ld, int out_c, const float* res,
             int relu, const char* label) {
        if (err) return err;
        ConvArgs a;
        const int cin = c.cin_p > 0 ? c.cin_p : c.cin;
        a.x = in; a.w = c.w; a.scale = c.scale; a.shift = c.shift; a.res = res; a.out = out;
        a.B = B; a.H = ih; a.W = iw; a.Cin = cin; a.ldx = in_ld;
        a.OH = conv_out(ih, c.k, c.stride, c.pad); a.OW = conv_out(iw, c.k, c.stride, c.pad);
        a.Cout = out_c; a.Npad = c.Npad; a.ldo = out_ld;
        a.KH = c.k; a.KW = c.k; a.stride = c.stride; a.pad = c.pad; a.relu = relu;
        a.force_variant = opt_i(h, "force_conv_variant", 0);
        a.wino_variant = opt_i(h, "force_wino_variant", 0);
        LaunchCtx ctx{s, &h->prof, label};
        int rc;
        if (c.wino && opt_i(h, "winograd", 1) && conv_wino_supported(a)) {
            a.w = c.wino;
            rc = launch_conv_wino(a, ctx);
        } else {
            rc = launch_conv_igemm(a, ctx);
        }
        if (rc) err = fail(h, SPECMI_ERR_HIP, "launch %s failed: %s", label, hipGetErrorString((hipError_t)rc));
        return err;
    }
    int fuse_sum(const Ten* terms, const int* ld, const int* sh, int nt, int lvl, int C, float* out, int ldo, int relu,
                 const char* label) {
        if (err) return err;
        FuseArgs a;
        for (int k = 0; k < 4; ++k) { a.p[k] = k < nt ? terms[k].p : nullptr; a.ld[k] = k < nt ? ld[k] : 0; a.sh[k] = k < nt ? sh[k] : 0; }
        a.n = nt; a.C4 = C / 4; a.H = hh[lvl]; a.W = ww[lvl]; a.relu = relu; a.out = out; a.ldo = ldo;
        a.total = (long)B * a.H * a.W * a.C4;
        if (a.total >= (1L << 31)) return err = fail(h, SPECMI_ERR_ARG, "hrnet: %s has too many elements for 32-bit indexing", label);
        LaunchCtx ctx{s, &h->prof, label};
        ProfScope ps(ctx, "hrnet_fuse_sum", 0.0, 4.0 * a.total * 4 * (nt + 1));
        hipLaunchKernelGGL(hr_fuse_sum_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, s, a);
        const int rc = (int)hipGetLastError();
        if (rc) err = fail(h, SPECMI_ERR_HIP, "launch %s failed: %s", label, hipGetErrorString((hipError_t)rc));
        return err;
    }
};

}  // namespace

static int hr_ensure_pool(specmi_handle* h, HrNet* n, int B, int H, int W, const int* hh, const int* ww) {
    if (B <= n->pool_B && H == n->pool_H && W == n->pool_W) return SPECMI_OK;
    HIPCHK(h, hipDeviceSynchronize());
    for (void* p : n->allocs) (void)hipFree(p);
    n->allocs.clear();
    const int Bw = B > n->pool_B ? B : n->pool_B;
    // the pool is gone from here on: a failed hipMalloc below must not leave the old geometry (and with it the early
    // return above) pointing at freed slots
    n->pool_B = n->pool_H = n->pool_W = 0;
    n->feat_ws = nullptr;
    for (int l = 0; l < 4; ++l)
        for (int i = 0; i < HR_SLOTS; ++i) { n->slot[l][i] = nullptr; n->used[l][i] = false; }
    for (int l = 0; l < 4; ++l) {
        const size_t bytes = (size_t)Bw * hh[l] * ww[l] * n->Cp[l] * 4;
        for (int i = 0; i < HR_SLOTS; ++i) {
            void* p = nullptr;
            HIPCHK(h, hipMalloc(&p, bytes));
            n->allocs.push_back(p);
            n->slot[l][i] = (float*)p;
            n->used[l][i] = false;
        }
    }
    void* p = nullptr;
    HIPCHK(h, hipMalloc(&p, (size_t)Bw * hh[3] * ww[3] * h->feat_ch * 4));
    n->allocs.push_back(p);
    n->feat_ws = (float*)p;
    n->pool_B = Bw; n->pool_H = H; n->pool_W = W;
    return SPECMI_OK;
}

int hrnet_forward(specmi_handle* h, const float* images, int B, int H, int W, float* feat_out, int* fh, int* fw, hipStream_t s) {
    HrNet* n = h->hrnet;
    if (!n) return fail(h, SPECMI_ERR_STATE, "no HRNet trunk committed");
    if (H % 32 || W % 32 || H < 32 || W < 32)
        return fail(h, SPECMI_ERR_ARG, "HRNet trunks need H and W to be multiples of 32, got %dx%d (the exchange units add a "
                                       "2^k-upsampled map to the higher-resolution one, which only fits then)", H, W);
    Runner r;
    r.h = h; r.n = n; r.s = s; r.B = B;
    const int oh1 = conv_out(H, 3, 2, 1), ow1 = conv_out(W, 3, 2, 1);
    r.hh[0] = conv_out(oh1, 3, 2, 1); r.ww[0] = conv_out(ow1, 3, 2, 1);
    for (int l = 1; l < 4; ++l) { r.hh[l] = conv_out(r.hh[l - 1], 3, 2, 1); r.ww[l] = conv_out(r.ww[l - 1], 3, 2, 1); }
    int rc;
    if ((rc = hr_ensure_pool(h, n, B, H, W, r.hh, r.ww))) return rc;
    for (int l = 0; l < 4; ++l)
        for (int i = 0; i < HR_SLOTS; ++i) n->used[l][i] = false;
    float* feat = feat_out ? feat_out : n->feat_ws;
    const int Ctot = h->feat_ch;

    // ---- stem ------------------------------------------------------------------------------------------------------
    {
        const long npix = (long)B * oh1 * ow1;
        if (npix >= (1L << 31) || (long)3 * H * W >= (1L << 31)) return fail(h, SPECMI_ERR_ARG, "hrnet stem: batch too large for 32-bit indexing");
        LaunchCtx ctx{s, &h->prof, "backbone.conv1"};
        ProfScope ps(ctx, "hrnet_stem3x3_f32", 2.0 * npix * 64 * 27, 4.0 * ((double)B * 3 * H * W + (double)npix * 64));
        hipLaunchKernelGGL(hr_stem3x3_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(256), 0, s, images, n->conv1.w,
                           n->conv1.scale, n->conv1.shift, h->act[0], H, W, oh1, ow1, npix);
        if (int e = (int)hipGetLastError()) return fail(h, SPECMI_ERR_HIP, "hrnet stem launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    const int h0 = r.hh[0], w0 = r.ww[0];
    r.conv(n->conv2, h->act[0], oh1, ow1, 64, h->act[1], 64, 64, nullptr, 1, "backbone.conv2");
    int xi = 1;
    for (size_t b = 0; b < n->layer1.size(); ++b) {
        const HrBlock& k = n->layer1[b];
        int fr[3], nf = 0;
        for (int i = 0; i < 4; ++i) if (i != xi) fr[nf++] = i;
        const int t1 = fr[0], t2 = fr[1], idb = fr[2];
        const std::string p = "backbone.layer1." + std::to_string(b);
        r.conv(k.c1, h->act[xi], h0, w0, k.c1.cin, h->act[t1], 64, 64, nullptr, 1, (p + ".conv1").c_str());
        r.conv(k.c2, h->act[t1], h0, w0, 64, h->act[t2], 64, 64, nullptr, 1, (p + ".conv2").c_str());
        int identity = xi;
        if (k.has_ds) {
            r.conv(k.ds, h->act[xi], h0, w0, k.ds.cin, h->act[idb], 256, 256, nullptr, 0, (p + ".downsample").c_str());
            identity = idb;
        }
        r.conv(k.c3, h->act[t2], h0, w0, 64, h->act[t1], 256, 256, h->act[identity], 1, (p + ".conv3").c_str());
        xi = t1;
    }
    if (r.err) return r.err;

    // ---- transition1 + stages --------------------------------------------------------------------------------------
    std::vector<Ten> x;
    x.push_back(r.alloc(0));
    x.push_back(r.alloc(1));
    if (r.err) return r.err;
    r.conv(n->trans1[0], h->act[xi], h0, w0, 256, x[0].p, n->Cp[0], n->Cp[0], nullptr, 1, "backbone.transition1.0");
    r.conv(n->trans1[1], h->act[xi], h0, w0, 256, x[1].p, n->Cp[1], n->Cp[1], nullptr, 1, "backbone.transition1.1");

    for (int st = 0; st < 3; ++st) {
        for (size_t mi = 0; mi < n->stages[st].size(); ++mi) {
            const HrModule& m = n->stages[st][mi];
            const std::string mp = "backbone.stage" + std::to_string(st + 2) + "." + std::to_string(mi);
            // branches: 4 BasicBlocks each
            for (int b = 0; b < m.nb; ++b)
                for (size_t k = 0; k < m.branches[b].size(); ++k) {
                    const HrBlock& blk = m.branches[b][k];
                    const std::string p = mp + ".branches." + std::to_string(b) + "." + std::to_string(k);
                    Ten t = r.alloc(b), y = r.alloc(b);
                    if (r.err) return r.err;
                    r.conv(blk.c1, x[b].p, r.hh[b], r.ww[b], n->Cp[b], t.p, n->Cp[b], n->Cp[b], nullptr, 1, (p + ".conv1").c_str());
                    r.conv(blk.c2, t.p, r.hh[b], r.ww[b], n->Cp[b], y.p, n->Cp[b], n->Cp[b], x[b].p, 1, (p + ".conv2").c_str());
                    r.release(t);
                    r.release(x[b]);
                    x[b] = y;
                }
            // exchange unit
            std::vector<Ten> outs(m.nb);
            for (int i = 0; i < m.nb; ++i) {
                Ten terms[4];
                int ld[4], sh[4];
                std::vector<Ten> temps;
                for (int j = 0; j < m.nb; ++j) {
                    const std::string p = mp + ".fuse_layers." + std::to_string(i) + "." + std::to_string(j);
                    if (j == i) {
                        terms[j] = x[j]; ld[j] = n->Cp[j]; sh[j] = 0;
                    } else if (j > i) {
                        Ten t = r.alloc(j);
                        if (r.err) return r.err;
                        r.conv(m.fuse[i][j][0], x[j].p, r.hh[j], r.ww[j], n->Cp[j], t.p, n->Cp[i], n->Cp[i], nullptr, 0, p.c_str());
                        temps.push_back(t);
                        terms[j] = t; ld[j] = n->Cp[i]; sh[j] = j - i;
                    } else {
                        Ten cur = x[j];
                        int cur_c = n->Cp[j];
                        for (int k = 0; k < i - j; ++k) {
                            const bool last = (k == i - j - 1);
                            const int lv = j + k + 1, oc = last ? n->Cp[i] : n->Cp[j];
                            Ten t = r.alloc(lv);
                            if (r.err) return r.err;
                            r.conv(m.fuse[i][j][k], cur.p, r.hh[lv - 1], r.ww[lv - 1], cur_c, t.p, oc, oc, nullptr, last ? 0 : 1,
                                   (p + "." + std::to_string(k)).c_str());
                            temps.push_back(t);
                            cur = t; cur_c = oc;
                        }
                        terms[j] = cur; ld[j] = n->Cp[i]; sh[j] = 0;
                    }
                }
                outs[i] = r.alloc(i);
                if (r.err) return r.err;
                r.fuse_sum(terms, ld, sh, m.nb, i, n->Cp[i], outs[i].p, n->Cp[i], 1, (mp + ".fuse." + std::to_string(i)).c_str());
                for (Ten& t : temps) r.release(t);
            }
            for (int b = 0; b < m.nb; ++b) { r.release(x[b]); x[b] = outs[b]; }
            if (r.err) return r.err;
        }
        if (st < 2) {   // transition2 / transition3: a new branch from the lowest-resolution output
            const int nb = st + 2;
            const ConvW& tc = st == 0 ? n->trans2 : n->trans3;
            Ten t = r.alloc(nb);
            if (r.err) return r.err;
            r.conv(tc, x[nb - 1].p, r.hh[nb - 1], r.ww[nb - 1], n->Cp[nb - 1], t.p, n->Cp[nb], n->Cp[nb], nullptr, 1,
                   (std::string("backbone.transition") + std::to_string(st + 2)).c_str());
            x.push_back(t);
        }
    }

    // ---- multi-scale head: everything to the 1/32 resolution, channel-concatenated into feat ---------------------------
    int coff = 0;
    for (int l = 0; l < 3; ++l) {
        if (n->use_conv) {
            Ten cur = x[l];
            const int steps = 3 - l;
            for (int k = 0; k < steps; ++k) {
                const bool last = (k == steps - 1);
                const int lv = l + k + 1;
                const std::string p = "backbone.downsample_stage_" + std::to_string(l + 1) + "." + std::to_string(3 * k);
                if (last) {
                    r.conv(n->down[l][k], cur.p, r.hh[lv - 1], r.ww[lv - 1], n->Cp[l], feat + coff, Ctot, n->C[l], nullptr, 1, p.c_str());
                } else {
                    Ten t = r.alloc(lv);
                    if (r.err) return r.err;
                    r.conv(n->down[l][k], cur.p, r.hh[lv - 1], r.ww[lv - 1], n->Cp[l], t.p, n->Cp[l], n->Cp[l], nullptr, 1, p.c_str());
                    if (cur.p != x[l].p) r.release(cur);
                    cur = t;
                }
            }
            if (cur.p != x[l].p) r.release(cur);
        } else {
            const int OH = r.hh[3], OW = r.ww[3], C4 = n->C[l] / 4;
            const long total = (long)B * OH * OW * C4;
            const float sy = OH > 1 ? (float)(r.hh[l] - 1) / (float)(OH - 1) : 0.f;
            const float sx = OW > 1 ? (float)(r.ww[l] - 1) / (float)(OW - 1) : 0.f;
            LaunchCtx ctx{s, &h->prof, "backbone.interp"};
            ProfScope ps(ctx, "hrnet_resize_bilinear_ac", 0.0, 4.0 * total * 4 * 5);
            hipLaunchKernelGGL(hr_resize_ac_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x[l].p, r.hh[l], r.ww[l],
                               n->Cp[l], C4, feat + coff, OH, OW, Ctot, sy, sx, total);
            if (int e = (int)hipGetLastError()) return fail(h, SPECMI_ERR_HIP, "hrnet resize launch failed: %s", hipGetErrorString((hipError_t)e));
        }
        coff += n->C[l];
    }
    {   // x[3] itself: plain copy into its channel slice
        Ten terms[4] = {x[3]};
        int ld[4] = {n->Cp[3], 0, 0, 0}, sh[4] = {0, 0, 0, 0};
        r.fuse_sum(terms, ld, sh, 1, 3, n->C[3], feat + coff, Ctot, 0, "backbone.concat");
    }
    for (Ten& t : x) r.release(t);
    if (r.err) return r.err;
    *fh = r.hh[3]; *fw = r.ww[3];
    return SPECMI_OK;
}
