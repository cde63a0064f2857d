// api.hip - host side of libspecmi.so: the C ABI of include/specmi.h, parameter staging, workspace management and the launch
// sequences of the two networks.  Weight packing / BatchNorm folding / the composed regressor live in commit.hip, the option table
// in options.hip, device code in conv_igemm.hip / conv_wsplit.hip / conv_wino.hip / stem.hip / head.hip / smpl.hip / ...
#include <cmath>
#include <cstdarg>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "handle.h"

using namespace specmi;

thread_local std::string g_create_err;   // specmi_create's message when no handle exists yet, per calling thread

int fail(specmi_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}

// Every entry point runs on the handle's device and puts the caller's current device back on return (the caller -
// PyTorch - reads hipGetDevice to decide where its next allocation goes; a library must not move it).
struct DeviceGuard {
    int prev = -1;
    hipError_t enter(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) { prev = -1; return e; }
        if (prev == dev) { prev = -1; return hipSuccess; }
        return hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};


int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------
// device memory helpers
// ------------------------------------------------------------------------------------------
int dev_upload(specmi_handle* h, const void* src, size_t bytes, void** out, std::vector<void*>& pool) {
    void* p = nullptr;
    HIPCHK(h, hipMalloc(&p, bytes ? bytes : 4));
    pool.push_back(p);
    if (bytes) HIPCHK(h, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *out = p;
    return SPECMI_OK;
}

int dev_alloc(specmi_handle* h, size_t bytes, void** out, std::vector<void*>& pool) {
    void* p = nullptr;
    HIPCHK(h, hipMalloc(&p, bytes ? bytes : 4));
    pool.push_back(p);
    *out = p;
    return SPECMI_OK;
}

void free_pool(std::vector<void*>& pool) {
    for (void* p : pool) (void)hipFree(p);
    pool.clear();
}

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

// Growing a workspace synchronises the device, frees and allocates - all of which a stream capture forbids.  Say so in words
// instead of a bare HIP error: the remedy is one eager forward of the shape before capturing (GraphedPipeline / GraphedStep do it).
static int sync_for_growth(specmi_handle* h, const char* what) {
    const hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) return SPECMI_OK;
    (void)hipGetLastError();
    if (e == hipErrorStreamCaptureUnsupported || e == hipErrorStreamCaptureImplicit || e == hipErrorStreamCaptureInvalidated ||
        e == hipErrorStreamCaptureIsolation || e == hipErrorStreamCaptureWrongThread || e == hipErrorStreamCaptureUnjoined)
        return fail(h, SPECMI_ERR_STATE, "%s must grow for this batch / resolution, which is not possible while a stream is being "
                    "captured: run one eager forward of this shape first (the capture is invalid now)", what);
    return fail(h, SPECMI_ERR_HIP, "hipDeviceSynchronize failed while growing %s: %s", what, hipGetErrorString(e));
}

// The in-launch hand-offs (split-K tile tickets, the persistent walker's completion counters) rely on counters that every
// launch leaves at zero.  A launch that dies mid-flight does not: zero them whenever that may have happened (after any failed
// forward), at specmi_commit and on demand (specmi_sync_reset).  Enqueued on `s`.
static void reset_sync_state(specmi_handle* h, hipStream_t s) {
    if (h->sk.cnt) (void)hipMemsetAsync(h->sk.cnt, 0, (size_t)h->sk.ncnt * 4, s);
    if (h->pctl) (void)hipMemsetAsync(h->pctl, 0, sizeof(PersistCtl), s);
    if (h->tail_ctl) (void)hipMemsetAsync(h->tail_ctl, 0, (size_t)specmi_handle::kTailCtlWords * 4, s);
    (void)hipGetLastError();
}

static int ensure_ws(specmi_handle* h, int B, int H, int W) {
    const int oh1 = conv_out(H, 7, 2, 3), ow1 = conv_out(W, 7, 2, 3);
    // largest activation: stem output (B,oh1,ow1,64) == layer1 output (B,oh1/2,ow1/2,256) rounded up
    const int oh2 = conv_out(oh1, 3, 2, 1), ow2 = conv_out(ow1, 3, 2, 1);
    size_t elems = (size_t)B * oh1 * ow1 * 64;
    const size_t l1 = (size_t)B * oh2 * ow2 * 256;
    if (l1 > elems) elems = l1;
    const bool grow_act = elems > h->act_elems;
    const bool grow_b = B > h->ws_B;
    if (!grow_act && !grow_b) return SPECMI_OK;
    if (int rc0 = sync_for_growth(h, "the activation workspace")) return rc0;
    // the outgrown buffers stay alive until specmi_destroy (as the split-K workspace does): a hipGraph captured at a smaller
    // batch / resolution has their addresses baked into its kernel nodes, and replaying it after an eager call at a larger size
    // must not write into freed memory.  Growth is geometric (a dimension that must grow grows to at least 1.5x its old size), so
    // an ascending sweep of batch sizes or resolutions retires a geometric series: at most ~3x the final size in total
    // instead of one full copy per step.
    h->ws_retired.insert(h->ws_retired.end(), h->ws_allocs.begin(), h->ws_allocs.end());
    h->ws_allocs.clear();
    if (grow_act && h->act_elems && elems < h->act_elems + h->act_elems / 2) elems = h->act_elems + h->act_elems / 2;
    if (elems < h->act_elems) elems = h->act_elems;
    int Bw = B > h->ws_B ? B : h->ws_B;
    if (grow_b && h->ws_B && Bw < h->ws_B + h->ws_B / 2) Bw = h->ws_B + h->ws_B / 2;
    // the workspace is gone from here on: if an allocation below fails, the next call must not take the early return
    // above on the strength of the old sizes and launch kernels on freed memory
    h->act_elems = 0; h->ws_B = 0;
    for (int i = 0; i < 4; ++i) h->act[i] = nullptr;
    int rc;
    for (int i = 0; i < 4; ++i)
        if ((rc = dev_alloc(h, elems * 4, (void**)&h->act[i], h->ws_allocs))) return rc;
    h->act_elems = elems;
    const int Bp = round_up(Bw, 8);
    const int V = h->smpl.V > 0 ? h->smpl.V : 1;
    if ((rc = dev_alloc(h, (size_t)Bp * h->xc_ld * 4, (void**)&h->xc, h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp * 1024 * 4, (void**)&h->h1, h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp * 1024 * 4, (void**)&h->h2, h->ws_allocs))) return rc;
    for (int i = 0; i < 2; ++i)   // hidden rows of the three CamCalib Linear chains when they run as one launch per layer
        if ((rc = dev_alloc(h, (size_t)3 * Bp * 1024 * 4, (void**)&h->fc_hidden[i], h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp * 2048 * 4, (void**)&h->xf, h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp * 216 * 4, (void**)&h->rot_ws, h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp * 10 * 4, (void**)&h->betas_ws, h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp * 3 * 4, (void**)&h->cam_ws, h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp * V * 3 * 4, (void**)&h->verts_ws, h->ws_allocs))) return rc;
    const int Bp32 = round_up(Bw, 32);   // the skinning kernel's operands come in tiles of 32 images (smpl.hip)
    if ((rc = dev_alloc(h, (size_t)Bp32 * SMPL_KQ * 8 * 4, (void**)&h->pf_ws, h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp32 * 288 * 4, (void**)&h->A_ws, h->ws_allocs))) return rc;
    if ((rc = dev_alloc(h, (size_t)Bp * 72 * 4, (void**)&h->pj_ws, h->ws_allocs))) return rc;
    if (!h->tail_ctl) {   // (once: captured graphs keep naming it)
        HIPCHK(h, hipMalloc((void**)&h->tail_ctl, (size_t)specmi_handle::kTailCtlWords * 4));
        HIPCHK(h, hipMemset(h->tail_ctl, 0, (size_t)specmi_handle::kTailCtlWords * 4));
    }
    HIPCHK(h, hipMemset(h->pf_ws, 0, (size_t)Bp32 * SMPL_KQ * 8 * 4));   // rows 218..223 of the feature operand stay zero
    HIPCHK(h, hipMemset(h->A_ws, 0, (size_t)Bp32 * 288 * 4));
    h->ws_B = Bw;
    return SPECMI_OK;
}

// split-K workspace (partial tiles + arrival counters), grown on demand and never shrunk.  Growing frees and allocates, which a
// stream capture forbids: the eager warm-up call of a shape (GraphedPipeline runs two) sizes it, captures find it large enough.
static int ensure_sk(specmi_handle* h, size_t floats, int ncnt) {
    if (floats <= h->sk.floats && ncnt <= h->sk.ncnt) return SPECMI_OK;
    if (int rc0 = sync_for_growth(h, "the split-K workspace")) return rc0;
    if (floats < h->sk.floats) floats = h->sk.floats;
    if (ncnt < h->sk.ncnt) ncnt = h->sk.ncnt;
    if (h->sk.floats && floats > h->sk.floats && floats < h->sk.floats + h->sk.floats / 2) floats = h->sk.floats + h->sk.floats / 2;   // geometric: see ensure_ws
    floats = (floats + ((size_t)1 << 20) - 1) >> 20 << 20;
    ncnt = round_up(ncnt < 4096 ? 4096 : ncnt, 4096);
    // the outgrown buffers stay alive until specmi_destroy: a hipGraph captured earlier (GraphedPipeline at another batch
    // size) has their addresses baked into its kernel nodes
    if (h->sk.ws) h->sk_retired.push_back(h->sk.ws);
    if (h->sk.cnt) h->sk_retired.push_back(h->sk.cnt);
    h->sk = SkWs{};
    float* ws = nullptr;
    unsigned* cnt = nullptr;
    HIPCHK(h, hipMalloc((void**)&ws, floats * 4));
    hipError_t e = hipMalloc((void**)&cnt, (size_t)ncnt * 4);
    if (e == hipSuccess) e = hipMemset(cnt, 0, (size_t)ncnt * 4);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(ws);
        if (cnt) (void)hipFree(cnt);
        return fail(h, SPECMI_ERR_HIP, "split-K workspace: %s", hipGetErrorString(e));
    }
    h->sk.ws = ws; h->sk.floats = floats; h->sk.cnt = cnt; h->sk.ncnt = ncnt;
    return SPECMI_OK;
}

// ------------------------------------------------------------------------------------------
// launch sequences
// ------------------------------------------------------------------------------------------
static int run_fc(specmi_handle* h, const FcW& fc, const float* x, int ldx, int B, const float* res, float* out,
                  int ldo, hipStream_t s, const char* label) {
    // Rows are processed in blocks of at most 1024 so that the K slicing (and with it the summation order of every
    // output) is the same for every batch size: an image's result must not depend on how many images share its launch
    // (rank-sharded 8 x 256 == unsharded 2048, bit for bit).
    constexpr int RB = 1024;
    for (int b0 = 0; b0 < B; b0 += RB) {
        const int nb = B - b0 < RB ? B - b0 : RB;
        ConvArgs a;
        a.x = x + (size_t)b0 * ldx; a.w = fc.w; a.scale = fc.scale; a.shift = fc.shift;
        a.res = res ? res + (size_t)b0 * ldo : nullptr; a.out = out + (size_t)b0 * ldo;
        a.B = nb; a.H = 1; a.W = 1; a.Cin = fc.Kp; a.ldx = ldx;
        a.OH = 1; a.OW = 1; a.Cout = fc.nout; a.Npad = fc.Npad; a.ldo = ldo;
        a.KH = 1; a.KW = 1; a.stride = 1; a.pad = 0; a.relu = 0;
        a.force_variant = opt_i(h, "force_conv_variant", 0);
        LaunchCtx ctx{s, &h->prof, label};
        const int S = opt_i(h, "fc_splitk", 1) ? conv_igemm_splitk_plan(a) : 1;
        if (S > 1) {
            int rc;
            if ((rc = ensure_sk(h, conv_igemm_sk_ws_floats(a, S, 1), conv_igemm_sk_tiles(a, 1)))) return rc;
            SkPlan pl;
            pl.leaves = S;      // flat fold of S slices, the order these GEMMs have had since round 1
            LAUNCHCHK(h, launch_conv_igemm_sk(a, pl, h->sk, ctx), label);
            continue;
        }
        LAUNCHCHK(h, launch_conv_igemm(a, ctx), label);
    }
    return SPECMI_OK;
}

// images NCHW -> layer4 map NHWC in *feat (a workspace buffer unless feat_out given)
// One launch of the trunk's op list: the stem, the max-pool or a fused conv, with the buffer each
// operand lives in (act[] index, -1 = none, -2 = the caller's feature buffer) and its per-image size.
struct TrunkOp {
    int kind;  // 0 stem, 1 maxpool, 2 conv
    const ConvW* c;
    int in_buf, out_buf, res_buf;
    int H, W;            // input spatial size
    int OH, OW;          // output spatial size
    int relu;
    std::string label;
    size_t in_img, out_img;  // floats per image of input / output (and residual)
    // fused downsample branch (conv3 ops only): the block input as a second A source
    const Bneck* fused = nullptr;
    int in2_buf = -1, H2 = 0, W2 = 0;
    size_t in2_img = 0;
};

// What one op of the list launches: the fused-conv arguments and the kernel family they go to
struct OpLaunch {
    int kind = 2;             // 0 stem, 1 maxpool, 2 conv
    ConvArgs a;
    int family = 0;           // conv: 0 implicit GEMM, 1 Winograd, 2 split-bf16 (optional path)
    int sk = 1;               // latency plan: K slices of the implicit GEMM (1 = the throughput kernel)
    const void* wsplit = nullptr;
    int terms = 0;
    const float *x = nullptr, *w = nullptr, *scale = nullptr, *shift = nullptr;   // stem / maxpool operands
    float* out = nullptr;
};

static OpLaunch prepare_op(specmi_handle* h, const TrunkOp& op, const float* images, float* feat_out, int b0, int nb,
                           int Himg, int Wimg, int mode = 0) {
    const bool latency = mode != 0;
    auto buf = [&](int idx) -> float* { return idx == -2 ? feat_out : h->act[idx]; };
    OpLaunch L;
    L.kind = op.kind;
    if (op.kind == 0) {
        L.x = images + (size_t)b0 * 3 * Himg * Wimg; L.w = h->stem.w; L.scale = h->stem.scale; L.shift = h->stem.shift;
        L.out = buf(op.out_buf) + (size_t)b0 * op.out_img;
        return L;
    }
    if (op.kind == 1) {
        L.x = buf(op.in_buf) + (size_t)b0 * op.in_img;
        L.out = buf(op.out_buf) + (size_t)b0 * op.out_img;
        return L;
    }
    const ConvW& c = *op.c;
    ConvArgs& a = L.a;
    a.x = buf(op.in_buf) + (size_t)b0 * op.in_img;
    a.w = c.w; a.scale = c.scale; a.shift = c.shift;
    a.res = op.res_buf == -1 ? nullptr : buf(op.res_buf) + (size_t)b0 * op.out_img;
    a.out = buf(op.out_buf) + (size_t)b0 * op.out_img;
    a.B = nb; a.H = op.H; a.W = op.W; a.Cin = c.cin; a.ldx = c.cin;
    a.OH = op.OH; a.OW = op.OW; a.Cout = c.cout; a.Npad = c.Npad; a.ldo = c.cout;
    a.KH = c.k; a.KW = c.k; a.stride = c.stride; a.pad = c.pad; a.relu = op.relu;
    a.force_variant = opt_i(h, "force_conv_variant", 0);
    a.wino_variant = opt_i(h, "force_wino_variant", 0);
    if (op.fused) {
        const Bneck& bk = *op.fused;
        a.w = bk.f_w; a.scale = bk.f_scale; a.shift = bk.f_shift; a.Npad = bk.f_Npad; a.res = nullptr;
        a.x2 = buf(op.in2_buf) + (size_t)b0 * op.in2_img;
        a.H2 = op.H2; a.W2 = op.W2; a.ldx2 = bk.ds.cin; a.Cin2 = bk.ds.cin; a.stride2 = bk.ds.stride;
    }
    if (const void* wsplit = op.fused ? op.fused->f_wsplit : c.wsplit) {
        const int terms = opt_i(h, "conv_precision", 0);
        // 3x3 layers: the bf16 implicit GEMM spends 9 taps x terms / 16 of an fp32 MFMA cycle per MAC, the fp32 Winograd kernel
        // 16 / 36: with three terms the bf16 path wins on every 3x3 layer, with six only where Winograd does not apply (stride 2)
        const bool wino_ok = c.wino && opt_i(h, "winograd", 1) && conv_wino_supported(a);
        const bool take = c.k == 1 || terms == 3 || !wino_ok || opt_i(h, "conv_precision_3x3", 0);
        if ((terms == 3 || terms == 6) && take && conv_bf16s_supported(a)) {
            L.family = 2; L.wsplit = wsplit; L.terms = terms;
            return L;
        }
    }
    bool wino = c.wino && opt_i(h, "winograd", 1) && conv_wino_supported(a);
    if (latency && !a.force_variant) {
        // Latency plan (batch <= 10 by default).  Every choice below is a function of the layer's per-image shape, never of the batch:
        // an image's bits are the same at batch 1 and 16.  Winograd only where an image alone brings enough 2x2 tiles to
        // fill its 32-tile rows (layer1 / layer2 at 224^2); layer3 / layer4 (49 / 16 tiles per image, U = 16/9 of the
        // weight bytes) run the direct kernel over K slices.
        // (wide layers need proportionally more tiles: layer4 of a 600 x 1066 frame has 170 tiles per image = 48 Winograd workgroups
        // walking Cin = 512 for 75 us, the sliced direct kernel takes 45)
        const int min_tiles = opt_i(h, "latency_wino_min_tiles", 128);
        if (wino && ((a.OH + 1) / 2) * ((a.OW + 1) / 2) < (min_tiles > a.Cin || min_tiles == 0 || min_tiles >= 100000 ? min_tiles : a.Cin)) wino = false;
        if (mode == 2) wino = false;   // 'single': batch 1-2, the direct kernel wins on layer1 / layer2 too
        if (!wino) L.sk = conv_igemm_sk_slices(a, opt_i(h, "latency_target_wgs", 256), opt_i(h, "latency_min_chunks", 4));
    }
    if (wino) {
        a.w = c.wino;
        L.family = 1;
    }
    return L;
}

// plan: 0 auto, 1 throughput, 2 latency, 3 single.  Returns the trunk mode of this call: 0 throughput kernels, 1 latency (sliced
// direct kernels on layer3 / layer4, Winograd where an image fills its rows), 2 single (round 5: batch 1-2, the reference's demo
// granularity, scripts/camcalib_demo.py:95-102 - every 3x3 convolution on the sliced direct kernel, which is faster there by
// 21 / 12 us at batch 1 / 2 (profiles/r04_b_latency_layers.txt) and leaves the whole trunk behind the max-pool as one run of
// implicit-GEMM layers for the persistent walker).  auto = single up to "single_max_batch" (2) images' worth of pixels, latency up
// to N = "latency_max_batch" (10) for the trunk PAIR / "latency_max_batch_single" (16) for one trunk - measured per batch size
// (profiles/r04_l_plan_crossover.jsonl: pair 1.67 vs 1.78 ms at 10 images, 2.18 vs 2.15 at 12; one trunk after the other 3.08 vs
// 3.32 ms still at 16; a single CamCalib frame at 600 x 1066 = 12.7 crops' worth of rows: 2.10 vs 2.27 ms for the demo's one-frame step)
static int trunk_mode(specmi_handle* h, int B, int H, int W, bool pair = false) {
    const int plan = opt_i(h, "plan", 0);
    if (plan == 1) return 0;
    if (plan == 2) return 1;
    if (plan == 3) return 2;
    const long px = (long)B * H * W, crop = 224L * 224;
    if (px <= (long)opt_i(h, "single_max_batch", 2) * crop) return 2;
    const int nmax = pair ? opt_i(h, "latency_max_batch", 10) : opt_i(h, "latency_max_batch_single", 16);
    return px <= (long)nmax * crop ? 1 : 0;
}
// the FC layers behind the trunk see batch rows only: the small-batch GEMV kernel (head.hip) up to "latency_max_batch" rows
// the unit rule of the sliced 64x64 kernel (conv_igemm_sk_plan): "latency_unit_model" = 1 -> the round model (fill <= 0 = -slots),
// 0 -> the threshold "latency_fill_wgs"
// px = pixels of the trunk call (B x H x W): beyond ten 224 x 224 crops' worth - a single trunk keeps the latency plan up to 16 - the
// threshold is "latency_fill_wgs_large" (400): measured per batch under the auto structure (profiles/r05_n_unit_rule_sweep.jsonl:
// 1.99 -> 1.80 ms at batch 11, 2.13 -> 1.91 at 12, 2.20 -> 2.10 at 14; at batch <= 10 240 stays ahead: 1.55 vs 2.06 ms at 10)
static int sk_fill(specmi_handle* h, long px = 0) {
    if (opt_i(h, "latency_unit_model", 0)) return -opt_i(h, "latency_unit_slots", 256);
    return px > 10L * 224 * 224 ? opt_i(h, "latency_fill_wgs_large", 400) : opt_i(h, "latency_fill_wgs", 240);
}

static bool use_latency_heads(specmi_handle* h, int B) {
    const int plan = opt_i(h, "plan", 0);
    return opt_i(h, "fc_gemv", 1) && (plan == 2 || plan == 3 || (plan == 0 && B <= opt_i(h, "latency_max_batch", 10)));
}

// partner != nullptr: the same op of a second network, launched together (one grouped launch); the caller has checked
// that both ops have the same kind / family / shape
static int launch_op(specmi_handle* h, const TrunkOp& op, const OpLaunch& L, const OpLaunch* partner, int nb, int Himg, int Wimg,
                     hipStream_t s) {
    LaunchCtx ctx{s, &h->prof, op.label.c_str()};
    if (L.kind == 0) {
        StemPair sp;
        if (partner) { sp.x = partner->x; sp.w = partner->w; sp.scale = partner->scale; sp.shift = partner->shift; sp.out = partner->out; }
        LAUNCHCHK(h, launch_stem(L.x, L.w, L.scale, L.shift, L.out, nb, Himg, Wimg, op.OH, op.OW, 1, ctx, partner ? &sp : nullptr), "stem");
        return SPECMI_OK;
    }
    if (L.kind == 1) {
        LAUNCHCHK(h, launch_maxpool3x3s2(L.x, L.out, nb, op.H, op.W, 64, op.OH, op.OW, ctx, partner ? partner->x : nullptr,
                                         partner ? partner->out : nullptr), "maxpool");
        return SPECMI_OK;
    }
    if (L.family == 2) {
        LAUNCHCHK(h, launch_conv_bf16s(L.a, L.wsplit, L.terms, ctx), op.label.c_str());
        if (partner) LAUNCHCHK(h, launch_conv_bf16s(partner->a, partner->wsplit, partner->terms, ctx), op.label.c_str());
        return SPECMI_OK;
    }
    if (L.sk > 1) {
        const int groups = partner ? 2 : 1;
        int rc;
        SkPlan pl = conv_igemm_sk_plan(L.a, groups, opt_i(h, "latency_target_wgs", 256), opt_i(h, "latency_min_chunks", 4),
                                       sk_fill(h, (long)nb * Himg * Wimg));
        const int fu = opt_i(h, "latency_force_unit", 0);   // tests: 1 leaf / 2 group / 3 whole K per workgroup, whatever the batch
        if (fu) pl.unit = fu == 1 ? 1 : (fu == 2 ? pl.G : pl.leaves);
        // The wave-split unit (conv_wsplit.hip, round 5): a 32x32 tile per workgroup, the G leaves of a group on its waves side by
        // side - four times the tiles, slabs of 4 KB per GROUP or none.  Same canonical tree, same bits: a pure speed choice, made
        // from the per-layer tables of profiles/r05_c_wsplit_layers.txt:
        //   * taken when the 64x64 kernel would have to slice K across workgroups (its unit is not the whole K), the layer has one A
        //     source (the strided second source of a folded downsample branch measured slower at every batch) and the launch stays
        //     under "wsplit_max_units" (1400 for the trunk pair, "wsplit_max_units_single" 500 for one trunk) leaf-units = 32x32 tiles x
        //     groups: beyond, several workgroups share a CU and the
        //     8 KB of operands per wave-chunk (6 in the 64x64 kernel) cost more than the slabs (batch 8: layer3 1568 units, slower);
        //   * one group per workgroup (slab per group) or all groups (no slab): whichever needs fewer rounds of workgroups per CU,
        //     a round costing one leaf (L chunks of ~0.45 us) and the slab hand-off ~1 us - the rule that reproduces every row of
        //     the tables (batch 1-4, layer2-4).
        // "wsplit": 0 never, 1 (default) by this rule, 2 / 3 always with one group / all groups per workgroup (tests, tables).
        const int wsplit = opt_i(h, "wsplit", 1);
        if (wsplit && !fu) {
            SkPlan pw = pl;
            const long t32 = conv_wsplit_tiles(L.a, groups);
            const long ng = pl.leaves / (pl.G > 0 ? pl.G : 1);
            const int nch = L.a.KH * L.a.KW * (L.a.Cin / 32) + (L.a.x2 ? L.a.Cin2 / 32 : 0);
            const double t_leaf = 0.45 * (double)(nch / pl.leaves);
            const long slots = opt_i(h, "wsplit_slots", 256);
            const double cost_all = (double)((t32 + slots - 1) / slots) * (double)ng * t_leaf;
            const double cost_group = (double)((t32 * ng + slots - 1) / slots) * t_leaf + 1.0;
            pw.unit = (wsplit == 3 || (wsplit == 1 && cost_all <= cost_group)) ? pl.leaves : pl.G;
            // (whole-step sweep, profiles/r05_d_structure_sweep.jsonl: a single trunk's launches - two trunks on two streams, batch
            // 4-10 - want the cap at 500 units, 1.245 vs 1.285 ms at batch 6, 1.441 vs 1.487 at 8; the pair's grouped launches between 1300 and 1500:
            // 0.884 vs 0.913 ms at batch 3 and 1.312 vs 1.421 at 6 with layer3 / layer4 on the unit (<= 1280 units), 1.546 vs 1.573 at
            // batch 8 without it (1568 / 1664 units))
            const long max_units = groups == 2 ? opt_i(h, "wsplit_max_units", 1400) : opt_i(h, "wsplit_max_units_single", 500);
            const bool take = wsplit > 1 || (pl.unit != pl.leaves && !L.a.x2 && t32 * ng <= max_units);
            if (take && conv_wsplit_supported(L.a, pw)) {
                if ((rc = ensure_sk(h, conv_wsplit_ws_floats(L.a, pw.leaves / pw.unit, groups), conv_wsplit_tiles(L.a, groups)))) return rc;
                const int wrc = launch_conv_wsplit(L.a, pw, h->sk, ctx, partner ? &partner->a : nullptr);
                if (wrc != SK_NEEDS_BATCH_SPLIT) {
                    LAUNCHCHK(h, wrc, op.label.c_str());
                    return SPECMI_OK;
                }
            }
        }
        if ((rc = ensure_sk(h, conv_igemm_sk_ws_floats(L.a, pl.leaves / pl.unit, groups), conv_igemm_sk_tiles(L.a, groups)))) return rc;
        rc = launch_conv_igemm_sk(L.a, pl, h->sk, ctx, partner ? &partner->a : nullptr);
        if (rc != SK_NEEDS_BATCH_SPLIT) {       // misalignment, a bad plan, an undersized workspace: errors, not a silent change of kernel
            LAUNCHCHK(h, rc, op.label.c_str());
            return SPECMI_OK;
        }
        // the sliced launcher does not split the batch (activations past 32-bit addressing, plan = 'latency' pinned at a large
        // batch or resolution): the throughput launcher below does - other bits (the plans differ anyway), never an error
    }
    int rc = L.family == 1 ? launch_conv_wino(L.a, ctx, partner ? &partner->a : nullptr)
                           : launch_conv_igemm(L.a, ctx, partner ? &partner->a : nullptr);
    if (rc == (int)hipErrorInvalidValue && partner) {
        // a grouped launch is refused when the batch's activations pass the 32-bit addressing limit of one launch (the
        // separate launchers split the batch instead): two launches, same kernels, same bits
        (void)hipGetLastError();
        rc = L.family == 1 ? launch_conv_wino(L.a, ctx) : launch_conv_igemm(L.a, ctx);
        if (!rc) rc = L.family == 1 ? launch_conv_wino(partner->a, ctx) : launch_conv_igemm(partner->a, ctx);
    }
    LAUNCHCHK(h, rc, op.label.c_str());
    return SPECMI_OK;
}

static int exec_op(specmi_handle* h, const TrunkOp& op, const float* images, float* feat_out, int b0, int nb,
                   int Himg, int Wimg, hipStream_t s, int mode = 0) {
    const OpLaunch L = prepare_op(h, op, images, feat_out, b0, nb, Himg, Wimg, mode);
    return launch_op(h, op, L, nullptr, nb, Himg, Wimg, s);
}

struct TrunkPlan {
    std::vector<TrunkOp> ops;
    std::vector<int> stage_of;   // resnet stage (0 = stem/pool, 1..4) of each op
    int final_buf = 1, fh = 0, fw = 0;
};

// the trunk's op list for an (H, W) input: the stem, the max-pool and one fused conv per layer, with the ping-pong
// buffers they read and write
static void plan_trunk(specmi_handle* h, int H, int W, bool to_caller, TrunkPlan& P) {
    float* const feat_out = to_caller ? reinterpret_cast<float*>(1) : nullptr;   // only its null-ness matters here
    std::vector<TrunkOp>& ops = P.ops;
    std::vector<int>& stage_of = P.stage_of;
    const int oh1 = conv_out(H, 7, 2, 3), ow1 = conv_out(W, 7, 2, 3);
    int ch = conv_out(oh1, 3, 2, 1), cw = conv_out(ow1, 3, 2, 1);
    ops.push_back({0, nullptr, -1, 0, -1, H, W, oh1, ow1, 1, "backbone.conv1", (size_t)3 * H * W, (size_t)oh1 * ow1 * 64});
    stage_of.push_back(0);
    ops.push_back({1, nullptr, 0, 1, -1, oh1, ow1, ch, cw, 0, "backbone.maxpool", (size_t)oh1 * ow1 * 64, (size_t)ch * cw * 64});
    stage_of.push_back(0);
    int xi = 1;  // index of the buffer holding the block input
    int final_buf = 1;
    for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
        const Bneck& bk = h->blocks[bi];
        const bool last = (bi + 1 == h->blocks.size());
        int free_idx[3], nf = 0;
        for (int i = 0; i < 4; ++i)
            if (i != xi) free_idx[nf++] = i;
        const int t1 = free_idx[0], t2 = free_idx[1], idb = free_idx[2];
        const int stage = bk.c1.name[5] - '0';   // "layerN.b.conv1"
        const std::string p = "backbone." + bk.c1.name.substr(0, bk.c1.name.rfind('.'));
        const int bstride = bk.basic ? bk.c1.stride : bk.c2.stride;
        const int oh = conv_out(ch, 3, bstride, 1), ow = conv_out(cw, 3, bstride, 1);
        auto add = [&](const ConvW& c, int in, int out, int res, int ih, int iw, int ooh, int oow, int relu, const char* nm) {
            ops.push_back({2, &c, in, out, res, ih, iw, ooh, oow, relu, p + nm, (size_t)ih * iw * c.cin, (size_t)ooh * oow * c.cout});
            stage_of.push_back(stage);
        };
        if (bk.basic) {
            // BasicBlock: relu(bn1(conv3x3_s(x))) -> relu(bn2(conv3x3(.)) + identity | downsample(x))
            add(bk.c1, xi, t1, -1, ch, cw, oh, ow, 1, ".conv1");
            int identity = xi;
            if (bk.has_ds) {
                add(bk.ds, xi, idb, -1, ch, cw, oh, ow, 0, ".downsample");
                identity = idb;
            }
            const int out = (last && feat_out) ? -2 : t2;
            add(bk.c2, t1, out, identity, oh, ow, oh, ow, 1, ".conv2");
            ch = oh; cw = ow;
            xi = out;
            final_buf = out;
            continue;
        }
        add(bk.c1, xi, t1, -1, ch, cw, ch, cw, 1, ".conv1");
        add(bk.c2, t1, t2, -1, ch, cw, oh, ow, 1, ".conv2");
        int identity = xi;
        const bool fuse = bk.has_ds && bk.f_w && opt_i(h, "fuse_downsample", 1);
        if (bk.has_ds && !fuse) {
            add(bk.ds, xi, idb, -1, ch, cw, oh, ow, 0, ".downsample");
            identity = idb;
        }
        const int out = (last && feat_out) ? -2 : t1;  // t1 is dead after conv2
        add(bk.c3, t2, out, fuse ? -1 : identity, oh, ow, oh, ow, 1, fuse ? ".conv3+downsample" : ".conv3");
        if (fuse) {
            TrunkOp& o = ops.back();
            o.fused = &bk; o.in2_buf = xi; o.H2 = ch; o.W2 = cw; o.in2_img = (size_t)ch * cw * bk.ds.cin;
        }
        ch = oh; cw = ow;
        xi = t1;
        final_buf = out;
    }
    P.final_buf = final_buf; P.fh = ch; P.fw = cw;
}

// ---- persistent runs (conv_persist.hip) ----------------------------------------------------------------------------------
// Ops [i0, i1) of a trunk (Lb == nullptr) or of a trunk pair - all implicit-GEMM convolutions of the latency / single plan - as
// ONE launch.  Returns SPECMI_OK, an error, or -1: "not for the walker" (a shape it does not take: the caller launches the
// layers one by one).
static int persist_run(specmi_handle* h, const std::vector<OpLaunch>& La, const std::vector<OpLaunch>* Lb, const std::vector<TrunkOp>& ops,
                       size_t i0, size_t i1, float* feat_a, float* feat_b, hipStream_t s) {
    const int nl = (int)(i1 - i0);
    const int groups = Lb ? 2 : 1;
    std::vector<PersistLayerHost> lay((size_t)nl);
    float *out0 = nullptr, *out1 = nullptr;
    const int fill = opt_i(h, "persist_fill_wgs", 0) > 0 ? opt_i(h, "persist_fill_wgs", 0) : sk_fill(h);      // 0 = the latency fill
    const int fu = opt_i(h, "latency_force_unit", 0);
    for (int l = 0; l < nl; ++l) {
        PersistLayerHost& P = lay[(size_t)l];
        P.a = La[i0 + l].a;
        P.pair = Lb != nullptr;
        if (Lb) P.b = (*Lb)[i0 + l].a;
        P.pl = conv_igemm_sk_plan(P.a, groups, opt_i(h, "latency_target_wgs", 256), opt_i(h, "latency_min_chunks", 4), fill);
        if (fu) P.pl.unit = fu == 1 ? 1 : (fu == 2 ? P.pl.G : P.pl.leaves);
        if (ops[i0 + l].out_buf == -2) {
            // the caller's feature buffer changes from call to call (and between the warm-up and the capture of a graph): it is a
            // launch argument, the table holds a placeholder
            if ((reinterpret_cast<uintptr_t>(feat_a) & 15) || (Lb && (reinterpret_cast<uintptr_t>(feat_b) & 15))) return -1;
            P.out_arg = 1;
            out0 = feat_a; out1 = feat_b;
            P.a.out = nullptr;
            if (Lb) P.b.out = nullptr;
        }
    }
    const int nwg = opt_i(h, "persist_wgs", 0) > 0 ? opt_i(h, "persist_wgs", 0) : (Lb ? 512 : 256);           // 0 = by context
    const int l2pf = opt_i(h, "persist_l2_prefetch", 0);
    size_t ws_need = 0;
    int cnt_need = 0;
    if (persist_fill_table(lay.data(), nl, SkWs{}, nwg, l2pf, nullptr, &ws_need, &cnt_need)) return -1;
    int rc;
    if ((rc = ensure_sk(h, ws_need ? ws_need : 1, cnt_need))) return rc;
    std::vector<unsigned char> img(persist_table_bytes(nl));
    if (persist_fill_table(lay.data(), nl, h->sk, nwg, l2pf, img.data(), &ws_need, &cnt_need)) return -1;
    const specmi_handle::PersistTable* tab = nullptr;
    for (const auto& t : h->persist_tables)
        if (t.nl == nl && t.img.size() == img.size() && memcmp(t.img.data(), img.data(), img.size()) == 0) { tab = &t; break; }
    if (!tab || !h->pctl) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
        if (st != hipStreamCaptureStatusNone)
            return fail(h, SPECMI_ERR_STATE, "the layer table of this batch / resolution is not on the device yet, which is not possible while "
                        "a stream is being captured: run one eager forward of this shape first (the capture is invalid now)");
        if (!h->pctl) {
            HIPCHK(h, hipMalloc((void**)&h->pctl, sizeof(PersistCtl)));
            HIPCHK(h, hipMemset(h->pctl, 0, sizeof(PersistCtl)));
        }
        if (!tab) {
            specmi_handle::PersistTable t;
            t.nl = nl;
            HIPCHK(h, hipMalloc(&t.dev, img.size()));
            HIPCHK(h, hipMemcpy(t.dev, img.data(), img.size(), hipMemcpyHostToDevice));
            t.img = std::move(img);
            h->persist_tables.push_back(std::move(t));
            tab = &h->persist_tables.back();
        }
    }
    LaunchCtx ctx{s, &h->prof, ops[i0].label.c_str()};
    LAUNCHCHK(h, launch_persist(tab->dev, nl, h->pctl, out0, out1, nwg, (unsigned)opt_i(h, "persist_spin_limit", 400000), ctx, 0.0, 0.0,
                                opt_i(h, "persist_allow_full", 0) != 0),
              "persistent trunk run");
    return SPECMI_OK;
}

// Launch ops [first, n) of a trunk (hb == nullptr) or of a trunk pair.  mode != 0 and option "persist" (default 0: opt-in): maximal runs of
// implicit-GEMM convolutions go to the persistent walker (one launch per run of up to 64 layers), everything else - the stem,
// the max-pool, Winograd layers, the optional split-bf16 path - is launched op by op as before.
static int launch_ops(specmi_handle* ha, specmi_handle* hb, const TrunkPlan& Pa, const TrunkPlan* Pb, const float* img_a, const float* img_b,
                      int B, int H, int W, float* feat_a, float* feat_b, size_t first, int mode, hipStream_t s) {
    const size_t n = Pa.ops.size();
    std::vector<OpLaunch> La(n), Lb(hb ? n : 0);
    for (size_t i = first; i < n; ++i) {
        La[i] = prepare_op(ha, Pa.ops[i], img_a, feat_a, 0, B, H, W, mode);
        if (!hb) continue;
        Lb[i] = prepare_op(hb, Pb->ops[i], img_b, feat_b, 0, B, H, W, mode);
        OpLaunch &A = La[i], &Bp = Lb[i];
        if (A.kind == 2 && Bp.kind == 2 && A.family != 2 && Bp.family != 2 && (Bp.family != A.family || Bp.sk != A.sk)) {
            // the kernel choice reads per-handle options ("winograd", "latency_*"): the pair follows the first handle
            Bp.family = A.family; Bp.sk = A.sk;
            Bp.a.w = A.family == 1 ? Pb->ops[i].c->wino : (Pb->ops[i].fused ? Pb->ops[i].fused->f_w : Pb->ops[i].c->w);
            if (A.family == 1 && !Bp.a.w) return fail(ha, SPECMI_ERR_ARG, "op %zu: the second trunk has no Winograd filters", i);
        }
        const TrunkOp &oa = Pa.ops[i], &ob = Pb->ops[i];
        if (A.kind != Bp.kind || A.family != Bp.family || oa.H != ob.H || oa.W != ob.W || oa.OH != ob.OH || oa.OW != ob.OW ||
            (A.kind == 2 && (oa.c->cin != ob.c->cin || oa.c->cout != ob.c->cout || oa.c->k != ob.c->k || oa.c->stride != ob.c->stride ||
                             (oa.fused != nullptr) != (ob.fused != nullptr) || oa.relu != ob.relu)))
            return fail(ha, SPECMI_ERR_ARG, "op %zu (%s) differs between the two trunks: grouped launches need identical layer shapes",
                        i, oa.label.c_str());
    }
    // (the built-in launch profiler wants one record per layer: it sees the per-layer launches)
    const bool persist = mode != 0 && opt_i(ha, "persist", 0) && !ha->prof.on;   // opt-in: measured slower than the per-layer launches (profiles/r05_a_*)
    const size_t min_run = (size_t)opt_i(ha, "persist_min_run", 2);
    size_t max_run = (size_t)opt_i(ha, "persist_max_run", 64);     // 1: every layer its own walker launch (tiles walked, no in-launch waits)
    if (max_run < 1 || max_run > (size_t)kPersistMaxLayers) max_run = (size_t)kPersistMaxLayers;
    auto eligible = [&](size_t i) { return La[i].kind == 2 && La[i].family == 0 && !La[i].a.force_variant; };
    int rc;
    size_t i = first;
    while (i < n) {
        if (persist && eligible(i)) {
            size_t j = i;
            while (j < n && eligible(j) && j - i < max_run) ++j;
            if (j - i >= min_run) {
                rc = persist_run(ha, La, hb ? &Lb : nullptr, Pa.ops, i, j, feat_a, feat_b, s);
                if (rc == SPECMI_OK) { i = j; continue; }
                if (rc != -1) return rc;
            }
        }
        if ((rc = launch_op(ha, Pa.ops[i], La[i], hb ? &Lb[i] : nullptr, B, H, W, s))) return rc;
        ++i;
    }
    return SPECMI_OK;
}

// images NCHW -> layer4 map NHWC in *feat (a workspace buffer unless feat_out given).
// Option "trunk_subbatch" = S > 0: the stem, the max-pool and the first "trunk_subbatch_layers"
// ResNet stages are run S images at a time (activations of a slice are <= 103 MB at S = 32 and stay
// resident in the 256 MiB Infinity Cache between layers); later stages see the whole batch.
static int run_trunk(specmi_handle* h, const float* images, int B, int H, int W, float* feat_out, const float** feat,
                     int* fh, int* fw, hipStream_t s) {
    int rc;
    if (H < 32 || W < 32) return fail(h, SPECMI_ERR_ARG, "image size %dx%d too small", H, W);
    if ((rc = ensure_ws(h, B, H, W))) return rc;
    if (h->hrnet) {
        if ((rc = hrnet_forward(h, images, B, H, W, feat_out, fh, fw, s))) return rc;
        *feat = feat_out ? feat_out : hrnet_feat_ws(h);
        return SPECMI_OK;
    }
    TrunkPlan P;
    plan_trunk(h, H, W, feat_out != nullptr, P);
    const std::vector<TrunkOp>& ops = P.ops;
    const std::vector<int>& stage_of = P.stage_of;
    const int final_buf = P.final_buf, ch = P.fh, cw = P.fw;
    const int S = opt_i(h, "trunk_subbatch", 0);
    const int Lsplit = opt_i(h, "trunk_subbatch_layers", 2);
    const int mode = trunk_mode(h, B, H, W);
    size_t first_full = 0;
    if (S > 0 && S < B) {
        while (first_full < ops.size() && stage_of[first_full] <= Lsplit) ++first_full;
        for (int b0 = 0; b0 < B; b0 += S) {
            const int nb = (B - b0 < S) ? B - b0 : S;
            for (size_t i = 0; i < first_full; ++i)
                if ((rc = exec_op(h, ops[i], images, feat_out, b0, nb, H, W, s, mode))) return rc;
        }
    }
    if ((rc = launch_ops(h, nullptr, P, nullptr, images, nullptr, B, H, W, feat_out, nullptr, first_full, mode, s))) return rc;
    *feat = final_buf == -2 ? feat_out : h->act[final_buf];
    *fh = ch; *fw = cw;
    return SPECMI_OK;
}


// The two ResNet trunks of the path (CamCalib + SPEC: same depth, same input size) walked in lockstep, every layer of both
// as ONE grouped launch (SURVEY.md 7, step 7's alternative to two streams): half the launches, no side-stream join, and the
// last, partially filled round of workgroups of one network is filled by the other.  Each network keeps its own weights and
// activation buffers; an image's result is bit-identical to the separate launches (same kernels, same k order).
static int run_trunk_pair(specmi_handle* ha, specmi_handle* hb, const float* img_a, const float* img_b, int B, int H, int W,
                          float* feat_a, float* feat_b, hipStream_t s) {
    int rc;
    if (H < 32 || W < 32) return fail(ha, SPECMI_ERR_ARG, "image size %dx%d too small", H, W);
    if ((rc = ensure_ws(ha, B, H, W))) return rc;
    if ((rc = ensure_ws(hb, B, H, W))) { ha->err = hb->err; return rc; }
    TrunkPlan Pa, Pb;
    plan_trunk(ha, H, W, true, Pa);
    plan_trunk(hb, H, W, true, Pb);
    if (Pa.ops.size() != Pb.ops.size()) return fail(ha, SPECMI_ERR_ARG, "the two trunks have different depths");
    const int mode = trunk_mode(ha, B, H, W, true);   // the first handle's options decide for the pair
    return launch_ops(ha, hb, Pa, &Pb, img_a, img_b, B, H, W, feat_a, feat_b, 0, mode, s);
}

// per-image strides of the HMR outputs: dense, or all equal to option "output_ld" (the outputs are then columns of
// one caller-owned (B, output_ld) record, e.g. the packed all-gather record of SURVEY.md 8e)
struct OutLd {
    long pose = 216, shape = 10, cam = 3, p6d = 144, verts = 0, j3d = 147, j2d = 98, camt = 3;
};
static OutLd out_ld(specmi_handle* h) {
    OutLd o;
    const long ld = opt_i(h, "output_ld", 0);
    if (ld > 0) o.pose = o.shape = o.cam = o.p6d = o.verts = o.j3d = o.j2d = o.camt = ld;
    return o;
}

// defer != nullptr: head_final is not launched; *defer describes it for the SMPL pose kernel, which does its work (run_smpl)
static int run_head(specmi_handle* h, const float* feat, int B, int fh, int fw, const float* R, const float* K,
                    const float* img_h, float* pred_pose, float* pred_shape, float* pred_cam, float* pred_pose_6d,
                    const OutLd& old, hipStream_t s, HeadFinal* defer = nullptr, bool* pose_done = nullptr) {
    int rc;
    const int ucf = opt_i(h, "use_cam_feats", 0);
    const int F = h->feat_ch, LD = h->xc_ld;
    if (ucf && (!R || !K || !img_h))
        return fail(h, SPECMI_ERR_ARG, "use_cam_feats needs cam_rotmat, cam_intrinsics and img_h");
    if (pose_done) *pose_done = false;
    // Small batches (round 5, option "tail_fuse", default 0): avg-pool + state init -> composed regressor map -> pose chains as ONE
    // launch (head.hip: tail_gemv_kernel; the code of the three kernels, same bits).  Needs the collapsed head, the GEMV path,
    // head_final deferred into the pose chain and a map that pools in one part.  Opt-in: measured 15.2 us against 16.3 us for the
    // three kernels at batch 1 and the whole step 0-1 % SLOWER at batch 1-10 (profiles/r05_f_tail_check.jsonl) - an in-launch hop
    // costs what a kernel boundary costs on this part.
    if (defer && pose_done && !h->has_var && opt_i(h, "tail_fuse", 0) && use_latency_heads(h, B) && h->has_head_c && opt_i(h, "head_collapse", 1) &&
        h->head_c.w_rm && fh * fw < 64 && (B + 1) / 2 + 3 <= specmi_handle::kTailCtlWords && h->tail_ctl && !h->prof.on) {
        const HeadInit hi{h->xc, h->init_pose, h->init_shape, h->init_cam, R, K, img_h, ucf, F, LD};
        defer->state = h->h1; defer->ld_state = 1024;
        defer->pred_pose = pred_pose; defer->pred_shape = pred_shape; defer->pred_cam = pred_cam; defer->pred_pose_6d = pred_pose_6d;
        defer->ld_pose = old.pose; defer->ld_shape = old.shape; defer->ld_cam = old.cam; defer->ld_p6d = old.p6d;
        defer->rot_ws = h->rot_ws; defer->betas_ws = h->betas_ws; defer->cam_ws = h->cam_ws;
        const FcGemv hd{h->xc, h->head_c.w_rm, h->head_c.shift, nullptr, h->h1};
        LaunchCtx ctx{s, &h->prof, "head.tail"};
        const int lrc = launch_tail_hmr(hd, h->head_c.nout, h->head_c.Kp, LD, 1024, B, feat, h->xc, fh * fw, F, hi, h->tail_ctl,
                                        specmi_handle::kTailCtlWords, h->smpl, h->pf_ws, h->A_ws, h->pj_ws, *defer, ctx);
        if (lrc == 0) { *pose_done = true; return SPECMI_OK; }
        if (lrc != (int)hipErrorInvalidValue) LAUNCHCHK(h, lrc, "hmr tail");
        (void)hipGetLastError();
    }
    {
        // the IEF state columns are written by extra workgroups of the pooling launch (option "head_fuse" bit 0, default on; large
        // maps pool in parts and keep head_init as its own launch)
        const HeadInit hi{h->xc, h->init_pose, h->init_shape, h->init_cam, R, K, img_h, ucf, F, LD};
        bool init_done = false;
        {
            LaunchCtx ctx{s, &h->prof, "head.avgpool"};
            LAUNCHCHK(h, launch_avgpool(feat, h->xc, B, fh * fw, F, LD, ctx, (opt_i(h, "head_fuse", 3) & 1) ? &hi : nullptr, &init_done), "avgpool");
        }
        if (!init_done) {
            LaunchCtx ctx{s, &h->prof, "head.init"};
            LAUNCHCHK(h, launch_head_init(h->xc, h->init_pose, h->init_shape, h->init_cam, R, K, img_h, ucf, B, F, LD, ctx),
                      "head_init");
        }
    }
    const float* state = h->xc + F;
    long ld_state = LD;
    const bool gemv = use_latency_heads(h, B);
    auto fc = [&](const FcW& w, const float* x, int ldx, const float* res, float* out, int ldo, const char* label) -> int {
        if (gemv && w.w_rm) {     // latency plan: one wave per output column (head.hip: fc_gemv_kernel)
            const FcGemv hd{x, w.w_rm, w.shift, res, out};
            LaunchCtx ctx{s, &h->prof, label};
            LAUNCHCHK(h, launch_fc_gemv(&hd, 1, w.nout, w.Kp, ldx, ldo, B, ctx), label);
            return SPECMI_OK;
        }
        return run_fc(h, w, x, ldx, B, res, out, ldo, s, label);
    };
    if (h->has_head_c && opt_i(h, "head_collapse", 1)) {
        // the three IEF iterations as one composed affine map (commit_head_collapsed)
        if ((rc = fc(h->head_c, h->xc, LD, nullptr, h->h1, 1024, "head.ief_collapsed"))) return rc;
        state = h->h1;
        ld_state = 1024;
        h->last_var = h->has_var ? h->h1 + 157 : nullptr;      // (the composed map's 154 extra output columns)
        h->last_ld_var = 1024;
    } else {
        for (int it = 0; it < 3; ++it) {
            if ((rc = fc(h->fc1, h->xc, LD, nullptr, h->h1, 1024, "head.fc1"))) return rc;
            if ((rc = fc(h->fc2, h->h1, 1024, nullptr, h->h2, 1024, "head.fc2"))) return rc;
            // the variance decoders read the SAME xc as this iteration's mean decoders; only the last iteration's survive
            if (it == 2 && h->has_var && (rc = fc(h->head_var, h->h2, 1024, nullptr, h->h1, 1024, "head.dec_var"))) return rc;
            float* st = h->xc + F;  // dec* + running estimate, in place
            if ((rc = fc(h->dec, h->h2, 1024, st, st, LD, "head.dec"))) return rc;
        }
        h->last_var = h->has_var ? h->h1 : nullptr;
        h->last_ld_var = 1024;
    }
    h->last_state = state; h->last_ld_state = ld_state; h->last_B = B;
    if (defer) {
        defer->state = state; defer->ld_state = ld_state;
        defer->pred_pose = pred_pose; defer->pred_shape = pred_shape; defer->pred_cam = pred_cam; defer->pred_pose_6d = pred_pose_6d;
        defer->ld_pose = old.pose; defer->ld_shape = old.shape; defer->ld_cam = old.cam; defer->ld_p6d = old.p6d;
        defer->rot_ws = h->rot_ws; defer->betas_ws = h->betas_ws; defer->cam_ws = h->cam_ws;
    } else {
        LaunchCtx ctx{s, &h->prof, "head.final"};
        const long ld[4] = {old.pose, old.shape, old.cam, old.p6d};
        LAUNCHCHK(h, launch_head_final(state, ld_state, pred_pose, pred_shape, pred_cam, pred_pose_6d, ld, h->rot_ws,
                                       h->betas_ws, h->cam_ws, B, ctx),
                  "head_final");
    }
    return SPECMI_OK;
}

static int run_smpl(specmi_handle* h, const float* rotmat, const float* betas, const float* cam, int B, const float* R,
                    const float* K, const float* bbox_scale, const float* bbox_center, const float* img_w,
                    const float* img_h, float* vertices, float* joints3d, float* joints2d, float* cam_t,
                    const OutLd& old, hipStream_t s, const HeadFinal* final_ = nullptr, bool pose_done = false) {
    const int use_cam = opt_i(h, "use_cam", 0);
    if (use_cam && (!R || !K || !bbox_scale || !bbox_center || !img_w || !img_h))
        return fail(h, SPECMI_ERR_ARG, "use_cam needs cam_rotmat, cam_intrinsics, bbox_scale, bbox_center, img_w, img_h");
    SmplArgs a;
    a.rotmat = rotmat; a.betas = betas; a.cam = cam; a.cam_rotmat = R; a.cam_intrinsics = K;
    a.bbox_scale = bbox_scale; a.bbox_center = bbox_center; a.img_w = img_w; a.img_h = img_h;
    a.vertices = vertices ? vertices : h->verts_ws;
    a.ld_verts = vertices ? old.verts : 0;
    a.joints3d = joints3d; a.joints2d = joints2d; a.cam_t = cam_t;
    a.ld_j3d = old.j3d; a.ld_j2d = old.j2d; a.ld_camt = old.camt;
    a.pose_feat = h->pf_ws; a.A = h->A_ws; a.posed_j = h->pj_ws;
    a.B = B;
    a.mode = use_cam ? 0 : 1;
    a.focal_length = opt_f(h, "focal_length", 5000.f);
    a.img_res = (float)opt_i(h, "img_res", 224);
    a.normalize_joints2d = use_cam ? 0 : 1;  // spec/models/hmr.py:111 vs :119
    a.skin_split = opt_i(h, "smpl_skin_split", -1);
    a.final_ = final_;
    a.pose_done = pose_done;
    LaunchCtx ctx{s, &h->prof, "smpl"};
    LAUNCHCHK(h, launch_smpl(h->smpl, a, ctx), "smpl");
    return SPECMI_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* specmi_version(void) { return "specmi 0.6 (gfx950, fp32 MFMA)"; }

int specmi_create(specmi_handle** out, int device_id, int model_kind) {
    if (!out) return fail(nullptr, SPECMI_ERR_ARG, "out is NULL");
    if (model_kind != SPECMI_MODEL_CAMCALIB && model_kind != SPECMI_MODEL_HMR && model_kind != SPECMI_MODEL_SMPL)
        return fail(nullptr, SPECMI_ERR_ARG, "unknown model kind %d", model_kind);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, SPECMI_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, SPECMI_ERR_ARG, "device %d out of range [0,%d)", device_id, n);
    specmi_handle* h = new specmi_handle();
    h->device = device_id;
    h->kind = model_kind;
    build_resnet(h, 50);
    *out = h;
    return SPECMI_OK;
}

int specmi_destroy(specmi_handle* h) {
    if (!h) return SPECMI_OK;
    DeviceGuard guard;
    (void)guard.enter(h->device);
    (void)hipDeviceSynchronize();
    for (auto& r : h->prof.log) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    free_pool(h->ws_allocs);
    free_pool(h->param_allocs);
    if (h->sk.ws) (void)hipFree(h->sk.ws);
    if (h->sk.cnt) (void)hipFree(h->sk.cnt);
    free_pool(h->sk_retired);
    free_pool(h->ws_retired);
    for (auto& t : h->persist_tables) if (t.dev) (void)hipFree(t.dev);
    if (h->pctl) (void)hipFree(h->pctl);
    if (h->tail_ctl) (void)hipFree(h->tail_ctl);
    if (h->resize_tab) (void)hipFree(h->resize_tab);
    hrnet_free(h->hrnet);
    delete h;
    return SPECMI_OK;
}

const char* specmi_last_error(const specmi_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

static int stage(specmi_handle* h, const char* name, const void* data, const int64_t* shape, int ndim, bool is_int) {
    if (!h || !name || !data || (ndim > 0 && !shape) || ndim < 0 || ndim > 8) return fail(h, SPECMI_ERR_ARG, "bad argument to set_tensor");
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(h, SPECMI_ERR_ARG, "negative dimension in '%s'", name);
        n *= (size_t)shape[i];
    }
    HostTensor t;
    t.is_int = is_int;
    t.shape.assign(shape, shape + ndim);
    if (is_int) t.i.assign((const int32_t*)data, (const int32_t*)data + n);
    else t.f.assign((const float*)data, (const float*)data + n);
    h->staged[name] = std::move(t);
    return SPECMI_OK;
}

int specmi_set_tensor_f32(specmi_handle* h, const char* name, const float* d, const int64_t* shape, int ndim) {
    return stage(h, name, d, shape, ndim, false);
}
int specmi_set_tensor_i32(specmi_handle* h, const char* name, const int32_t* d, const int64_t* shape, int ndim) {
    return stage(h, name, d, shape, ndim, true);
}

int specmi_commit(specmi_handle* h) {
    if (!h) return SPECMI_ERR_ARG;
    DeviceGuard guard;
    HIPCHK(h, guard.enter(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    reset_sync_state(h, nullptr);
    HIPCHK(h, hipDeviceSynchronize());
    free_pool(h->param_allocs);
    h->committed = false;
    int rc;
    if (h->kind == SPECMI_MODEL_SMPL) {   // the body model alone (evaluation: ground-truth meshes / joints)
        if ((rc = commit_smpl(h))) return rc;
        free_pool(h->ws_allocs);
        h->act_elems = 0; h->ws_B = 0;
        h->committed = true;
        return SPECMI_OK;
    }
    const std::string bp = "backbone.";
    const int depth = opt_i(h, "backbone", 50);
    if (depth != 50 && depth != 34 && depth != 18 && depth != 101 && depth != 152 && depth != 32 && depth != 48)
        return fail(h, SPECMI_ERR_ARG, "backbone %d: resnet18 / 34 / 50 / 101 / 152, hrnet_w32 (32) and hrnet_w48 (48) are built", depth);
    if ((depth == 32 || depth == 48) && h->kind != SPECMI_MODEL_HMR)
        return fail(h, SPECMI_ERR_ARG, "the HRNet trunks belong to HMR (camcalib/model.py:33 builds resnet trunks only)");
    if (h->hrnet) { hrnet_free(h->hrnet); h->hrnet = nullptr; }
    if (depth == 32 || depth == 48) {
        h->blocks.clear();
        if ((rc = hrnet_commit(h, bp, depth, opt_i(h, "hrnet_use_conv", 1)))) return rc;
    } else {
        build_resnet(h, depth);
        if ((rc = commit_conv(h, bp, h->stem))) return rc;
        for (Bneck& b : h->blocks) {
            if ((rc = commit_conv(h, bp, b.c1))) return rc;
            if ((rc = commit_conv(h, bp, b.c2))) return rc;
            if (!b.basic && (rc = commit_conv(h, bp, b.c3))) return rc;
            if (b.has_ds && (rc = commit_conv(h, bp, b.ds))) return rc;
            if (!b.basic && b.has_ds && b.c3.cin % 32 == 0 && b.ds.cin % 32 == 0 && (rc = commit_fused_ds(h, bp, b))) return rc;
        }
    }
    if (h->kind == SPECMI_MODEL_CAMCALIB) {
        // camcalib/model.py:39-57: one Linear per angle, or Sequential(Linear x num_fc_layers) WITHOUT activations
        const char* names[3] = {"fc_vfov", "fc_pitch", "fc_roll"};
        const int L = opt_i(h, "num_fc_layers", 1), hc = opt_i(h, "num_fc_channels", 1024);
        if (L < 1 || L > 3 || hc < 1 || hc > 1024) return fail(h, SPECMI_ERR_ARG, "num_fc_layers in 1..3 and num_fc_channels <= 1024 are built");
        h->fc_layers = L;
        const std::string lastw = L == 1 ? std::string("fc_vfov.weight") : "fc_vfov." + std::to_string(L - 1) + ".weight";
        const HostTensor* w0 = find(h, lastw);
        if (!w0) return fail(h, SPECMI_ERR_MISSING, "missing tensor '%s'", lastw.c_str());
        const int last_in = L == 1 ? h->feat_ch : hc;
        const int nb = (int)(w0->numel() / last_in);
        h->opt_i["nbins"] = nb;
        for (int i = 0; i < 3; ++i)
            for (int l = 0; l < L; ++l) {
                const std::string nm = L == 1 ? std::string(names[i]) : std::string(names[i]) + "." + std::to_string(l);
                const int nin = l == 0 ? h->feat_ch : hc, nout = l == L - 1 ? nb : hc;
                if ((rc = commit_fc(h, {nm}, {nout}, nin, h->fc_cam[i][l]))) return rc;
            }
    } else {
        const int ucf = opt_i(h, "use_cam_feats", 0);
        const int nin = h->feat_ch + 144 + 13 + (ucf ? 7 : 0);
        h->xc_ld = round_up(h->feat_ch + 164, 32);
        h->has_head_c = false;
        if ((rc = commit_fc(h, {"head.fc1"}, {1024}, nin, h->fc1))) return rc;
        if ((rc = commit_fc(h, {"head.fc2"}, {1024}, 1024, h->fc2))) return rc;
        if ((rc = commit_fc(h, {"head.decpose", "head.decshape", "head.deccam"}, {144, 10, 3}, 1024, h->dec))) return rc;
        h->has_var = opt_i(h, "estimate_var", 0) != 0;
        if (h->has_var && (rc = commit_fc(h, {"head.decpose_var", "head.decshape_var"}, {144, 10}, 1024, h->head_var))) return rc;
        h->last_state = h->last_var = nullptr; h->last_B = 0;
        const HostTensor *ip, *is, *ic;
        if ((rc = need(h, "head.init_pose", {144}, false, &ip))) return rc;
        if ((rc = need(h, "head.init_shape", {10}, false, &is))) return rc;
        if ((rc = need(h, "head.init_cam", {3}, false, &ic))) return rc;
        if ((rc = dev_upload(h, ip->f.data(), 144 * 4, (void**)&h->init_pose, h->param_allocs))) return rc;
        if ((rc = dev_upload(h, is->f.data(), 10 * 4, (void**)&h->init_shape, h->param_allocs))) return rc;
        if ((rc = dev_upload(h, ic->f.data(), 3 * 4, (void**)&h->init_cam, h->param_allocs))) return rc;
        if (opt_i(h, "head_collapse", 1) && (rc = commit_head_collapsed(h, h->feat_ch, ucf))) return rc;
        if ((rc = commit_smpl(h))) return rc;
        // the SMPL vertex count sizes the workspace
        free_pool(h->ws_allocs);
        h->act_elems = 0; h->ws_B = 0;
    }
    h->committed = true;
    return SPECMI_OK;
}

#define ENTER(h)                                                                   \
    if (!(h)) return SPECMI_ERR_ARG;                                               \
    DeviceGuard dev_guard__;                                                       \
    HIPCHK(h, dev_guard__.enter((h)->device));

#define NEED_COMMIT(h) \
    if (!(h)->committed) return fail(h, SPECMI_ERR_STATE, "specmi_commit has not succeeded on this handle");

int specmi_trunk_forward(specmi_handle* h, const float* images, int B, int H, int W, float* feat, void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (!images || !feat || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    const float* f; int fh, fw;
    const int rc = run_trunk(h, images, B, H, W, feat, &f, &fh, &fw, (hipStream_t)stream);
    if (rc) reset_sync_state(h, (hipStream_t)stream);
    return rc;
}

int specmi_trunk_forward_pair(specmi_handle* ha, specmi_handle* hb, const float* images_a, const float* images_b, int B, int H,
                              int W, float* feat_a, float* feat_b, void* stream) {
    ENTER(ha); NEED_COMMIT(ha);
    if (!hb) return fail(ha, SPECMI_ERR_ARG, "second handle is NULL");
    if (!hb->committed) return fail(ha, SPECMI_ERR_STATE, "second handle not committed");
    if (hb->device != ha->device) return fail(ha, SPECMI_ERR_ARG, "the two handles live on different devices");
    if (!images_a || !images_b || !feat_a || !feat_b || B <= 0) return fail(ha, SPECMI_ERR_ARG, "bad argument");
    if (ha->hrnet || hb->hrnet || ha->blocks.size() != hb->blocks.size() || ha->blocks.empty() ||
        ha->blocks[0].basic != hb->blocks[0].basic)
        return fail(ha, SPECMI_ERR_ARG, "grouped trunk launches need two ResNet trunks of the same depth");
    const int rc = run_trunk_pair(ha, hb, images_a, images_b, B, H, W, feat_a, feat_b, (hipStream_t)stream);
    if (rc) reset_sync_state(ha, (hipStream_t)stream);
    return rc;
}

static int run_camcalib_head(specmi_handle* h, const float* f, int B, int fh, int fw, float* lv, float* lp, float* lr, hipStream_t s);

int specmi_camcalib_head_forward(specmi_handle* h, const float* feat, int B, int fh, int fw, float* lv, float* lp, float* lr,
                                 void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_CAMCALIB) return fail(h, SPECMI_ERR_STATE, "handle is not a CamCalib model");
    if (!feat || !lv || !lp || !lr || B <= 0 || fh <= 0 || fw <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    int rc;
    if ((rc = ensure_ws(h, B, 32, 32))) return rc;
    return run_camcalib_head(h, feat, B, fh, fw, lv, lp, lr, (hipStream_t)stream);
}

int specmi_camcalib_forward(specmi_handle* h, const float* images, int B, int H, int W, float* lv, float* lp, float* lr,
                            void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_CAMCALIB) return fail(h, SPECMI_ERR_STATE, "handle is not a CamCalib model");
    if (!images || !lv || !lp || !lr || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const float* f; int fh, fw, rc;
    if ((rc = run_trunk(h, images, B, H, W, nullptr, &f, &fh, &fw, s)) || (rc = run_camcalib_head(h, f, B, fh, fw, lv, lp, lr, s)))
        reset_sync_state(h, s);
    return rc;
}

static int run_camcalib_head(specmi_handle* h, const float* f, int B, int fh, int fw, float* lv, float* lp, float* lr, hipStream_t s) {
    int rc;
    {
        LaunchCtx ctx{s, &h->prof, "avgpool"};
        LAUNCHCHK(h, launch_avgpool(f, h->xf, B, fh * fw, h->feat_ch, 2048, ctx), "avgpool");
    }
    float* outs[3] = {lv, lp, lr};
    const char* labels[3] = {"fc_vfov", "fc_pitch", "fc_roll"};
    if (use_latency_heads(h, B) && h->fc_cam[0][0].w_rm) {
        // latency plan: the three heads of a layer as ONE launch (head.hip: fc_gemv_kernel); Linear chains layer by layer
        const float* x[3] = {h->xf, h->xf, h->xf};
        int ldx = 2048;
        for (int l = 0; l < h->fc_layers; ++l) {
            const bool lastl = (l + 1 == h->fc_layers);
            FcGemv hd[3];
            // hidden rows of the three heads of layer l: fc_hidden[l & 1] holds them (3 x B x 1024), alternating with the layer that reads them
            const FcW& f0 = h->fc_cam[0][l];
            const int ldy = lastl ? f0.nout : 1024;
            for (int i = 0; i < 3; ++i) {
                const FcW& fc = h->fc_cam[i][l];
                if (fc.nout != f0.nout || fc.Kp != f0.Kp) return fail(h, SPECMI_ERR_STATE, "CamCalib heads differ in shape");
                hd[i] = FcGemv{x[i], fc.w_rm, fc.shift, nullptr, lastl ? outs[i] : h->fc_hidden[l & 1] + (size_t)i * B * 1024};
            }
            LaunchCtx ctx{s, &h->prof, "fc_vfov|pitch|roll"};
            LAUNCHCHK(h, launch_fc_gemv(hd, 3, f0.nout, f0.Kp, ldx, ldy, B, ctx), "fc heads");
            for (int i = 0; i < 3; ++i) x[i] = hd[i].out;
            ldx = ldy;
        }
        return SPECMI_OK;
    }
    for (int i = 0; i < 3; ++i) {
        // Linear chain (no activation in between, camcalib/model.py:59-70); hidden rows live in h1 / h2 (1024 wide)
        const float* x = h->xf;
        int ldx = 2048;
        for (int l = 0; l < h->fc_layers; ++l) {
            const FcW& fc = h->fc_cam[i][l];
            const bool lastl = (l + 1 == h->fc_layers);
            float* y = lastl ? outs[i] : (l == 0 ? h->h1 : h->h2);
            const int ldy = lastl ? fc.nout : 1024;
            if ((rc = run_fc(h, fc, x, ldx, B, nullptr, y, ldy, s, labels[i]))) return rc;
            x = y; ldx = ldy;
        }
    }
    return SPECMI_OK;
}

int specmi_camcalib_head_decode(specmi_handle* h, const float* feat, int B, int fh, int fw, float* lv, float* lp, float* lr,
                                const float* img_h, const float* img_w, float* vfov, float* pitch, float* roll, float* f_pix,
                                float* R, float* K, void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_CAMCALIB) return fail(h, SPECMI_ERR_STATE, "handle is not a CamCalib model");
    if (!feat || !lv || !lp || !lr || B <= 0 || fh <= 0 || fw <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = ensure_ws(h, B, 32, 32))) return rc;
    const long ld_ang = opt_i(h, "angle_ld", 0) > 0 ? opt_i(h, "angle_ld", 0) : 1;
    const FcW& f0 = h->fc_cam[0][0];
    // small batches (round 5, option "tail_fuse", opt-in): avg-pool -> the three heads -> decode as ONE launch (head.hip: tail_gemv_kernel)
    if (opt_i(h, "tail_fuse", 0) && use_latency_heads(h, B) && h->fc_layers == 1 && f0.w_rm && fh * fw < 64 && f0.Kp == h->feat_ch &&
        h->fc_cam[1][0].nout == f0.nout && h->fc_cam[2][0].nout == f0.nout && h->fc_cam[1][0].Kp == f0.Kp && h->fc_cam[2][0].Kp == f0.Kp &&
        (B + 1) / 2 + 3 <= specmi_handle::kTailCtlWords && h->tail_ctl && !h->prof.on) {
        float* outs[3] = {lv, lp, lr};
        FcGemv hd[3];
        for (int i = 0; i < 3; ++i) hd[i] = FcGemv{h->xf, h->fc_cam[i][0].w_rm, h->fc_cam[i][0].shift, nullptr, outs[i]};
        LaunchCtx ctx{s, &h->prof, "camcalib.tail"};
        const int lrc = launch_tail_camcalib(hd, f0.nout, f0.Kp, B, feat, h->xf, fh * fw, h->feat_ch, h->tail_ctl, specmi_handle::kTailCtlWords,
                                             img_h, img_w, vfov, pitch, roll, f_pix, R, K, ld_ang, ctx);
        if (lrc == 0) return SPECMI_OK;
        if (lrc != (int)hipErrorInvalidValue) LAUNCHCHK(h, lrc, "camcalib tail");
        (void)hipGetLastError();
    }
    if ((rc = run_camcalib_head(h, feat, B, fh, fw, lv, lp, lr, s))) { reset_sync_state(h, s); return rc; }
    LaunchCtx ctx{s, &h->prof, "camcalib.decode"};
    // the bins are the LAST Linear of a head's chain (num_fc_layers > 1: the first one is num_fc_channels wide, camcalib/model.py:59-70)
    const int nbins = h->fc_cam[0][h->fc_layers - 1].nout;
    LAUNCHCHK(h, launch_camcalib_decode(lv, lp, lr, B, nbins, img_h, img_w, vfov, pitch, roll, f_pix, R, K, ld_ang, ctx), "camcalib_decode");
    return SPECMI_OK;
}

int specmi_camcalib_decode(specmi_handle* h, const float* lv, const float* lp, const float* lr, int B, int nbins,
                           const float* img_h, const float* img_w, float* vfov, float* pitch, float* roll, float* f_pix,
                           float* R, float* K, void* stream) {
    ENTER(h);
    if (!lv || !lp || !lr || B <= 0 || nbins < 2) return fail(h, SPECMI_ERR_ARG, "bad argument");
    if ((f_pix || K) && !img_h) return fail(h, SPECMI_ERR_ARG, "f_pix / K need img_h");
    if (K && !img_w) return fail(h, SPECMI_ERR_ARG, "K needs img_w");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "camcalib.decode"};
    LAUNCHCHK(h, launch_camcalib_decode(lv, lp, lr, B, nbins, img_h, img_w, vfov, pitch, roll, f_pix, R, K,
                                        (long)(opt_i(h, "angle_ld", 0) > 0 ? opt_i(h, "angle_ld", 0) : 1), ctx), "decode");
    return SPECMI_OK;
}

int specmi_camcalib_bins(specmi_handle* h, const float* logits, int rows, int nbins, int32_t* argmax_idx,
                         float* soft_idx, void* stream) {
    ENTER(h);
    if (!logits || rows <= 0 || nbins < 2 || (!argmax_idx && !soft_idx)) return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "camcalib.bins"};
    LAUNCHCHK(h, launch_bins_reduce(logits, rows, nbins, argmax_idx, soft_idx, ctx), "bins_reduce");
    return SPECMI_OK;
}

int specmi_cam_params(specmi_handle* h, const float* pitch, const float* roll, const float* f_pix, const float* img_w,
                      const float* img_h, int B, float* R, float* K, void* stream) {
    ENTER(h);
    if (!pitch || !roll || !f_pix || !img_w || !img_h || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "camcalib.cam_params"};
    LAUNCHCHK(h, launch_cam_params(pitch, roll, f_pix, img_w, img_h, B, R, K, ctx), "cam_params");
    return SPECMI_OK;
}

int specmi_hmr_head_forward(specmi_handle* h, const float* feat, int B, int fh, int fw, const float* R, const float* K,
                            const float* img_h, float* pred_pose, float* pred_shape, float* pred_cam,
                            float* pred_pose_6d, void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_HMR) return fail(h, SPECMI_ERR_STATE, "handle is not an HMR model");
    if (!feat || B <= 0 || fh <= 0 || fw <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    int rc;
    if ((rc = ensure_ws(h, B, 32, 32))) return rc;
    return run_head(h, feat, B, fh, fw, R, K, img_h, pred_pose, pred_shape, pred_cam, pred_pose_6d, out_ld(h),
                    (hipStream_t)stream);
}

int specmi_smpl_forward(specmi_handle* h, const float* rotmat, const float* betas, const float* cam, int B,
                        const float* R, const float* K, const float* bbox_scale, const float* bbox_center,
                        const float* img_w, const float* img_h, float* vertices, float* joints3d, float* joints2d,
                        float* cam_t, void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_HMR && h->kind != SPECMI_MODEL_SMPL) return fail(h, SPECMI_ERR_STATE, "handle has no body model");
    if (!rotmat || !betas || !cam || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    int rc;
    if ((rc = ensure_ws(h, B, 32, 32))) return rc;
    return run_smpl(h, rotmat, betas, cam, B, R, K, bbox_scale, bbox_center, img_w, img_h, vertices, joints3d, joints2d,
                    cam_t, out_ld(h), (hipStream_t)stream);
}

int specmi_smpl_native(specmi_handle* h, const float* pose, int pose_is_axis_angle, const float* betas, int B,
                       float* vertices, float* joints24, void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_HMR && h->kind != SPECMI_MODEL_SMPL) return fail(h, SPECMI_ERR_STATE, "handle has no body model");
    if (!pose || !betas || B <= 0 || (!vertices && !joints24)) return fail(h, SPECMI_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = ensure_ws(h, B, 32, 32))) return rc;
    const float* rot = pose;
    if (pose_is_axis_angle) {
        LaunchCtx ctx{s, &h->prof, "smpl.rodrigues"};
        LAUNCHCHK(h, launch_rodrigues(pose, h->rot_ws, B * 24, ctx), "rodrigues");
        rot = h->rot_ws;
    }
    SmplArgs a;
    a.rotmat = rot; a.betas = betas; a.cam = nullptr; a.cam_rotmat = nullptr; a.cam_intrinsics = nullptr;
    a.bbox_scale = a.bbox_center = a.img_w = a.img_h = nullptr;
    a.vertices = vertices; a.joints3d = a.joints2d = a.cam_t = nullptr;
    a.pose_feat = h->pf_ws; a.A = h->A_ws; a.posed_j = h->pj_ws;
    a.B = B; a.mode = 1; a.focal_length = 0.f; a.img_res = 0.f; a.normalize_joints2d = 0;
    a.skin_split = opt_i(h, "smpl_skin_split", -1);
    LaunchCtx ctx{s, &h->prof, "smpl.native"};
    LAUNCHCHK(h, launch_smpl_native(h->smpl, a, joints24, ctx), "smpl_native");
    return SPECMI_OK;
}

int specmi_hmr_forward(specmi_handle* h, const float* images, int B, int H, int W, const float* R, const float* K,
                       const float* bbox_scale, const float* bbox_center, const float* img_w, const float* img_h,
                       const specmi_hmr_outputs* out, void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_HMR) return fail(h, SPECMI_ERR_STATE, "handle is not an HMR model");
    if (!images || !out || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const float* f; int fh, fw, rc;
    if ((rc = run_trunk(h, images, B, H, W, nullptr, &f, &fh, &fw, s))) { reset_sync_state(h, s); return rc; }
    const OutLd old = out_ld(h);
    HeadFinal fin;
    const bool fuse = (opt_i(h, "head_fuse", 3) & 2) != 0;     // head_final's work inside the SMPL pose kernel (same bits, one node less)
    bool pose_done = false;
    if ((rc = run_head(h, f, B, fh, fw, R, K, img_h, out->pred_pose, out->pred_shape, out->pred_cam, out->pred_pose_6d, old, s,
                       fuse ? &fin : nullptr, fuse ? &pose_done : nullptr)) ||
        (rc = run_smpl(h, h->rot_ws, h->betas_ws, h->cam_ws, B, R, K, bbox_scale, bbox_center, img_w, img_h,
                       out->smpl_vertices, out->smpl_joints3d, out->smpl_joints2d, out->pred_cam_t, old, s, fuse ? &fin : nullptr, pose_done)))
        reset_sync_state(h, s);      // include/specmi.h: the hand-off counters are reset after any forward that returned an error
    return rc;
}

int specmi_hmr_regress(specmi_handle* h, const float* feat, int B, int fh, int fw, const float* R, const float* K,
                       const float* bbox_scale, const float* bbox_center, const float* img_w, const float* img_h,
                       const specmi_hmr_outputs* out, void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_HMR) return fail(h, SPECMI_ERR_STATE, "handle is not an HMR model");
    if (!feat || !out || B <= 0 || fh <= 0 || fw <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = ensure_ws(h, B, 32, 32))) return rc;
    const OutLd old = out_ld(h);
    HeadFinal fin;
    const bool fuse = (opt_i(h, "head_fuse", 3) & 2) != 0;     // head_final's work inside the SMPL pose kernel (same bits, one node less)
    bool pose_done = false;
    if ((rc = run_head(h, feat, B, fh, fw, R, K, img_h, out->pred_pose, out->pred_shape, out->pred_cam, out->pred_pose_6d, old, s,
                       fuse ? &fin : nullptr, fuse ? &pose_done : nullptr)) ||
        (rc = run_smpl(h, h->rot_ws, h->betas_ws, h->cam_ws, B, R, K, bbox_scale, bbox_center, img_w, img_h,
                       out->smpl_vertices, out->smpl_joints3d, out->smpl_joints2d, out->pred_cam_t, old, s, fuse ? &fin : nullptr, pose_done)))
        reset_sync_state(h, s);
    return rc;
}

int specmi_hmr_uncertainty(specmi_handle* h, int B, float* pred_pose_var, float* pred_shape_var, void* stream) {
    ENTER(h); NEED_COMMIT(h);
    if (h->kind != SPECMI_MODEL_HMR) return fail(h, SPECMI_ERR_STATE, "handle is not an HMR model");
    if (!h->has_var) return fail(h, SPECMI_ERR_STATE, "the model was committed without option \"estimate_var\"");
    if (!pred_pose_var || !pred_shape_var || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    if (!h->last_state || !h->last_var || h->last_B != B)
        return fail(h, SPECMI_ERR_STATE, "specmi_hmr_uncertainty follows a head forward of the same batch (%d) on the same stream; the last one had %d",
                    B, h->last_B);
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "head.var"};
    LAUNCHCHK(h, launch_head_var(h->last_state, h->last_ld_state, h->last_var, h->last_ld_var, opt_i(h, "uncertainty_activation", 0), pred_pose_var,
                                 pred_shape_var, B, ctx), "head_var");
    return SPECMI_OK;
}

int specmi_conv2d(specmi_handle* h, const float* x, int B, int H, int W, int Cin, const float* w_host,
                  const float* scale_host, const float* shift_host, int Cout, int KH, int KW, int stride, int pad,
                  const float* residual, int relu, float* out, void* stream) {
    ENTER(h);
    if (!x || !w_host || !scale_host || !shift_host || !out || B <= 0 || KH != KW)
        return fail(h, SPECMI_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<void*> tmp;
    std::vector<float> packed, sc, sh;
    const bool stem = (Cin == 3 && KH == 7 && Cout == 64 && stride == 2 && pad == 3);
    int rc = SPECMI_OK;
    float *dw = nullptr, *dsc = nullptr, *dsh = nullptr;
    const int Npad = round_up(Cout, 64);
    sc.assign(Npad, 0.f); sh.assign(Npad, 0.f);
    std::memcpy(sc.data(), scale_host, (size_t)Cout * 4);
    std::memcpy(sh.data(), shift_host, (size_t)Cout * 4);
    // option "winograd": 1 (default) = F(2x2,3x3) where the shape allows it, 0 = always the direct implicit GEMM
    const bool wino = !stem && opt_i(h, "winograd", 1) && KH == 3 && stride == 1 && pad == 1 && Cin % 16 == 0 &&
                      (Cout % 64 == 0 || (Cout % 32 == 0 && Cout > 64));
    const int terms = opt_i(h, "conv_precision", 0);
    const bool split = !stem && (terms == 3 || terms == 6) && Cin % 16 == 0 && Cout % 4 == 0;
    std::vector<unsigned short> pieces;
    void* dsplit = nullptr;
    if (split) {
        pack_bf16_split_weights_oihw(w_host, Cout, Cin, KH, KW, Npad, pieces);
        if ((rc = dev_upload(h, pieces.data(), pieces.size() * 2, &dsplit, tmp))) { free_pool(tmp); return rc; }
    }
    if (stem) pack_stem_weights(w_host, packed);
    else if (wino) pack_wino_weights(w_host, Cout, Cin, packed);
    else {
        if (Cin % 32) { free_pool(tmp); return fail(h, SPECMI_ERR_ARG, "Cin must be a multiple of 32 (got %d)", Cin); }
        pack_gemm_weights(w_host, Cout, Cin, KH, KW, Cin * KH * KW, Npad, packed);
    }
    if ((rc = dev_upload(h, packed.data(), packed.size() * 4, (void**)&dw, tmp)) ||
        (rc = dev_upload(h, sc.data(), sc.size() * 4, (void**)&dsc, tmp)) ||
        (rc = dev_upload(h, sh.data(), sh.size() * 4, (void**)&dsh, tmp))) {
        free_pool(tmp);
        return rc;
    }
    const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
    LaunchCtx ctx{s, &h->prof, "conv2d"};
    int lrc;
    if (stem) {
        lrc = launch_stem(x, dw, dsc, dsh, out, B, H, W, OH, OW, relu, ctx);
    } else {
        ConvArgs a;
        a.x = x; a.w = dw; a.scale = dsc; a.shift = dsh; a.res = residual; a.out = out;
        a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.ldx = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.Npad = Npad;
        a.ldo = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.relu = relu;
        a.force_variant = opt_i(h, "force_conv_variant", 0);
        a.wino_variant = opt_i(h, "force_wino_variant", 0);
        // option "conv2d_sk" (tests): > 1 = that many leaves, -1 = the latency plan's own rule, 0 = the throughput kernel;
        // "latency_force_unit": 0 = by batch, 1 / 2 / 3 = a leaf / a group / the whole K per workgroup
        int S = wino ? 0 : opt_i(h, "conv2d_sk", 0);
        if (split && conv_bf16s_supported(a)) lrc = launch_conv_bf16s(a, dsplit, terms, ctx);
        else if (S != 0) {
            SkPlan pl = conv_igemm_sk_plan(a, 1, opt_i(h, "latency_target_wgs", 256), opt_i(h, "latency_min_chunks", 4),
                                           sk_fill(h));
            if (S > 0) {
                pl.leaves = S; pl.G = 1;
                for (int g = 2; g <= 4; ++g)
                    if (S % g == 0) pl.G = g;
                pl.unit = 1;
            }
            const int fu = opt_i(h, "latency_force_unit", 0);
            if (fu) pl.unit = fu == 1 ? 1 : (fu == 2 ? pl.G : pl.leaves);
            // "conv2d_wsplit" (tests): 2 / 3 = the wave-split unit of the same tree, one group / all groups per workgroup
            const int ws = opt_i(h, "conv2d_wsplit", 0);
            SkPlan pw = pl;
            pw.unit = ws == 3 ? pl.leaves : pl.G;
            if (ws >= 2 && conv_wsplit_supported(a, pw)) {
                if ((rc = ensure_sk(h, conv_wsplit_ws_floats(a, pw.leaves / pw.unit, 1), conv_wsplit_tiles(a, 1)))) { free_pool(tmp); return rc; }
                lrc = launch_conv_wsplit(a, pw, h->sk, ctx);
            } else {
                if ((rc = ensure_sk(h, conv_igemm_sk_ws_floats(a, pl.leaves / pl.unit, 1), conv_igemm_sk_tiles(a, 1)))) { free_pool(tmp); return rc; }
                lrc = pl.leaves > 1 ? launch_conv_igemm_sk(a, pl, h->sk, ctx) : launch_conv_igemm(a, ctx);
            }
        } else lrc = wino ? launch_conv_wino(a, ctx) : launch_conv_igemm(a, ctx);
    }
    hipError_t se = hipStreamSynchronize(s);
    free_pool(tmp);
    if (lrc) return fail(h, SPECMI_ERR_HIP, "conv2d launch failed: %s", hipGetErrorString((hipError_t)lrc));
    if (se != hipSuccess) return fail(h, SPECMI_ERR_HIP, "conv2d failed: %s", hipGetErrorString(se));
    return SPECMI_OK;
}

int specmi_maxpool3x3s2(specmi_handle* h, const float* x, int B, int H, int W, int C, float* out, void* stream) {
    ENTER(h);
    if (!x || !out || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "maxpool"};
    LAUNCHCHK(h, launch_maxpool3x3s2(x, out, B, H, W, C, conv_out(H, 3, 2, 1), conv_out(W, 3, 2, 1), ctx), "maxpool");
    return SPECMI_OK;
}

int specmi_avgpool(specmi_handle* h, const float* x, int B, int HW, int C, float* out, void* stream) {
    ENTER(h);
    if (!x || !out || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "avgpool"};
    LAUNCHCHK(h, launch_avgpool(x, out, B, HW, C, C, ctx), "avgpool");
    return SPECMI_OK;
}

int specmi_resize_normalize(specmi_handle* h, const uint8_t* frame, int H, int W, int OH, int OW, float* out, uint8_t* raw,
                            void* stream) {
    ENTER(h);
    if (!frame || !out || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    // tables: [hb (2 OW) | hk (OW ksh) | vb (2 OH) | vk (OH ksv)], rebuilt only when the geometry changes
    std::vector<int> hb, hk, vb, vk;
    const int ksh = pillow_coeffs(W, OW, hb, hk), ksv = pillow_coeffs(H, OH, vb, vk);
    const size_t o_hk = hb.size(), o_vb = o_hk + hk.size(), o_vk = o_vb + vb.size(), total = o_vk + vk.size();
    if (h->resize_geom[0] != H || h->resize_geom[1] != W || h->resize_geom[2] != OH || h->resize_geom[3] != OW || !h->resize_tab) {
        // a resize enqueued earlier on ANY stream may still read the old tables (the handle is shared by callers on
        // different streams): wait for the whole device before freeing / overwriting them
        HIPCHK(h, hipDeviceSynchronize());
        if (total > h->resize_tab_ints) {
            if (h->resize_tab) (void)hipFree(h->resize_tab);
            h->resize_tab = nullptr;
            HIPCHK(h, hipMalloc((void**)&h->resize_tab, total * 4));
            h->resize_tab_ints = total;
        }
        h->resize_host.resize(total);
        std::memcpy(h->resize_host.data(), hb.data(), hb.size() * 4);
        std::memcpy(h->resize_host.data() + o_hk, hk.data(), hk.size() * 4);
        std::memcpy(h->resize_host.data() + o_vb, vb.data(), vb.size() * 4);
        std::memcpy(h->resize_host.data() + o_vk, vk.data(), vk.size() * 4);
        HIPCHK(h, hipMemcpyAsync(h->resize_tab, h->resize_host.data(), total * 4, hipMemcpyHostToDevice, s));
        h->resize_geom[0] = H; h->resize_geom[1] = W; h->resize_geom[2] = OH; h->resize_geom[3] = OW;
    }
    LaunchCtx ctx{s, &h->prof, "resize_normalize"};
    LAUNCHCHK(h, launch_resize_normalize(frame, H, W, OH, OW, h->resize_tab, h->resize_tab + o_hk, ksh, h->resize_tab + o_vb,
                                         h->resize_tab + o_vk, ksv, out, raw, ctx), "resize_normalize");
    return SPECMI_OK;
}

int specmi_crop_normalize(specmi_handle* h, const uint8_t* frame, int H, int W, const float* bboxes, int n, float scale,
                          int crop_size, float* out, uint8_t* raw, float* bbox_scale, float* bbox_center, void* stream) {
    ENTER(h);
    if (!frame || !bboxes || !out || H <= 0 || W <= 0 || n <= 0 || crop_size <= 0 || !(scale > 0.f))
        return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "preprocess.crop"};
    LAUNCHCHK(h, launch_crop_normalize(frame, H, W, bboxes, n, scale, crop_size, out, raw, bbox_scale, bbox_center, ctx),
              "crop_normalize");
    return SPECMI_OK;
}

int specmi_crop_normalize_batch(specmi_handle* h, const uint8_t* frames, int nframes, int H, int W, const int32_t* frame_index,
                                const float* bboxes, int n, float scale, int crop_size, float* out, uint8_t* raw,
                                float* bbox_scale, float* bbox_center, void* stream) {
    ENTER(h);
    if (!frames || !frame_index || !bboxes || !out || nframes <= 0 || H <= 0 || W <= 0 || n <= 0 || crop_size <= 0 || !(scale > 0.f))
        return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "preprocess.crop_batch"};
    LAUNCHCHK(h, launch_crop_normalize(frames, H, W, bboxes, n, scale, crop_size, out, raw, bbox_scale, bbox_center, ctx,
                                       frame_index, nframes),
              "crop_normalize_batch");
    return SPECMI_OK;
}

int specmi_crop_resize_normalize(specmi_handle* h, const uint8_t* frame, int H, int W, const int32_t* boxes, int n,
                                 int crop_size, float* out, void* stream) {
    ENTER(h);
    if (!frame || !boxes || !out || H <= 0 || W <= 0 || n <= 0 || crop_size <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "preprocess.dataset_crop"};
    LAUNCHCHK(h, launch_crop_resize_normalize(frame, H, W, boxes, n, crop_size, out, ctx), "crop_resize_normalize");
    return SPECMI_OK;
}

int specmi_eval_mesh(specmi_handle* h, const float* pred, const float* gt, int B, int V, const float* Jr, int J,
                     const int32_t* sel, int nsel, float* mpjpe, float* pampjpe, float* v2v, void* stream) {
    ENTER(h);
    if (!pred || !gt || !Jr || B <= 0 || V <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    if (J < 1 || J > 32 || nsel < 1 || nsel > 32) return fail(h, SPECMI_ERR_ARG, "J and nsel must be in [1,32]");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "eval.mesh"};
    LAUNCHCHK(h, launch_eval_mesh(pred, gt, B, V, Jr, J, sel, nsel, mpjpe, pampjpe, v2v, ctx), "eval_mesh");
    return SPECMI_OK;
}

int specmi_eval_joints(specmi_handle* h, const float* pred, const float* gt, int B, int J, float* mpjpe, float* pampjpe,
                       void* stream) {
    ENTER(h);
    if (!pred || !gt || B <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    if (J < 1 || J > 32) return fail(h, SPECMI_ERR_ARG, "J must be in [1,32]");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "eval.joints"};
    LAUNCHCHK(h, launch_eval_joints(pred, gt, B, J, mpjpe, pampjpe, ctx), "eval_joints");
    return SPECMI_OK;
}

int specmi_regress_joints(specmi_handle* h, const float* vertices, int B, int V, const float* Jr, int J, float* joints,
                          void* stream) {
    ENTER(h);
    if (!vertices || !Jr || !joints || B <= 0 || V <= 0 || J <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "eval.regress_joints"};
    LAUNCHCHK(h, launch_regress_joints(vertices, B, V, Jr, J, joints, ctx), "regress_joints");
    return SPECMI_OK;
}

int specmi_rotate_points(specmi_handle* h, const float* R, const float* points, int B, int N, float* out, void* stream) {
    ENTER(h);
    if (!R || !points || !out || B <= 0 || N <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    LaunchCtx ctx{(hipStream_t)stream, &h->prof, "eval.rotate_points"};
    LAUNCHCHK(h, launch_rotate_points(R, points, B, N, out, ctx), "rotate_points");
    return SPECMI_OK;
}

int specmi_trunk_plan(specmi_handle* h, int B, int H, int W, int pair, int32_t* mode) {
    if (!h) return SPECMI_ERR_ARG;
    if (!mode || B <= 0 || H <= 0 || W <= 0) return fail(h, SPECMI_ERR_ARG, "bad argument");
    *mode = h->hrnet ? 0 : trunk_mode(h, B, H, W, pair != 0);
    return SPECMI_OK;
}

int specmi_sync_status(specmi_handle* h, int32_t* persist_err) {
    ENTER(h);
    if (!persist_err) return fail(h, SPECMI_ERR_ARG, "null argument");
    HIPCHK(h, hipDeviceSynchronize());
    *persist_err = 0;
    if (h->pctl) {
        PersistCtl c;
        HIPCHK(h, hipMemcpy(&c, h->pctl, sizeof(c), hipMemcpyDeviceToHost));
        *persist_err = (int32_t)c.err;
        if (!c.err) {   // a finished launch leaves the control block clean: anything else is a protocol error too
            if (c.exit) *persist_err = -1;
            for (int i = 0; i < 2 * kPersistMaxLayers; ++i)
                if (c.done[i]) *persist_err = -2;
        }
    }
    if (h->tail_ctl && !*persist_err) {   // the fused tails (option "tail_fuse"): error word last, every counter zero between launches
        unsigned v[specmi_handle::kTailCtlWords];
        HIPCHK(h, hipMemcpy(v, h->tail_ctl, sizeof(v), hipMemcpyDeviceToHost));
        if (v[specmi_handle::kTailCtlWords - 1]) *persist_err = 3;
        else
            for (int i = 0; i < specmi_handle::kTailCtlWords - 1; ++i)
                if (v[i]) *persist_err = -3;
    }
    return SPECMI_OK;
}

int specmi_sync_reset(specmi_handle* h, void* stream) {
    ENTER(h);
    reset_sync_state(h, (hipStream_t)stream);
    return SPECMI_OK;
}

int specmi_debug_poison_sync(specmi_handle* h, uint32_t value) {
    ENTER(h);
    HIPCHK(h, hipDeviceSynchronize());
    if (h->sk.cnt) {
        std::vector<unsigned> v((size_t)h->sk.ncnt, value);
        HIPCHK(h, hipMemcpy(h->sk.cnt, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    }
    if (h->tail_ctl) {
        std::vector<unsigned> v((size_t)specmi_handle::kTailCtlWords, value);
        HIPCHK(h, hipMemcpy(h->tail_ctl, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    }
    if (h->pctl) {
        PersistCtl c;
        for (int i = 0; i < 2 * kPersistMaxLayers; ++i) c.done[i] = value;
        c.exit = value; c.err = 0;
        HIPCHK(h, hipMemcpy(h->pctl, &c, sizeof(c), hipMemcpyHostToDevice));
    }
    return SPECMI_OK;
}

int specmi_profile_enable(specmi_handle* h, int on) {
    if (!h) return SPECMI_ERR_ARG;
    h->prof.on = on != 0;
    return SPECMI_OK;
}

int specmi_profile_read(specmi_handle* h, specmi_prof_entry* entries, int max_entries, int* n) {
    ENTER(h);
    if (!n) return fail(h, SPECMI_ERR_ARG, "n is NULL");
    std::vector<specmi_prof_entry> acc;
    for (auto& r : h->prof.log) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
        specmi_prof_entry* e = nullptr;
        for (auto& a : acc)
            if (!std::strncmp(a.kernel, r.kernel, sizeof(a.kernel) - 1) && !std::strncmp(a.label, r.label.c_str(), sizeof(a.label) - 1)) {
                e = &a;
                break;
            }
        if (!e) {
            specmi_prof_entry ne;
            std::memset(&ne, 0, sizeof(ne));
            std::strncpy(ne.kernel, r.kernel, sizeof(ne.kernel) - 1);
            std::strncpy(ne.label, r.label.c_str(), sizeof(ne.label) - 1);
            acc.push_back(ne);
            e = &acc.back();
        }
        e->ms += ms; e->flops += r.flops; e->bytes += r.bytes; e->launches += 1;
    }
    h->prof.log.clear();
    *n = (int)acc.size();
    for (int i = 0; i < (int)acc.size() && i < max_entries && entries; ++i) entries[i] = acc[i];
    return SPECMI_OK;
}

}  // extern "C"
