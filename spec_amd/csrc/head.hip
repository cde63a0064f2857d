// head.hip - small per-image kernels around the regressor GEMMs (gfx950).
//
//   * head_init_kernel   : builds the IEF state row of HMRHead (pare, call site
//     spec/models/hmr.py:94-98): xc = [xf(2048) | pose6d(144) | shape(10) | cam(3) |
//     rot6d(cam_rotmat)(6) | vfov(1) | zero pad] with vfov = 2*atan(img_h / (2*K[0,0]))
//     (spec/models/hmr.py:95).  xf is written by the avg-pool kernel; the three dec* GEMMs
//     update the state columns in place (residual epilogue), so torch.cat is never executed.
//   * head_final_kernel  : rot6d_to_rotmat (Gram-Schmidt, F.normalize eps 1e-12) and the
//     output gather pred_pose / pred_shape / pred_cam / pred_pose_6d.
//   * camcalib_decode_kernel : soft-argmax decode of the three 256-bin logit rows
//     (camcalib/cam_utils.py:110-133: softmax expectation -> [-1,1] -> affine to radians),
//     f = h/2/tan(vfov/2) (scripts/camcalib_demo.py:129), R = euler2matrix([pitch,0,roll])
//     via the quaternion route and K with K[2,2] = 0 (spec/utils/cam_params.py:37-46).
//     One wave per (image, head): 64-lane shuffle reductions, no LDS traffic for the softmax.
#include "smpl_pose_body.h"

namespace specmi {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) head_init_kernel(const HeadInit a) { head_init_row(a, blockIdx.x, threadIdx.x, blockDim.x); }

int launch_head_init(float* xc, const float* init_pose, const float* init_shape, const float* init_cam,
                     const float* cam_rotmat, const float* cam_intrinsics, const float* img_h, int use_cam_feats,
                     int B, int state_off, int ld, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "head_init", 0.0, 4.0 * B * (ld - state_off));
    const HeadInit a{xc, init_pose, init_shape, init_cam, cam_rotmat, cam_intrinsics, img_h, use_cam_feats, state_off, ld};
    hipLaunchKernelGGL(head_init_kernel, dim3(B), dim3(256), 0, ctx.stream, a);
    return (int)hipGetLastError();
}

// `state` = the 157 regressor outputs [pose6d | shape | cam] of image b at state[b * ld_state]; every output
// pointer addresses image 0 and is advanced by its own per-image stride (dense, or the row stride of a packed record)
__global__ void __launch_bounds__(256) head_final_kernel(const float* __restrict__ state, long ld_state,
                                                          float* __restrict__ pred_pose, float* __restrict__ pred_shape,
                                                          float* __restrict__ pred_cam, float* __restrict__ pred_pose_6d,
                                                          long ld_pose, long ld_shape, long ld_cam, long ld_p6d,
                                                          float* __restrict__ rot_ws, float* __restrict__ betas_ws,
                                                          float* __restrict__ cam_ws, int B) {
    const int b = blockIdx.x, t = threadIdx.x;
    const float* s = state + (size_t)b * ld_state;
    if (t < 144 && pred_pose_6d) pred_pose_6d[(size_t)b * ld_p6d + t] = s[t];
    if (t >= 144 && t < 154) {
        const float v = s[t];
        if (pred_shape) pred_shape[(size_t)b * ld_shape + t - 144] = v;
        if (betas_ws) betas_ws[(size_t)b * 10 + t - 144] = v;
    }
    if (t >= 154 && t < 157) {
        const float v = s[t];
        if (pred_cam) pred_cam[(size_t)b * ld_cam + t - 154] = v;
        if (cam_ws) cam_ws[(size_t)b * 3 + t - 154] = v;
    }
    if (t >= 192 && t < 216) {
        const int j = t - 192;
        float Rm[9];
        rot6d_joint(s + 6 * j, Rm);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (pred_pose) pred_pose[(size_t)b * ld_pose + j * 9 + k] = Rm[k];
            if (rot_ws) rot_ws[((size_t)b * 24 + j) * 9 + k] = Rm[k];
        }
    }
}

int launch_head_final(const float* state, long ld_state, float* pred_pose, float* pred_shape, float* pred_cam,
                      float* pred_pose_6d, const long ld[4], float* rotmat_ws, float* betas_ws, float* cam_ws, int B,
                      const LaunchCtx& ctx) {
    ProfScope ps(ctx, "head_final_rot6d", 0.0, 4.0 * B * (157 + 157 + 216));
    hipLaunchKernelGGL(head_final_kernel, dim3(B), dim3(256), 0, ctx.stream, state, ld_state, pred_pose, pred_shape,
                       pred_cam, pred_pose_6d, ld[0], ld[1], ld[2], ld[3], rotmat_ws, betas_ws, cam_ws, B);
    return (int)hipGetLastError();
}

// HMRHead with estimate_var (un-vendored pare; constructor flags at spec/models/hmr.py:35-38,57-64, consumer spec/losses.py:61-62):
// 'pred_pose_var' = cat([pred_pose_6d, var_pose]), 'pred_shape_var' = cat([pred_shape, var_shape]), the variances being the
// (optionally activated) extra decoder outputs of the LAST iteration
__device__ __forceinline__ float uncertainty_act(float x, int act) {
    switch (act) {
        case 1: return fmaxf(x, 0.f);                                  // F.relu
        case 2: return x > 20.f ? x : log1pf(expf(x));                 // F.softplus (beta 1, threshold 20)
        case 3: return 1.f / (1.f + expf(-x));                         // F.sigmoid
        case 4: return tanhf(x);                                       // F.tanh
        case 5: return x > 0.f ? x : expm1f(x);                        // F.elu (alpha 1)
        default: return x;
    }
}
__global__ void __launch_bounds__(320) head_var_kernel(const float* __restrict__ state, long ld_state, const float* __restrict__ var, long ld_var,
                                                       int act, float* __restrict__ pose_var, float* __restrict__ shape_var) {
    const int b = blockIdx.x, t = threadIdx.x;
    const float* s = state + (size_t)b * ld_state;
    const float* v = var + (size_t)b * ld_var;
    if (t < 144) pose_var[(size_t)b * 288 + t] = s[t];
    else if (t < 288) pose_var[(size_t)b * 288 + t] = uncertainty_act(v[t - 144], act);
    else if (t < 298) shape_var[(size_t)b * 20 + t - 288] = s[144 + t - 288];
    else if (t < 308) shape_var[(size_t)b * 20 + t - 288] = uncertainty_act(v[144 + t - 298], act);
}
int launch_head_var(const float* state, long ld_state, const float* var, long ld_var, int act, float* pose_var, float* shape_var,
                    int B, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "head_var", 0.0, 4.0 * B * (308 + 308));
    hipLaunchKernelGGL(head_var_kernel, dim3(B), dim3(320), 0, ctx.stream, state, ld_state, var, ld_var, act, pose_var, shape_var);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// R = batch_euler2matrix([pitch, 0, roll]) through the quaternion route and K with K[2,2] = 0
// (spec/utils/cam_params.py:37-46)
__device__ __forceinline__ void build_cam_RK(float pt, float rl, float f, float w, float h, float* R, float* K) {
    if (R) {
        const float hx = pt / 2.0f, hy = 0.0f / 2.0f, hz = rl / 2.0f;
        const float cz = cosf(hz), sz = sinf(hz), cy = cosf(hy), sy = sinf(hy), cx = cosf(hx), sx = sinf(hx);
        float qw = cx * cy * cz - sx * sy * sz;
        float qx = cx * sy * sz + cy * cz * sx;
        float qy = cx * cz * sy - sx * cy * sz;
        float qz = cx * cy * sz + sx * cz * sy;
        const float nq = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
        qw /= nq; qx /= nq; qy /= nq; qz /= nq;
        const float w2 = qw * qw, x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        const float wx = qw * qx, wy = qw * qy, wz = qw * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
        R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
        R[3] = 2 * wz + 2 * xy;     R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
        R[6] = 2 * xz - 2 * wy;     R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
    }
    if (K) {
        K[0] = f;   K[1] = 0.f; K[2] = w / 2.0f;
        K[3] = 0.f; K[4] = f;   K[5] = h / 2.0f;
        K[6] = 0.f; K[7] = 0.f; K[8] = 0.f;   // K[2,2] stays 0 as in the reference
    }
}

// ---- FC layers at small batch (latency plan) -------------------------------------------------------------------------
// The three CamCalib heads (camcalib/model.py:77-79: fc_vfov / fc_pitch / fc_roll on the same pooled features) and the HMR
// regressor's composed map (spec/models/hmr.py:96) are M = batch-row GEMMs: on the matrix-core kernel each is a launch of 4-24
// tiles that are 1/64 ... 8/64 full plus a fold of K slices - 11-12 us per GEMM for 0.2 us of arithmetic, and three graph nodes
// where one will do.  Here: ONE launch for up to three heads (blockIdx.y), one WAVE per output column, lanes across K (16-byte
// loads of the row-major weight row: 1 KiB per instruction, coalesced), one or two images per wave, fixed lane-local k
// order and a fixed xor-shuffle tree - an image's result does not depend on the batch it travels in.
struct GemvHead { const float* x; const float* w; const float* bias; const float* res; float* out; };
struct GemvArgs {
    GemvHead hd[3];
    int nheads, N, Kp, ldx, ldo, B;   // w: (N, Kp) row-major, zero padded beyond the true K; x rows ldx apart (>= Kp floats readable)
};

// NB = images per wave (1 or 2; more images = more waves, blockIdx.z).  The kernel is pure latency - a wave's whole k range is
// 9 steps of (1 + NB) 16-byte loads - so GKU steps are requested before the first is used (8 images per wave at two steps in
// flight took 16 us at batch 8 for 1.4 MB of weights).
constexpr int GKU = 9;
template <int NB>
__global__ void __launch_bounds__(256) fc_gemv_kernel(const GemvArgs a) {
    const GemvHead& hd = a.hd[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    const int b0 = blockIdx.z * NB;
    if (n >= a.N) return;
    const int nb = min(NB, a.B - b0);
    const float* wrow = hd.w + (size_t)n * a.Kp;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    // rows past the batch re-read the last image (results discarded) and k steps past the row re-read its last step with weight
    // 0 (acc + 0 * x = acc): no branch around a load, so all of them are independent and in flight together
    const float* xrow[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) xrow[b] = hd.x + (size_t)(b0 + (b < nb ? b : nb - 1)) * a.ldx;
    for (int k0 = lane * 4; k0 < a.Kp; k0 += 256 * GKU) {
        float4 wv[GKU], xv[GKU][NB];
#pragma unroll
        for (int u = 0; u < GKU; ++u) {
            const int k = min(k0 + u * 256, a.Kp - 4);
            wv[u] = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
            for (int b = 0; b < NB; ++b) xv[u][b] = *reinterpret_cast<const float4*>(xrow[b] + k);
        }
#pragma unroll
        for (int u = 0; u < GKU; ++u) {
            // a step past the row re-read the last real quad: BOTH operands are zeroed there, so that it adds exactly +0 even when
            // that x quad holds Inf / NaN (0 * Inf = NaN would reach outputs the matrix-core path leaves finite)
            const bool in = k0 + u * 256 < a.Kp;
            const float wx = in ? wv[u].x : 0.f, wy = in ? wv[u].y : 0.f, wz = in ? wv[u].z : 0.f, ww = in ? wv[u].w : 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[b] = fmaf(wx, in ? xv[u][b].x : 0.f, acc[b]);
                acc[b] = fmaf(wy, in ? xv[u][b].y : 0.f, acc[b]);
                acc[b] = fmaf(wz, in ? xv[u][b].z : 0.f, acc[b]);
                acc[b] = fmaf(ww, in ? xv[u][b].w : 0.f, acc[b]);
            }
        }
        // all loads of the body first, then the arithmetic (the scheduler otherwise puts each load next to its use: serial round trips)
        __builtin_amdgcn_sched_group_barrier(0x020, GKU * (1 + NB), 0);
        __builtin_amdgcn_sched_group_barrier(0x002, GKU * NB * 8, 0);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float v = acc[b];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[b] = v;
    }
    if (lane < nb) {
        float v = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (lane == b) v = acc[b];
        const size_t o = (size_t)(b0 + lane) * a.ldo + n;
        v += hd.bias[n];
        if (hd.res) v += hd.res[o];
        hd.out[o] = v;
    }
}

int launch_fc_gemv(const FcGemv* heads, int nheads, int N, int Kp, int ldx, int ldo, int B, const LaunchCtx& ctx) {
    if (nheads < 1 || nheads > 3 || N < 1 || B < 1 || Kp % 4 != 0 || ldx % 4 != 0) return (int)hipErrorInvalidValue;
    GemvArgs a;
    a.nheads = nheads; a.N = N; a.Kp = Kp; a.ldx = ldx; a.ldo = ldo; a.B = B;
    for (int i = 0; i < 3; ++i) {
        const FcGemv& f = heads[i < nheads ? i : 0];
        if ((reinterpret_cast<uintptr_t>(f.x) | reinterpret_cast<uintptr_t>(f.w)) & 15) return (int)hipErrorInvalidValue;
        a.hd[i] = GemvHead{f.x, f.w, f.bias, f.res, f.out};
    }
    ProfScope ps(ctx, "fc_gemv_f32", 2.0 * nheads * (double)B * N * Kp, 4.0 * nheads * ((double)N * Kp + (double)B * (Kp + N)));
    // an image's sum is the same in every instantiation (lane-local k order, then the shuffle tree): NB only sets how many share a wave
    const dim3 blk(256);
    if (B == 1) hipLaunchKernelGGL(fc_gemv_kernel<1>, dim3((N + 3) / 4, nheads, 1), blk, 0, ctx.stream, a);
    else hipLaunchKernelGGL(fc_gemv_kernel<2>, dim3((N + 3) / 4, nheads, (B + 1) / 2), blk, 0, ctx.stream, a);
    return (int)hipGetLastError();
}

// soft-argmax decode of image b: waves 0..2 = vfov / pitch / roll (a fourth wave idles), `ang` = 3 floats of LDS.  COHERENT: the
// logits were written earlier in the SAME launch by other workgroups (fused tail: write-through stores) -> agent-scope loads.
struct DecodeArgs {
    const float *lv, *lp, *lr; int nbins; const float *img_h, *img_w;
    float *vfov, *pitch, *roll, *f_pix, *R, *K; long ld_ang;
};
template <bool COHERENT>
__device__ __forceinline__ void camcalib_decode_image(const DecodeArgs& a, int b, float* ang) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 3) {
        const float* row = (wave == 0 ? a.lv : wave == 1 ? a.lp : a.lr) + (size_t)b * a.nbins;
        auto X = [&](int i) { return COHERENT ? __hip_atomic_load(row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : row[i]; };
        float mx = -INFINITY;
        for (int i = lane; i < a.nbins; i += 64) mx = fmaxf(mx, X(i));
        mx = wave_max(mx);
        float se = 0.f, sp = 0.f;
        for (int i = lane; i < a.nbins; i += 64) {
            const float e = expf(X(i) - mx);
            se += e;
        }
        se = wave_sum(se);
        for (int i = lane; i < a.nbins; i += 64) {
            const float pr = expf(X(i) - mx) / se;  // softmax, then expectation of the index
            sp += pr * (float)i;
        }
        sp = wave_sum(sp);
        if (lane == 0) {
            const float s = sp / (float)(a.nbins - 1) * 2.0f - 1.0f;                  // softargmax1d normalisation
            // soft_idx_to_angle: (max - min) is a python double rounded to fp32 when it meets the tensor
            const float span = wave == 0 ? (float)(2.1 - 0.2617) : (float)(0.6 - (-0.6));
            const float lo = wave == 0 ? (float)0.2617 : (float)-0.6;
            ang[wave] = span * ((s + 1.0f) / 2.0f) + lo;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float vf = ang[0], pt = ang[1], rl = ang[2];
        if (a.vfov) a.vfov[(size_t)b * a.ld_ang] = vf;
        if (a.pitch) a.pitch[(size_t)b * a.ld_ang] = pt;
        if (a.roll) a.roll[(size_t)b * a.ld_ang] = rl;
        const float h = a.img_h ? a.img_h[b] : 0.f, w = a.img_w ? a.img_w[b] : 0.f;
        const float f = h / 2.0f / tanf(vf / 2.0f);
        if (a.f_pix) a.f_pix[b] = f;
        build_cam_RK(pt, rl, f, w, h, a.R ? a.R + (size_t)b * 9 : nullptr, a.K ? a.K + (size_t)b * 9 : nullptr);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(192) camcalib_decode_kernel(const DecodeArgs a) {
    __shared__ float ang[3];
    camcalib_decode_image<false>(a, blockIdx.x, ang);
}

int launch_camcalib_decode(const float* lv, const float* lp, const float* lr, int B, int nbins, const float* img_h,
                           const float* img_w, float* vfov, float* pitch, float* roll, float* f_pix, float* R,
                           float* K, long ld_ang, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "camcalib_decode", 0.0, 4.0 * B * (3.0 * nbins + 24));
    const DecodeArgs a{lv, lp, lr, nbins, img_h, img_w, vfov, pitch, roll, f_pix, R, K, ld_ang};
    hipLaunchKernelGGL(camcalib_decode_kernel, dim3(B), dim3(192), 0, ctx.stream, a);
    return (int)hipGetLastError();
}

// ---- fused tails (round 5) ---------------------------------------------------------------------------------------------------
// Behind each trunk the small-batch step ran 3 (CamCalib: avg-pool -> three heads -> decode) and 3 (HMR: avg-pool + state init ->
// composed regressor map -> pose chain) graph nodes of 4.5-7 us each whose work is a few hundred KB.  Here each tail is ONE launch
// of the GEMV grid:
//   phase A  the first workgroups pool the trunk's feature map (one (image, channel quad) per thread, the order of avgpool_kernel)
//            and write the IEF state columns (head_init_row) with write-through stores, drain them and bump ONE counter;
//   phase B  every workgroup requests its weight rows FIRST (they depend on nothing), waits for the counter (bounded spin), reads
//            the pooled rows with agent-scope loads and runs the GEMV exactly as fc_gemv_kernel does (same lane-local k order, same
//            shuffle tree: same bits); outputs leave with write-through stores, one ticket per image slab (blockIdx.z);
//   epilogue the last workgroup of a slab to arrive decodes its images (CamCalib: camcalib_decode_image) or runs their pose chains
//            (HMR: smpl_pose_body, one wave per image) - the code of the stand-alone kernels on agent-scope loads.
// The hand-offs are the write-through form of conv_igemm_tile.h (gfx950: MI355X_MICROARCH.md, inter-workgroup visibility).  The
// producers are the lowest-numbered workgroups of the grid and wait for nobody; every spin is bounded.
struct PoseTail {
    const float *Jt, *Jd; const int* parents; float *feat, *Afrag, *posed_j; HeadFinal fin;
};
struct TailArgs {
    GemvArgs g;
    const float* map; float* pool_out; int HW, C4, pool_ld, pool_total, npool, ninit;
    HeadInit init;
    unsigned* ctl;        // [0] phase-A arrivals, [1] slabs finished, [2 + z] arrivals of slab z; all zero between launches
    unsigned* err;        // set to 1 when the bounded wait for phase A gave up (results of that launch are garbage): specmi_sync_status
    unsigned spin_limit;
    DecodeArgs dec;
    PoseTail pose;
};

template <int NB, int EPI>   // EPI: 1 = CamCalib decode, 2 = HMR pose chain
__global__ void __launch_bounds__(256) tail_gemv_kernel(const TailArgs a) {
    __shared__ float ang[3];
    __shared__ int flag;
    const GemvHead& hd = a.g.hd[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    // ---- phase A ---------------------------------------------------------------------------------------------------------------
    if (w < a.npool + a.ninit) {
        if (w < a.npool) {
            const int i = w * 256 + threadIdx.x;
            if (i < a.pool_total) {
                const int c4 = i % a.C4, b = i / a.C4;
                const f32x4* src = reinterpret_cast<const f32x4*>(a.map) + (size_t)b * a.HW * a.C4 + c4;
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 7
                for (int p = 0; p < a.HW; ++p) {
                    const f32x4 v = src[(size_t)p * a.C4];
                    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
                }
                f32x4 q;    // PyTorch's mean is sum / count
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = s[e] / (float)a.HW;
                const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.pool_out, 0, (unsigned)a.g.B * (unsigned)a.pool_ld * 4u, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q), prs, (unsigned)((b * a.pool_ld + c4 * 4) * 4), 0, /*sc1*/ 16);
            }
        } else {
            head_init_row<true>(a.init, w - a.npool, threadIdx.x, 256);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.ctl + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- phase B: fc_gemv_kernel with the weights requested before the wait ------------------------------------------------------
    const int n = blockIdx.x * 4 + wave;
    const int b0 = blockIdx.z * NB;
    const int nb = min(NB, a.g.B - b0);
    const bool col = n < a.g.N;
    const float* wrow = hd.w + (size_t)(col ? n : 0) * a.g.Kp;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    int xr[NB];    // rows past the batch re-read the last image (results discarded)
#pragma unroll
    for (int b = 0; b < NB; ++b) xr[b] = b0 + (b < nb ? b : nb - 1);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hd.x), 0, (unsigned)a.g.B * (unsigned)a.g.ldx * 4u, 0x00020000);
    bool waited = false;
    for (int k0 = lane * 4; k0 < a.g.Kp; k0 += 256 * GKU) {
        float4 wv[GKU], xv[GKU][NB];
#pragma unroll
        for (int u = 0; u < GKU; ++u) {
            const int k = min(k0 + u * 256, a.g.Kp - 4);
            wv[u] = *reinterpret_cast<const float4*>(wrow + k);
        }
        if (!waited) {
            if (threadIdx.x == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(a.ctl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(a.npool + a.ninit)) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > a.spin_limit) {
                        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
            waited = true;
        }
#pragma unroll
        for (int u = 0; u < GKU; ++u) {
            const int k = min(k0 + u * 256, a.g.Kp - 4);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (unsigned)((xr[b] * a.g.ldx + k) * 4), 0, /*sc1*/ 16));
                xv[u][b].x = q[0]; xv[u][b].y = q[1]; xv[u][b].z = q[2]; xv[u][b].w = q[3];
            }
        }
#pragma unroll
        for (int u = 0; u < GKU; ++u) {
            const bool in = k0 + u * 256 < a.g.Kp;
            const float wx = in ? wv[u].x : 0.f, wy = in ? wv[u].y : 0.f, wz = in ? wv[u].z : 0.f, ww = in ? wv[u].w : 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[b] = fmaf(wx, in ? xv[u][b].x : 0.f, acc[b]);
                acc[b] = fmaf(wy, in ? xv[u][b].y : 0.f, acc[b]);
                acc[b] = fmaf(wz, in ? xv[u][b].z : 0.f, acc[b]);
                acc[b] = fmaf(ww, in ? xv[u][b].w : 0.f, acc[b]);
            }
        }
        // all x loads of the body first, then the arithmetic (as in fc_gemv_kernel: the scheduler otherwise pairs each load with its use)
        __builtin_amdgcn_sched_group_barrier(0x020, GKU * NB, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, GKU * NB * 8, 0);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float v = acc[b];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[b] = v;
    }
    if (col && lane < nb) {
        float v = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (lane == b) v = acc[b];
        const size_t o = (size_t)(b0 + lane) * a.g.ldo + n;
        v += hd.bias[n];
        if (hd.res) v += hd.res[o];
        __hip_atomic_store(hd.out + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- epilogue by the slab's last arriver ----------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned per_slab = gridDim.x * gridDim.y;
        const unsigned ticket = __hip_atomic_fetch_add(a.ctl + 2 + blockIdx.z, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == per_slab - 1;
        if (last) {
            __hip_atomic_store(a.ctl + 2 + blockIdx.z, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the last slab to finish frees the phase-A counter: every workgroup has passed its wait by then
            const unsigned done = __hip_atomic_fetch_add(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == gridDim.z - 1) {
                __hip_atomic_store(a.ctl + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        flag = last;
    }
    __syncthreads();
    if (!flag) return;
    if (EPI == 1) {
        for (int b = 0; b < nb; ++b) camcalib_decode_image<true>(a.dec, b0 + b, ang);
    } else {
        if (wave < nb)
            smpl_pose_body<true, true>(b0 + wave, lane, nullptr, nullptr, a.pose.Jt, a.pose.Jd, a.pose.parents, a.pose.feat, a.pose.Afrag,
                                       a.pose.posed_j, a.pose.fin);
    }
}

// One launch for: avg-pool of `map` (B, HW, C) into pool_out rows (stride pool_ld) [+ IEF state columns `init`] -> up to three GEMV
// heads over those rows -> epilogue (dec != nullptr: CamCalib decode; pose != nullptr: HMR pose chain).  ctl: >= 2 + ceil(B / 2) zeroed
// words.  Returns hipErrorInvalidValue for shapes the fused form does not take (the caller launches the separate kernels).
int launch_tail_gemv(const FcGemv* heads, int nheads, int N, int Kp, int ldx, int ldo, int B, const float* map, float* pool_out, int HW,
                     int C, int pool_ld, const HeadInit* init, unsigned* ctl, int ctl_words, const DecodeArgs* dec, const PoseTail* pose,
                     const LaunchCtx& ctx) {
    if (nheads < 1 || nheads > 3 || N < 1 || B < 1 || Kp % 4 != 0 || ldx % 4 != 0 || C % 4 != 0 || pool_ld % 4 != 0 || HW >= 64 ||
        (dec != nullptr) == (pose != nullptr) || !ctl)
        return (int)hipErrorInvalidValue;
    const int NBv = B == 1 ? 1 : 2;
    const int nz = (B + NBv - 1) / NBv;
    if (ctl_words < 3 + nz) return (int)hipErrorInvalidValue;      // (the last word is the error word)
    TailArgs a;
    a.g.nheads = nheads; a.g.N = N; a.g.Kp = Kp; a.g.ldx = ldx; a.g.ldo = ldo; a.g.B = B;
    for (int i = 0; i < 3; ++i) {
        const FcGemv& f = heads[i < nheads ? i : 0];
        if ((reinterpret_cast<uintptr_t>(f.x) | reinterpret_cast<uintptr_t>(f.w)) & 15) return (int)hipErrorInvalidValue;
        a.g.hd[i] = GemvHead{f.x, f.w, f.bias, f.res, f.out};
    }
    a.map = map; a.pool_out = pool_out; a.HW = HW; a.C4 = C / 4; a.pool_ld = pool_ld;
    a.pool_total = B * (C / 4);
    a.npool = (a.pool_total + 255) / 256;
    a.ninit = init ? B : 0;
    a.init = init ? *init : HeadInit{};
    a.ctl = ctl;
    a.err = ctl + ctl_words - 1;
    a.spin_limit = 2000000u;
    a.dec = dec ? *dec : DecodeArgs{};
    a.pose = pose ? *pose : PoseTail{};
    const dim3 grid((N + 3) / 4, nheads, nz), blk(256);
    if ((int)(grid.x * grid.y * grid.z) < a.npool + a.ninit) return (int)hipErrorInvalidValue;   // (every phase-A item needs a workgroup)
    ProfScope ps(ctx, dec ? "tail_pool_gemv_decode" : "tail_pool_gemv_pose", 2.0 * nheads * (double)B * N * Kp,
                 4.0 * (nheads * ((double)N * Kp + (double)B * (Kp + N)) + (double)B * HW * C));
    if (dec) {
        if (NBv == 1) hipLaunchKernelGGL((tail_gemv_kernel<1, 1>), grid, blk, 0, ctx.stream, a);
        else hipLaunchKernelGGL((tail_gemv_kernel<2, 1>), grid, blk, 0, ctx.stream, a);
    } else {
        if (NBv == 1) hipLaunchKernelGGL((tail_gemv_kernel<1, 2>), grid, blk, 0, ctx.stream, a);
        else hipLaunchKernelGGL((tail_gemv_kernel<2, 2>), grid, blk, 0, ctx.stream, a);
    }
    return (int)hipGetLastError();
}

int launch_tail_camcalib(const FcGemv* heads, int N, int Kp, int B, const float* map, float* pooled, int HW, int C, unsigned* ctl, int ctl_words,
                         const float* img_h, const float* img_w, float* vfov, float* pitch, float* roll, float* f_pix, float* R, float* K,
                         long ld_ang, const LaunchCtx& ctx) {
    const DecodeArgs d{heads[0].out, heads[1].out, heads[2].out, N, img_h, img_w, vfov, pitch, roll, f_pix, R, K, ld_ang};
    return launch_tail_gemv(heads, 3, N, Kp, Kp, N, B, map, pooled, HW, C, Kp, nullptr, ctl, ctl_words, &d, nullptr, ctx);
}

int launch_tail_hmr(const FcGemv& head, int N, int Kp, int ldx, int ldo, int B, const float* map, float* xc, int HW, int C, const HeadInit& init,
                    unsigned* ctl, int ctl_words, const SmplDev& m, float* feat, float* Afrag, float* posed_j, const HeadFinal& fin,
                    const LaunchCtx& ctx) {
    const PoseTail pt{m.J_template, m.J_shapedirs, m.parents, feat, Afrag, posed_j, fin};
    return launch_tail_gemv(&head, 1, N, Kp, ldx, ldo, B, map, xc, HW, C, ldx, &init, ctl, ctl_words, nullptr, &pt, ctx);
}

// The two per-row reductions camcalib/cam_utils.py applies to a (rows, nbins) logit tensor, one wave per row:
//   idx[row]  = np.argmax(row)            (bins2vfov / bins2pitch / bins2roll / bins2horizon, cam_utils.py:66-91):
//               FIRST index of the maximum, a NaN counts as the maximum (NumPy semantics) - index work, bit-exact;
//   soft[row] = softargmax1d(row, normalize_keypoints=True) in [-1, 1] (get_softargmax, cam_utils.py:110-118).
__global__ void __launch_bounds__(256) bins_reduce_kernel(const float* __restrict__ x, int rows, int nbins,
                                                           int* __restrict__ idx, float* __restrict__ soft) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* r = x + (size_t)row * nbins;
    float mx = -INFINITY;
    int mi = 0x7fffffff, nan_i = 0x7fffffff;
    for (int i = lane; i < nbins; i += 64) {
        const float v = r[i];
        if (v != v) nan_i = min(nan_i, i);
        if (v > mx || (v == mx && i < mi)) { mx = v; mi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float omx = __shfl_xor(mx, o, 64);
        const int omi = __shfl_xor(mi, o, 64);
        nan_i = min(nan_i, __shfl_xor(nan_i, o, 64));
        if (omx > mx || (omx == mx && omi < mi)) { mx = omx; mi = omi; }
    }
    if (idx && lane == 0) idx[row] = nan_i != 0x7fffffff ? nan_i : (mi == 0x7fffffff ? 0 : mi);
    if (soft) {
        float se = 0.f, sp = 0.f;
        for (int i = lane; i < nbins; i += 64) se += expf(r[i] - mx);
        se = wave_sum(se);
        for (int i = lane; i < nbins; i += 64) sp += expf(r[i] - mx) / se * (float)i;
        sp = wave_sum(sp);
        if (lane == 0) soft[row] = sp / (float)(nbins - 1) * 2.0f - 1.0f;
    }
}

int launch_bins_reduce(const float* x, int rows, int nbins, int* idx, float* soft, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "bins_reduce", 0.0, 4.0 * rows * (nbins + 2.0));
    hipLaunchKernelGGL(bins_reduce_kernel, dim3((rows + 3) / 4), dim3(256), 0, ctx.stream, x, rows, nbins, idx, soft);
    return (int)hipGetLastError();
}

// read_cam_params (spec/utils/cam_params.py:24-50) for already-decoded angles: (pitch, roll, f_pix, w, h) -> R, K
__global__ void cam_params_kernel(const float* __restrict__ pitch, const float* __restrict__ roll,
                                  const float* __restrict__ f_pix, const float* __restrict__ img_w,
                                  const float* __restrict__ img_h, int B, float* __restrict__ R, float* __restrict__ K) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    build_cam_RK(pitch[b], roll[b], f_pix[b], img_w[b], img_h[b], R ? R + (size_t)b * 9 : nullptr,
                 K ? K + (size_t)b * 9 : nullptr);
}

int launch_cam_params(const float* pitch, const float* roll, const float* f_pix, const float* img_w, const float* img_h,
                      int B, float* R, float* K, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "cam_params", 0.0, 4.0 * B * 23);
    hipLaunchKernelGGL(cam_params_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx.stream, pitch, roll, f_pix, img_w, img_h, B, R, K);
    return (int)hipGetLastError();
}

}  // namespace specmi
