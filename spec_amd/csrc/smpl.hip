// smpl.hip - SMPL linear blend skinning + 49 joints + camera + projection (gfx950).
//
// Replaces SMPLCamHead / SMPLHead of the reference path (call sites spec/models/hmr.py:101-120;
// arithmetic = smplx 0.1.28 lbs(pose2rot=False) + the SPIN/PARE 49-joint wrapper +
// convert_pare_to_full_img_cam + perspective_projection).  ~25 framework launches and a
// 23-step Python loop upstream become three kernels:
//
//   1. smpl_pose_kernel   (one wave per image): rest joints J = J_template + J_shapedirs*beta,
//      pose features (R_j - I), and the 24-joint kinematic chain.  Lane j owns joint j; the
//      tree is walked level by level and a child reads its parent's 3x4 world transform with
//      wave shuffles (ds_bpermute), so the chain never touches memory.
//   2. smpl_skin_kernel   (a wave = 32 vertices x 32 images): everything that is a contraction runs on the fp32 matrix
//      cores.  v_posed = [pose features | betas | 1] (32 images x 224) x [posedirs ; shapedirs ; v_template] (224 x 32
//      vertices, one GEMM per coordinate) is 3 x 112 v_mfma_f32_32x32x2_f32; the blended transforms T = lbs_weights (32
//      vertices x 24 joints) x A (24 joints x 32 images, one GEMM per entry of the 3x4 matrix) are 12 x 12 more.  Both
//      products leave (vertex row, image column) in the same accumulator slot of a lane, so the skinned vertex
//      T[:, :3] v_posed + T[:, 3] is lane-local arithmetic on the accumulators.  Operands are stored in MFMA fragment
//      order - the body model at commit (SmplDev::dirsT / wT), the per-image features and transforms by the pose kernel -
//      so every operand fetch is one coalesced 16-byte load per lane for four MFMA steps.
//   3. smpl_joints_kernel (one workgroup per image): J_regressor_extra @ vertices (9 dot
//      products of length V, wave-shuffle + LDS reduction), the 21 vertex-picked joints, the
//      49-entry joint_map gather, the full-image camera translation and the projection
//      p = K ((R X + t) / z).
#include "smpl_pose_body.h"

namespace specmi {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool FUSED>
__global__ void __launch_bounds__(64) smpl_pose_kernel(const float* __restrict__ rotmat, const float* __restrict__ betas,
                                                        const float* __restrict__ Jt, const float* __restrict__ Jd,
                                                        const int* __restrict__ parents, float* __restrict__ feat,
                                                        float* __restrict__ Afrag, float* __restrict__ posed_j, const HeadFinal fin) {
    smpl_pose_body<FUSED>(blockIdx.x, threadIdx.x, rotmat, betas, Jt, Jd, parents, feat, Afrag, posed_j, fin);
}

// SPLIT = false: grid (ceil(groups / 4), image tiles), 256 threads; wave w of a workgroup owns vertex group 4 blockIdx.x + w
// and needs nothing from the other waves (its LDS slice is private: no barrier).
// SPLIT = true (few image tiles: the 216 vertex groups x 1 wave leave most SIMDs idle and one wave's 480 dependent-by-pipe
// MFMAs are the whole kernel time): grid (groups, image tiles), 192 threads; wave c of a workgroup computes coordinate c of
// v_posed (its own 112-step chain) and row c of the blended transform, the three v_posed coordinates are exchanged through
// LDS.  Every accumulator sees the same operands in the same order as in the other variant: the results are the same bits.
template <bool SPLIT>
__global__ void __launch_bounds__(SPLIT ? 192 : 256) smpl_skin_kernel(const float* __restrict__ dirsT, const float* __restrict__ wT,
                                                                      const float* __restrict__ feat, const float* __restrict__ Afrag,
                                                                      float* __restrict__ verts, long ld_verts, int V, int B, int G) {
    constexpr int NC = SPLIT ? 1 : 3;          // coordinates of v_posed this wave accumulates
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = SPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (g >= G) return;                        // (SPLIT: the grid is exactly G wide, no workgroup is cut by this)
    const int c0 = SPLIT ? wave : 0;
    const int tile = blockIdx.y;
    const f32x4* __restrict__ fq = reinterpret_cast<const f32x4*>(feat + (size_t)tile * FEAT_TILE) + lane;     // [quad][64]
    const f32x4* __restrict__ dq = reinterpret_cast<const f32x4*>(dirsT + (size_t)g * 3 * FEAT_TILE) + (size_t)c0 * KQ * 64 + lane;   // [coord][quad][64]

    // v_posed[vertex][image] per coordinate: A operand = the model's directions (row = vertex), B operand = features (column = image)
    f32x16 vp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) vp[c][r] = 0.f;
    constexpr int PD = 3;                      // quads in flight ahead of the MFMAs (12 MFMAs = 768 cycles each)
    f32x4 fb[PD], dd[NC][PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        fb[i] = fq[i * 64];
#pragma unroll
        for (int k = 0; k < NC; ++k) dd[k][i] = dq[(k * KQ + i) * 64];
    }
#pragma unroll
    for (int s4 = 0; s4 < KQ; ++s4) {
        const int cur = s4 % PD;
        const f32x4 f = fb[cur];
        f32x4 av[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) av[k] = dd[k][cur];
        if (s4 + PD < KQ) {
            fb[cur] = fq[(s4 + PD) * 64];
#pragma unroll
            for (int k = 0; k < NC; ++k) dd[k][cur] = dq[(k * KQ + s4 + PD) * 64];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < NC; ++k) vp[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k][q], f[q], vp[k], 0, 0, 0);
    }

    // blended transform entry e = 4 c + d of (vertex, image): sum_j w[vertex][j] A[image][j][e]; then the skinned coordinate c
    const f32x4* __restrict__ wq = reinterpret_cast<const f32x4*>(wT + (size_t)g * 768) + lane;                // [quad (3)][64]
    const f32x4* __restrict__ aq = reinterpret_cast<const f32x4*>(Afrag + (size_t)tile * A_TILE) + lane;       // [entry][quad][64]
    const f32x4 w0 = wq[0], w1 = wq[64], w2 = wq[128];
    // the 32 images x 32 vertices x 3 coordinates go through LDS (wave-private without SPLIT: no barrier) so that global
    // memory sees, per image, the 384 contiguous bytes of the 32 vertices instead of 4-byte pieces 80 KB apart
    extern __shared__ __attribute__((aligned(16))) float skin_lds[];
    float* const tw = skin_lds + (SPLIT ? 0 : wave * (IT * SKIN_LD));
    float* const xch = skin_lds + IT * SKIN_LD;                      // SPLIT: v_posed exchange [coordinate][register][lane]
    if (SPLIT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(wave * 16 + r) * 64 + lane] = vp[0][r];
    }
    const int vloc0 = 4 * (lane >> 5);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c = c0 + k;
        f32x16 T[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[d][r] = 0.f;
            const f32x4 b0 = aq[((4 * c + d) * 3 + 0) * 64], b1 = aq[((4 * c + d) * 3 + 1) * 64], b2 = aq[((4 * c + d) * 3 + 2) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) T[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[q], b0[q], T[d], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) T[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[q], b1[q], T[d], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) T[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[q], b2[q], T[d], 0, 0, 0);
        }
        if (SPLIT) {
            __syncthreads();                                         // the three coordinates of v_posed are in LDS
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                for (int r = 0; r < 16; ++r) vp[cc][r] = xch[(cc * 16 + r) * 64 + lane];
        }
        // accumulator r of this lane = image lane % 32, vertex vloc0 + (r & 3) + 8 (r >> 2) of the group
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float o = T[0][r] * vp[0][r] + T[1][r] * vp[1][r] + T[2][r] * vp[2][r] + T[3][r];
            tw[(lane & 31) * SKIN_LD + (vloc0 + (r & 3) + 8 * (r >> 2)) * 3 + c] = o;
        }
    }
    if (SPLIT) __syncthreads();                                      // rows of tw hold all three coordinates
    // (without SPLIT one wave's LDS instructions execute in order: the reads below see the writes above)
    const int nimg = min(IT, B - tile * IT);
    const int nfl = min(96, (V - g * 32) * 3);                       // floats of this group that exist
    float* const obase = verts + (size_t)tile * IT * ld_verts + (size_t)g * 96;
    const int i0 = SPLIT ? wave : 0, istep = SPLIT ? 3 : 1;          // SPLIT: the three waves take every third image
    if (((ld_verts & 1) == 0) && ((reinterpret_cast<uintptr_t>(verts) & 7) == 0)) {
        // rows of the output start on 8-byte boundaries: 48 lanes x 8 bytes per image
        const bool on = 2 * lane + 1 < nfl;
        const bool half = 2 * lane + 1 == nfl;                        // (V * 3 odd: the last float alone)
#pragma unroll 4
        for (int i = i0; i < nimg; i += istep) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(tw + i * SKIN_LD + 2 * (lane < 48 ? lane : 0));
            float* const o = obase + (size_t)i * ld_verts + 2 * lane;
            if (on) *reinterpret_cast<f32x2*>(o) = t;
            else if (half) o[0] = t[0];
        }
    } else {
        for (int i = i0; i < nimg; i += istep) {
            float* const o = obase + (size_t)i * ld_verts;
            if (lane < nfl) o[lane] = tw[i * SKIN_LD + lane];
            if (lane + 64 < nfl) o[lane + 64] = tw[i * SKIN_LD + lane + 64];
        }
    }
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct JointArgs {
    const float* verts; const float* posed_j; const float* J_extra; const int* extra_ids; const int* joint_map;
    const float* cam; const float* R; const float* K; const float* bbox_scale; const float* bbox_center;
    const float* img_w; const float* img_h;
    float* joints3d; float* joints2d; float* cam_t;
    long ld_verts, ld_j3d, ld_j2d, ld_camt;
    int V, mode, normalize; float focal, img_res;
};

// One workgroup per image is latency-bound (12 independent loads per vertex and thread): 8 waves x 7 vertices in flight put
// the whole mesh in two rounds (16 waves leave 128 registers per lane, not enough for that).  JT is part of the arithmetic
// (it fixes which vertices a lane sums and the 8-partial fold).
constexpr int JT = 512;
__global__ void __launch_bounds__(JT) smpl_joints_kernel(const JointArgs a) {
    __shared__ float red[JT / 64][27];
    __shared__ float J54[54][3];
    __shared__ float ct[3];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* vb = a.verts + (size_t)b * a.ld_verts;
    // everything the tail needs is requested before the vertex loop, so that its latency (two dependent round trips for the
    // vertex-picked joints) hides behind the loop's
    float pre = 0.f;
    if (t < 72) pre = a.posed_j[(size_t)b * 72 + t];
    else if (t < 72 + 63) pre = vb[(size_t)a.extra_ids[(t - 72) / 3] * 3 + (t - 72) % 3];
    const int jm = t < 49 ? a.joint_map[t] : 0;
    // (block-uniform addresses: scalar loads, no vector registers held across the loop)
    const float cam_s = a.cam[b * 3 + 0], cam_x = a.cam[b * 3 + 1], cam_y = a.cam[b * 3 + 2];
    float cam_f = 0.f, bb_s = 0.f, bb_x = 0.f, bb_y = 0.f, im_w = 0.f, im_h = 0.f;
    if (a.mode == 0) {
        cam_f = a.K[(size_t)b * 9]; bb_s = a.bbox_scale[b]; bb_x = a.bbox_center[b * 2 + 0]; bb_y = a.bbox_center[b * 2 + 1];
        im_w = a.img_w[b]; im_h = a.img_h[b];
    }
    float acc[9][3];
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[e][0] = acc[e][1] = acc[e][2] = 0.f;
    // JU vertices per lane requested before the first is used (12 independent loads each; written out because the unroller
    // interleaves loads and uses, and this kernel is pure latency).  Past-the-end slots read vertex V - 1 with weight 0.
    constexpr int JU = 7;
    for (int v0 = t; v0 < a.V; v0 += JT * JU) {
        float x[JU], y[JU], z[JU], wgt[JU][9];
#pragma unroll
        for (int u = 0; u < JU; ++u) {
            const int v = v0 + u * JT, vc = min(v, a.V - 1);
            x[u] = vb[vc * 3 + 0]; y[u] = vb[vc * 3 + 1]; z[u] = vb[vc * 3 + 2];
#pragma unroll
            for (int e = 0; e < 9; ++e) wgt[u][e] = a.J_extra[(size_t)e * a.V + vc];
        }
        __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks loads next to their uses (serial round trips)
#pragma unroll
        for (int u = 0; u < JU; ++u) {
            const bool in = v0 + u * JT < a.V;
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const float w = in ? wgt[u][e] : 0.f;
                acc[e][0] = fmaf(w, x[u], acc[e][0]);
                acc[e][1] = fmaf(w, y[u], acc[e][1]);
                acc[e][2] = fmaf(w, z[u], acc[e][2]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = wsum(acc[e][c]);
            if ((t & 63) == 0) red[t >> 6][e * 3 + c] = s;
        }
    if (t < 72) J54[t / 3][t % 3] = pre;
    else if (t < 72 + 63) J54[24 + (t - 72) / 3][(t - 72) % 3] = pre;
    if (t == 255) {   // (wave 3: neither of the J54 writers above)
        const float s = cam_s, tx = cam_x, ty = cam_y;
        if (a.mode == 0) {  // convert_pare_to_full_img_cam, res = 224 literal
            const float f = cam_f;
            const float bh = bb_s * 200.0f;
            const float r = bh / 224.0f;
            const float tz = 2.0f * f / (r * 224.0f * s);
            const float cx = 2.0f * (bb_x - (im_w / 2.0f)) / (s * bh);
            const float cy = 2.0f * (bb_y - (im_h / 2.0f)) / (s * bh);
            ct[0] = tx + cx; ct[1] = ty + cy; ct[2] = tz;
        } else {            // convert_weak_perspective_to_perspective
            ct[0] = tx; ct[1] = ty; ct[2] = 2.0f * a.focal / (a.img_res * s + 1e-9f);
        }
    }
    __syncthreads();
    if (t < 27) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < JT / 64; ++w) s += red[w][t];
        J54[45 + t / 3][t % 3] = s;
    }
    __syncthreads();
    if (t < 3 && a.cam_t) a.cam_t[(size_t)b * a.ld_camt + t] = ct[t];
    if (t < 49) {
        const float X0 = J54[jm][0], X1 = J54[jm][1], X2 = J54[jm][2];
        if (a.joints3d) {
            float* o = a.joints3d + (size_t)b * a.ld_j3d + t * 3;
            o[0] = X0; o[1] = X1; o[2] = X2;
        }
        if (a.joints2d) {
            float Rm[9], Km[9];
            if (a.mode == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) { Rm[k] = a.R[(size_t)b * 9 + k]; Km[k] = a.K[(size_t)b * 9 + k]; }
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) { Rm[k] = (k % 4 == 0) ? 1.f : 0.f; Km[k] = 0.f; }
                Km[0] = a.focal; Km[4] = a.focal; Km[8] = 1.f;
            }
            float P[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) P[i] = Rm[i * 3 + 0] * X0 + Rm[i * 3 + 1] * X1 + Rm[i * 3 + 2] * X2 + ct[i];
            const float x0 = P[0] / P[2], x1 = P[1] / P[2], x2 = P[2] / P[2];
            float u = Km[0] * x0 + Km[1] * x1 + Km[2] * x2;
            float vv = Km[3] * x0 + Km[4] * x1 + Km[5] * x2;
            if (a.normalize) { u = u / (a.img_res / 2.0f); vv = vv / (a.img_res / 2.0f); }
            a.joints2d[(size_t)b * a.ld_j2d + t * 2 + 0] = u;
            a.joints2d[(size_t)b * a.ld_j2d + t * 2 + 1] = vv;
        }
    }
}

// smplx.lbs.batch_rodrigues (0.1.28): angle = |r + 1e-8| (the epsilon is added to every component), axis = r / angle,
// R = I + sin(angle) K + (1 - cos(angle)) K K with K the cross-product matrix of the axis
__global__ void rodrigues_kernel(const float* __restrict__ aa, float* __restrict__ rot, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = aa[i * 3 + 0], y = aa[i * 3 + 1], z = aa[i * 3 + 2];
    const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float rx = x / angle, ry = y / angle, rz = z / angle;
    const float c = cosf(angle), sn = sinf(angle);
    const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    float* R = rot + (size_t)i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) {
            const float kk = K[r * 3 + 0] * K[0 * 3 + cidx] + K[r * 3 + 1] * K[1 * 3 + cidx] + K[r * 3 + 2] * K[2 * 3 + cidx];
            R[r * 3 + cidx] = ((r == cidx) ? 1.0f : 0.0f) + sn * K[r * 3 + cidx] + (1.0f - c) * kk;
        }
}

int launch_rodrigues(const float* aa, float* rot, int n, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "smpl_rodrigues", 0.0, 4.0 * n * 12);
    hipLaunchKernelGGL(rodrigues_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx.stream, aa, rot, n);
    return (int)hipGetLastError();
}

// The skinning launch of both entry points.  a.skin_split: -1 = by the number of image tiles, 0 / 1 = never / always the
// three-waves-per-vertex-group variant (same bits either way; tests force both).
static void launch_skin(const SmplDev& m, const SmplArgs& a, long ld_verts, int tiles, int G, const LaunchCtx& ctx) {
    const bool split = a.skin_split < 0 ? tiles <= SKIN_SPLIT_MAX_TILES : a.skin_split != 0;
    if (split)
        hipLaunchKernelGGL(smpl_skin_kernel<true>, dim3(G, tiles), dim3(192), (IT * SKIN_LD + 3 * 16 * 64) * sizeof(float), ctx.stream,
                           m.dirsT, m.wT, a.pose_feat, a.A, a.vertices, ld_verts, m.V, a.B, G);
    else
        hipLaunchKernelGGL(smpl_skin_kernel<false>, dim3((G + 3) / 4, tiles), dim3(256), 4 * IT * SKIN_LD * sizeof(float), ctx.stream,
                           m.dirsT, m.wT, a.pose_feat, a.A, a.vertices, ld_verts, m.V, a.B, G);
}

// smplx.SMPL.forward(pose2rot=False) without the 49-joint wrapper: vertices + the 24 posed kinematic-chain joints
// (`.joints[:, :24]` of the reference's smpl_native / body_model_orig, spec/trainer.py:249-254, compute_error.py:156-160)
int launch_smpl_native(const SmplDev& m, const SmplArgs& a, float* joints24, const LaunchCtx& ctx) {
    const int B = a.B, V = m.V;
    const long ld_verts = a.ld_verts > 0 ? a.ld_verts : (long)V * 3;
    const int tiles = (B + IT - 1) / IT, G = (V + 31) / 32;
    {
        ProfScope ps(ctx, "smpl_pose_chain", 0.0, 4.0 * B * (216 + 10 + 207 + 288 + 72));
        hipLaunchKernelGGL(smpl_pose_kernel<false>, dim3(B), dim3(64), 0, ctx.stream, a.rotmat, a.betas, m.J_template,
                           m.J_shapedirs, m.parents, a.pose_feat, a.A, joints24 ? joints24 : a.posed_j, HeadFinal{});
    }
    if (a.vertices) {
        const double flops = 2.0 * (double)B * V * (3.0 * 207 + 30 + 288 + 9);
        // algorithmic HBM bytes: the body model once + every output once (the kernel re-reads its model slice once per
        // tile of IT images - that re-use is served by L2 - and is not counted)
        const double bytes = 4.0 * ((double)B * V * 3 + (double)V * (3.0 * 207 + 3 + 30 + 24) + (double)B * (207 + 10 + 288));
        ProfScope ps(ctx, "smpl_skin_lbs", flops, bytes);
        launch_skin(m, a, ld_verts, tiles, G, ctx);
    }
    return (int)hipGetLastError();
}

int launch_smpl(const SmplDev& m, const SmplArgs& a, const LaunchCtx& ctx) {
    const int B = a.B, V = m.V;
    const long ld_verts = a.ld_verts > 0 ? a.ld_verts : (long)V * 3;
    const int tiles = (B + IT - 1) / IT, G = (V + 31) / 32;
    if (!a.pose_done) {   // (pose_done: the fused HMR tail - head.hip - ran the pose chains as its epilogue)
        ProfScope ps(ctx, "smpl_pose_chain", 0.0, 4.0 * B * (216 + 10 + 207 + 288 + 72));
        if (a.final_)
            hipLaunchKernelGGL(smpl_pose_kernel<true>, dim3(B), dim3(64), 0, ctx.stream, a.rotmat, a.betas, m.J_template,
                               m.J_shapedirs, m.parents, a.pose_feat, a.A, a.posed_j, *a.final_);
        else
            hipLaunchKernelGGL(smpl_pose_kernel<false>, dim3(B), dim3(64), 0, ctx.stream, a.rotmat, a.betas, m.J_template,
                               m.J_shapedirs, m.parents, a.pose_feat, a.A, a.posed_j, HeadFinal{});
    }
    {
        const double flops = 2.0 * (double)B * V * (3.0 * 207 + 30 + 288 + 9);
        // algorithmic HBM bytes: the body model once + every output once (the kernel re-reads its model slice once per
        // tile of IT images - that re-use is served by L2 - and is not counted)
        const double bytes = 4.0 * ((double)B * V * 3 + (double)V * (3.0 * 207 + 3 + 30 + 24) + (double)B * (207 + 10 + 288));
        ProfScope ps(ctx, "smpl_skin_lbs", flops, bytes);
        launch_skin(m, a, ld_verts, tiles, G, ctx);
    }
    {
        JointArgs j;
        j.verts = a.vertices; j.posed_j = a.posed_j; j.J_extra = m.J_extra; j.extra_ids = m.extra_ids;
        j.joint_map = m.joint_map; j.cam = a.cam; j.R = a.cam_rotmat; j.K = a.cam_intrinsics;
        j.bbox_scale = a.bbox_scale; j.bbox_center = a.bbox_center; j.img_w = a.img_w; j.img_h = a.img_h;
        j.joints3d = a.joints3d; j.joints2d = a.joints2d; j.cam_t = a.cam_t;
        j.ld_verts = ld_verts; j.ld_j3d = a.ld_j3d; j.ld_j2d = a.ld_j2d; j.ld_camt = a.ld_camt;
        j.V = V; j.mode = a.mode; j.normalize = a.normalize_joints2d; j.focal = a.focal_length; j.img_res = a.img_res;
        // algorithmic bytes: every mesh once + the 9 x V extra-joint regressor once (its per-image re-reads are L2 hits) + outputs
        ProfScope ps(ctx, "smpl_joints_project", 2.0 * B * V * 27.0, 4.0 * ((double)B * V * 3 + 9.0 * V + (double)B * 49 * 5));
        hipLaunchKernelGGL(smpl_joints_kernel, dim3(B), dim3(JT), 0, ctx.stream, j);
    }
    return (int)hipGetLastError();
}

}  // namespace specmi
