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
#include "specmi_internal.h"

namespace specmi {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IT = 32;                     // images per tile = columns of one MFMA
constexpr int KQ = SMPL_KQ;                // K = 224 = 207 pose features + 10 betas + 1 (template) + 6 zeros = 112 MFMA steps = 28 quads
constexpr int FEAT_TILE = KQ * 64 * 4;     // floats of one image tile of features  [quad][lane][4]   (224 per image)
constexpr int SKIN_LD = 98;                // floats per image row of the output transpose (96 + 2: even, so rows stay 8-byte aligned)
constexpr int A_TILE = 12 * 3 * 64 * 4;    // floats of one image tile of transforms [entry][quad][lane][4] (288 per image)


__global__ void __launch_bounds__(64) smpl_pose_kernel(const float* __restrict__ rotmat, const float* __restrict__ betas,
                                                        const float* __restrict__ Jt, const float* __restrict__ Jd,
                                                        const int* __restrict__ parents, float* __restrict__ feat,
                                                        float* __restrict__ Afrag, float* __restrict__ posed_j) {
    const int b = blockIdx.x;
    const int j = threadIdx.x;
    const bool act = j < 24;
    const int jj = act ? j : 0;
    int par = (act && j > 0) ? parents[jj] : -1;
    int depth = 0;
    for (int pp = par; pp >= 0; pp = (pp > 0 ? parents[pp] : -1)) ++depth;
    int maxd = act ? depth : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxd = max(maxd, __shfl_xor(maxd, o, 64));

    float R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = rotmat[((size_t)b * 24 + jj) * 9 + k];

    float beta[10];
#pragma unroll
    for (int l = 0; l < 10; ++l) beta[l] = betas[(size_t)b * 10 + l];
    // row b of the skin kernel's feature operand, in fragment order: k < 207 pose features, 207..216 betas, 217 the constant 1
    // that multiplies v_template (218..223 stay 0 from the allocation)
    float* const ft = feat + (size_t)(b / IT) * FEAT_TILE;
    if (j < 10) ft[frag_slot(207 + j, b % IT)] = beta[j];
    if (j == 10) ft[frag_slot(217, b % IT)] = 1.0f;

    float J[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 10; ++l) s = fmaf(beta[l], Jd[(jj * 3 + c) * 10 + l], s);
        J[c] = Jt[jj * 3 + c] + s;
    }

    // pose feature (R_j - I) for j >= 1
    if (act && j > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) ft[frag_slot((j - 1) * 9 + k, b % IT)] = R[k] - ((k % 4 == 0) ? 1.0f : 0.0f);
    }

    const int src = par >= 0 ? par : 0;
    float rel[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float pj = __shfl(J[c], src, 64);
        rel[c] = (par >= 0) ? J[c] - pj : J[c];
    }
    // local transform L = [R | rel]; world transform G starts as L (root) and is finalised level by level
    float G[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        G[r * 4 + 0] = R[r * 3 + 0]; G[r * 4 + 1] = R[r * 3 + 1]; G[r * 4 + 2] = R[r * 3 + 2]; G[r * 4 + 3] = rel[r];
    }
    for (int d = 1; d <= maxd; ++d) {
        float P[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) P[e] = __shfl(G[e], src, 64);
        if (act && depth == d) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    G[r * 4 + c] = P[r * 4 + 0] * R[0 * 3 + c] + P[r * 4 + 1] * R[1 * 3 + c] + P[r * 4 + 2] * R[2 * 3 + c];
                G[r * 4 + 3] = P[r * 4 + 0] * rel[0] + P[r * 4 + 1] * rel[1] + P[r * 4 + 2] * rel[2] + P[r * 4 + 3];
            }
        }
    }
    if (act) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            posed_j[((size_t)b * 24 + j) * 3 + r] = G[r * 4 + 3];
            // entry e = 4 r + c of joint j's relative transform: operand [e] of the skin kernel, k = joint, column = image
            float* const at = Afrag + (size_t)(b / IT) * A_TILE + frag_slot(j, b % IT);
            at[(size_t)(r * 4 + 0) * 768] = G[r * 4 + 0]; at[(size_t)(r * 4 + 1) * 768] = G[r * 4 + 1]; at[(size_t)(r * 4 + 2) * 768] = G[r * 4 + 2];
            at[(size_t)(r * 4 + 3) * 768] = G[r * 4 + 3] - (G[r * 4 + 0] * J[0] + G[r * 4 + 1] * J[1] + G[r * 4 + 2] * J[2]);
        }
    }
}

// grid (ceil(groups / 4), image tiles); wave w of a workgroup owns vertex group 4 blockIdx.x + w and needs nothing from the
// other waves (its LDS slice is private: no barrier).
__global__ void __launch_bounds__(256) smpl_skin_kernel(const float* __restrict__ dirsT, const float* __restrict__ wT,
                                                         const float* __restrict__ feat, const float* __restrict__ Afrag,
                                                         float* __restrict__ verts, long ld_verts, int V, int B, int G) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int tile = blockIdx.y;
    const f32x4* __restrict__ fq = reinterpret_cast<const f32x4*>(feat + (size_t)tile * FEAT_TILE) + lane;     // [quad][64]
    const f32x4* __restrict__ dq = reinterpret_cast<const f32x4*>(dirsT + (size_t)g * 3 * FEAT_TILE) + lane;   // [coord][quad][64]

    // v_posed[vertex][image] per coordinate: A operand = the model's directions (row = vertex), B operand = features (column = image)
    f32x16 vp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) vp[c][r] = 0.f;
    constexpr int PD = 3;                      // quads in flight ahead of the MFMAs (12 MFMAs = 768 cycles each)
    f32x4 fb[PD], d0[PD], d1[PD], d2[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        fb[i] = fq[i * 64]; d0[i] = dq[i * 64]; d1[i] = dq[(KQ + i) * 64]; d2[i] = dq[(2 * KQ + i) * 64];
    }
#pragma unroll
    for (int s4 = 0; s4 < KQ; ++s4) {
        const int cur = s4 % PD;
        const f32x4 f = fb[cur], a0 = d0[cur], a1 = d1[cur], a2 = d2[cur];
        if (s4 + PD < KQ) {
            fb[cur] = fq[(s4 + PD) * 64]; d0[cur] = dq[(s4 + PD) * 64];
            d1[cur] = dq[(KQ + s4 + PD) * 64]; d2[cur] = dq[(2 * KQ + s4 + PD) * 64];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vp[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], f[q], vp[0], 0, 0, 0);
            vp[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], f[q], vp[1], 0, 0, 0);
            vp[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[q], f[q], vp[2], 0, 0, 0);
        }
    }

    // blended transform entry e = 4 c + d of (vertex, image): sum_j w[vertex][j] A[image][j][e]; then the skinned coordinate c
    const f32x4* __restrict__ wq = reinterpret_cast<const f32x4*>(wT + (size_t)g * 768) + lane;                // [quad (3)][64]
    const f32x4* __restrict__ aq = reinterpret_cast<const f32x4*>(Afrag + (size_t)tile * A_TILE) + lane;       // [entry][quad][64]
    const f32x4 w0 = wq[0], w1 = wq[64], w2 = wq[128];
    // the wave's 32 images x 32 vertices x 3 coordinates go through LDS (wave-private: no barrier) so that global memory sees,
    // per image, the 384 contiguous bytes of the 32 vertices instead of 4-byte pieces 80 KB apart
    extern __shared__ __attribute__((aligned(16))) float skin_lds[];
    float* const tw = skin_lds + (threadIdx.x >> 6) * (IT * SKIN_LD);
    const int vloc0 = 4 * (lane >> 5);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
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
        // accumulator r of this lane = image lane % 32, vertex vloc0 + (r & 3) + 8 (r >> 2) of the group
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float o = T[0][r] * vp[0][r] + T[1][r] * vp[1][r] + T[2][r] * vp[2][r] + T[3][r];
            tw[(lane & 31) * SKIN_LD + (vloc0 + (r & 3) + 8 * (r >> 2)) * 3 + c] = o;
        }
    }
    // one wave's LDS instructions execute in order: the reads below see the writes above
    const int nimg = min(IT, B - tile * IT);
    const int nfl = min(96, (V - g * 32) * 3);                       // floats of this group that exist
    float* const obase = verts + (size_t)tile * IT * ld_verts + (size_t)g * 96;
    if (((ld_verts & 1) == 0) && ((reinterpret_cast<uintptr_t>(verts) & 7) == 0)) {
        // rows of the output start on 8-byte boundaries: 48 lanes x 8 bytes per image
        const bool on = 2 * lane + 1 < nfl;
        const bool half = 2 * lane + 1 == nfl;                        // (V * 3 odd: the last float alone)
#pragma unroll 4
        for (int i = 0; i < nimg; ++i) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(tw + i * SKIN_LD + 2 * (lane < 48 ? lane : 0));
            float* const o = obase + (size_t)i * ld_verts + 2 * lane;
            if (on) *reinterpret_cast<f32x2*>(o) = t;
            else if (half) o[0] = t[0];
        }
    } else {
        for (int i = 0; i < nimg; ++i) {
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

__global__ void __launch_bounds__(256) smpl_joints_kernel(const JointArgs a) {
    __shared__ float red[4][27];
    __shared__ float J54[54][3];
    __shared__ float ct[3];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* vb = a.verts + (size_t)b * a.ld_verts;
    float acc[9][3];
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[e][0] = acc[e][1] = acc[e][2] = 0.f;
#pragma unroll 4   // 12 independent loads per iteration: keep several iterations in flight (one workgroup per image is latency-bound)
    for (int v = t; v < a.V; v += 256) {
        const float x = vb[v * 3 + 0], y = vb[v * 3 + 1], z = vb[v * 3 + 2];
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const float wgt = a.J_extra[(size_t)e * a.V + v];
            acc[e][0] = fmaf(wgt, x, acc[e][0]);
            acc[e][1] = fmaf(wgt, y, acc[e][1]);
            acc[e][2] = fmaf(wgt, z, acc[e][2]);
        }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = wsum(acc[e][c]);
            if ((t & 63) == 0) red[t >> 6][e * 3 + c] = s;
        }
    if (t < 72) J54[t / 3][t % 3] = a.posed_j[(size_t)b * 72 + t];
    if (t >= 72 && t < 72 + 63) {
        const int i = t - 72, e = i / 3, c = i % 3;
        J54[24 + e][c] = vb[(size_t)a.extra_ids[e] * 3 + c];
    }
    if (t == 255) {
        const float s = a.cam[b * 3 + 0], tx = a.cam[b * 3 + 1], ty = a.cam[b * 3 + 2];
        if (a.mode == 0) {  // convert_pare_to_full_img_cam, res = 224 literal
            const float f = a.K[(size_t)b * 9];
            const float bh = a.bbox_scale[b] * 200.0f;
            const float r = bh / 224.0f;
            const float tz = 2.0f * f / (r * 224.0f * s);
            const float cx = 2.0f * (a.bbox_center[b * 2 + 0] - (a.img_w[b] / 2.0f)) / (s * bh);
            const float cy = 2.0f * (a.bbox_center[b * 2 + 1] - (a.img_h[b] / 2.0f)) / (s * bh);
            ct[0] = tx + cx; ct[1] = ty + cy; ct[2] = tz;
        } else {            // convert_weak_perspective_to_perspective
            ct[0] = tx; ct[1] = ty; ct[2] = 2.0f * a.focal / (a.img_res * s + 1e-9f);
        }
    }
    __syncthreads();
    if (t < 27) J54[45 + t / 3][t % 3] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    __syncthreads();
    if (t < 3 && a.cam_t) a.cam_t[(size_t)b * a.ld_camt + t] = ct[t];
    if (t < 49) {
        const int jm = a.joint_map[t];
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

// smplx.SMPL.forward(pose2rot=False) without the 49-joint wrapper: vertices + the 24 posed kinematic-chain joints
// (`.joints[:, :24]` of the reference's smpl_native / body_model_orig, spec/trainer.py:249-254, compute_error.py:156-160)
int launch_smpl_native(const SmplDev& m, const SmplArgs& a, float* joints24, const LaunchCtx& ctx) {
    const int B = a.B, V = m.V;
    const long ld_verts = a.ld_verts > 0 ? a.ld_verts : (long)V * 3;
    const int tiles = (B + IT - 1) / IT, G = (V + 31) / 32;
    {
        ProfScope ps(ctx, "smpl_pose_chain", 0.0, 4.0 * B * (216 + 10 + 207 + 288 + 72));
        hipLaunchKernelGGL(smpl_pose_kernel, dim3(B), dim3(64), 0, ctx.stream, a.rotmat, a.betas, m.J_template,
                           m.J_shapedirs, m.parents, a.pose_feat, a.A, joints24 ? joints24 : a.posed_j);
    }
    if (a.vertices) {
        const double flops = 2.0 * (double)B * V * (3.0 * 207 + 30 + 288 + 9);
        // algorithmic HBM bytes: the body model once + every output once (the kernel re-reads its model slice once per
        // tile of IT images - that re-use is served by L2 - and is not counted)
        const double bytes = 4.0 * ((double)B * V * 3 + (double)V * (3.0 * 207 + 3 + 30 + 24) + (double)B * (207 + 10 + 288));
        ProfScope ps(ctx, "smpl_skin_lbs", flops, bytes);
        hipLaunchKernelGGL(smpl_skin_kernel, dim3((G + 3) / 4, tiles), dim3(256), 4 * IT * SKIN_LD * sizeof(float), ctx.stream, m.dirsT,
                           m.wT, a.pose_feat, a.A, a.vertices, ld_verts, V, B, G);
    }
    return (int)hipGetLastError();
}

int launch_smpl(const SmplDev& m, const SmplArgs& a, const LaunchCtx& ctx) {
    const int B = a.B, V = m.V;
    const long ld_verts = a.ld_verts > 0 ? a.ld_verts : (long)V * 3;
    const int tiles = (B + IT - 1) / IT, G = (V + 31) / 32;
    {
        ProfScope ps(ctx, "smpl_pose_chain", 0.0, 4.0 * B * (216 + 10 + 207 + 288 + 72));
        hipLaunchKernelGGL(smpl_pose_kernel, dim3(B), dim3(64), 0, ctx.stream, a.rotmat, a.betas, m.J_template,
                           m.J_shapedirs, m.parents, a.pose_feat, a.A, a.posed_j);
    }
    {
        const double flops = 2.0 * (double)B * V * (3.0 * 207 + 30 + 288 + 9);
        // algorithmic HBM bytes: the body model once + every output once (the kernel re-reads its model slice once per
        // tile of IT images - that re-use is served by L2 - and is not counted)
        const double bytes = 4.0 * ((double)B * V * 3 + (double)V * (3.0 * 207 + 3 + 30 + 24) + (double)B * (207 + 10 + 288));
        ProfScope ps(ctx, "smpl_skin_lbs", flops, bytes);
        hipLaunchKernelGGL(smpl_skin_kernel, dim3((G + 3) / 4, tiles), dim3(256), 4 * IT * SKIN_LD * sizeof(float), ctx.stream, m.dirsT,
                           m.wT, a.pose_feat, a.A, a.vertices, ld_verts, V, B, G);
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
        hipLaunchKernelGGL(smpl_joints_kernel, dim3(B), dim3(256), 0, ctx.stream, j);
    }
    return (int)hipGetLastError();
}

}  // namespace specmi
