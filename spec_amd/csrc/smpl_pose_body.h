// smpl_pose_body.h - the SMPL pose chain of one image on one wave (smpl.hip: kernel 1), as a device function shared by
// smpl_pose_kernel and by the last arriver of the fused HMR tail (head.hip: tail_gemv_kernel) - same arithmetic, same bits.
// Reference: smplx 0.1.28 lbs(pose2rot=False) batch_rigid_transform, call site spec/models/hmr.py:101-120.
#pragma once
#include "specmi_internal.h"

namespace specmi {

constexpr int IT = 32;                     // images per tile = columns of one MFMA
constexpr int KQ = SMPL_KQ;                // K = 224 = 207 pose features + 10 betas + 1 (template) + 6 zeros = 112 MFMA steps = 28 quads
constexpr int FEAT_TILE = KQ * 64 * 4;     // floats of one image tile of features  [quad][lane][4]   (224 per image)
constexpr int SKIN_LD = 98;                // floats per image row of the output transpose (96 + 2: even, so rows stay 8-byte aligned)
constexpr int SKIN_SPLIT_MAX_TILES = 2;    // image tiles up to which the skin kernel runs three waves per vertex group (measured: DESIGN.md section 4)
constexpr int A_TILE = 12 * 3 * 64 * 4;    // floats of one image tile of transforms [entry][quad][lane][4] (288 per image)


// a0 b0 + a1 b1 + a2 b2 with the roundings written out (one product, two fused steps): the two instantiations of the pose kernel
// must agree to the bit, and left to itself the compiler contracts the same source differently in each (-ffp-contract=fast
// with aggressive FMA fusion picks by context)
__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
#pragma clang fp contract(off)
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

// FUSED: the wave first does head_final's work for its image (head.hip: rot6d -> rotmat of joint j on lane j, the pred_* output
// gather, the rot / betas / cam workspaces) and keeps rotmat / betas in registers - the regressor's last graph node folded into
// this one; same arithmetic (rot6d_joint), same bits.
// COHERENT: the regressor state was written earlier in the SAME launch by other workgroups (the fused tail of head.hip: write-through
// stores): it is read with agent-scope (sc1) loads instead of plain ones.  b = image, j = lane (0..63) of the wave that owns it.
template <bool FUSED, bool COHERENT = false>
__device__ __forceinline__ void smpl_pose_body(const int b, const int j, const float* __restrict__ rotmat, const float* __restrict__ betas,
                                               const float* __restrict__ Jt, const float* __restrict__ Jd,
                                               const int* __restrict__ parents, float* __restrict__ feat,
                                               float* __restrict__ Afrag, float* __restrict__ posed_j, const HeadFinal& fin) {
#pragma clang fp contract(off)   // every fused step below is an explicit fmaf (see dot3)
    const bool act = j < 24;
    const int jj = act ? j : 0;
    int par = (act && j > 0) ? parents[jj] : -1;
    int depth = 0;
    for (int pp = par; pp >= 0; pp = (pp > 0 ? parents[pp] : -1)) ++depth;
    int maxd = act ? depth : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxd = max(maxd, __shfl_xor(maxd, o, 64));

    float R[9], beta[10];
    if (FUSED) {
        const float* st = fin.state + (size_t)b * fin.ld_state;
        auto S = [&](int i) { return COHERENT ? __hip_atomic_load(st + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : st[i]; };
        for (int i = j; i < 157; i += 64) {
            const float v = S(i);
            if (i < 144) {
                if (fin.pred_pose_6d) fin.pred_pose_6d[(size_t)b * fin.ld_p6d + i] = v;
            } else if (i < 154) {
                if (fin.pred_shape) fin.pred_shape[(size_t)b * fin.ld_shape + i - 144] = v;
                if (fin.betas_ws) fin.betas_ws[(size_t)b * 10 + i - 144] = v;
            } else {
                if (fin.pred_cam) fin.pred_cam[(size_t)b * fin.ld_cam + i - 154] = v;
                if (fin.cam_ws) fin.cam_ws[(size_t)b * 3 + i - 154] = v;
            }
        }
        float s6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) s6[k] = S(6 * jj + k);
        rot6d_joint(s6, R);
        if (act) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (fin.pred_pose) fin.pred_pose[(size_t)b * fin.ld_pose + j * 9 + k] = R[k];
                if (fin.rot_ws) fin.rot_ws[((size_t)b * 24 + j) * 9 + k] = R[k];
            }
        }
#pragma unroll
        for (int l = 0; l < 10; ++l) beta[l] = S(144 + l);
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = rotmat[((size_t)b * 24 + jj) * 9 + k];
#pragma unroll
        for (int l = 0; l < 10; ++l) beta[l] = betas[(size_t)b * 10 + l];
    }
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
                    G[r * 4 + c] = dot3(P[r * 4 + 0], R[0 * 3 + c], P[r * 4 + 1], R[1 * 3 + c], P[r * 4 + 2], R[2 * 3 + c]);
                G[r * 4 + 3] = dot3(P[r * 4 + 0], rel[0], P[r * 4 + 1], rel[1], P[r * 4 + 2], rel[2]) + P[r * 4 + 3];
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
            at[(size_t)(r * 4 + 3) * 768] = G[r * 4 + 3] - dot3(G[r * 4 + 0], J[0], G[r * 4 + 1], J[1], G[r * 4 + 2], J[2]);
        }
    }
}


}  // namespace specmi
