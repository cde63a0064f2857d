// eval.hip - evaluation metrics on the device, directly on the path's outputs.
//
// Replaces the host NumPy scoring that follows the hot path in the reference:
//   * eval_single  (spec/utils/compute_error.py:52-86; validation step spec/trainer.py:272-316):
//     joints = J_regressor @ vertices, pelvis alignment, joint selection (H36M_TO_J14), MPJPE,
//     PA-MPJPE (similarity Procrustes), pelvis-aligned V2V - all in millimetres.
//   * eval_j_24    (spec/utils/compute_error.py:33-49): pelvis-aligned MPJPE / PA-MPJPE of two
//     joint sets; with joints = J_regressor(24xV) @ vertices this is the README's W-MPJPE.
// The reference copies 21 MB of vertices per batch to the host for this; here the vertices never
// leave HBM and only 3 floats per image come back.
//
// One workgroup per image: joint regression is a set of length-V dot products (wave shuffle +
// LDS reduction, the vertices stay L2-resident between joints), then one lane solves the
// Procrustes problem with Horn's quaternion method (largest eigenpair of a symmetric 4x4 via
// cyclic Jacobi in fp64) - the same optimum as the SVD/Kabsch solution with its det(R)=+1 fix.
#include "specmi_internal.h"

namespace specmi {

__device__ __forceinline__ float block_sum_256(float v, float* red) {   // blockDim.x == 256
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Largest eigenvalue / eigenvector of a symmetric 4x4 (cyclic Jacobi, fp64).
__device__ void sym4_max_eig(double A[4][4], double q[4], double* lam) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = 0;
        for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        double diag = 0;
        for (int i = 0; i < 4; ++i) diag += A[i][i] * A[i][i];
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < 4; ++p)
            for (int r = p + 1; r < 4; ++r) {
                if (fabs(A[p][r]) < 1e-300) continue;
                const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akr = A[k][r];
                    A[k][p] = c * akp - s * akr;
                    A[k][r] = s * akp + c * akr;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], ark = A[r][k];
                    A[p][k] = c * apk - s * ark;
                    A[r][k] = s * apk + c * ark;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkr = V[k][r];
                    V[k][p] = c * vkp - s * vkr;
                    V[k][r] = s * vkp + c * vkr;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i) if (A[i][i] > A[best][best]) best = i;
    *lam = A[best][best];
    for (int k = 0; k < 4; ++k) q[k] = V[k][best];
}

// MPJPE and PA-MPJPE (mm) of N pelvis-aligned joints p, g (N x 3 floats, stride 3).
__device__ void joint_errors(const float* p, const float* g, int N, float* mpjpe, float* pampjpe) {
    double e = 0, mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
    for (int i = 0; i < N; ++i) {
        double d = 0;
        for (int c = 0; c < 3; ++c) {
            const double df = (double)p[i * 3 + c] - (double)g[i * 3 + c];
            d += df * df;
            mu1[c] += p[i * 3 + c];
            mu2[c] += g[i * 3 + c];
        }
        e += sqrt(d);
    }
    *mpjpe = (float)(e / N * 1000.0);
    for (int c = 0; c < 3; ++c) { mu1[c] /= N; mu2[c] /= N; }
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0;
    for (int i = 0; i < N; ++i) {
        double x1[3], x2[3];
        for (int c = 0; c < 3; ++c) { x1[c] = p[i * 3 + c] - mu1[c]; x2[c] = g[i * 3 + c] - mu2[c]; var1 += x1[c] * x1[c]; }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a][b] += x1[a] * x2[b];
    }
    double Nm[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double q[4], lam;
    sym4_max_eig(Nm, q, &lam);
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double R[3][3] = {{w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)},
                            {2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)},
                            {2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z}};
    const double scale = lam / var1;
    double pe = 0;
    for (int i = 0; i < N; ++i) {
        double d = 0;
        for (int a = 0; a < 3; ++a) {
            double v = 0;
            for (int b = 0; b < 3; ++b) v += R[a][b] * (p[i * 3 + b] - mu1[b]);
            const double df = scale * v + mu2[a] - g[i * 3 + a];
            d += df * df;
        }
        pe += sqrt(d);
    }
    *pampjpe = (float)(pe / N * 1000.0);
}

constexpr int kMaxJ = 32;

__global__ void __launch_bounds__(256) eval_mesh_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         int V, const float* __restrict__ Jr, int J,
                                                         const int* __restrict__ sel, int nsel, float* __restrict__ mpjpe,
                                                         float* __restrict__ pampjpe, float* __restrict__ v2v) {
    __shared__ float red[4];
    __shared__ float jp[kMaxJ][3], jg[kMaxJ][3];
    __shared__ float sp[kMaxJ * 3], sg[kMaxJ * 3];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* pv = pred + (size_t)b * V * 3;
    const float* gv = gt + (size_t)b * V * 3;
    for (int j = 0; j < J; ++j) {
        float a[6] = {0, 0, 0, 0, 0, 0};
        for (int v = t; v < V; v += 256) {
            const float w = Jr[(size_t)j * V + v];
            a[0] = fmaf(w, pv[v * 3 + 0], a[0]); a[1] = fmaf(w, pv[v * 3 + 1], a[1]); a[2] = fmaf(w, pv[v * 3 + 2], a[2]);
            a[3] = fmaf(w, gv[v * 3 + 0], a[3]); a[4] = fmaf(w, gv[v * 3 + 1], a[4]); a[5] = fmaf(w, gv[v * 3 + 2], a[5]);
        }
        for (int c = 0; c < 6; ++c) {
            const float s = block_sum_256(a[c], red);
            if (t == 0) { if (c < 3) jp[j][c] = s; else jg[j][c - 3] = s; }
        }
    }
    __syncthreads();
    // pelvis-aligned vertex-to-vertex error
    const float ppx = jp[0][0], ppy = jp[0][1], ppz = jp[0][2], gpx = jg[0][0], gpy = jg[0][1], gpz = jg[0][2];
    float acc = 0.f;
    for (int v = t; v < V; v += 256) {
        const float dx = (gv[v * 3 + 0] - gpx) - (pv[v * 3 + 0] - ppx);
        const float dy = (gv[v * 3 + 1] - gpy) - (pv[v * 3 + 1] - ppy);
        const float dz = (gv[v * 3 + 2] - gpz) - (pv[v * 3 + 2] - ppz);
        acc += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    const float vs = block_sum_256(acc, red);
    if (t < nsel * 3) {
        const int i = t / 3, c = t % 3, j = sel ? sel[i] : i;
        sp[t] = jp[j][c] - jp[0][c];
        sg[t] = jg[j][c] - jg[0][c];
    }
    __syncthreads();
    if (t == 0) {
        if (v2v) v2v[b] = vs / (float)V * 1000.0f;
        float m, pa;
        joint_errors(sp, sg, nsel, &m, &pa);
        if (mpjpe) mpjpe[b] = m;
        if (pampjpe) pampjpe[b] = pa;
    }
}

__global__ void __launch_bounds__(64) eval_joints_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int B,
                                                          int J, float* __restrict__ mpjpe, float* __restrict__ pampjpe) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float p[kMaxJ * 3], g[kMaxJ * 3];
    for (int i = 0; i < J; ++i)
        for (int c = 0; c < 3; ++c) {
            p[i * 3 + c] = pred[((size_t)b * J + i) * 3 + c] - pred[(size_t)b * J * 3 + c];
            g[i * 3 + c] = gt[((size_t)b * J + i) * 3 + c] - gt[(size_t)b * J * 3 + c];
        }
    float m, pa;
    joint_errors(p, g, J, &m, &pa);
    if (mpjpe) mpjpe[b] = m;
    if (pampjpe) pampjpe[b] = pa;
}

// joints (B,J,3) = J_regressor (J,V) @ vertices (B,V,3)  (torch.einsum('bik,ji->bjk'), compute_error.py:184,187;
// torch.matmul(J_regressor_batch, vertices), :53,58): one workgroup per (image, joint)
__global__ void __launch_bounds__(256) regress_joints_kernel(const float* __restrict__ verts, int V,
                                                              const float* __restrict__ Jr, int J, float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.x / J, j = blockIdx.x % J, t = threadIdx.x;
    const float* v = verts + (size_t)b * V * 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = t; i < V; i += 256) {
        const float w = Jr[(size_t)j * V + i];
        a0 = fmaf(w, v[i * 3 + 0], a0); a1 = fmaf(w, v[i * 3 + 1], a1); a2 = fmaf(w, v[i * 3 + 2], a2);
    }
    const float s0 = block_sum_256(a0, red), s1 = block_sum_256(a1, red), s2 = block_sum_256(a2, red);
    if (t == 0) {
        float* o = out + ((size_t)b * J + j) * 3;
        o[0] = s0; o[1] = s1; o[2] = s2;
    }
}

int launch_regress_joints(const float* verts, int B, int V, const float* Jr, int J, float* out, const LaunchCtx& ctx) {
    ProfScope ps(ctx, "regress_joints", 2.0 * B * (double)V * 3 * J, 4.0 * ((double)B * V * 3 + (double)J * V + (double)B * J * 3));
    hipLaunchKernelGGL(regress_joints_kernel, dim3(B * J), dim3(256), 0, ctx.stream, verts, V, Jr, J, out);
    return (int)hipGetLastError();
}

// out[b,n,:] = R[b] @ x[b,n,:]  (torch.bmm(R, x.transpose(2,1)).transpose(2,1), compute_error.py:164-165,189)
__global__ void __launch_bounds__(256) rotate_points_kernel(const float* __restrict__ R, const float* __restrict__ x,
                                                             int N, long total, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float* r = R + (i / N) * 9;
    const float x0 = x[i * 3 + 0], x1 = x[i * 3 + 1], x2 = x[i * 3 + 2];
#pragma unroll
    for (int c = 0; c < 3; ++c) out[i * 3 + c] = r[c * 3 + 0] * x0 + r[c * 3 + 1] * x1 + r[c * 3 + 2] * x2;
}

int launch_rotate_points(const float* R, const float* x, int B, int N, float* out, const LaunchCtx& ctx) {
    const long total = (long)B * N;
    ProfScope ps(ctx, "rotate_points", 18.0 * total, 4.0 * (6.0 * total + 9.0 * B));
    hipLaunchKernelGGL(rotate_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx.stream, R, x, N, total, out);
    return (int)hipGetLastError();
}

int launch_eval_mesh(const float* pred, const float* gt, int B, int V, const float* Jr, int J, const int* sel, int nsel,
                     float* mpjpe, float* pampjpe, float* v2v, const LaunchCtx& ctx) {
    if (J > kMaxJ || nsel > kMaxJ || nsel < 1 || J < 1) return (int)hipErrorInvalidValue;
    ProfScope ps(ctx, "eval_mesh_metrics", 2.0 * B * (double)V * 6 * J, 4.0 * B * ((double)V * 6 * (J + 1) + (double)J * V));
    hipLaunchKernelGGL(eval_mesh_kernel, dim3(B), dim3(256), 0, ctx.stream, pred, gt, V, Jr, J, sel, nsel, mpjpe, pampjpe, v2v);
    return (int)hipGetLastError();
}

int launch_eval_joints(const float* pred, const float* gt, int B, int J, float* mpjpe, float* pampjpe, const LaunchCtx& ctx) {
    if (J > kMaxJ || J < 1) return (int)hipErrorInvalidValue;
    ProfScope ps(ctx, "eval_joint_metrics", 0.0, 4.0 * B * J * 6.0);
    hipLaunchKernelGGL(eval_joints_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx.stream, pred, gt, B, J, mpjpe, pampjpe);
    return (int)hipGetLastError();
}

}  // namespace specmi
