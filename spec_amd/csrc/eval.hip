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
// One workgroup per image: joint regression walks the vertices ONCE per chunk of joints (every lane keeps a vertex
// pair in registers and accumulates kJC joints x 6 coordinates; all loads of the loop are independent, so they pipeline;
// one wave-shuffle + LDS reduction per chunk), then one lane solves the Procrustes problem with Horn's quaternion
// method (largest eigenpair of a symmetric 4x4 via cyclic Jacobi in fp64) - the same optimum as the SVD/Kabsch
// solution with its det(R)=+1 fix.
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

// Sum N per-lane values over a 256-thread block: wave shuffles, then one LDS exchange.  Result i lands in dst[i]
// (LDS, valid after the call's trailing barrier).  part = LDS scratch of 4 * N floats.
template <int N>
__device__ __forceinline__ void block_sum_many_256(float (&a)[N], float* part, float* dst) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float v = a[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        a[i] = v;
    }
    __syncthreads();   // part may still be read from the previous call
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) part[(threadIdx.x >> 6) * N + i] = a[i];
    }
    __syncthreads();
    if (threadIdx.x < N) dst[threadIdx.x] = part[threadIdx.x] + part[N + threadIdx.x] + part[2 * N + threadIdx.x] + part[3 * N + threadIdx.x];
    __syncthreads();
}

// Largest eigenvalue / eigenvector of a symmetric 4x4 (cyclic Jacobi, fp64).
// All matrix indices are compile-time constants after unrolling, so A and V live in registers (no scratch).
__device__ void sym4_max_eig(double A[4][4], double q[4], double* lam) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = 0, diag = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            diag += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-30 * (diag + 1e-300)) break;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int r = p + 1; r < 4; ++r) {
                // tan of the rotation angle: t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (a_rr - a_pp) / (2 a_pr),
                // written without the division that forms theta: one sqrt, one division, one rsqrt per rotation; an
                // already-zero a_pr gives t = 0 (identity rotation) through the guarded denominator
                const double apr2 = 2.0 * A[p][r], diff = A[r][r] - A[p][p];
                const double den = fabs(diff) + sqrt(diff * diff + apr2 * apr2);
                const double t = fabs(A[p][r]) < 1e-300 ? 0.0 : (diff >= 0 ? apr2 : -apr2) / den;
                const double c = rsqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akr = A[k][r];
                    A[k][p] = c * akp - s * akr;
                    A[k][r] = s * akp + c * akr;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], ark = A[r][k];
                    A[p][k] = c * apk - s * ark;
                    A[r][k] = s * apk + c * ark;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkr = V[k][r];
                    V[k][p] = c * vkp - s * vkr;
                    V[k][r] = s * vkp + c * vkr;
                }
            }
        }
    }
    double best = A[0][0];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = V[k][0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (A[i][i] > best) {
            best = A[i][i];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = V[k][i];
        }
    *lam = best;
}

// MPJPE and PA-MPJPE (mm) of N pelvis-aligned joints p, g: coordinate c of joint i at p[(i * 3 + c) * ST].
template <int ST>
__device__ void joint_errors(const float* p, const float* g, int N, float* mpjpe, float* pampjpe) {
    double e = 0, mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
    for (int i = 0; i < N; ++i) {
        double d = 0;
        for (int c = 0; c < 3; ++c) {
            const double df = (double)p[(i * 3 + c) * ST] - (double)g[(i * 3 + c) * ST];
            d += df * df;
            mu1[c] += p[(i * 3 + c) * ST];
            mu2[c] += g[(i * 3 + c) * ST];
        }
        e += sqrt(d);
    }
    *mpjpe = (float)(e / N * 1000.0);
    for (int c = 0; c < 3; ++c) { mu1[c] /= N; mu2[c] /= N; }
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0;
    for (int i = 0; i < N; ++i) {
        double x1[3], x2[3];
        for (int c = 0; c < 3; ++c) { x1[c] = p[(i * 3 + c) * ST] - mu1[c]; x2[c] = g[(i * 3 + c) * ST] - mu2[c]; var1 += x1[c] * x1[c]; }
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
            for (int b = 0; b < 3; ++b) v += R[a][b] * (p[(i * 3 + b) * ST] - mu1[b]);
            const double df = scale * v + mu2[a] - g[(i * 3 + a) * ST];
            d += df * df;
        }
        pe += sqrt(d);
    }
    *pampjpe = (float)(pe / N * 1000.0);
}

constexpr int kMaxJ = 32;
constexpr int kJC = 6;     // joints regressed per pass over the vertices (kJC * 6 accumulators per lane)

__global__ void __launch_bounds__(256) eval_mesh_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         int V, const float* __restrict__ Jr, int J,
                                                         const int* __restrict__ sel, int nsel, float* __restrict__ mpjpe,
                                                         float* __restrict__ pampjpe, float* __restrict__ v2v) {
    __shared__ float red[4];
    __shared__ float jp[kMaxJ][3], jg[kMaxJ][3];
    __shared__ float sp[kMaxJ * 3], sg[kMaxJ * 3];
    __shared__ float part[4 * kJC * 6], sums[kJC * 6];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* pv = pred + (size_t)b * V * 3;
    const float* gv = gt + (size_t)b * V * 3;
    for (int j0 = 0; j0 < J; j0 += kJC) {
        float a[kJC * 6];
#pragma unroll
        for (int i = 0; i < kJC * 6; ++i) a[i] = 0.f;
        // rows past J alias row J - 1 (unconditional loads - a predicated load would sit in its own exec-masked branch
        // and serialise on the memory latency); their sums are never stored
        const float* wr[kJC];
#pragma unroll
        for (int jj = 0; jj < kJC; ++jj) wr[jj] = Jr + (size_t)min(j0 + jj, J - 1) * V;
#pragma unroll 3
        for (int v = t; v < V; v += 256) {
            const float x[6] = {pv[v * 3 + 0], pv[v * 3 + 1], pv[v * 3 + 2], gv[v * 3 + 0], gv[v * 3 + 1], gv[v * 3 + 2]};
#pragma unroll
            for (int jj = 0; jj < kJC; ++jj) {
                const float w = wr[jj][v];
#pragma unroll
                for (int c = 0; c < 6; ++c) a[jj * 6 + c] = fmaf(w, x[c], a[jj * 6 + c]);
            }
        }
        block_sum_many_256<kJC * 6>(a, part, sums);
        if (t < kJC * 6 && j0 + t / 6 < J) {
            const int jj = t / 6, c = t % 6;
            if (c < 3) jp[j0 + jj][c] = sums[t]; else jg[j0 + jj][c - 3] = sums[t];
        }
    }
    __syncthreads();
    // pelvis-aligned vertex-to-vertex error
    const float ppx = jp[0][0], ppy = jp[0][1], ppz = jp[0][2], gpx = jg[0][0], gpy = jg[0][1], gpz = jg[0][2];
    float acc = 0.f;
#pragma unroll 4
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
        joint_errors<1>(sp, sg, nsel, &m, &pa);
        if (mpjpe) mpjpe[b] = m;
        if (pampjpe) pampjpe[b] = pa;
    }
}

__global__ void __launch_bounds__(64) eval_joints_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int B,
                                                          int J, float* __restrict__ mpjpe, float* __restrict__ pampjpe) {
    // one lane per pose; the pelvis-aligned joints sit in LDS as [coordinate][lane] (conflict-free, no scratch arrays)
    __shared__ float sp[kMaxJ * 3 * 64], sg[kMaxJ * 3 * 64];
    const int t = threadIdx.x, b0 = blockIdx.x * 64, nb = min(64, B - b0), n3 = J * 3;
    for (int i = t; i < nb * n3; i += 64) {   // coalesced over the block's nb * J * 3 floats
        const int l = i / n3, k = i - l * n3;
        const size_t base = (size_t)(b0 + l) * n3;
        sp[k * 64 + l] = pred[base + k] - pred[base + k % 3];
        sg[k * 64 + l] = gt[base + k] - gt[base + k % 3];
    }
    __syncthreads();
    if (t >= nb) return;
    float m, pa;
    joint_errors<64>(sp + t, sg + t, J, &m, &pa);
    if (mpjpe) mpjpe[b0 + t] = m;
    if (pampjpe) pampjpe[b0 + t] = pa;
}

// joints (B,J,3) = J_regressor (J,V) @ vertices (B,V,3)  (torch.einsum('bik,ji->bjk'), compute_error.py:184,187;
// torch.matmul(J_regressor_batch, vertices), :53,58): one workgroup per (image, chunk of kRC joints); the vertices are
// read once per chunk and every load of the loop is independent
constexpr int kRC = 8;
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) regress_joints_kernel(const float* __restrict__ verts, int V,
                                                              const float* __restrict__ Jr, int J, int nchunk,
                                                              float* __restrict__ out) {
    __shared__ float part[4 * kRC * 3], sums[kRC * 3];
    const int b = blockIdx.x / nchunk, j0 = (blockIdx.x % nchunk) * kRC, t = threadIdx.x;
    const float* v = verts + (size_t)b * V * 3;
    float a[kRC * 3];
#pragma unroll
    for (int i = 0; i < kRC * 3; ++i) a[i] = 0.f;
    const float* wr[kRC];      // rows past J alias row J - 1 (unconditional loads); their sums are never stored
#pragma unroll
    for (int jj = 0; jj < kRC; ++jj) wr[jj] = Jr + (size_t)min(j0 + jj, J - 1) * V;
    // two vertices per trip, all 2 * (3 + kRC) loads issued before the first use (the second vertex of a lane's last trip is
    // clamped to a valid index and weighted with zero)
    for (int i = t; i < V; i += 512) {
        const bool two = i + 256 < V;
        const int i2 = two ? i + 256 : i;
        float xa[3], xb[3], wa[kRC], wb[kRC];
#pragma unroll
        for (int c = 0; c < 3; ++c) { xa[c] = v[i * 3 + c]; xb[c] = v[i2 * 3 + c]; }
#pragma unroll
        for (int jj = 0; jj < kRC; ++jj) { wa[jj] = wr[jj][i]; wb[jj] = wr[jj][i2]; }
#pragma unroll
        for (int jj = 0; jj < kRC; ++jj) {
            const float w2 = two ? wb[jj] : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) a[jj * 3 + c] = fmaf(w2, xb[c], fmaf(wa[jj], xa[c], a[jj * 3 + c]));
        }
    }
    block_sum_many_256<kRC * 3>(a, part, sums);
    if (t < kRC * 3 && j0 + t / 3 < J) out[((size_t)b * J + j0) * 3 + t] = sums[t];
}

int launch_regress_joints(const float* verts, int B, int V, const float* Jr, int J, float* out, const LaunchCtx& ctx) {
    const int nchunk = (J + kRC - 1) / kRC;
    ProfScope ps(ctx, "regress_joints", 2.0 * B * (double)V * 3 * J, 4.0 * ((double)B * V * 3 + (double)J * V + (double)B * J * 3));
    hipLaunchKernelGGL(regress_joints_kernel, dim3(B * nchunk), dim3(256), 0, ctx.stream, verts, V, Jr, J, nchunk, out);
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
    // algorithmic HBM bytes: both meshes once + the regressor once (re-reads per joint chunk / for V2V come through L2)
    ProfScope ps(ctx, "eval_mesh_metrics", 2.0 * B * (double)V * 6 * J, 4.0 * ((double)B * V * 6 + (double)J * V + 3.0 * B));
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
