// Can the fp32 MFMA pipe and the fp32 VALU FMA pipe of a SIMD run at full rate together?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// mode: 0 all waves MFMA, 1 all waves VALU, 2 waves 0-3 MFMA + 4-7 VALU
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, float a0, float b0) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    float s = 0;
    if (do_mfma) {
        f32x16 acc[2];
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = a0 + threadIdx.x, b = b0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0); }
        }
        for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    } else {
        f32x2 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
        f32x2 a = {a0 + threadIdx.x, a0}, b = {b0, b0 * 0.5f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)          // 16 x 16 pk_fma = 256 instr = 65536 flop/wave per iter (MFMA branch: 16 x 4096)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 512 * 4);
    const int iters = 4000, grid = 256;   // one 8-wave block per CU: 2 waves per SIMD
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, iters, mode, 1.f, 2.f);
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, iters, mode, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_wave = (double)iters * 65536.0;
        const double fl = (double)grid * 8 * per_wave;
        printf("mode %d (%s): %.3f ms  %.1f TF/s total\n", mode, mode == 0 ? "8 MFMA waves" : mode == 1 ? "8 VALU waves" : "4 MFMA + 4 VALU waves", ms, fl / ms / 1e9);
    }
    return 0;
}
