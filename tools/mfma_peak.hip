// MFMA issue micro-benchmark: fp32 32x32x2 MFMA throughput vs accumulators per wave and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, float* d) {
    int iters = 20000; int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 4 * iters * 16 * 4096.0;
    printf("acc/wave=%d waves/SIMD=%d : %.1f TF/s (%.3f ms)\n", NACC, blocks_per_cu, fl / ms / 1e9, ms);
}
int main() { float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
    for (int b = 1; b <= 8; b *= 2) { run<1>(b, d); run<2>(b, d); run<4>(b, d); }
    return 0; }
