// What does a lone wave's dependent fp32 MFMA chain cost, in shader cycles and in nanoseconds, at different chip occupancies?
// (round 4: the small-batch kernels run 16 MFMAs in 0.62 us whatever the loop around them looks like - is that the pipe at a low clock?)
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/bin/clock_probe && tools/bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) chain(float* out, long long* cyc, int iters, float a0, float b0) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* d; long long* c; hipMalloc(&d, 4096 * 256 * 4); hipMalloc(&c, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;   // 32000 MFMAs per wave: ~0.85 ms at 2.4 GHz
    for (int rep = 0; rep < 2; ++rep)
    for (int grid : {1, 16, 64, 200, 256, 512, 1024}) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(chain, dim3(grid), dim3(256), 0, 0, d, c, iters, 1.f, 2.f);
        hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(grid), dim3(256), 0, 0, d, c, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        const double nm = (double)iters * 16;
        printf("grid %5d x 256 threads: %.3f ms  %.1f ns per MFMA  s_memtime ticks per MFMA %.1f  (ticks/us %.0f)\n", grid, ms, ms * 1e6 / nm, hc / nm, hc / (ms * 1e3));
    }
    // short kernels back to back (the small-batch regime: ~10 us of work per launch)
    for (int grid : {16, 200}) {
        const int it2 = 16;    // 256 MFMAs per wave ~ 7 us
        for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(chain, dim3(grid), dim3(256), 0, 0, d, c, it2, 1.f, 2.f);
        hipDeviceSynchronize();
        long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("short kernels, grid %d: s_memtime ticks per MFMA %.1f\n", grid, hc / (16.0 * it2));
    }
    return 0;
}
