// Where does the dispatcher put the workgroups of a persistent 2-per-CU grid?  Logs (XCC, SE, SH, CU, wave slot, start tick)
// per workgroup of a 512 x 256-thread launch with ~70 KB of LDS each (the conv_wino<8> footprint).
//   hipcc --offload-arch=gfx950 -O3 tools/cu_probe.hip -o tools/bin/cu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { unsigned hw, xcc; unsigned long long t0, t1; };
__global__ void __launch_bounds__(256, 2) probe(Rec* out, int spin) {
    extern __shared__ char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x].hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        out[blockIdx.x].xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    }
    smem[threadIdx.x] = (char)threadIdx.x;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);
    __syncthreads();
    if (threadIdx.x == 0) { out[blockIdx.x].t0 = t0; out[blockIdx.x].t1 = __builtin_amdgcn_s_memtime() + smem[5]; }
}
int main() {
    const int G = 512;
    Rec* d; hipMalloc(&d, G * sizeof(Rec));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 70672);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 70672, 0, d, 20);
        hipDeviceSynchronize();
    }
    std::vector<Rec> h(G);
    hipMemcpy(h.data(), d, G * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;
    unsigned long long tmin = ~0ull;
    for (auto& r : h) tmin = std::min(tmin, r.t0);
    for (int i = 0; i < G; ++i) {
        const unsigned key = (h[i].xcc << 8) | ((h[i].hw >> 8) & 0xff);
        per_cu[key].push_back(i);
        if (i < 40) printf("wg %3d xcc %u se %u sh %u cu %2u waveslot %u simd %u start %llu dur %llu\n", i, h[i].xcc, (h[i].hw >> 13) & 7,
                           (h[i].hw >> 12) & 1, (h[i].hw >> 8) & 15, h[i].hw & 15, (h[i].hw >> 4) & 3, h[i].t0 - tmin, h[i].t1 - h[i].t0);
    }
    int hist[8] = {};
    for (auto& kv : per_cu) hist[std::min<size_t>(kv.second.size(), 7)]++;
    printf("distinct CUs %zu; CUs with n workgroups: 1:%d 2:%d 3:%d 4:%d 5+:%d\n", per_cu.size(), hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7]);
    int shown = 0;
    for (auto& kv : per_cu) { if (shown++ >= 12) break; printf("cu key %03x:", kv.first); for (int i : kv.second) printf(" %d(slot %u)", i, h[i].hw & 15); printf("\n"); }
    return 0;
}
