// Micro-benchmark of the split-bf16 1x1 convolution (spec_amd/csrc/conv_bf16s.hip) beside the exact fp32 kernel, per layer
// shape of the ResNet-50 trunk (no Python).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf16s_bench.hip -o tools/bin/bf16s_bench && tools/bin/bf16s_bench [B] [check]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../spec_amd/csrc/conv_igemm.hip"
#define smem smem_wino        // (conv_igemm.hip and conv_wino.hip both call their dynamic LDS array smem: one translation unit here)
#include "../spec_amd/csrc/conv_wino.hip"
#undef smem
#include "../spec_amd/csrc/conv_bf16s.hip"

using namespace specmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Layer { const char* name; int cin, cout, hw, res, count; int k = 1, stride = 1; };

__global__ void ref_1x1(const float* x, const float* w, const float* sc, const float* sh, const float* res, double* out, int Cin, int Cout, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = i % Cout; const long m = i / Cout;
    double acc = 0;
    for (int c = 0; c < Cin; ++c) acc += (double)x[m * Cin + c] * (double)w[(size_t)n * Cin + c];
    double v = acc * sc[n] + sh[n];
    if (res) v += res[i];
    out[i] = v > 0 ? v : 0;
}
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, check = argc > 2 ? atoi(argv[2]) : 1, only = argc > 3 ? atoi(argv[3]) : -1;
    std::vector<Layer> layers = {
        {"l1.0.conv1  64->64  56", 64, 64, 56, 0, 1}, {"l1.conv1   256->64  56", 256, 64, 56, 0, 2},
        {"l1.conv3    64->256 56 +res", 64, 256, 56, 1, 2},
        {"l2.0.conv1 256->128 56", 256, 128, 56, 0, 1}, {"l2.conv1   512->128 28", 512, 128, 28, 0, 3},
        {"l2.conv3   128->512 28 +res", 128, 512, 28, 1, 3},
        {"l3.0.conv1 512->256 28", 512, 256, 28, 0, 1}, {"l3.conv1  1024->256 14", 1024, 256, 14, 0, 5},
        {"l3.conv3   256->1024 14 +res", 256, 1024, 14, 1, 5},
        {"l4.0.conv1 1024->512 14", 1024, 512, 14, 0, 1}, {"l4.conv1  2048->512 7", 2048, 512, 7, 0, 2},
        {"l4.conv3   512->2048 7 +res", 512, 2048, 7, 1, 2},
        {"l1.conv2 3x3  64->64  56", 64, 64, 56, 0, 3, 3, 1}, {"l2.0.conv2 3x3 s2 128 56", 128, 128, 56, 0, 1, 3, 2},
        {"l2.conv2 3x3 128->128 28", 128, 128, 28, 0, 3, 3, 1}, {"l3.0.conv2 3x3 s2 256 28", 256, 256, 28, 0, 1, 3, 2},
        {"l3.conv2 3x3 256->256 14", 256, 256, 14, 0, 5, 3, 1}, {"l4.0.conv2 3x3 s2 512 14", 512, 512, 14, 0, 1, 3, 2},
        {"l4.conv2 3x3 512->512 7", 512, 512, 7, 0, 2, 3, 1},
    };
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot[3] = {0, 0, 0};
    int li = -1;
    for (auto& L : layers) {
        ++li; if (only >= 0 && li != only) continue;
        const int H = L.hw, Npad = (L.cout + 63) / 64 * 64, KK = L.k * L.k, pad = L.k == 3 ? 1 : 0;
        const int OH = (H + 2 * pad - L.k) / L.stride + 1;
        const int Kd = L.cin * KK;
        const size_t M = (size_t)B * OH * OH, nx = (size_t)B * H * H * L.cin, no = M * L.cout, nw = (size_t)L.cout * Kd;
        std::vector<float> hx(nx), hw(nw), hsc(Npad, 0.f), hsh(Npad, 0.f), hp((size_t)Kd * Npad, 0.f);
        unsigned seed = 99 + L.cin + 3 * L.cout + L.k;
        for (auto& v : hx) { v = frand(seed) * 2.f; if (v < 0.f) v = 0.f; }
        for (auto& v : hw) v = frand(seed) * 2.f / sqrtf((float)Kd);     // OIHW
        for (int n = 0; n < L.cout; ++n) { hsc[n] = 1.f + 0.2f * frand(seed); hsh[n] = 0.1f * frand(seed); }
        for (int n = 0; n < L.cout; ++n) for (int ci = 0; ci < L.cin; ++ci) for (int t = 0; t < KK; ++t) {
            const int k = t * L.cin + ci;
            hp[((size_t)(k / 4) * Npad + n) * 4 + (k % 4)] = hw[((size_t)n * L.cin + ci) * KK + t];
        }
        std::vector<unsigned short> pieces;
        pack_bf16_split_weights_oihw(hw.data(), L.cout, L.cin, L.k, L.k, Npad, pieces);
        const bool use_wino = L.k == 3 && L.stride == 1;
        std::vector<float> hu;
        if (use_wino) pack_wino_weights(hw.data(), L.cout, L.cin, hu);
        float *dx, *dw, *dp, *dsc, *dsh, *dres = nullptr, *dout, *du = nullptr; void* dsp; double* dref;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dp, hp.size() * 4)); CK(hipMalloc(&dsp, pieces.size() * 2));
        CK(hipMalloc(&dsc, Npad * 4)); CK(hipMalloc(&dsh, Npad * 4)); CK(hipMalloc(&dout, no * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsp, pieces.data(), pieces.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsc, hsc.data(), Npad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, hsh.data(), Npad * 4, hipMemcpyHostToDevice));
        if (use_wino) { CK(hipMalloc(&du, hu.size() * 4)); CK(hipMemcpy(du, hu.data(), hu.size() * 4, hipMemcpyHostToDevice)); }
        if (L.res) { std::vector<float> hr(no); for (auto& v : hr) v = frand(seed); CK(hipMalloc(&dres, no * 4)); CK(hipMemcpy(dres, hr.data(), no * 4, hipMemcpyHostToDevice)); }
        ConvArgs a; a.x = dx; a.w = dp; a.scale = dsc; a.shift = dsh; a.res = dres; a.out = dout;
        a.B = B; a.H = H; a.W = H; a.Cin = L.cin; a.ldx = L.cin; a.OH = OH; a.OW = OH; a.Cout = L.cout; a.Npad = Npad; a.ldo = L.cout;
        a.KH = L.k; a.KW = L.k; a.stride = L.stride; a.pad = pad; a.relu = 1;
        ConvArgs aw = a; aw.w = du;
        LaunchCtx ctx{s, nullptr, "bench"};
        std::vector<double> href;
        const size_t ns = no < (size_t)2000000 ? no : 2000000;
        if (check && L.k == 1) {
            CK(hipMalloc(&dref, no * 8));
            hipLaunchKernelGGL(ref_1x1, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, s, dx, dw, dsc, dsh, dres, dref, L.cin, L.cout, (long)no);
            CK(hipStreamSynchronize(s));
            href.resize(ns); CK(hipMemcpy(href.data(), dref + (no - ns), ns * 8, hipMemcpyDeviceToHost)); CK(hipFree(dref));
        }
        float ms[3]; double err[3] = {-1, -1, -1};
        for (int v = 0; v < 3; ++v) {           // 0: exact fp32, 1: 6 terms, 2: 3 terms
            auto go = [&]() { return v == 0 ? (use_wino ? launch_conv_wino(aw, ctx) : launch_conv_igemm(a, ctx)) : launch_conv_bf16s(a, dsp, v == 1 ? 6 : 3, ctx); };
            CK(hipMemsetAsync(dout, 0xFF, no * 4, s));
            int rc = go(); if (rc) { printf("launch failed %d\n", rc); return 1; }
            CK(hipStreamSynchronize(s));
            if (check && L.k == 1) {
                std::vector<float> ho(ns); CK(hipMemcpy(ho.data(), dout + (no - ns), ns * 4, hipMemcpyDeviceToHost));
                double e = 0, mx = 0; for (size_t i = 0; i < ns; ++i) { double d = fabs((double)ho[i] - href[i]); if (!(d <= e)) e = d; if (href[i] > mx) mx = href[i]; }
                err[v] = e / mx;
            }
            for (int i = 0; i < 40; ++i) go();
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 20; ++i) go();
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[v], e0, e1)); ms[v] /= 20; tot[v] += ms[v] * L.count;
        }
        const double fl = 2.0 * M * (double)L.cout * Kd, by = 4.0 * ((double)nx + (double)no * (L.res ? 2 : 1));
        printf("%-30s fp32 %6.3f ms %6.1f TF/s | 6t %6.3f ms %6.1f TF/s x%.2f | 3t %6.3f ms %6.1f TF/s x%.2f | hbm floor %.3f ms | err %.1e %.1e %.1e  x%d\n",
               L.name, ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, ms[0] / ms[1], ms[2], fl / ms[2] / 1e9, ms[0] / ms[2], by / 8e9, err[0], err[1], err[2], L.count);
        (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dp); (void)hipFree(dsp); (void)hipFree(dsc); (void)hipFree(dsh); (void)hipFree(dout); if (dres) (void)hipFree(dres); if (du) (void)hipFree(du);
    }
    printf("TRUNK plain 1x1 (weighted): fp32 %.3f ms | 6 terms %.3f ms | 3 terms %.3f ms   (B=%d)\n", tot[0], tot[1], tot[2], B);
    return 0;
}
