// Standalone micro-benchmark + correctness check of the implicit-GEMM conv kernels on a GPU box
// (no Python): builds against spec_amd/csrc/conv_igemm.hip directly.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/igemm_bench.hip -o /tmp/igemm_bench && /tmp/igemm_bench [variant] [B]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SPECMI_TUNE 1
#include "../spec_amd/csrc/conv_igemm.hip"

using namespace specmi;
static const char* PH[7] = {"issue", "mfma", "wait+store", "barrier", "prologue", "loop", "epilogue"};

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Layer { const char* name; int cin, cout, k, stride, hw; int res; int count; };

__global__ void ref_conv(const float* x, const float* w /*OIHW*/, const float* sc, const float* sh, const float* res,
                         float* out, int B, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int pad,
                         int relu, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int n = i % Cout; long m = i / Cout;
    int ox = m % OW; long t = m / OW; int oy = t % OH; int b = t / OH;
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) for (int kx = 0; kx < k; ++kx) {
        int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float* xp = x + ((size_t)(b * H + iy) * W + ix) * Cin;
        const float* wp = w + ((size_t)n * Cin * k + ky) * k + kx;
        for (int c = 0; c < Cin; ++c) acc = fmaf(xp[c], wp[(size_t)c * k * k], acc);
    }
    float v = fmaf(acc, sc[n], sh[n]);
    if (res) v += res[i];
    if (relu) v = fmaxf(v, 0.f);
    out[i] = v;
}

__global__ void read_clock(long long* out) { if (threadIdx.x == 0) out[0] = __builtin_amdgcn_s_memtime(); }

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(int argc, char** argv) {
    int variant = argc > 1 ? atoi(argv[1]) : 0;
    int B = argc > 2 ? atoi(argv[2]) : 256;
    int check = argc > 3 ? atoi(argv[3]) : 1;
    int relu_in = 1;
    int ablate = argc > 4 ? atoi(argv[4]) : 0;
    int only = argc > 5 ? atoi(argv[5]) : -1;   // run a single layer index
    int reps = argc > 6 ? atoi(argv[6]) : 1;     // weight replicas
    long skew = argc > 7 ? atol(argv[7]) : 4096;  // extra bytes between replicas
    conv_igemm_set_ablate(ablate);
    unsigned long long* dtp; CK(hipMalloc(&dtp, 128));   // inputs are post-ReLU activations (as in the trunk)
    long long* dclk; CK(hipMalloc(&dclk, 16)); long long hclk[2];
    std::vector<Layer> layers = {
        {"l1.conv1  1x1  64->64  56", 64, 64, 1, 1, 56, 0, 1},
        {"l1.conv2  3x3  64->64  56", 64, 64, 3, 1, 56, 0, 3},
        {"l1.conv3  1x1  64->256 56 +res", 64, 256, 1, 1, 56, 1, 4},
        {"l1.conv1b 1x1 256->64  56", 256, 64, 1, 1, 56, 0, 2},
        {"l2.0.c1   1x1 256->128 56", 256, 128, 1, 1, 56, 0, 1},
        {"l2.0.c2   3x3 128 s2   56", 128, 128, 3, 2, 56, 0, 1},
        {"l2.0.ds   1x1 256->512 s2", 256, 512, 1, 2, 56, 0, 1},
        {"l2.conv3  1x1 128->512 28 +res", 128, 512, 1, 1, 28, 1, 4},
        {"l2.conv1  1x1 512->128 28", 512, 128, 1, 1, 28, 0, 3},
        {"l2.conv2  3x3 128      28", 128, 128, 3, 1, 28, 0, 3},
        {"l3.0.c1   1x1 512->256 28", 512, 256, 1, 1, 28, 0, 1},
        {"l3.0.c2   3x3 256 s2   28", 256, 256, 3, 2, 28, 0, 1},
        {"l3.0.ds   1x1 512->1024 s2", 512, 1024, 1, 2, 28, 0, 1},
        {"l3.conv3  1x1 256->1024 14 +res", 256, 1024, 1, 1, 14, 1, 6},
        {"l3.conv1  1x1 1024->256 14", 1024, 256, 1, 1, 14, 0, 5},
        {"l3.conv2  3x3 256      14", 256, 256, 3, 1, 14, 0, 5},
        {"l4.0.c1   1x1 1024->512 14", 1024, 512, 1, 1, 14, 0, 1},
        {"l4.0.c2   3x3 512 s2   14", 512, 512, 3, 2, 14, 0, 1},
        {"l4.0.ds   1x1 1024->2048 s2", 1024, 2048, 1, 2, 14, 0, 1},
        {"l4.conv3  1x1 512->2048 7 +res", 512, 2048, 1, 1, 7, 1, 3},
        {"l4.conv1  1x1 2048->512 7", 2048, 512, 1, 1, 7, 0, 2},
        {"l4.conv2  3x3 512      7", 512, 512, 3, 1, 7, 0, 2},
        {"bigK      1x1 8192->256 14", 8192, 256, 1, 1, 14, 0, 0},
        {"bigK3x3   3x3 1024->256 14", 1024, 256, 3, 1, 14, 0, 0},
    };
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot_ms = 0, tot_fl = 0;
    int li = -1;
    for (auto& L : layers) {
        ++li; if (only >= 0 && li != only) continue;
        int H = L.hw, W = L.hw, pad = L.k == 3 ? 1 : 0;
        int OH = (H + 2 * pad - L.k) / L.stride + 1, OW = OH;
        size_t nx = (size_t)B * H * W * L.cin, no = (size_t)B * OH * OW * L.cout, nw = (size_t)L.cout * L.cin * L.k * L.k;
        int K = L.cin * L.k * L.k, Npad = (L.cout + 63) / 64 * 64;
        std::vector<float> hx(nx), hw(nw), hsc(Npad, 0.f), hsh(Npad, 0.f), hp((size_t)K * Npad, 0.f);
        unsigned seed = 1234 + L.cin + L.cout * 3 + L.k;
        for (auto& v : hx) { v = frand(seed) * 2.f; if (relu_in && v < 0.f) v = 0.f; }
        for (auto& v : hw) v = frand(seed) * 2.f / sqrtf((float)K);
        for (int n = 0; n < L.cout; ++n) { hsc[n] = 1.f + 0.2f * frand(seed); hsh[n] = 0.1f * frand(seed); }
        for (int n = 0; n < L.cout; ++n) for (int ci = 0; ci < L.cin; ++ci) for (int ky = 0; ky < L.k; ++ky) for (int kx = 0; kx < L.k; ++kx) {
            int k = (ky * L.k + kx) * L.cin + ci;
            hp[((size_t)(k / 4) * Npad + n) * 4 + (k % 4)] = hw[(((size_t)n * L.cin + ci) * L.k + ky) * L.k + kx];
        }
        float *dx, *dw, *dp, *dsc, *dsh, *dres = nullptr, *dout, *dref;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); const size_t rep_stride = ((hp.size() * 4 + 4095) / 4096 * 4096 + skew) / 4; CK(hipMalloc(&dp, rep_stride * 4 * reps));
        CK(hipMalloc(&dsc, Npad * 4)); CK(hipMalloc(&dsh, Npad * 4)); CK(hipMalloc(&dout, no * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        for (int r = 0; r < reps; ++r) CK(hipMemcpy(dp + (size_t)r * rep_stride, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsc, hsc.data(), Npad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, hsh.data(), Npad * 4, hipMemcpyHostToDevice));
        if (L.res) { std::vector<float> hr(no); for (auto& v : hr) v = frand(seed); CK(hipMalloc(&dres, no * 4)); CK(hipMemcpy(dres, hr.data(), no * 4, hipMemcpyHostToDevice)); }
        ConvArgs a; a.x = dx; a.w = dp; a.scale = dsc; a.shift = dsh; a.res = dres; a.out = dout;
        a.B = B; a.H = H; a.W = W; a.Cin = L.cin; a.ldx = L.cin; a.OH = OH; a.OW = OW; a.Cout = L.cout; a.Npad = Npad; a.ldo = L.cout;
        a.KH = L.k; a.KW = L.k; a.stride = L.stride; a.pad = pad; a.relu = 1; a.force_variant = variant;
        LaunchCtx ctx{s, nullptr, "bench"};
        int rc = launch_conv_igemm(a, ctx); if (rc) { printf("launch failed %d\n", rc); return 1; }
        CK(hipStreamSynchronize(s));
        double maxerr = -1;
        if (check) {
            CK(hipMalloc(&dref, no * 4));
            long total = (long)no;
            hipLaunchKernelGGL(ref_conv, dim3((total + 255) / 256), dim3(256), 0, s, dx, dw, dsc, dsh, dres, dref, B, H, W, L.cin, OH, OW, L.cout, L.k, L.stride, pad, 1, total);
            CK(hipStreamSynchronize(s));
            size_t ns = no < (size_t)4000000 ? no : 4000000;   // compare a prefix + a suffix
            std::vector<float> ho(ns), hr(ns);
            for (int part = 0; part < 2; ++part) {
                size_t off = part ? no - ns : 0;
                CK(hipMemcpy(ho.data(), dout + off, ns * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), dref + off, ns * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < ns; ++i) { double d = fabs((double)ho[i] - hr[i]); if (d > maxerr) maxerr = d; }
            }
            CK(hipFree(dref));
        }
        const int iters = 20;
        for (int i = 0; i < 60; ++i) launch_conv_igemm(a, ctx);   // long warm-up: DVFS ramps over milliseconds
        hipLaunchKernelGGL(read_clock, dim3(1), dim3(64), 0, s, dclk);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) launch_conv_igemm(a, ctx);
        CK(hipEventRecord(e1, s));
        hipLaunchKernelGGL(read_clock, dim3(1), dim3(64), 0, s, dclk + 1);
        CK(hipEventSynchronize(e1)); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        unsigned long long htp[8];
        CK(hipMemset(dtp, 0, 64)); conv_igemm_set_tprof(dtp); launch_conv_igemm(a, ctx); CK(hipStreamSynchronize(s)); conv_igemm_set_tprof(nullptr);
        CK(hipMemcpy(htp, dtp, 64, hipMemcpyDeviceToHost));
        double fl = 2.0 * B * OH * OW * (double)L.cout * K;
        double by = 4.0 * ((double)nx + (double)no * (L.res ? 2 : 1) + (double)nw);
        printf("%-34s %8.3f ms %7.1f TF/s %7.0f GB/s  x%d  err %.2e %s\n", L.name, ms, fl / ms / 1e9, by / ms / 1e6, L.count, maxerr, conv_igemm_variant(a));
        if (only >= 0 && htp[7]) { printf("   per-wave cycles (avg of %llu waves):", htp[7]); for (int q = 0; q < 7; ++q) printf(" %s=%.0f", PH[q], (double)htp[q] / htp[7]); printf("\n"); }
        tot_ms += ms * L.count; tot_fl += fl * L.count;
        (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dp); (void)hipFree(dsc); (void)hipFree(dsh); (void)hipFree(dout); if (dres) (void)hipFree(dres);
    }
    printf("TRUNK(convs, weighted) %.3f ms  %.1f TF/s  (variant %d, B=%d)\n", tot_ms, tot_fl / tot_ms / 1e9, variant, B);
    return 0;
}
