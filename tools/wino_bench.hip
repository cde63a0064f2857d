// Standalone micro-benchmark + correctness check of the Winograd F(2x2,3x3) conv kernel (no Python).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_bench.hip -o tools/bin/wino_bench
//   tools/bin/wino_bench [B] [check] [only_layer]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define WINO_PROF 1
#include "../spec_amd/csrc/conv_wino.hip"

using namespace specmi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Layer { const char* name; int cin, cout, h, w; int count; int b; };

__global__ void ref_conv3(const float* x, const float* w /*OIHW*/, const float* sc, const float* sh, double* out,
                          int B, int H, int W, int Cin, int Cout, int relu, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int n = i % Cout; long m = i / Cout;
    int ox = m % W; long t = m / W; int oy = t % H; int b = t / H;
    double acc = 0.0;
    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
        int iy = oy - 1 + ky, ix = ox - 1 + kx;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float* xp = x + ((size_t)(b * H + iy) * W + ix) * Cin;
        const float* wp = w + ((size_t)n * Cin * 3 + ky) * 3 + kx;
        for (int c = 0; c < Cin; ++c) acc += (double)xp[c] * (double)wp[(size_t)c * 9];
    }
    double v = acc * sc[n] + sh[n];
    if (relu) v = v > 0 ? v : 0;
    out[i] = v;
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 256;
    int check = argc > 2 ? atoi(argv[2]) : 1;
    int only = argc > 3 ? atoi(argv[3]) : -1;
    int variant = argc > 4 ? atoi(argv[4]) : 0;
    conv_wino_set_persistent(argc > 5 ? atoi(argv[5]) : 1);
    std::vector<Layer> layers = {
        {"odd      3x3  16->128  7x9 ", 16, 128, 7, 9, 0, 3},
        {"odd2     3x3  32->256  5x3 ", 32, 256, 5, 3, 0, 5},
        {"odd3     3x3  48->64   9x7 ", 48, 64, 9, 7, 0, 4},
        {"odd4     3x3  96->64   6x5 ", 96, 64, 6, 5, 0, 70},
        {"l1.conv2 3x3  64->64  56   ", 64, 64, 56, 56, 3, 0},
        {"l2.conv2 3x3 128->128 28   ", 128, 128, 28, 28, 3, 0},
        {"l3.conv2 3x3 256->256 14   ", 256, 256, 14, 14, 5, 0},
        {"l4.conv2 3x3 512->512 7    ", 512, 512, 7, 7, 2, 0},
    };
    double tot_ms = 0, tot_fl = 0;
    for (size_t li = 0; li < layers.size(); ++li) {
        if (only >= 0 && (int)li != only) continue;
        const Layer& L = layers[li];
        const int b = L.b ? L.b : B;
        const size_t nx = (size_t)b * L.h * L.w * L.cin, no = (size_t)b * L.h * L.w * L.cout, nw = (size_t)L.cout * L.cin * 9;
        std::vector<float> hx(nx), hw(nw), hs(L.cout), hb(L.cout), packed;
        unsigned s = 1234 + li;
        for (auto& v : hx) { v = frand(s) * 2.f; if (v < 0) v = 0; }
        const float wstd = sqrtf(2.f / (9.f * L.cin)) * 3.4f;
        for (auto& v : hw) v = frand(s) * wstd;
        for (auto& v : hs) v = 1.f + frand(s) * 0.2f;
        for (auto& v : hb) v = frand(s) * 0.2f;
        pack_wino_weights(hw.data(), L.cout, L.cin, packed);
        float *dx, *dw, *du, *ds, *db, *dout; double* dref;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&du, packed.size() * 4));
        CK(hipMalloc(&ds, L.cout * 4)); CK(hipMalloc(&db, L.cout * 4)); CK(hipMalloc(&dout, no * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(du, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(ds, hs.data(), L.cout * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), L.cout * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dout, 0xFF, no * 4));
        ConvArgs a;
        a.x = dx; a.w = du; a.scale = ds; a.shift = db; a.res = nullptr; a.out = dout;
        a.B = b; a.H = L.h; a.W = L.w; a.Cin = L.cin; a.ldx = L.cin; a.OH = L.h; a.OW = L.w; a.Cout = L.cout;
        a.Npad = L.cout; a.ldo = L.cout; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.relu = 1; a.wino_variant = variant;
        LaunchCtx ctx{nullptr, nullptr, "bench"};
        int rc = launch_conv_wino(a, ctx);
        if (rc) { printf("launch failed rc=%d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
        double err = -1, refmax = 0;
        if (check) {
            CK(hipMalloc(&dref, no * 8));
            ref_conv3<<<(unsigned)((no + 255) / 256), 256>>>(dx, dw, ds, db, dref, b, L.h, L.w, L.cin, L.cout, 1, (long)no);
            CK(hipDeviceSynchronize());
            std::vector<float> ho(no); std::vector<double> hr(no);
            CK(hipMemcpy(ho.data(), dout, no * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), dref, no * 8, hipMemcpyDeviceToHost));
            err = 0;
            for (size_t i = 0; i < no; ++i) {
                const double d = std::fabs((double)ho[i] - hr[i]);
                if (!(d <= err)) err = d;   // catches NaN
                if (hr[i] > refmax) refmax = hr[i];
            }
            CK(hipFree(dref));
        }
        float ms = 0;
        unsigned long long* dtp; CK(hipMalloc(&dtp, 128)); CK(hipMemset(dtp, 0, 128));
        if (L.count) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 60; ++i) launch_conv_wino(a, ctx);
            const int iters = 40;
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) launch_conv_wino(a, ctx);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            conv_wino_set_tprof(dtp);
            launch_conv_wino(a, ctx);
            CK(hipDeviceSynchronize());
            conv_wino_set_tprof(nullptr);
            unsigned long long tp[9]; CK(hipMemcpy(tp, dtp, 72, hipMemcpyDeviceToHost));
            if (tp[3]) printf("    epilogue split: transform+send %.0f  barrier %.0f  finish+stores %.0f  barrier %.0f | stage barriers %.0f\n", (double)tp[4] / tp[3],
                              (double)tp[5] / tp[3], (double)tp[6] / tp[3], (double)tp[7] / tp[3], (double)tp[8] / tp[3]);
            if (tp[3]) printf("    per-WG ticks(10ns): prologue %.0f  loop %.0f  epilogue %.0f   (n=%llu)\n", (double)tp[0] / tp[3], (double)tp[1] / tp[3], (double)tp[2] / tp[3], tp[3]);
        }
        const double fl = 2.0 * b * L.h * L.w * (double)L.cout * 9.0 * L.cin;
        printf("%s B=%-3d %8.3f ms  %7.1f TF/s(direct-equiv)  x%d  max|err| %.3g (ref max %.3g)\n", L.name, b, ms,
               ms > 0 ? fl / ms * 1e-9 : 0.0, L.count, err, refmax);
        tot_ms += ms * L.count; tot_fl += fl * L.count;
        hipFree(dx); hipFree(dw); hipFree(du); hipFree(ds); hipFree(db); hipFree(dout);
    }
    if (tot_ms > 0) printf("TOTAL (weighted) %.3f ms  %.1f TF/s direct-equivalent\n", tot_ms, tot_fl / tot_ms * 1e-9);
    return 0;
}
