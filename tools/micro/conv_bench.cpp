// Isolated timing of the MFMA convolution entries (forward with statistics, data gradient, weight gradient) through
// the C ABI at the shapes of the 16 x 1024 x 1024 train step; cold = rotating buffer sets (> Infinity Cache).
//   hipcc -O2 -o conv_bench conv_bench.cpp -I../../include -L../../led-net_amd/csrc -lledn_hip
//   ./conv_bench [shape-index | -1] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ledn.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Shape { const char* kind; int ci, co, k, s, g, N, H, W; };
static const Shape SHAPES[] = {
    {"fwd", 32, 32, 3, 1, 1, 16, 256, 256}, {"dgrad", 32, 32, 3, 1, 1, 16, 256, 256}, {"wgrad", 32, 32, 3, 1, 1, 16, 256, 256},
    {"fwd", 64, 64, 3, 1, 1, 16, 128, 128}, {"dgrad", 64, 64, 3, 1, 1, 16, 128, 128}, {"wgrad", 64, 64, 3, 1, 1, 16, 128, 128},
    {"fwd", 32, 32, 1, 1, 1, 16, 512, 512}, {"wgrad", 32, 32, 1, 1, 1, 16, 512, 512},
    {"fwd", 32, 32, 3, 2, 1, 16, 512, 512}, {"dgrad", 32, 32, 3, 2, 1, 16, 512, 512}, {"wgrad", 32, 32, 3, 2, 1, 16, 512, 512},
    {"fwd", 64, 64, 1, 1, 4, 16, 128, 128}, {"dgrad", 64, 64, 1, 1, 4, 16, 128, 128}, {"wgrad", 64, 64, 1, 1, 4, 16, 128, 128},
    {"fwd", 128, 128, 1, 1, 4, 16, 128, 128}, {"fwd", 128, 64, 3, 1, 1, 16, 128, 128}, {"fwd", 64, 16, 1, 1, 4, 16, 128, 128},
    {"fwd", 64, 128, 3, 2, 1, 16, 128, 128}, {"wgrad", 64, 128, 3, 2, 1, 16, 128, 128},
    // 1x1 stride 1 is layout-agnostic: the same tensors viewed as [N*H*W/512][16][32] make every 16 x 32 tile a
    // LINEAR 512-pixel chunk of memory instead of 16 segments a row pitch apart
    {"fwd", 64, 64, 1, 1, 4, 512, 16, 32}, {"dgrad", 64, 64, 1, 1, 4, 512, 16, 32}, {"fwd", 32, 32, 1, 1, 1, 8192, 16, 32},
    {"fwd", 128, 128, 1, 1, 4, 512, 16, 32}, {"fwd", 64, 16, 1, 1, 4, 512, 16, 32},
    // under-filled grids (fewer 16-row tiles than workgroups wanted): inference batch 8 and the narrow 1/8-resolution convs
    {"fwd", 64, 16, 1, 1, 4, 8, 128, 128}, {"fwd", 16, 64, 1, 1, 4, 8, 128, 128}, {"fwd", 64, 32, 1, 1, 1, 8, 128, 128},
    {"fwd", 64, 32, 3, 1, 1, 8, 128, 128}, {"fwd", 32, 32, 3, 1, 1, 8, 128, 128}, {"dgrad", 64, 16, 1, 1, 4, 16, 128, 128},
    {"fwd", 16, 64, 1, 1, 4, 16, 128, 128}, {"fwd", 128, 32, 1, 1, 1, 16, 64, 64},
};

#include <cstdlib>
int main(int argc, char** argv) {
    if (getenv("LEDN_STREAM_FAST")) ledn_set_option(LEDN_OPT_STREAM_FAST, atoi(getenv("LEDN_STREAM_FAST")));

    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const int iters = argc > 2 ? atoi(argv[2]) : 30;
    hipStream_t s; CK(hipStreamCreate(&s));
    float* ws; CK(hipMalloc(&ws, 128 << 20)); CK(hipMemset(ws, 0, 128 << 20)); ledn_set_workspace(ws, 32 << 20);
    const int NSET = 6;
    float* stats; CK(hipMalloc(&stats, 4096 * 4)); CK(hipMemset(stats, 0, 4096 * 4));
    int idx = -1;
    for (const Shape& sh : SHAPES) {
        ++idx;
        if (only >= 0 && idx != only) continue;
        const int pad = sh.k / 2;
        const int Ho = (sh.H + 2 * pad - sh.k) / sh.s + 1, Wo = (sh.W + 2 * pad - sh.k) / sh.s + 1;
        const size_t xb = (size_t)sh.N * sh.H * sh.W * sh.ci * 2, zb = (size_t)sh.N * Ho * Wo * sh.co * 2;
        const size_t wn = (size_t)sh.co * (sh.ci / sh.g) * sh.k * sh.k;
        void *x[NSET], *z[NSET];
        for (int i = 0; i < NSET; ++i) { CK(hipMalloc(&x[i], xb)); CK(hipMalloc(&z[i], zb)); CK(hipMemset(x[i], 0x3c, xb)); CK(hipMemset(z[i], 0x3c, zb)); }
        float* w; CK(hipMalloc(&w, wn * 4)); std::vector<float> hw(wn, 0.01f); CK(hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice));
        float* dw; CK(hipMalloc(&dw, wn * 4)); CK(hipMemset(dw, 0, wn * 4));
        void *wp0, *wp1; CK(hipMalloc(&wp0, (size_t)sh.co * sh.ci * sh.k * sh.k * 2)); CK(hipMalloc(&wp1, (size_t)sh.co * sh.ci * sh.k * sh.k * 2));
        ledn_pack_conv_weights(w, wp0, sh.co, sh.ci, sh.k, sh.k, 0, sh.g, s);
        ledn_pack_conv_weights(w, wp1, sh.co, sh.ci, sh.k, sh.k, 1, sh.g, s);
        for (int cold = 0; cold <= 1; ++cold) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            int mf = -1;
            for (int it = -3; it < iters; ++it) {
                if (it == 0) hipEventRecord(e0, s);
                const int k = cold ? ((it + 6) % NSET) : 0;
                int rc = 0;
                if (!strcmp(sh.kind, "wgrad")) {
                    ledn_wgrad_desc d; memset(&d, 0, sizeof d);
                    d.x = x[k]; d.dz = z[k]; d.dw = dw; d.ws_co = (long long)(sh.ci / sh.g) * sh.k * sh.k; d.ws_ci = sh.k * sh.k; d.ws_tap = 1;
                    d.N = sh.N; d.H = sh.H; d.W = sh.W; d.Cin = sh.ci; d.Ho = Ho; d.Wo = Wo; d.Cout = sh.co; d.KH = d.KW = sh.k; d.stride = sh.s;
                    d.pad = pad; d.dil = 1; d.groups = sh.g; d.dtype_x = LEDN_BF16; d.dtype_dz = LEDN_BF16;
                    if (mf < 0) mf = ledn_conv2d_wgrad_uses_mfma(&d);
                    rc = ledn_conv2d_wgrad(&d, s);
                } else {
                    const bool T = !strcmp(sh.kind, "dgrad");
                    ledn_conv_desc d; memset(&d, 0, sizeof d);
                    d.x = T ? z[k] : x[k]; d.y = T ? x[(k + 1) % NSET] : z[(k + 1) % NSET]; d.w = w; d.w_bf16 = T ? wp1 : wp0;
                    if (!T) { d.stat_sum = stats; d.stat_sqsum = stats + 2048; }
                    d.ws_co = T ? sh.k * sh.k : (long long)(sh.ci / sh.g) * sh.k * sh.k; d.ws_ci = T ? (long long)(sh.ci / sh.g) * sh.k * sh.k : sh.k * sh.k; d.ws_tap = 1;
                    d.N = sh.N; d.KH = d.KW = sh.k; d.stride = sh.s; d.pad = pad; d.dil = 1; d.groups = sh.g; d.dtype_x = d.dtype_y = LEDN_BF16; d.transposed = T;
                    if (!T) { d.H = sh.H; d.W = sh.W; d.Cin = sh.ci; d.Ho = Ho; d.Wo = Wo; d.Cout = sh.co; }
                    else { d.H = Ho; d.W = Wo; d.Cin = sh.co; d.Ho = sh.H; d.Wo = sh.W; d.Cout = sh.ci; }
                    if (mf < 0) mf = ledn_conv2d_uses_mfma(&d);
                    rc = ledn_conv2d(&d, s);
                }
                if (rc) { printf("[%d] %s rc=%d\n", idx, sh.kind, rc); return 1; }
            }
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / iters, flops = 2.0 * sh.N * Ho * Wo * sh.co * (sh.ci / sh.g) * sh.k * sh.k;
            printf("[%2d] %-5s %dx%d %3d->%3d g%d s%d %dx%dx%d %-4s mfma=%d  %7.1f us  %6.0f GB/s  %6.1f TF/s\n", idx, sh.kind, sh.k, sh.k, sh.ci, sh.co,
                   sh.g, sh.s, sh.N, sh.H, sh.W, cold ? "cold" : "hot", mf, us, (xb + zb) / us * 1e-3, flops / us * 1e-6);
        }
        for (int i = 0; i < NSET; ++i) { hipFree(x[i]); hipFree(z[i]); }
        hipFree(w); hipFree(dw); hipFree(wp0); hipFree(wp1);
    }
    return 0;
}
