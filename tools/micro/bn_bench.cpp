// Isolated timing of the BatchNorm / activation streaming entry points through the C ABI, hot (same buffers every
// launch: Infinity-Cache resident) and cold (rotating through NSET buffer sets, > 256 MB in total).
//   hipcc -O2 -o bn_bench bn_bench.cpp -I../../include -L../../led-net_amd/csrc -lledn_hip -Wl,-rpath,$PWD/../../led-net_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "ledn.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* ws; CK(hipMalloc(&ws, 128 << 20)); CK(hipMemset(ws, 0, 128 << 20)); ledn_set_workspace(ws, 32 << 20);
    const int NSET = 8;
    struct Shape { int C; long P; } shapes[] = {{64, 262144}, {128, 262144}, {32, 1048576}, {16, 262144}, {128, 65536}};
    float *par; CK(hipMalloc(&par, 16 * 512 * 4)); CK(hipMemset(par, 0, 16 * 512 * 4));
    std::vector<float> h(16 * 512, 0.5f);
    CK(hipMemcpy(par, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (auto sh : shapes) {
        const size_t bytes = (size_t)sh.P * sh.C * 2;
        void *z[NSET], *dy[NSET], *dz[NSET];
        for (int i = 0; i < NSET; ++i) { CK(hipMalloc(&z[i], bytes)); CK(hipMalloc(&dy[i], bytes)); CK(hipMalloc(&dz[i], bytes));
            CK(hipMemset(z[i], 0x3c, bytes)); CK(hipMemset(dy[i], 0x3d, bytes)); }
        for (int fast = 1; fast >= 0; --fast) {
            ledn_set_option(LEDN_OPT_STREAM_FAST, fast);
            for (int cold = 0; cold <= 1; ++cold) {
                for (int op = 0; op < 4; ++op) {   // 0 reduce, 1 apply, 2 affine, 3 stats
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    const int iters = 40;
                    for (int it = -4; it < iters; ++it) {
                        if (it == 0) hipEventRecord(e0, s);
                        const int k = cold ? ((it + 8) % NSET) : 0;
                        if (op <= 1) {
                            ledn_bnbwd_desc d; memset(&d, 0, sizeof d);
                            d.z = z[k]; d.dy = dy[k]; d.scale = par; d.shift = par + 512; d.mean = par + 1024; d.invstd = par + 1536;
                            d.sum_g = par + 2048; d.sum_gx = par + 2560; d.dz = dz[k]; d.count = (double)sh.P; d.P = sh.P; d.C = sh.C;
                            d.act = LEDN_ACT_RELU; d.res_mode = LEDN_RES_NONE; d.bn_mode = 1; d.dtype_z = LEDN_BF16; d.dtype_y = LEDN_BF16;
                            int rc = op == 0 ? ledn_bn_act_bwd_reduce(&d, s) : ledn_bn_act_bwd_apply(&d, s);
                            if (rc) { printf("rc %d\n", rc); return 1; }
                        } else if (op == 2) {
                            ledn_affine_desc d; memset(&d, 0, sizeof d);
                            d.x = z[k]; d.y = dz[k]; d.scale = par; d.shift = par + 512; d.P = sh.P; d.C = sh.C; d.act = LEDN_ACT_RELU;
                            d.res_mode = LEDN_RES_NONE; d.dtype_x = LEDN_BF16; d.dtype_y = LEDN_BF16;
                            if (ledn_affine_act(&d, s)) { printf("rc\n"); return 1; }
                        } else {
                            if (ledn_channel_stats(z[k], nullptr, sh.P, sh.C, LEDN_BF16, par + 3072, par + 3584, s)) { printf("rc\n"); return 1; }
                        }
                    }
                    hipEventRecord(e1, s); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    const char* names[] = {"bn_bwd_reduce(+finish)", "bn_bwd_apply", "affine_act", "channel_stats(+finish)"};
                    const double nb = (op == 0 ? 2.0 : op == 1 ? 3.0 : op == 2 ? 2.0 : 1.0) * bytes;
                    printf("C%-4d P%-8ld %-5s %-5s %-24s %7.1f us  %6.0f GB/s\n", sh.C, sh.P, fast ? "fast" : "gen", cold ? "cold" : "hot",
                           names[op], ms * 1e3 / iters, nb / (ms * 1e-3 / iters) * 1e-9);
                }
            }
        }
        for (int i = 0; i < NSET; ++i) { hipFree(z[i]); hipFree(dy[i]); hipFree(dz[i]); }
    }
    return 0;
}
