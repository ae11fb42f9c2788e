// Calibration: how fast can a trivially simple streaming kernel go on MI355X at the tensor sizes of the
// LED-Net train step (33 / 67 / 134 / 268 MB bf16), back-to-back on one stream, as a function of launch shape?
//   hipcc --offload-arch=gfx950 -O3 -o stream_cal stream_cal.hip && ./stream_cal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// y = a*x (+ z): NR reads, 1 write, 16 B per lane, UNR independent 16-B loads in flight per lane
template <int NR, int UNR>
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ x, const uint4* __restrict__ z, uint4* __restrict__ y, long n16) {
    const long stride = (long)gridDim.x * 256 * UNR;
    for (long base = (long)blockIdx.x * 256 * UNR + threadIdx.x; base < n16; base += stride) {
        uint4 v[UNR], w[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            long i = base + u * 256;
            if (i < n16) { v[u] = x[i]; if (NR > 1) w[u] = z[i]; }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            long i = base + u * 256;
            if (i < n16) {
                uint4 o = v[u];
                o.x ^= 0x10001u; o.y += 3u;
                if (NR > 1) { o.x += w[u].x; o.y ^= w[u].y; o.z += w[u].z; o.w ^= w[u].w; }
                y[i] = o;
            }
        }
    }
}

template <int NR, int UNR>
float run(const uint4* x, const uint4* z, uint4* y, long n16, int blocks, int iters, hipStream_t s) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<NR, UNR>), dim3(blocks), dim3(256), 0, s, x, z, y, n16);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_stream<NR, UNR>), dim3(blocks), dim3(256), 0, s, x, z, y, n16);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const long maxb = 268435456L;   // 268 MB
    uint4 *x, *z, *y;
    CK(hipMalloc(&x, maxb)); CK(hipMalloc(&z, maxb)); CK(hipMalloc(&y, maxb));
    CK(hipMemset(x, 1, maxb)); CK(hipMemset(z, 2, maxb));
    const long sizes[] = {8388608L, 33554432L, 67108864L, 134217728L, 268435456L};
    printf("%10s %4s %4s %8s %9s %9s\n", "bytes", "NR", "UNR", "blocks", "us", "GB/s");
    for (long nb : sizes) {
        long n16 = nb / 16;
        for (int nr = 1; nr <= 2; ++nr) {
            std::vector<int> grids = {256, 512, 1024, 2048, 4096, 8192, (int)((n16 + 255) / 256), (int)((n16 + 1023) / 1024)};
            for (int g : grids) {
                if ((long)g * 256 > n16 * 2) continue;
                float a = nr == 1 ? run<1, 1>(x, z, y, n16, g, 30, s) : run<2, 1>(x, z, y, n16, g, 30, s);
                float b = nr == 1 ? run<1, 4>(x, z, y, n16, g, 30, s) : run<2, 4>(x, z, y, n16, g, 30, s);
                double bytes = (double)nb * (nr + 1);
                printf("%10ld %4d %4d %8d %9.1f %9.0f\n", nb, nr, 1, g, a, bytes / a * 1e-3);
                printf("%10ld %4d %4d %8d %9.1f %9.0f\n", nb, nr, 4, g, b, bytes / b * 1e-3);
            }
        }
    }
    // launch floor: empty-ish kernel back to back
    float t = run<1, 1>(x, z, y, 256, 1, 200, s);
    printf("tiny kernel back-to-back: %.2f us per launch\n", t);
    return 0;
}
