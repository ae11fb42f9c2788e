// Does the end-of-kernel write-back of dirty L2 lines cost the step?  A chain of dependent streaming kernels (the shape
// of affine_fast / bn_apply_fast: one-shot workgroups, 16 B per lane, all loads before the first use), ping-ponging
// between tensors, with (a) plain stores, (b) nontemporal stores, (c) write-through (sc1) stores by inline asm.
// Reports us per kernel in the chain (HIP events around 40 back-to-back launches) for 1 read + 1 write and 2 reads + 1 write.
//   hipcc --offload-arch=gfx950 -O3 -o wt_store wt_store.hip && ./wt_store
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void store16(u32x4* p, u32x4 v) {
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

template <int NR, int MODE, int UNR>
__global__ __launch_bounds__(256) void k_chain(const u32x4* __restrict__ x, const u32x4* __restrict__ z, u32x4* __restrict__ y, long n16) {
    const long base = (long)blockIdx.x * 256 * UNR + threadIdx.x;
    u32x4 v[UNR], w[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long i = base + u * 256;
        const long j = i < n16 ? i : 0;
        v[u] = x[j];
        if (NR > 1) w[u] = z[j];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long i = base + u * 256;
        if (i >= n16) break;
        u32x4 o = v[u];
        o.x ^= 0x10001u; o.y += 3u;
        if (NR > 1) { o.x += w[u].x; o.z ^= w[u].z; }
        store16<MODE>(y + i, o);
    }
}

template <int NR, int MODE>
float run(u32x4* a, u32x4* b, u32x4* c, long n16, int iters, hipStream_t s) {
    constexpr int UNR = 4;
    const int blocks = (int)((n16 + 256 * UNR - 1) / (256 * UNR));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k_chain<NR, MODE, UNR>), dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, c, (i & 1) ? a : b, n16);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_chain<NR, MODE, UNR>), dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, c, (i & 1) ? a : b, n16);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const long maxb = 268435456L;
    u32x4 *a, *b, *c;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&c, maxb));
    CK(hipMemset(a, 1, maxb)); CK(hipMemset(b, 2, maxb)); CK(hipMemset(c, 3, maxb));
    const long sizes[] = {8388608L, 16777216L, 33554432L, 67108864L, 134217728L, 268435456L};
    printf("%10s %3s  %9s %9s %9s   (us per kernel of a dependent chain; GB/s of the plain form)\n", "bytes", "NR", "plain", "nt", "sc1");
    for (long nb : sizes) {
        const long n16 = nb / 16;
        for (int rep = 0; rep < 2; ++rep) {
            float p1 = run<1, 0>(a, b, c, n16, 40, s), n1 = run<1, 1>(a, b, c, n16, 40, s), w1 = run<1, 2>(a, b, c, n16, 40, s);
            printf("%10ld %3d  %9.2f %9.2f %9.2f   %7.0f\n", nb, 1, p1, n1, w1, 2.0 * nb / p1 * 1e-3);
            float p2 = run<2, 0>(a, b, c, n16, 40, s), n2 = run<2, 1>(a, b, c, n16, 40, s), w2 = run<2, 2>(a, b, c, n16, 40, s);
            printf("%10ld %3d  %9.2f %9.2f %9.2f   %7.0f\n", nb, 2, p2, n2, w2, 3.0 * nb / p2 * 1e-3);
        }
    }
    return 0;
}
