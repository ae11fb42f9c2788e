// getb.hip -- GETB (global-local attention block) specific kernels:
// windowed multi-head attention with relative-position bias, and the
// "avgpool(ws,1) + avgpool(1,ws) + local" mixing pass.
//
// window_attn: one 64-lane wavefront per (window, head); lane = query token.
// K and V of the window/head (64 x d f32) sit in LDS, every lane walks the 64
// keys with its q in registers; softmax is lane-local (no cross-lane traffic).
// Reflect padding to a multiple of ws is folded into the load indices; the
// crop back to H x W is folded into the store.
#include "ledn_rt.h"

namespace ledn {

template <typename T, int D>
__global__ void __launch_bounds__(64) window_attn_kernel(const T* qkv, const float* biasT, T* out, int N,
                                                         int H, int W, int C, int heads, int hh,
                                                         int ww) {
    constexpr int WS = 8, T2 = 64;
    __shared__ float s_k[T2 * D];
    __shared__ float s_v[T2 * D];
    const int win = blockIdx.x, head = blockIdx.y;
    const int wx = win % ww, wy = (win / ww) % hh, n = win / (ww * hh);
    const int t = threadIdx.x;  // token = ws1*8 + ws2
    int y = wy * WS + t / WS, x = wx * WS + t % WS;
    const bool inside = y < H && x < W;
    const int ys = y < H ? y : 2 * H - 2 - y;  // reflect (pad < H guaranteed by the host)
    const int xs = x < W ? x : 2 * W - 2 - x;
    const T* p = qkv + (((long)n * H + ys) * W + xs) * (3L * C) + head * D;
    float q[D];
#pragma unroll
    for (int j = 0; j < D; j += 4) {
        float kv[4], vv[4];
        ldv<4>(p + j, q + j);
        ldv<4>(p + C + j, kv);
        ldv<4>(p + 2 * C + j, vv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s_k[t * D + j + i] = kv[i];
            s_v[t * D + j + i] = vv[i];
        }
    }
    __syncthreads();
    const float scale = rsqrtf((float)D);
    const float* b = biasT + (long)head * T2 * T2 + t;  // biasT[head][key][query]
    // two passes over the 64 keys, scores recomputed instead of kept (64 live score
    // registers spill; 2 x 16 FMAs per key are cheaper than scratch traffic)
    float m = -3.0e38f;
#pragma unroll 4
    for (int k = 0; k < T2; ++k) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) dot = fmaf(q[j], s_k[k * D + j], dot);
        m = fmaxf(m, dot * scale + b[k * T2]);
    }
    float o[D];
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = 0.f;
    float l = 0.f;
#pragma unroll 4
    for (int k = 0; k < T2; ++k) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) dot = fmaf(q[j], s_k[k * D + j], dot);
        const float e = __expf(dot * scale + b[k * T2] - m);
        l += e;
#pragma unroll
        for (int j = 0; j < D; ++j) o[j] = fmaf(e, s_v[k * D + j], o[j]);
    }
    if (!inside) return;
    const float inv = 1.f / l;
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] *= inv;
    T* op = out + (((long)n * H + y) * W + x) * C + head * D;
#pragma unroll
    for (int j = 0; j < D; j += 4) stv<4>(op + j, o + j);
}

int window_attn_impl(const void* qkv, const float* biasT, void* out, int N, int H, int W, int C,
                     int heads, int ws, int dtype, hipStream_t s) {
    LEDN_REQUIRE(qkv && biasT && out && N > 0 && H > 0 && W > 0 && C > 0 && heads > 0);
    LEDN_REQUIRE(ws == 8 && C % heads == 0);
    const int D = C / heads;
    const int hh = (H + ws - 1) / ws, ww = (W + ws - 1) / ws;
    LEDN_REQUIRE(hh * ws - H < H && ww * ws - W < W);  // reflect pad must be < size
    const dim3 grid((unsigned)(N * hh * ww), (unsigned)heads);
#define LEDN_WA(T, DD)                                                                          \
    LEDN_LAUNCH((window_attn_kernel<T, DD>), grid, dim3(64), 0, s, (const T*)qkv, biasT, (T*)out, N, H, \
                W, C, heads, hh, ww)
#define LEDN_WAD(T)                    \
    do {                               \
        if (D == 8) LEDN_WA(T, 8);     \
        else if (D == 16) LEDN_WA(T, 16); \
        else if (D == 32) LEDN_WA(T, 32); \
        else return LEDN_EINVAL;       \
    } while (0)
    if (dtype == LEDN_F32) LEDN_WAD(float);
    else if (dtype == LEDN_BF16) LEDN_WAD(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_WAD
#undef LEDN_WA
    return check_launch();
}

// out[y,x] = 1/ws * sum_{i<ws} P(y-ws/2+1+i, x) + 1/ws * sum_{i<ws} Q(y, x-ws/2+1+i) + local
// P = a reflect-extended by one row at the bottom, zero outside; Q likewise by one column.
template <typename T, int V>
__global__ void __launch_bounds__(256) getb_pool_kernel(const T* a, const T* local, T* out, int N, int H,
                                                        int W, int C, int ws) {
    const int cv = C / V;
    const long total = (long)N * H * W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cv) * V;
    const long pix = idx / cv;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int n = (int)(pix / ((long)W * H));
    const T* base = a + (long)n * H * W * C + c;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    const int p = ws / 2 - 1;
    for (int i = 0; i < ws; ++i) {
        int r = y - p + i;
        if (r >= 0 && r <= H) {
            if (r == H) r = H - 2;
            float t[V];
            ldv<V>(base + ((long)r * W + x) * C, t);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += t[v];
        }
        int q = x - p + i;
        if (q >= 0 && q <= W) {
            if (q == W) q = W - 2;
            float t[V];
            ldv<V>(base + ((long)y * W + q) * C, t);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += t[v];
        }
    }
    float lv[V];
    ldv<V>(local + pix * C + c, lv);
    const float inv = 1.f / (float)ws;
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = acc[v] * inv + lv[v];
    stv<V>(out + pix * C + c, acc);
}

int getb_pool_impl(const void* a, const void* local, void* out, int N, int H, int W, int C, int ws,
                   int dtype, hipStream_t s) {
    LEDN_REQUIRE(a && local && out && N > 0 && H >= 2 && W >= 2 && C > 0 && ws >= 2 && C % 4 == 0);
    const long total = (long)N * H * W * (C / 4);
    const dim3 grid((unsigned)cdiv(total, 256));
    if (dtype == LEDN_F32)
        LEDN_LAUNCH((getb_pool_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)a,
                    (const float*)local, (float*)out, N, H, W, C, ws);
    else if (dtype == LEDN_BF16)
        LEDN_LAUNCH((getb_pool_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (const bf16_t*)a,
                    (const bf16_t*)local, (bf16_t*)out, N, H, W, C, ws);
    else return LEDN_EINVAL;
    return check_launch();
}

}  // namespace ledn
