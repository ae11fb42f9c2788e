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

// Matrix-core variant (bf16, head dim 16, windows tile the map; see window_attn_bwd_mfma_kernel in
// attn_loss_opt.hip for the layout): S^T = K Q^T with fragments straight from global memory, softmax
// in-lane over the accumulator registers (+ one exchange with lane ^ 32), O^T = V^T P^T with the
// probability registers as the B operand and V^T gathered from a 2 KB LDS tile in the matching k order.
__global__ void __launch_bounds__(64) window_attn_mfma_kernel(const bf16_t* qkv, const float* biasT, bf16_t* out,
                                                              int N, int H, int W, int C, int heads, int hh,
                                                              int ww) {
    constexpr int D = 16, T2 = 64, LP = 72;
    __shared__ __attribute__((aligned(16))) unsigned short s_vt[D * LP];
    const int head = blockIdx.y;
    const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
    const int nwin = N * hh * ww;
    const float scale = rsqrtf((float)D);
    float bias[2][2][16];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bias[jt][it][r] = biasT[(long)head * T2 * T2 + (jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * T2 + it * 32 + lr];
    const bf16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
        const int wx = win % ww, wy = (win / ww) % hh, n = win / (ww * hh);
        bf16x8_t kf[2], vf[2], qf[2];
        long tokoff[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int tok = tl * 32 + lr;
            const long pix = ((long)n * H + wy * 8 + tok / 8) * W + wx * 8 + tok % 8;
            tokoff[tl] = pix;
            const bf16_t* p = qkv + pix * (3L * C) + head * D + lh * 8;
            qf[tl] = *reinterpret_cast<const bf16x8_t*>(p);
            kf[tl] = *reinterpret_cast<const bf16x8_t*>(p + C);
            vf[tl] = *reinterpret_cast<const bf16x8_t*>(p + 2 * C);
        }
        __syncthreads();     // previous window's readers of s_vt are done
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int e = 0; e < 8; ++e) s_vt[(lh * 8 + e) * LP + tl * 32 + lr] = (unsigned short)vf[tl][e];
        f32x16_t st[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                f32x16_t z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                st[jt][it] = mfma_32x32x16_bf16(kf[jt], qf[it], z);
            }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float m = -3.0e38f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[jt][it][r] = st[jt][it][r] * scale + bias[jt][it][r];
                    m = fmaxf(m, st[jt][it][r]);
                }
            m = fmaxf(m, __shfl_xor(m, 32));
            float l = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[jt][it][r] = __expf(st[jt][it][r] - m);
                    l += st[jt][it][r];
                }
            l += __shfl_xor(l, 32);
            const float inv = 1.f / l;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[jt][it][r] *= inv;
        }
        __syncthreads();     // V^T tile complete
        f32x16_t ot[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[it][r] = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8_t a = zero8;
                if (lr < D) {
                    const int j0 = jt * 32 + 16 * h + 4 * lh;
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = (short)s_vt[lr * LP + j0 + (e & 3) + 8 * (e >> 2)];
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    bf16x8_t b;
#pragma unroll
                    for (int e = 0; e < 8; ++e) b[e] = (short)f32_to_bf16(st[jt][it][8 * h + e]);
                    ot[it] = mfma_32x32x16_bf16(a, b, ot[it]);
                }
            }
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            bf16_t* o = out + tokoff[tl] * C + head * D + 4 * lh;
            float t4[4];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[e] = ot[tl][4 * g + e];
                st4(o + 8 * g, t4);
            }
        }
    }
}

int window_attn_impl(const void* qkv, const float* biasT, void* out, int N, int H, int W, int C,
                     int heads, int ws, int dtype, hipStream_t s) {
    LEDN_REQUIRE(qkv && biasT && out && N > 0 && H > 0 && W > 0 && C > 0 && heads > 0);
    LEDN_REQUIRE(ws == 8 && C % heads == 0);
    const int D = C / heads;
    const int hh = (H + ws - 1) / ws, ww = (W + ws - 1) / ws;
    LEDN_REQUIRE(hh * ws - H < H && ww * ws - W < W);  // reflect pad must be < size
    if (dtype == LEDN_BF16 && D == 16 && H % ws == 0 && W % ws == 0 && C % 8 == 0) {   // matrix-core variant
        long nb = (long)N * hh * ww;
        if (nb > 512) nb = 512;                          // wavefronts per head, each walks its windows
        LEDN_LAUNCH(window_attn_mfma_kernel, dim3((unsigned)nb, (unsigned)heads), dim3(64), 0, s, (const bf16_t*)qkv,
                    biasT, (bf16_t*)out, N, H, W, C, heads, hh, ww);
        return check_launch();
    }
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
    const NhwcIdx ix_ = nhwc_split(idx, cv, W, H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
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

// ---------------------------------------------------------------------------
// relative-position bias of the window attention (UNetFormer_GETB.py:181-187):
//   biasT[h][j][i] = table[index[i*T + j]][h]      (T = ws*ws tokens, table: (2ws-1)^2 x heads)
// and its adjoint in gather form (deterministic, no atomics): workgroup r sums dbiasT over the
// token pairs that index table row r.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) relpos_bias_kernel(const float* table, const long long* index, float* out,
                                                          int R, int heads, int T) {
    const long total = (long)heads * T * T;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % T), j = (int)((idx / T) % T), h = (int)(idx / ((long)T * T));
    const long long r = index[(long)i * T + j];
    out[idx] = (r >= 0 && r < R) ? table[r * heads + h] : 0.f;
}

// scatter-add adjoint of the gather: one workgroup per head accumulates its T x T gradient values into an LDS copy of
// the table column (LDS atomics: a few hundred distinct addresses) and adds the column to dtable -- one writer per
// (entry, head), no global atomics.  (The first version ran one workgroup per table entry, each scanning the whole
// index map once per head with a tree reduction: 61 us for 2 x 32 K values on the context branch's exposed tail.)
template <bool DET>
__global__ void __launch_bounds__(256) relpos_bias_bwd_kernel(const float* dbias, const long long* index, float* dtable,
                                                              int R, int heads, int T) {
    constexpr int RMAX = 1024;
    __shared__ float s_t[RMAX];
    const int h = blockIdx.x;
    if (DET) {
        // deterministic mode: the same scatter, accumulated as 64-bit FIXED-POINT integers (scale 2^40: resolution 9e-13,
        // range +-8e6) -- integer addition is associative, so the order in which the LDS atomics arrive does not matter.
        // (The first deterministic form, one thread per table row scanning the whole index map, took 414 us per launch.)
        __shared__ unsigned long long s_q[RMAX];
        for (int i = threadIdx.x; i < R; i += 256) s_q[i] = 0ull;
        __syncthreads();
        for (int e = threadIdx.x; e < T * T; e += 256) {
            const int i = e / T, j = e % T;
            const long long r = index[e];
            if (r >= 0 && r < R) {
                const float v = dbias[((long)h * T + j) * T + i];
                const long long q = (long long)llrintf(fminf(fmaxf(v, -8.0e6f), 8.0e6f) * 1099511627776.f);
                atomicAdd(&s_q[r], (unsigned long long)q);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < R; i += 256)
            dtable[(long)i * heads + h] += (float)((double)(long long)s_q[i] * (1.0 / 1099511627776.0));
        return;
    }
    for (int i = threadIdx.x; i < R; i += 256) s_t[i] = 0.f;
    __syncthreads();
    for (int e = threadIdx.x; e < T * T; e += 256) {
        const int i = e / T, j = e % T;
        const long long r = index[e];
        if (r >= 0 && r < R) atomicAdd(&s_t[r], dbias[((long)h * T + j) * T + i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += 256) dtable[(long)i * heads + h] += s_t[i];
}

int relpos_bias_impl(const float* table, const long long* index, float* out, int R, int heads, int T, hipStream_t s) {
    LEDN_REQUIRE(table && index && out && R > 0 && heads > 0 && T > 0);
    LEDN_LAUNCH(relpos_bias_kernel, dim3((unsigned)cdiv((long)heads * T * T, 256)), dim3(256), 0, s, table, index, out,
                R, heads, T);
    return check_launch();
}

int relpos_bias_bwd_impl(const float* dbias, const long long* index, float* dtable, int R, int heads, int T,
                         hipStream_t s) {
    LEDN_REQUIRE(dbias && index && dtable && R > 0 && R <= 1024 && heads > 0 && T > 0);
    if (det()) LEDN_LAUNCH(relpos_bias_bwd_kernel<true>, dim3((unsigned)heads), dim3(256), 0, s, dbias, index, dtable, R, heads, T);
    else LEDN_LAUNCH(relpos_bias_bwd_kernel<false>, dim3((unsigned)heads), dim3(256), 0, s, dbias, index, dtable, R, heads, T);
    return check_launch();
}

}  // namespace ledn
