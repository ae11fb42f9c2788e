// norm_act.hip -- per-channel statistics (wavefront / LDS reductions), BatchNorm
// bookkeeping and the elementwise affine + residual + activation pass.
// All HBM-bound: one read (+ one write) of the activation, 16 B per lane.
#include "ledn_rt.h"

namespace ledn {

// x: [P][C].  Thread (r, cv): channel vector cv, pixel rows r, r+rows, ...
template <typename T, int V>
__global__ void __launch_bounds__(256) channel_stats_kernel(const T* x, const T* xadd, long P, int C,
                                                            float* sum, float* sqsum, float* part) {
    __shared__ float s_part[2][256 * (V > 4 ? V : 4)];
    const int cvn = C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    float a[V], b[V];
#pragma unroll
    for (int v = 0; v < V; ++v) a[v] = b[v] = 0.f;
    const bool worker = r < rows;
    if (worker) {
        for (long p = (long)blockIdx.x * rows + r; p < P; p += (long)gridDim.x * rows) {
            float xv[V];
            ldv<V>(x + p * C + cv * V, xv);
            if (xadd) {
                float xa[V];
                ldv<V>(xadd + p * C + cv * V, xa);
#pragma unroll
                for (int v = 0; v < V; ++v) xv[v] += xa[v];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                a[v] += xv[v];
                b[v] = fmaf(xv[v], xv[v], b[v]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        s_part[0][threadIdx.x * V + v] = a[v];
        s_part[1][threadIdx.x * V + v] = b[v];
    }
    __syncthreads();
    if (threadIdx.x < cvn) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float sa = 0.f, sb = 0.f;
            for (int rr = 0; rr < rows; ++rr) {
                sa += s_part[0][(rr * cvn + cv) * V + v];
                sb += s_part[1][(rr * cvn + cv) * V + v];
            }
            if (part) {
                part[(long)blockIdx.x * 2 * C + cv * V + v] = sa;
                part[(long)blockIdx.x * 2 * C + C + cv * V + v] = sb;
            } else {
                atomicAdd(sum + cv * V + v, sa);
                if (sqsum) atomicAdd(sqsum + cv * V + v, sb);
            }
        }
    }
}

struct FinishOuts {
    float* o[3];
};
// second stage of the two-stage reductions: workgroup (k-tile of 16 outputs, row split) sums
// its share of the partial rows with 16 row-slots per output, LDS-reduces the slots and adds the
// result with ONE atomic per output (<= 32 splits per address).
template <bool SOLE>
__global__ void __launch_bounds__(256) finish_partials_kernel(const float* part, int nblk, int C, int nout,
                                                              FinishOuts outs) {
    __shared__ float s_red[256];
    const int K = C * nout;
    const int j = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + j;
    const int per = (nblk + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float acc = 0.f;
    if (k < K) {
        int b = b0 + slot;
        for (; b + 48 < b1; b += 64) {   // 4 independent loads in flight
            const float v0 = part[(long)b * K + k], v1 = part[(long)(b + 16) * K + k];
            const float v2 = part[(long)(b + 32) * K + k], v3 = part[(long)(b + 48) * K + k];
            acc += (v0 + v1) + (v2 + v3);
        }
        for (; b < b1; b += 16) acc += part[(long)b * K + k];
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 16 && k < K) {
        float t = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) t += s_red[sl * 16 + threadIdx.x];
        float* dst = outs.o[k / C];
        if (dst) {
            if (SOLE) dst[k % C] += t;      // deterministic mode, gridDim.y == 1: this lane is the only adder of its output
            else atomicAdd(dst + k % C, t);
        }
    }
}

int finish_partials(const float* part, int nblk, int C, int nout, float* o0, float* o1, float* o2,
                    hipStream_t s) {
    if (deferred_stats().want && nout == 2 && !o2) {   // (sum, sum of squares) rows: the caller's BatchNorm
        deferred_stats().part = const_cast<float*>(part);   // finalize sums them (ledn_bn_finalize_rows)
        deferred_stats().rows = nblk;
        return check_launch();
    }
    FinishOuts outs;
    outs.o[0] = o0; outs.o[1] = o1; outs.o[2] = o2;
    long split = cdiv(nblk, 16 * 16);      // <= 16 rows per slot per workgroup
    if (split > 32) split = 32;
    if (split < 1) split = 1;
    if (det())                             // one workgroup column walks every row: a fixed summation order, one adder
        LEDN_LAUNCH(finish_partials_kernel<true>, dim3((unsigned)cdiv((long)C * nout, 16), 1u), dim3(256), 0,
                    s, part, nblk, C, nout, outs);
    else
        LEDN_LAUNCH(finish_partials_kernel<false>, dim3((unsigned)cdiv((long)C * nout, 16), (unsigned)split), dim3(256), 0,
                    s, part, nblk, C, nout, outs);
    return check_launch();
}

int channel_stats_impl(const void* x, const void* xadd, long long P, int C, int dtype, float* sum,
                       float* sqsum, hipStream_t s) {
    LEDN_REQUIRE(x && sum && P > 0 && C > 0);
    if (options().stream_fast & 1) {
        const int rc = channel_stats_fast(x, xadd, P, C, dtype, sum, sqsum, s);
        if (rc >= 0) return rc;
    }
    // 8 B per lane (4 x bf16): measured faster than 16 B per lane here -- twice the threads in flight
    // beat the wider access on these 33..270 MB tensors (r01l: 16 B/lane cost +15..30 %)
    const int V = (C % 4 == 0) ? 4 : 1;
    LEDN_REQUIRE(C / V <= 256);
    long nb = cdiv(P, (256 / (C / V)) * 8);
    float* part = nullptr;
    if (nb > 2048) nb = 2048;
    if (nb > 64 || det()) part = ws_take(nb * 2 * C);
    if (!part && nb > 256) nb = 256;     // atomics fallback: one per channel per workgroup, bounded grid
    const dim3 grid((unsigned)nb);
#define LEDN_CS(T)                                                                              \
    do {                                                                                        \
        if (V == 8)                                                                             \
            LEDN_LAUNCH((channel_stats_kernel<T, 8>), grid, dim3(256), 0, s, (const T*)x,       \
                        (const T*)xadd, (long)P, C, sum, sqsum, part);                      \
        else if (V == 4)                                                                        \
            LEDN_LAUNCH((channel_stats_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)x,       \
                        (const T*)xadd, (long)P, C, sum, sqsum, part);                      \
        else                                                                                    \
            LEDN_LAUNCH((channel_stats_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)x,       \
                        (const T*)xadd, (long)P, C, sum, sqsum, part);                      \
    } while (0)
    if (dtype == LEDN_F32) LEDN_CS(float);
    else if (dtype == LEDN_BF16) LEDN_CS(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_CS
    if (part) return finish_partials(part, (int)nb, C, 2, sum, sqsum, nullptr, s);
    return check_launch();
}

__global__ void bn_finalize_kernel(const float* sum, const float* sqsum, double count,
                                   const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float* scale,
                                   float* shift, float* mean_o, float* invstd_o, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = (double)sum[c] / count;
    double var = (double)sqsum[c] / count - m * m;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)m * sc;
    if (mean_o) mean_o[c] = (float)m;
    if (invstd_o) invstd_o[c] = invstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

template <typename TX, typename TY, int V>
__global__ void __launch_bounds__(256) affine_act_kernel(ledn_affine_desc d) {
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    if (r >= rows) return;
    const int c = cv * V;
    float sc[V], sh[V], sl[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        sc[i] = d.scale ? d.scale[c + i] : 1.f;
        sh[i] = d.shift ? d.shift[c + i] : 0.f;
        sl[i] = d.slope ? d.slope[c + i] : 0.f;
    }
    for (long p = (long)blockIdx.x * rows + r; p < d.P; p += (long)gridDim.x * rows) {
        const long off = p * d.C + c;
        float v[V];
        ldv<V>(reinterpret_cast<const TX*>(d.x) + off, v);
        if (d.xadd) {
            float a[V];
            ldv<V>(reinterpret_cast<const TX*>(d.xadd) + off, a);
#pragma unroll
            for (int i = 0; i < V; ++i) v[i] += a[i];
        }
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = v[i] * sc[i] + sh[i];
        if (d.res_mode != LEDN_RES_NONE) {
            float rr[V];
            ldv<V>(reinterpret_cast<const TY*>(d.res) + off, rr);
#pragma unroll
            for (int i = 0; i < V; ++i) v[i] = d.res_mode == LEDN_RES_ADD ? v[i] + rr[i] : v[i] * rr[i] + rr[i];
        }
        if (d.act != LEDN_ACT_NONE) {
#pragma unroll
            for (int i = 0; i < V; ++i) v[i] = act_apply(d.act, v[i], sl[i]);
        }
        stv<V>(reinterpret_cast<TY*>(d.y) + off, v);
    }
}

int affine_act_impl(const ledn_affine_desc& d, hipStream_t s) {
    LEDN_REQUIRE(d.x && d.y && d.P > 0 && d.C > 0);
    LEDN_REQUIRE((d.scale == nullptr) == (d.shift == nullptr));
    LEDN_REQUIRE(d.res_mode == LEDN_RES_NONE || d.res != nullptr);
    LEDN_REQUIRE(d.act != LEDN_ACT_PRELU || d.slope != nullptr);
    LEDN_REQUIRE((d.stat_sum == nullptr) == (d.stat_sqsum == nullptr));
    if ((options().stream_fast & 1) && d.act != LEDN_ACT_SIGMOID) {
        const int rc = affine_act_fast(d, s);
        if (rc >= 0) return rc;
    }
    LEDN_REQUIRE(!d.stat_sum);      // output statistics: the streaming kernel only (include/ledn.h)
    const bool v4 = d.C % 4 == 0;
    const bool v8 = false;   // 16 B per lane measured slower than 8 B per lane (see channel_stats_impl)
    const int cvn = v8 ? d.C / 8 : (v4 ? d.C / 4 : d.C);
    LEDN_REQUIRE(cvn <= 256);
    long nb = cdiv(d.P, (256 / cvn) * 4);
    if (nb > 4096) nb = 4096;
    const dim3 grid((unsigned)nb);
#define LEDN_AF(TX, TY)                                                                  \
    do {                                                                                 \
        if (v4) LEDN_LAUNCH((affine_act_kernel<TX, TY, 4>), grid, dim3(256), 0, s, d);   \
        else LEDN_LAUNCH((affine_act_kernel<TX, TY, 1>), grid, dim3(256), 0, s, d);      \
    } while (0)
    if (v8) LEDN_LAUNCH((affine_act_kernel<bf16_t, bf16_t, 8>), grid, dim3(256), 0, s, d);
    else if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_F32) LEDN_AF(float, float);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_BF16) LEDN_AF(bf16_t, bf16_t);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_F32) LEDN_AF(bf16_t, float);
    else if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_BF16) LEDN_AF(float, bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_AF
    return check_launch();
}

// ---- planar (NCHW) -> interleaved (NHWC) with per-channel affine and channel map
__device__ __forceinline__ float ld(const unsigned char* p) { return (float)*p; }

template <typename TX, typename TY>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const TX* x, TY* y, int N, int C, int H, int W,
                                                           const float* scale, const float* shift,
                                                           const int* map, const int* valid_hw, float pad_val) {
    const long plane = (long)H * W;
    const long total = (long)N * plane;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total) return;
    const long n = pix / plane, hw = pix % plane;
    // batch padding (stack_batch): outside the image's valid extent the value is pad_val (normalised domain)
    const bool data = !valid_hw || ((int)(hw / W) < valid_hw[2 * n] && (int)(hw % W) < valid_hw[2 * n + 1]);
    for (int c = 0; c < C; ++c) {
        const int cs = map ? map[c] : c;
        float v = ld(x + (n * C + cs) * plane + hw);
        if (scale) v = v * scale[c] + shift[c];
        st(y + pix * C + c, data ? v : pad_val);
    }
}

int nchw_to_nhwc_impl(const void* x, int dtype_x, void* y, int dtype_y, int N, int C, int H, int W,
                      const float* scale, const float* shift, const int* map, const int* valid_hw, float pad_val,
                      hipStream_t s) {
    LEDN_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0);
    LEDN_REQUIRE((scale == nullptr) == (shift == nullptr));
    const dim3 grid((unsigned)cdiv((long)N * H * W, 256));
#define LEDN_L(TX, TY) \
    LEDN_LAUNCH((nchw_to_nhwc_kernel<TX, TY>), grid, dim3(256), 0, s, (const TX*)x, (TY*)y, N, C, H, W, scale, shift, map, \
                valid_hw, pad_val)
    if (dtype_x == LEDN_F32 && dtype_y == LEDN_F32) LEDN_L(float, float);
    else if (dtype_x == LEDN_F32 && dtype_y == LEDN_BF16) LEDN_L(float, bf16_t);
    else if (dtype_x == LEDN_BF16 && dtype_y == LEDN_BF16) LEDN_L(bf16_t, bf16_t);
    else if (dtype_x == LEDN_U8 && dtype_y == LEDN_F32) LEDN_L(unsigned char, float);
    else if (dtype_x == LEDN_U8 && dtype_y == LEDN_BF16) LEDN_L(unsigned char, bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_L
    return check_launch();
}

// BatchNorm finalize straight from the producer's per-workgroup statistic rows part[rows][2][C]
// (one launch instead of finish_partials + bn_finalize).  Workgroup = 2 channels: thread = (one of
// the 4 sums, one of 64 row slots), 4 independent loads in flight, LDS reduction over the slots.
__global__ void __launch_bounds__(256) bn_finalize_rows_kernel(const float* part, int rows, double count,
                                                               const float* gamma, const float* beta,
                                                               float* running_mean, float* running_var,
                                                               float momentum, float eps, float* scale,
                                                               float* shift, float* mean_o, float* invstd_o,
                                                               float* sum_o, float* sqsum_o, int C) {
    __shared__ float s_red[256];
    const int o = threadIdx.x & 3, slot = threadIdx.x >> 2;
    const int ch = o & 1, kind = o >> 1;
    const int c = blockIdx.x * 2 + ch;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        const float* src = part + (long)kind * C + c;
        const long rs = 2L * C;
        int r = slot;
        for (; r + 192 < rows; r += 256) {
            a0 += src[(long)r * rs];
            a1 += src[(long)(r + 64) * rs];
            a2 += src[(long)(r + 128) * rs];
            a3 += src[(long)(r + 192) * rs];
        }
        for (; r < rows; r += 64) a0 += src[(long)r * rs];
    }
    s_red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int st = 128; st >= 4; st >>= 1) {
        if (threadIdx.x < st) s_red[threadIdx.x] += s_red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 2 && c < C) {      // thread = channel ch: s_red[ch] = sum, s_red[2 + ch] = sum of squares
        const float sm = s_red[threadIdx.x], sq = s_red[2 + threadIdx.x];
        if (sum_o) sum_o[c] = sm;
        if (sqsum_o) sqsum_o[c] = sq;
        const double m = (double)sm / count;
        double var = (double)sq / count - m * m;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float sc = g * invstd;
        scale[c] = sc;
        shift[c] = b - (float)m * sc;
        if (mean_o) mean_o[c] = (float)m;
        if (invstd_o) invstd_o[c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

int bn_finalize_rows_impl(const float* part, int rows, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* scale,
                          float* shift, float* mean, float* invstd, float* sum, float* sqsum, int C,
                          hipStream_t s) {
    LEDN_REQUIRE(part && rows > 0 && scale && shift && C > 0 && count > 0);
    LEDN_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    LEDN_LAUNCH(bn_finalize_rows_kernel, dim3((unsigned)cdiv(C, 2)), dim3(256), 0, s, part, rows, count, gamma,
                beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd, sum, sqsum, C);
    return check_launch();
}

int bn_finalize_impl(const float* sum, const float* sqsum, double count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, float* scale, float* shift, float* mean, float* invstd, int C,
                     hipStream_t s) {
    LEDN_REQUIRE(sum && sqsum && scale && shift && C > 0 && count > 0);
    LEDN_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    LEDN_LAUNCH(bn_finalize_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, s, sum, sqsum, count,
                gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd, C);
    return check_launch();
}

}  // namespace ledn
