// resize_pool.hip -- bilinear resampling (+add, +NCHW/argmax output), adaptive
// average pooling and the 3x3/s2 average pool.  HBM-bound gathers; the output
// is written once, 16 B per lane where the channel count allows.
#include "ledn_rt.h"

namespace ledn {

template <typename TX, typename TY, int V>
__global__ void __launch_bounds__(256) bilinear_nhwc_kernel(ledn_resize_desc d) {
    const int cv = d.C / V;
    const long total = (long)d.N * d.Ho * d.Wo * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, d.Wo, d.Ho);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int wo = ix_.x, ho = ix_.y, n = ix_.n;
    const Lerp ly = lerp_coord(ho, d.H, d.Ho), lx = lerp_coord(wo, d.W, d.Wo);
    const TX* x = reinterpret_cast<const TX*>(d.x) + (long)n * d.H * d.W * d.C + c;
    float v00[V], v01[V], v10[V], v11[V], o[V];
    ldv<V>(x + ((long)ly.i0 * d.W + lx.i0) * d.C, v00);
    ldv<V>(x + ((long)ly.i0 * d.W + lx.i1) * d.C, v01);
    ldv<V>(x + ((long)ly.i1 * d.W + lx.i0) * d.C, v10);
    ldv<V>(x + ((long)ly.i1 * d.W + lx.i1) * d.C, v11);
#pragma unroll
    for (int i = 0; i < V; ++i)
        o[i] = ly.w0 * (lx.w0 * v00[i] + lx.w1 * v01[i]) + ly.w1 * (lx.w0 * v10[i] + lx.w1 * v11[i]);
    if (d.add) {
        float a[V];
        ldv<V>(reinterpret_cast<const TY*>(d.add) + pix * d.C + c, a);
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] += a[i];
    }
    stv<V>(reinterpret_cast<TY*>(d.y) + pix * d.C + c, o);
}

// small-C variant: one thread per output pixel, y written planar [N][C][Ho][Wo] f32,
// optional first-max argmax.
template <typename TX, int C>
__global__ void __launch_bounds__(256) bilinear_nchw_kernel(ledn_resize_desc d) {
    const long total = (long)d.N * d.Ho * d.Wo;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total) return;
    const NhwcIdx ix_ = pix_split(pix, d.Wo, d.Ho);
    const int wo = ix_.x, ho = ix_.y, n = ix_.n;
    const Lerp ly = lerp_coord(ho, d.H, d.Ho), lx = lerp_coord(wo, d.W, d.Wo);
    const TX* x = reinterpret_cast<const TX*>(d.x) + (long)n * d.H * d.W * C;
    float v00[C], v01[C], v10[C], v11[C], o[C];
    ldv<C>(x + ((long)ly.i0 * d.W + lx.i0) * C, v00);
    ldv<C>(x + ((long)ly.i0 * d.W + lx.i1) * C, v01);
    ldv<C>(x + ((long)ly.i1 * d.W + lx.i0) * C, v10);
    ldv<C>(x + ((long)ly.i1 * d.W + lx.i1) * C, v11);
    float* y = reinterpret_cast<float*>(d.y);
    const long plane = (long)d.Ho * d.Wo;
    int best = 0;
    float bv = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        o[i] = ly.w0 * (lx.w0 * v00[i] + lx.w1 * v01[i]) + ly.w1 * (lx.w0 * v10[i] + lx.w1 * v11[i]);
        if (d.add) o[i] += reinterpret_cast<const float*>(d.add)[pix * C + i];
        y[((long)n * C + i) * plane + (long)ho * d.Wo + wo] = o[i];
        if (i == 0 || o[i] > bv) {
            bv = o[i];
            best = i;
        }
    }
    if (d.argmax) d.argmax[pix] = (unsigned char)best;
}

int bilinear_impl(const ledn_resize_desc& d, hipStream_t s) {
    LEDN_REQUIRE(d.x && d.y);
    LEDN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.Ho > 0 && d.Wo > 0);
    if (d.out_nchw) {
        LEDN_REQUIRE(d.dtype_y == LEDN_F32);
        const dim3 grid((unsigned)cdiv((long)d.N * d.Ho * d.Wo, 256));
#define LEDN_BN(C)                                                                               \
    do {                                                                                         \
        if (d.dtype_x == LEDN_F32) LEDN_LAUNCH((bilinear_nchw_kernel<float, C>), grid, dim3(256), 0, s, d);  \
        else LEDN_LAUNCH((bilinear_nchw_kernel<bf16_t, C>), grid, dim3(256), 0, s, d);           \
    } while (0)
        switch (d.C) {
            case 1: LEDN_BN(1); break;
            case 2: LEDN_BN(2); break;
            case 3: LEDN_BN(3); break;
            case 4: LEDN_BN(4); break;
            case 5: LEDN_BN(5); break;
            case 8: LEDN_BN(8); break;
            case 19: LEDN_BN(19); break;
            default: return LEDN_EINVAL;
        }
#undef LEDN_BN
        return check_launch();
    }
    LEDN_REQUIRE(d.argmax == nullptr);
    const bool v4 = d.C % 4 == 0, v2 = d.C % 2 == 0;    // v2: the 2-class logit pyramid (one access per pixel)
    const long total = (long)d.N * d.Ho * d.Wo * (v4 ? d.C / 4 : (v2 ? d.C / 2 : d.C));
    const dim3 grid((unsigned)cdiv(total, 256));
#define LEDN_BL(TX, TY)                                                                    \
    do {                                                                                   \
        if (v4) LEDN_LAUNCH((bilinear_nhwc_kernel<TX, TY, 4>), grid, dim3(256), 0, s, d);  \
        else if (v2) LEDN_LAUNCH((bilinear_nhwc_kernel<TX, TY, 2>), grid, dim3(256), 0, s, d);  \
        else LEDN_LAUNCH((bilinear_nhwc_kernel<TX, TY, 1>), grid, dim3(256), 0, s, d);     \
    } while (0)
    if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_F32) LEDN_BL(float, float);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_BF16) LEDN_BL(bf16_t, bf16_t);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_F32) LEDN_BL(bf16_t, float);
    else if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_BF16) LEDN_BL(float, bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_BL
    return check_launch();
}

// ---- adaptive average pool to S x S (PyTorch window: [floor(i*H/S), ceil((i+1)*H/S)) )
// grid = (cells, row-chunks): a workgroup sums a slab of rows of one cell, lanes run
// along (pixel-in-row, channel-vector) so a wave reads whole NHWC row segments, then
// an LDS reduction over the pixel lanes and one f32 atomic per channel (y pre-zeroed).
template <typename T, int V>
__global__ void __launch_bounds__(256) adaptive_avgpool_kernel(const T* x, const T* xadd, float* y, int N,
                                                               int H, int W, int C, int S,
                                                               int rows_per_block, float* part) {
    __shared__ float s_acc[256 * V];
    const int cell = blockIdx.x;
    const int ox = cell % S, oy = (cell / S) % S, n = cell / (S * S);
    const int h0 = (oy * H) / S, h1 = ((oy + 1) * H + S - 1) / S;
    const int w0 = (ox * W) / S, w1 = ((ox + 1) * W + S - 1) / S;
    const int r0 = h0 + blockIdx.y * rows_per_block;
    const int r1 = min(h1, r0 + rows_per_block);
    const int cvn = C / V;
    const int lanes = 256 / cvn;  // pixel lanes per row sweep
    const int pl = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    if (pl < lanes) {
        for (int h = r0; h < r1; ++h)
            for (int w = w0 + pl; w < w1; w += lanes) {
                const long off = (((long)n * H + h) * W + w) * C + cv * V;
                float t[V];
                ldv<V>(x + off, t);
                if (xadd) {
                    float u[V];
                    ldv<V>(xadd + off, u);
#pragma unroll
                    for (int v = 0; v < V; ++v) t[v] += u[v];
                }
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] += t[v];
            }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) s_acc[threadIdx.x * V + v] = acc[v];
    __syncthreads();
    if (threadIdx.x < cvn && (part || r0 < r1)) {
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float t = 0.f;
            for (int p = 0; p < lanes; ++p) t += s_acc[(p * cvn + cv) * V + v];
            // part: one row of per-cell sums per row chunk, added up in chunk order by finish_partials
            // (a fixed summation order: inference is bit-reproducible; atomics onto y are not)
            if (part) part[((long)blockIdx.y * gridDim.x + cell) * C + cv * V + v] = r0 < r1 ? t * inv : 0.f;
            else if (gridDim.y == 1) y[(long)cell * C + cv * V + v] = t * inv;     // the cell's only workgroup (y zeroed)
            else atomicAdd(y + (long)cell * C + cv * V + v, t * inv);
        }
    }
}

int adaptive_avgpool_impl(const void* x, const void* xadd, float* y, int N, int H, int W, int C, int S,
                          int dtype, hipStream_t s) {
    LEDN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && S > 0);
    const int V = C % 4 == 0 ? 4 : 1;
    LEDN_REQUIRE(C / V <= 256);
    const int max_rows = (H + S - 1) / S + 1;                      // tallest adaptive window
    // row chunks per cell: a function of S only, so that an image's sums are added in the same order
    // whatever the batch size (batch-split invariance of inference, tests/test_full_size.py)
    int chunks = (int)cdiv(256, (long)S * S);
    if (chunks < 1) chunks = 1;
    if (chunks > max_rows) chunks = max_rows;
    const int rpb = (int)cdiv(max_rows, chunks);
    const dim3 grid((unsigned)(N * S * S), (unsigned)cdiv(max_rows, rpb));
    if (hipMemsetAsync(y, 0, sizeof(float) * (size_t)N * S * S * C, s) != hipSuccess) return LEDN_ELAUNCH;
    // <= 256 chunk rows: finish_partials adds each output's row sums in one thread group, in row order
    float* part = (grid.y > 1 && grid.y <= 256) ? ws_take((long)grid.y * N * S * S * C) : nullptr;
#define LEDN_AA(T)                                                                                    \
    do {                                                                                              \
        if (V == 4)                                                                                   \
            LEDN_LAUNCH((adaptive_avgpool_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)x,          \
                        (const T*)xadd, y, N, H, W, C, S, rpb, part);                                 \
        else                                                                                          \
            LEDN_LAUNCH((adaptive_avgpool_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)x,          \
                        (const T*)xadd, y, N, H, W, C, S, rpb, part);                                 \
    } while (0)
    if (dtype == LEDN_F32) LEDN_AA(float);
    else if (dtype == LEDN_BF16) LEDN_AA(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_AA
    if (part) return finish_partials(part, (int)grid.y, N * S * S * C, 1, y, nullptr, nullptr, s);
    return check_launch();
}

// ---- 3x3 / stride 2 / pad 1 average pool, divisor always 9
template <typename T, int V>
__global__ void __launch_bounds__(256) avgpool3x3s2_kernel(const T* x, T* y, int N, int H, int W, int C,
                                                           int Ho, int Wo) {
    const int cv = C / V;
    const long total = (long)N * Ho * Wo * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, Wo, Ho);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int wo = ix_.x, ho = ix_.y, n = ix_.n;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - 1 + kh;
        if (hi < 0 || hi >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - 1 + kw;
            if (wi < 0 || wi >= W) continue;
            float xv[V];
            ldv<V>(x + (((long)n * H + hi) * W + wi) * C + c, xv);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += xv[v];
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] *= (1.f / 9.f);
    stv<V>(y + pix * C + c, acc);
}

int avgpool3x3s2_impl(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int dtype,
                      hipStream_t s) {
    LEDN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0);
    LEDN_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1);
    const bool v4 = C % 4 == 0;
    const long total = (long)N * Ho * Wo * (v4 ? C / 4 : C);
    const dim3 grid((unsigned)cdiv(total, 256));
#define LEDN_AP(T)                                                                                 \
    do {                                                                                           \
        if (v4) LEDN_LAUNCH((avgpool3x3s2_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)x, (T*)y, N, H, W, C, Ho, Wo); \
        else LEDN_LAUNCH((avgpool3x3s2_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)x, (T*)y, N, H, W, C, Ho, Wo);    \
    } while (0)
    if (dtype == LEDN_F32) LEDN_AP(float);
    else if (dtype == LEDN_BF16) LEDN_AP(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_AP
    return check_launch();
}

// ---- k x k / stride s / pad p average pool, zero padding counted in the divisor (nn.AvgPool2d defaults:
// count_include_pad=True, ceil_mode=False): the pooling pyramid of DAPPM / PAPPM (utils/ppm.py:66-70)
template <typename T, int V>
__global__ void __launch_bounds__(256) avgpool2d_kernel(const T* x, T* y, int N, int H, int W, int C, int Ho,
                                                        int Wo, int k, int st, int pad) {
    const int cv = C / V;
    const long total = (long)N * Ho * Wo * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, Wo, Ho);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int wo = ix_.x, ho = ix_.y, n = ix_.n;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    const int h0 = max(ho * st - pad, 0), h1 = min(ho * st - pad + k, H);
    const int w0 = max(wo * st - pad, 0), w1 = min(wo * st - pad + k, W);
    for (int hi = h0; hi < h1; ++hi)
        for (int wi = w0; wi < w1; ++wi) {
            float xv[V];
            ldv<V>(x + (((long)n * H + hi) * W + wi) * C + c, xv);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += xv[v];
        }
    const float inv = 1.f / (float)(k * k);
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] *= inv;
    stv<V>(y + pix * C + c, acc);
}

int avgpool2d_impl(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int k, int st, int pad,
                   int dtype, hipStream_t s) {
    LEDN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && st > 0 && pad >= 0 && 2 * pad <= k);
    LEDN_REQUIRE(Ho == (H + 2 * pad - k) / st + 1 && Wo == (W + 2 * pad - k) / st + 1 && Ho > 0 && Wo > 0);
    const bool v4 = C % 4 == 0;
    const long total = (long)N * Ho * Wo * (v4 ? C / 4 : C);
    const dim3 grid((unsigned)cdiv(total, 256));
#define LEDN_AP(T)                                                                                          \
    do {                                                                                                    \
        if (v4) LEDN_LAUNCH((avgpool2d_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)x, (T*)y, N, H, W, C, Ho, Wo, k, st, pad); \
        else LEDN_LAUNCH((avgpool2d_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)x, (T*)y, N, H, W, C, Ho, Wo, k, st, pad);    \
    } while (0)
    if (dtype == LEDN_F32) LEDN_AP(float);
    else if (dtype == LEDN_BF16) LEDN_AP(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_AP
    return check_launch();
}

// adjoint in gather form: dx[y,x] = 1/k^2 * sum of dy over the windows that contain (y,x)
template <typename T, int V>
__global__ void __launch_bounds__(256) avgpool2d_bwd_kernel(const T* dy, T* dx, int N, int H, int W, int C, int Ho,
                                                            int Wo, int k, int st, int pad) {
    const int cv = C / V;
    const long total = (long)N * H * W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, W, H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int xx = ix_.x, yy = ix_.y, n = ix_.n;
    // windows ho with ho*st - pad <= yy < ho*st - pad + k
    const int ho1 = min((yy + pad) / st, Ho - 1), wo1 = min((xx + pad) / st, Wo - 1);
    const int ho0 = max((yy + pad - k + st) / st, 0), wo0 = max((xx + pad - k + st) / st, 0);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    for (int ho = ho0; ho <= ho1; ++ho)
        for (int wo = wo0; wo <= wo1; ++wo) {
            float g[V];
            ldv<V>(dy + (((long)n * Ho + ho) * Wo + wo) * C + c, g);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += g[v];
        }
    const float inv = 1.f / (float)(k * k);
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] *= inv;
    stv<V>(dx + pix * C + c, acc);
}

int avgpool2d_bwd_impl(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int k, int st, int pad,
                       int dtype, hipStream_t s) {
    LEDN_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && st > 0 && pad >= 0 && 2 * pad <= k);
    LEDN_REQUIRE(Ho == (H + 2 * pad - k) / st + 1 && Wo == (W + 2 * pad - k) / st + 1 && Ho > 0 && Wo > 0);
    const bool v4 = C % 4 == 0;
    const long total = (long)N * H * W * (v4 ? C / 4 : C);
    const dim3 grid((unsigned)cdiv(total, 256));
#define LEDN_AP(T)                                                                                          \
    do {                                                                                                    \
        if (v4) LEDN_LAUNCH((avgpool2d_bwd_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)dy, (T*)dx, N, H, W, C, Ho, Wo, k, st, pad); \
        else LEDN_LAUNCH((avgpool2d_bwd_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)dy, (T*)dx, N, H, W, C, Ho, Wo, k, st, pad);    \
    } while (0)
    if (dtype == LEDN_F32) LEDN_AP(float);
    else if (dtype == LEDN_BF16) LEDN_AP(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_AP
    return check_launch();
}

// ---------------------------------------------------------------------------
// evaluation histograms (see include/ledn.h: ledn_iou_hist): LDS counters per workgroup, one
// global atomic per touched counter per workgroup (f32 counts, exact below 2^24 per image)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) iou_hist_kernel(const unsigned char* pred, const long long* label, long P,
                                                       int C, int ignore_index, float* hist) {
    __shared__ unsigned s_h[3 * 256];
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) s_h[i] = 0u;
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const long long lb = label[p];
        if (lb == ignore_index) continue;
        const int pr = pred[p];
        if (pr < C) {
            atomicAdd(&s_h[C + pr], 1u);
            if (pr == lb) atomicAdd(&s_h[pr], 1u);
        }
        if (lb >= 0 && lb < C) atomicAdd(&s_h[2 * C + (int)lb], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x)
        if (s_h[i]) atomicAdd(hist + i, (float)s_h[i]);
}

int iou_hist_impl(const unsigned char* pred, const long long* label, long long P, int num_classes,
                  int ignore_index, float* hist, hipStream_t s) {
    LEDN_REQUIRE(pred && label && hist && P > 0 && num_classes > 0 && num_classes <= 256);
    long nb = cdiv((long)P, 256 * 16);
    if (nb > 1024) nb = 1024;
    LEDN_LAUNCH(iou_hist_kernel, dim3((unsigned)nb), dim3(256), 0, s, pred, label, (long)P, num_classes,
                ignore_index, hist);
    return check_launch();
}

}  // namespace ledn
