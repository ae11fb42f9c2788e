// dwconv.hip -- depthwise convolutions: generic KxK (SESP second stage, GETB 8x8)
// and the fused SESP pyramid (4 dilated branches + hierarchical adds).
//
// HBM-bound streaming stencils.  One thread = one output pixel x 4 channels
// (16 B f32 / 8 B bf16 per lane, lanes run along channels then pixels, so a
// wave reads whole NHWC rows).  Filter taps are read through L1 (wave-coherent).
#include "ledn_rt.h"

namespace ledn {

template <typename TX, typename TY, int V>
__global__ void __launch_bounds__(256) dwconv_kernel(ledn_dw_desc d) {
    const int cv = d.C / V;
    const long total = (long)d.N * d.Ho * d.Wo * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = idx < total;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    int c = 0;
    long pix = 0;
    if (active) {
        c = (int)(idx % cv) * V;
        pix = idx / cv;
        const int wo = (int)(pix % d.Wo);
        const int ho = (int)((pix / d.Wo) % d.Ho);
        const int n = (int)(pix / ((long)d.Wo * d.Ho));
        const int dl = d.dil[c / d.group_size];
        const int padh = d.pad >= 0 ? d.pad : dl * (d.KH - 1) / 2;
        const int padw = d.pad >= 0 ? d.pad : dl * (d.KW - 1) / 2;
        const int Hx = d.H + (d.ext1 ? 1 : 0), Wx = d.W + (d.ext1 ? 1 : 0);
        const TX* x = reinterpret_cast<const TX*>(d.x);
        for (int kh = 0; kh < d.KH; ++kh) {
            int hi = ho * d.stride - padh + kh * dl;
            if (hi < 0 || hi >= Hx) continue;
            if (hi == d.H) hi = d.H - 2;  // ext1 reflect row
            for (int kw = 0; kw < d.KW; ++kw) {
                int wi = wo * d.stride - padw + kw * dl;
                if (wi < 0 || wi >= Wx) continue;
                if (wi == d.W) wi = d.W - 2;
                float xv[V], wv[V];
                ldv<V>(x + (((long)n * d.H + hi) * d.W + wi) * d.C + c, xv);
                ldv<V>(d.w + (long)(kh * d.KW + kw) * d.C + c, wv);
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = fmaf(xv[v], wv[v], acc[v]);
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float s = d.out_scale ? d.out_scale[c + v] : 1.f;
            const float b = d.out_shift ? d.out_shift[c + v] : 0.f;
            acc[v] = acc[v] * s + b;
        }
    }
    if (d.stat_sum) {
        // lanes of a wave hold different channels: reduce through LDS per block
        __shared__ float s_sum[512 * 2];
        for (int i = threadIdx.x; i < d.C * 2; i += blockDim.x) s_sum[i] = 0.f;
        __syncthreads();
        if (active) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                atomicAdd(&s_sum[c + v], acc[v]);
                atomicAdd(&s_sum[d.C + c + v], acc[v] * acc[v]);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < d.C; i += blockDim.x) {
            atomicAdd(d.stat_sum + i, s_sum[i]);
            atomicAdd(d.stat_sqsum + i, s_sum[d.C + i]);
        }
    }
    if (!active) return;
    if (d.act_out != LEDN_ACT_NONE) {
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = act_apply(d.act_out, acc[v], d.slope ? d.slope[c + v] : 0.f);
    }
    stv<V>(reinterpret_cast<TY*>(d.y) + pix * d.C + c, acc);
}

int dwconv_impl(const ledn_dw_desc& d, hipStream_t s) {
    LEDN_REQUIRE(d.x && d.w && d.y);
    LEDN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.Ho > 0 && d.Wo > 0);
    LEDN_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.group_size > 0);
    LEDN_REQUIRE(d.C <= 512 && d.C <= 4 * d.group_size);
    LEDN_REQUIRE((d.stat_sum == nullptr) == (d.stat_sqsum == nullptr));
    LEDN_REQUIRE(d.act_out != LEDN_ACT_PRELU || d.slope != nullptr);
    LEDN_REQUIRE(!d.ext1 || (d.H >= 2 && d.W >= 2));
    for (int g = 0; g * d.group_size < d.C; ++g) {
        LEDN_REQUIRE(d.dil[g] > 0);
        const int Hx = d.H + (d.ext1 ? 1 : 0), Wx = d.W + (d.ext1 ? 1 : 0);
        const int ph = d.pad >= 0 ? d.pad : d.dil[g] * (d.KH - 1) / 2;
        const int pw = d.pad >= 0 ? d.pad : d.dil[g] * (d.KW - 1) / 2;
        LEDN_REQUIRE(d.Ho == (Hx + 2 * ph - ((d.KH - 1) * d.dil[g] + 1)) / d.stride + 1);
        LEDN_REQUIRE(d.Wo == (Wx + 2 * pw - ((d.KW - 1) * d.dil[g] + 1)) / d.stride + 1);
    }
    const bool v4 = d.C % 4 == 0 && d.group_size % 4 == 0;
    const long total = (long)d.N * d.Ho * d.Wo * (v4 ? d.C / 4 : d.C);
    const dim3 grid((unsigned)cdiv(total, 256));
#define LEDN_DW(TX, TY)                                                              \
    do {                                                                             \
        if (v4) LEDN_LAUNCH((dwconv_kernel<TX, TY, 4>), grid, dim3(256), 0, s, d);   \
        else LEDN_LAUNCH((dwconv_kernel<TX, TY, 1>), grid, dim3(256), 0, s, d);      \
    } while (0)
    if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_F32) LEDN_DW(float, float);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_BF16) LEDN_DW(bf16_t, bf16_t);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_F32) LEDN_DW(bf16_t, float);
    else if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_BF16) LEDN_DW(float, bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_DW
    return check_launch();
}

// ---- SESP pyramid: y[..., b*n + c] = sum_{b'<=b} dw3x3(dil[b'], stride)(x)[..., c]
template <typename TX, typename TY, int V>
__global__ void __launch_bounds__(256) sesp_pyramid_kernel(ledn_pyr_desc d) {
    const int cv = d.n / V;
    const long total = (long)d.N * d.Ho * d.Wo * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cv) * V;
    const long pix = idx / cv;
    const int wo = (int)(pix % d.Wo);
    const int ho = (int)((pix / d.Wo) % d.Ho);
    const int n = (int)(pix / ((long)d.Wo * d.Ho));
    const TX* x = reinterpret_cast<const TX*>(d.x);
    TY* y = reinterpret_cast<TY*>(d.y) + pix * (4L * d.n) + c;
    float run[V];
#pragma unroll
    for (int v = 0; v < V; ++v) run[v] = 0.f;
    for (int b = 0; b < 4; ++b) {
        const int dl = d.dil[b];
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * d.stride + (kh - 1) * dl;
            if (hi < 0 || hi >= d.H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * d.stride + (kw - 1) * dl;
                if (wi < 0 || wi >= d.W) continue;
                float xv[V], wv[V];
                ldv<V>(x + (((long)n * d.H + hi) * d.W + wi) * d.n + c, xv);
                ldv<V>(d.w + (long)((b * 3 + kh) * 3 + kw) * d.n + c, wv);
#pragma unroll
                for (int v = 0; v < V; ++v) run[v] = fmaf(xv[v], wv[v], run[v]);
            }
        }
        stv<V>(y + (long)b * d.n, run);
    }
}

int sesp_pyramid_impl(const ledn_pyr_desc& d, hipStream_t s) {
    LEDN_REQUIRE(d.x && d.w && d.y);
    LEDN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.n > 0 && (d.stride == 1 || d.stride == 2));
    LEDN_REQUIRE(d.Ho == (d.H - 1) / d.stride + 1 && d.Wo == (d.W - 1) / d.stride + 1);
    for (int b = 0; b < 4; ++b) LEDN_REQUIRE(d.dil[b] > 0);
    const bool v4 = d.n % 4 == 0;
    const long total = (long)d.N * d.Ho * d.Wo * (v4 ? d.n / 4 : d.n);
    const dim3 grid((unsigned)cdiv(total, 256));
#define LEDN_PY(TX, TY)                                                                   \
    do {                                                                                  \
        if (v4) LEDN_LAUNCH((sesp_pyramid_kernel<TX, TY, 4>), grid, dim3(256), 0, s, d);  \
        else LEDN_LAUNCH((sesp_pyramid_kernel<TX, TY, 1>), grid, dim3(256), 0, s, d);     \
    } while (0)
    if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_F32) LEDN_PY(float, float);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_BF16) LEDN_PY(bf16_t, bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_PY
    return check_launch();
}

}  // namespace ledn
