// dwconv.hip -- depthwise convolutions: generic KxK (SESP second stage, GETB 8x8)
// and the fused SESP pyramid (4 dilated branches + hierarchical adds).
//
// HBM-bound streaming stencils.  One thread = one output pixel x 4 channels
// (16 B f32 / 8 B bf16 per lane, lanes run along channels then pixels, so a
// wave reads whole NHWC rows).  Filter taps are read through L1 (wave-coherent).
#include "ledn_rt.h"

namespace ledn {

// Thread = (pixel row r, channel vector cv) walking pixels r, r+rows, ... (grid-stride):
// filter taps, epilogue parameters and the running statistics of its V channels stay in
// registers; statistics leave the workgroup once (LDS reduction over the rows, then a
// per-workgroup partial in the workspace or one atomic per channel).
template <typename TX, typename TY, int V, int KK>
__global__ void __launch_bounds__(256) dwconv_kernel(ledn_dw_desc d, float* part) {
    __shared__ float s_part[2][256 * 4];
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    float st1[V], st2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) st1[v] = st2[v] = 0.f;
    if (r < rows) {
        const int dl = d.dil[c / d.group_size];
        const int padh = d.pad >= 0 ? d.pad : dl * (d.KH - 1) / 2;
        const int padw = d.pad >= 0 ? d.pad : dl * (d.KW - 1) / 2;
        const int Hx = d.H + (d.ext1 ? 1 : 0), Wx = d.W + (d.ext1 ? 1 : 0);
        const TX* x = reinterpret_cast<const TX*>(d.x);
        float wreg[KK > 0 ? KK : 1][V];
        if (KK > 0) {
#pragma unroll
            for (int t = 0; t < KK; ++t) ldv<V>(d.w + (long)t * d.C + c, wreg[t]);
        }
        float sc[V], sh[V], sl[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            sc[v] = d.out_scale ? d.out_scale[c + v] : 1.f;
            sh[v] = d.out_shift ? d.out_shift[c + v] : 0.f;
            sl[v] = d.slope ? d.slope[c + v] : 0.f;
        }
        const long npix = (long)d.N * d.Ho * d.Wo;
        for (long pix = (long)blockIdx.x * rows + r; pix < npix; pix += (long)gridDim.x * rows) {
            const int wo = (int)(pix % d.Wo);
            const int ho = (int)((pix / d.Wo) % d.Ho);
            const int n = (int)(pix / ((long)d.Wo * d.Ho));
            float acc[V];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = 0.f;
#pragma unroll
            for (int kh = 0; kh < (KK > 0 ? 3 : 1); ++kh) {
                for (int kh2 = (KK > 0 ? kh : 0); kh2 < (KK > 0 ? kh + 1 : d.KH); ++kh2) {
                    int hi = ho * d.stride - padh + kh2 * dl;
                    if (hi < 0 || hi >= Hx) continue;
                    if (hi == d.H) hi = d.H - 2;  // ext1 reflect row
#pragma unroll
                    for (int kw = 0; kw < (KK > 0 ? 3 : 1); ++kw) {
                        for (int kw2 = (KK > 0 ? kw : 0); kw2 < (KK > 0 ? kw + 1 : d.KW); ++kw2) {
                            int wi = wo * d.stride - padw + kw2 * dl;
                            if (wi < 0 || wi >= Wx) continue;
                            if (wi == d.W) wi = d.W - 2;
                            float xv[V], wv[V];
                            ldv<V>(x + (((long)n * d.H + hi) * d.W + wi) * d.C + c, xv);
                            if (KK > 0) {
#pragma unroll
                                for (int v = 0; v < V; ++v) wv[v] = wreg[kh * 3 + kw][v];
                            } else {
                                ldv<V>(d.w + (long)(kh2 * d.KW + kw2) * d.C + c, wv);
                            }
#pragma unroll
                            for (int v = 0; v < V; ++v) acc[v] = fmaf(xv[v], wv[v], acc[v]);
                        }
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                acc[v] = acc[v] * sc[v] + sh[v];
                st1[v] += acc[v];
                st2[v] = fmaf(acc[v], acc[v], st2[v]);
            }
            if (d.act_out != LEDN_ACT_NONE) {
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = act_apply(d.act_out, acc[v], sl[v]);
            }
            stv<V>(reinterpret_cast<TY*>(d.y) + pix * d.C + c, acc);
        }
    }
    if (!d.stat_sum) return;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        s_part[0][threadIdx.x * V + v] = st1[v];
        s_part[1][threadIdx.x * V + v] = st2[v];
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
                part[(long)blockIdx.x * 2 * d.C + c + v] = sa;
                part[(long)blockIdx.x * 2 * d.C + d.C + c + v] = sb;
            } else {
                atomicAdd(d.stat_sum + c + v, sa);
                atomicAdd(d.stat_sqsum + c + v, sb);
            }
        }
    }
}

int finish_partials(const float* part, int nblk, int C, int nout, float* o0, float* o1, float* o2,
                    hipStream_t s);

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
    const int cvn = v4 ? d.C / 4 : d.C;
    LEDN_REQUIRE(cvn <= 256);
    const int rows = 256 / cvn;
    const long npix = (long)d.N * d.Ho * d.Wo;
    long nb = cdiv(npix, rows * 4);
    if (nb > 2048) nb = 2048;
    float* part = nullptr;
    if (d.stat_sum) {
        if (nb > 64) part = ws_take(nb * 2 * d.C);
        if (!part && nb > 256) nb = 256;
    }
    const dim3 grid((unsigned)nb);
    const bool k3 = d.KH == 3 && d.KW == 3;
#define LEDN_DW(TX, TY)                                                                          \
    do {                                                                                         \
        if (v4 && k3) LEDN_LAUNCH((dwconv_kernel<TX, TY, 4, 9>), grid, dim3(256), 0, s, d, part); \
        else if (v4) LEDN_LAUNCH((dwconv_kernel<TX, TY, 4, 0>), grid, dim3(256), 0, s, d, part);  \
        else LEDN_LAUNCH((dwconv_kernel<TX, TY, 1, 0>), grid, dim3(256), 0, s, d, part);          \
    } while (0)
    if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_F32) LEDN_DW(float, float);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_BF16) LEDN_DW(bf16_t, bf16_t);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_F32) LEDN_DW(bf16_t, float);
    else if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_BF16) LEDN_DW(float, bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_DW
    if (part) return finish_partials(part, (int)nb, d.C, 2, d.stat_sum, d.stat_sqsum, nullptr, s);
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
