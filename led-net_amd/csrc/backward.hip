// backward.hip -- adjoint kernels of the streaming (HBM-bound) forward ops:
// BatchNorm/activation, depthwise + SESP pyramid, bilinear, pools, GETB mixing,
// MFAF gate.  Gradients wrt activations are written in gather form (one store
// per element, deterministic); only per-channel / per-tap reductions use f32
// atomics (one per workgroup after an LDS reduction).
#include "ledn_rt.h"

namespace ledn {

// ===========================================================================
// BatchNorm (+ residual / gate) + activation backward
// Thread = (pixel row r, channel vector cv): the per-channel parameters are loaded ONCE
// into registers, then the thread walks pixels r, r+rows, ... (grid-stride), so the
// streaming loop issues only the 2-3 wide activation loads per element.
// ===========================================================================
template <int V>
struct BnParams {
    float sc[V], sh[V], sl[V], mean[V], invstd[V];
};

template <int V>
__device__ __forceinline__ BnParams<V> bn_params(const ledn_bnbwd_desc& d, int c) {
    BnParams<V> p;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        p.sc[i] = d.scale ? d.scale[c + i] : 1.f;
        p.sh[i] = d.shift ? d.shift[c + i] : 0.f;
        p.sl[i] = d.slope ? d.slope[c + i] : 0.f;
        p.mean[i] = d.bn_mode ? d.mean[c + i] : 0.f;
        p.invstd[i] = d.bn_mode ? d.invstd[c + i] : 0.f;
    }
    return p;
}

// raw operands of one (pixel, channel vector): loaded first for several pixels so that a thread has
// all its global loads in flight before the first one is consumed
template <int V>
struct BnRaw {
    float z[V], dy[V], r[V];
};
template <typename TZ, typename TY, int V>
__device__ __forceinline__ void bn_load(const ledn_bnbwd_desc& d, long off, BnRaw<V>& w) {
    ldv<V>(reinterpret_cast<const TZ*>(d.z) + off, w.z);
    ldv<V>(reinterpret_cast<const TY*>(d.dy) + off, w.dy);
    if (d.res_mode != LEDN_RES_NONE) ldv<V>(reinterpret_cast<const TY*>(d.res) + off, w.r);
}
template <int V>
__device__ __forceinline__ void bn_math(const ledn_bnbwd_desc& d, const BnParams<V>& p, const BnRaw<V>& w, float* gv,
                                        float* xh, float* gres, float* dsl) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const float v = w.z[i] * p.sc[i] + p.sh[i];
        float t = v;
        if (d.res_mode == LEDN_RES_ADD) t = v + w.r[i];
        else if (d.res_mode == LEDN_RES_GATE) t = v * w.r[i] + w.r[i];
        const float gt = w.dy[i] * act_grad(d.act, t, p.sl[i]);
        dsl[i] = (d.act == LEDN_ACT_PRELU && t <= 0.f) ? w.dy[i] * t : 0.f;
        if (d.res_mode == LEDN_RES_GATE) {
            gv[i] = gt * w.r[i];
            gres[i] = gt * (v + 1.f);
        } else {
            gv[i] = gt;
            gres[i] = gt;
        }
        xh[i] = (w.z[i] - p.mean[i]) * p.invstd[i];
    }
}
template <typename TZ, typename TY, int V>
__device__ __forceinline__ void bn_g(const ledn_bnbwd_desc& d, const BnParams<V>& p, long off, float* gv,
                                     float* xh, float* gres, float* dsl) {
    BnRaw<V> w;
    bn_load<TZ, TY, V>(d, off, w);
    bn_math<V>(d, p, w, gv, xh, gres, dsl);
}

constexpr int BN_U = 4;   // pixel rows per thread and loop trip (loads issued together)

template <typename TZ, typename TY, int V>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(ledn_bnbwd_desc d, float* part) {
    __shared__ float s_part[3][256 * (V > 4 ? V : 4)];
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    float a[V], b[V], e[V];
#pragma unroll
    for (int v = 0; v < V; ++v) a[v] = b[v] = e[v] = 0.f;
    if (r < rows) {
        const BnParams<V> prm = bn_params<V>(d, cv * V);
        const long stride = (long)gridDim.x * rows;
        for (long p0 = (long)blockIdx.x * rows + r; p0 < d.P; p0 += BN_U * stride) {
            BnRaw<V> w[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const long p = p0 + u * stride;
                bn_load<TZ, TY, V>(d, (p < d.P ? p : p0) * d.C + cv * V, w[u]);
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                if (p0 + u * stride >= d.P) break;
                float gv[V], xh[V], gres[V], dsl[V];
                bn_math<V>(d, prm, w[u], gv, xh, gres, dsl);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    a[v] += gv[v];
                    b[v] = fmaf(gv[v], xh[v], b[v]);
                    e[v] += dsl[v];
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        s_part[0][threadIdx.x * V + v] = a[v];
        s_part[1][threadIdx.x * V + v] = b[v];
        s_part[2][threadIdx.x * V + v] = e[v];
    }
    __syncthreads();
    if (threadIdx.x < cvn) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float sa = 0.f, sb = 0.f, se = 0.f;
            for (int rr = 0; rr < rows; ++rr) {
                sa += s_part[0][(rr * cvn + cv) * V + v];
                sb += s_part[1][(rr * cvn + cv) * V + v];
                se += s_part[2][(rr * cvn + cv) * V + v];
            }
            if (part) {
                float* pp = part + (long)blockIdx.x * 3 * d.C + cv * V + v;
                pp[0] = sa; pp[d.C] = sb; pp[2 * d.C] = se;
            } else {
                atomicAdd(d.sum_g + cv * V + v, sa);
                if (d.sum_gx) atomicAdd(d.sum_gx + cv * V + v, sb);
                if (d.dslope) atomicAdd(d.dslope + cv * V + v, se);
            }
        }
    }
}

template <typename TZ, typename TY, int V>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(ledn_bnbwd_desc d) {
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    if (r >= rows) return;
    const int c = cv * V;
    const BnParams<V> prm = bn_params<V>(d, c);
    const float invn = (float)(1.0 / d.count);
    float mg[V], mgx[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        mg[i] = d.bn_mode ? d.sum_g[c + i] * invn : 0.f;
        mgx[i] = d.bn_mode ? d.sum_gx[c + i] * invn : 0.f;
    }
    // (one row per trip: issuing several rows of loads up front cost occupancy and ran 10 % slower)
    for (long p = (long)blockIdx.x * rows + r; p < d.P; p += (long)gridDim.x * rows) {
        const long off = p * d.C + c;
        float gv[V], xh[V], gres[V], dsl[V], dz[V];
        bn_g<TZ, TY, V>(d, prm, off, gv, xh, gres, dsl);
#pragma unroll
        for (int i = 0; i < V; ++i)
            dz[i] = d.bn_mode ? prm.sc[i] * (gv[i] - mg[i] - xh[i] * mgx[i]) : gv[i] * prm.sc[i];
        if (d.dz_add) {
            float a[V];
            ldv<V>(reinterpret_cast<const TZ*>(d.dz_add) + off, a);
#pragma unroll
            for (int i = 0; i < V; ++i) dz[i] += a[i];
        }
        stv<V>(reinterpret_cast<TZ*>(d.dz) + off, dz);
        if (d.dres) {
            if (d.dres_add) {
                float a[V];
                ldv<V>(reinterpret_cast<const TY*>(d.dres_add) + off, a);
#pragma unroll
                for (int i = 0; i < V; ++i) gres[i] += a[i];
            }
            stv<V>(reinterpret_cast<TY*>(d.dres) + off, gres);
        }
    }
}

static int bnbwd_validate(const ledn_bnbwd_desc& d, bool apply) {
    LEDN_REQUIRE(d.z && d.dy && d.P > 0 && d.C > 0);
    LEDN_REQUIRE((d.scale == nullptr) == (d.shift == nullptr));
    LEDN_REQUIRE(d.res_mode == LEDN_RES_NONE || d.res != nullptr);
    LEDN_REQUIRE(d.act != LEDN_ACT_PRELU || d.slope != nullptr);
    LEDN_REQUIRE(!d.bn_mode || (d.mean && d.invstd && d.scale && d.sum_g && d.sum_gx && d.count > 0));
    if (apply) LEDN_REQUIRE(d.dz != nullptr && (d.dres_add == nullptr || d.dres != nullptr));
    else LEDN_REQUIRE(d.sum_g != nullptr);
    const int V = d.C % 4 == 0 ? 4 : 1;
    LEDN_REQUIRE(d.C / V <= 256);
    return LEDN_OK;
}

#define LEDN_BNB_DISPATCH(KERNEL, ...)                                                                   \
    do {                                                                                                 \
        const bool v4 = d.C % 4 == 0;                                                                    \
        if (false && d.dtype_z == LEDN_BF16 && d.dtype_y == LEDN_BF16 && d.C % 8 == 0) { /* 16 B/lane: slower */ \
            LEDN_LAUNCH((KERNEL<bf16_t, bf16_t, 8>), grid, dim3(256), 0, s, __VA_ARGS__);                \
        } else if (d.dtype_z == LEDN_F32 && d.dtype_y == LEDN_F32) {                                     \
            if (v4) LEDN_LAUNCH((KERNEL<float, float, 4>), grid, dim3(256), 0, s, __VA_ARGS__);          \
            else LEDN_LAUNCH((KERNEL<float, float, 1>), grid, dim3(256), 0, s, __VA_ARGS__);             \
        } else if (d.dtype_z == LEDN_BF16 && d.dtype_y == LEDN_BF16) {                                   \
            if (v4) LEDN_LAUNCH((KERNEL<bf16_t, bf16_t, 4>), grid, dim3(256), 0, s, __VA_ARGS__);        \
            else LEDN_LAUNCH((KERNEL<bf16_t, bf16_t, 1>), grid, dim3(256), 0, s, __VA_ARGS__);           \
        } else if (d.dtype_z == LEDN_BF16 && d.dtype_y == LEDN_F32) {                                    \
            if (v4) LEDN_LAUNCH((KERNEL<bf16_t, float, 4>), grid, dim3(256), 0, s, __VA_ARGS__);         \
            else LEDN_LAUNCH((KERNEL<bf16_t, float, 1>), grid, dim3(256), 0, s, __VA_ARGS__);            \
        } else if (d.dtype_z == LEDN_F32 && d.dtype_y == LEDN_BF16) {                                    \
            if (v4) LEDN_LAUNCH((KERNEL<float, bf16_t, 4>), grid, dim3(256), 0, s, __VA_ARGS__);         \
            else LEDN_LAUNCH((KERNEL<float, bf16_t, 1>), grid, dim3(256), 0, s, __VA_ARGS__);            \
        } else return LEDN_EINVAL;                                                                       \
    } while (0)

static long bn_rows(const ledn_bnbwd_desc& d) {
    const int V = d.C % 4 == 0 ? 4 : 1;
    return 256 / (d.C / V);
}

// d.rows (see ledn.h) is honoured only when BOTH passes run on the streaming kernels: the same pure function of the
// descriptor decides it in the reduce and in the apply call, so the two always agree
static bool bn_rows_mode(const ledn_bnbwd_desc& d) {
    if (!d.rows || !(options().stream_fast & 1) || d.act == LEDN_ACT_SIGMOID) return false;
    if (d.dtype_z != LEDN_BF16 || d.dtype_y != LEDN_BF16) return false;
    if (d.C < 8 || d.C > 512 || (d.C & (d.C - 1))) return false;
    if (d.P * d.C / 8 < 4096) return false;
    if (d.dres != nullptr && d.res_mode == LEDN_RES_NONE) return false;
    return true;
}

int bn_act_bwd_reduce_impl(const ledn_bnbwd_desc& d0, hipStream_t s) {
    ledn_bnbwd_desc d = d0;
    if (!bn_rows_mode(d)) d.rows = nullptr;
    const int rc = bnbwd_validate(d, false);
    if (rc != LEDN_OK) return rc;
    if ((options().stream_fast & 1) && d.act != LEDN_ACT_SIGMOID) {
        const int rf = bn_act_bwd_reduce_fast(d, s);
        if (rf >= 0) return rf;
    }
    long nb = cdiv(d.P, bn_rows(d) * 8);
    float* part = nullptr;
    if (nb > 2048) nb = 2048;
    if (nb > 64 || det()) part = ws_take(nb * 3 * d.C);
    if (!part && nb > 256) nb = 256;     // atomics fallback: keep the grid bounded
    const dim3 grid((unsigned)nb);
    LEDN_BNB_DISPATCH(bn_bwd_reduce_kernel, d, part);
    if (part) return finish_partials(part, (int)nb, d.C, 3, d.sum_g, d.sum_gx, d.dslope, s);
    return check_launch();
}

int bn_act_bwd_apply_impl(const ledn_bnbwd_desc& d0, hipStream_t s) {
    ledn_bnbwd_desc d = d0;
    if (!bn_rows_mode(d)) d.rows = nullptr;
    const int rc = bnbwd_validate(d, true);
    if (rc != LEDN_OK) return rc;
    if ((options().stream_fast & 1) && d.act != LEDN_ACT_SIGMOID) {
        const int rf = bn_act_bwd_apply_fast(d, s);
        if (rf >= 0) return rf;
    }
    long nb = cdiv(d.P, bn_rows(d) * 4);
    if (nb > 4096) nb = 4096;
    const dim3 grid((unsigned)nb);
    LEDN_BNB_DISPATCH(bn_bwd_apply_kernel, d);
    return check_launch();
}

// ===========================================================================
// depthwise convolution backward
// ===========================================================================
template <typename T, int V>
__global__ void __launch_bounds__(256) dw_bwd_data_kernel(ledn_dwbwd_desc d) {
    const int cv = d.C / V;
    const long total = (long)d.N * d.H * d.W * cv;
    const long idx = (long)xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, d.W, d.H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    const int dl = d.dil[c / d.group_size];
    const int padh = d.pad >= 0 ? d.pad : dl * (d.KH - 1) / 2;
    const int padw = d.pad >= 0 ? d.pad : dl * (d.KW - 1) / 2;
    const T* dz = reinterpret_cast<const T*>(d.dz);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    if (d.KH == 3 && d.KW == 3 && !d.ext1) {
        // nine unconditional tap loads (+ weights) in flight, then the FMAs
        float g[9][V], wv[9][V];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int th = y + padh - (t / 3) * dl, tw = x + padw - (t % 3) * dl;
            const int ho = th / d.stride, wo = tw / d.stride;
            const bool valid = th >= 0 && tw >= 0 && th % d.stride == 0 && tw % d.stride == 0 && ho < d.Ho && wo < d.Wo;
            ldv_if<V>(dz, (((long)n * d.Ho + ho) * d.Wo + wo) * d.C + c, valid, g[t]);
            ldv<V>(d.w + (long)t * d.C + c, wv[t]);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fmaf(g[t][v], wv[t][v], acc[v]);
        if (d.add) {
            float a[V];
            ldv<V>(reinterpret_cast<const T*>(d.add) + pix * d.C + c, a);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += a[v];
        }
        stv<V>(reinterpret_cast<T*>(d.dx) + pix * d.C + c, acc);
        return;
    }
    // virtual source positions: (y,x) itself plus the reflected row H / column W it feeds
    const int ny = (d.ext1 && y == d.H - 2) ? 2 : 1, nx = (d.ext1 && x == d.W - 2) ? 2 : 1;
    for (int iy = 0; iy < ny; ++iy) {
        const int yy = iy ? d.H : y;
        for (int ix = 0; ix < nx; ++ix) {
            const int xx = ix ? d.W : x;
            for (int kh = 0; kh < d.KH; ++kh) {
                const int th = yy + padh - kh * dl;
                if (th < 0 || th % d.stride) continue;
                const int ho = th / d.stride;
                if (ho >= d.Ho) continue;
                for (int kw0 = 0; kw0 < d.KW; kw0 += 8) {   // eight taps of the row, loads in flight together
                    float g[8][V], wv[8][V];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kw = kw0 + j;
                        const int tw = xx + padw - kw * dl;
                        const int wo = tw / d.stride;
                        const bool ok = kw < d.KW && tw >= 0 && (tw % d.stride) == 0 && wo < d.Wo;
                        ldv_if<V>(dz, (((long)n * d.Ho + ho) * d.Wo + wo) * d.C + c, ok, g[j]);
                        ldv<V>(d.w + (long)(kh * d.KW + (kw < d.KW ? kw : 0)) * d.C + c, wv[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int v = 0; v < V; ++v) acc[v] = fmaf(g[j][v], wv[j][v], acc[v]);
                }
            }
        }
    }
    if (d.add) {
        float a[V];
        ldv<V>(reinterpret_cast<const T*>(d.add) + pix * d.C + c, a);
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] += a[v];
    }
    stv<V>(reinterpret_cast<T*>(d.dx) + pix * d.C + c, acc);
}

// one workgroup = one tap x a chunk of output pixels; thread (row r, channel vector cv)
template <typename T, int V>
__global__ void __launch_bounds__(256) dw_bwd_weight_kernel(ledn_dwbwd_desc d, int pix_per_block, float* part) {
    __shared__ float s_part[256 * 4];
    const int tap = blockIdx.y;
    const int kh = tap / d.KW, kw = tap % d.KW;
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    if (r < rows) {
        const int dl = d.dil[c / d.group_size];
        const int padh = d.pad >= 0 ? d.pad : dl * (d.KH - 1) / 2;
        const int padw = d.pad >= 0 ? d.pad : dl * (d.KW - 1) / 2;
        const int Hx = d.H + (d.ext1 ? 1 : 0), Wx = d.W + (d.ext1 ? 1 : 0);
        const T* x = reinterpret_cast<const T*>(d.x);
        const T* dz = reinterpret_cast<const T*>(d.dz);
        const long npix = (long)d.N * d.Ho * d.Wo;
        const long p0 = (long)blockIdx.x * pix_per_block;
        const long p1 = min(npix, p0 + (long)pix_per_block);
        // four pixels per trip, their eight loads unconditional and in flight together (the tap-major
        // grid makes every thread a chain of dependent trips: 8x8 GETB filters took 215 us one at a time)
        constexpr int U = 4;
        for (long pb = p0 + r; pb < p1; pb += (long)U * rows) {
            float xv[U][V], g[U][V];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long p = pb + (long)u * rows;
                const long pc = p < p1 ? p : pb;
                const int wo = (int)(pc % d.Wo);
                const int ho = (int)((pc / d.Wo) % d.Ho);
                const int n = (int)(pc / ((long)d.Wo * d.Ho));
                int hi = ho * d.stride - padh + kh * dl, wi = wo * d.stride - padw + kw * dl;
                ok[u] = p < p1 && hi >= 0 && hi < Hx && wi >= 0 && wi < Wx;
                if (hi == d.H) hi = d.H - 2;
                if (wi == d.W) wi = d.W - 2;
                const long xoff = ok[u] ? (((long)n * d.H + hi) * d.W + wi) * d.C + c : (long)c;
                ldv<V>(x + xoff, xv[u]);
                ldv<V>(dz + pc * d.C + c, g[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = fmaf(ok[u] ? xv[u][v] : 0.f, g[u][v], acc[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) s_part[threadIdx.x * V + v] = acc[v];
    __syncthreads();
    if (threadIdx.x < cvn) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float t = 0.f;
            for (int rr = 0; rr < rows; ++rr) t += s_part[(rr * cvn + cv) * V + v];
            if (part) part[((long)blockIdx.x * gridDim.y + tap) * d.C + c + v] = t;
            else atomicAdd(d.dw + (long)tap * d.C + c + v, t);
        }
    }
}

// Wide filters (GETB 8x8, stride 1): one workgroup = one filter ROW (kh) x a chunk of output pixels;
// thread (row r, channel vector cv) keeps the KW taps of that row in registers, reads dz once per
// pixel and the KW consecutive input pixels of the row (the tap-major kernel above re-read dz and x
// once per tap: 64 passes over both tensors for 8x8, L2-bound at 0.39 ms on the backward critical path).
template <typename T, int V, int KW>
__global__ void __launch_bounds__(256) dw_bwd_weight_row_kernel(ledn_dwbwd_desc d, int pix_per_block, float* part) {
    __shared__ float s_part[256 * V];
    const int kh = blockIdx.y;
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    float acc[KW][V];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[k][v] = 0.f;
    if (r < rows) {
        const int dl = d.dil[c / d.group_size];
        const int padh = d.pad >= 0 ? d.pad : dl * (d.KH - 1) / 2;
        const int padw = d.pad >= 0 ? d.pad : dl * (KW - 1) / 2;
        const int Hx = d.H + (d.ext1 ? 1 : 0), Wx = d.W + (d.ext1 ? 1 : 0);
        const T* x = reinterpret_cast<const T*>(d.x);
        const T* dz = reinterpret_cast<const T*>(d.dz);
        const long npix = (long)d.N * d.Ho * d.Wo;
        const long p0 = (long)blockIdx.x * pix_per_block;
        const long p1 = min(npix, p0 + (long)pix_per_block);
        for (long p = p0 + r; p < p1; p += rows) {
            const int wo = (int)(p % d.Wo);
            const int ho = (int)((p / d.Wo) % d.Ho);
            const int n = (int)(p / ((long)d.Wo * d.Ho));
            int hi = ho * d.stride - padh + kh * dl;
            const bool hok = hi >= 0 && hi < Hx;
            if (hi == d.H) hi = d.H - 2;
            float g[V], xv[KW][V];
            ldv<V>(dz + p * d.C + c, g);
#pragma unroll
            for (int k = 0; k < KW; ++k) {             // KW unconditional loads in flight
                int wi = wo * d.stride - padw + k * dl;
                const bool ok = hok && wi >= 0 && wi < Wx;
                if (wi == d.W) wi = d.W - 2;
                ldv_if<V>(x, (((long)n * d.H + hi) * d.W + wi) * d.C + c, ok, xv[k]);
            }
#pragma unroll
            for (int k = 0; k < KW; ++k)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[k][v] = fmaf(xv[k][v], g[v], acc[k][v]);
        }
    }
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < V; ++v) s_part[threadIdx.x * V + v] = acc[k][v];
        __syncthreads();
        if (threadIdx.x < cvn) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float t = 0.f;
                for (int rr = 0; rr < rows; ++rr) t += s_part[(rr * cvn + cv) * V + v];
                const int tap = kh * KW + k;
                if (part) part[((long)blockIdx.x * (gridDim.y * KW) + tap) * d.C + c + v] = t;
                else atomicAdd(d.dw + (long)tap * d.C + c + v, t);
            }
        }
    }
}

// 3x3 single pass: thread (row r, channel vector cv) keeps all 9 taps x V sums in registers
// and reads dz once per pixel; LDS reduction over the rows; per-workgroup partial (workspace)
// or one atomic per element.
template <typename T, int V>
__global__ void __launch_bounds__(256) dw_bwd_weight3x3_kernel(ledn_dwbwd_desc d, float* part) {
    __shared__ float s_part[9 * 256 * 4];
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    float acc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[t][v] = 0.f;
    if (r < rows) {
        const int dl = d.dil[c / d.group_size];
        const int pad = d.pad >= 0 ? d.pad : dl;
        const T* x = reinterpret_cast<const T*>(d.x);
        const T* dz = reinterpret_cast<const T*>(d.dz);
        const long npix = (long)d.N * d.Ho * d.Wo;
        const long ppb = cdiv(cdiv(npix, (long)gridDim.x), (long)rows) * rows;   // contiguous, XCD-aware range
        const long p0 = (long)xcd_block(blockIdx.x, gridDim.x) * ppb, p1 = min(npix, p0 + ppb);
        for (long p = p0 + r; p < p1; p += rows) {
            const int wo = (int)(p % d.Wo);
            const int ho = (int)((p / d.Wo) % d.Ho);
            const int n = (int)(p / ((long)d.Wo * d.Ho));
            float g[V], xv[9][V];
            ldv<V>(dz + p * d.C + c, g);
#pragma unroll
            for (int t = 0; t < 9; ++t) {   // nine unconditional tap loads in flight
                const int hi = ho * d.stride - pad + (t / 3) * dl, wi = wo * d.stride - pad + (t % 3) * dl;
                ldv_if<V>(x, (((long)n * d.H + hi) * d.W + wi) * d.C + c, hi >= 0 && hi < d.H && wi >= 0 && wi < d.W,
                          xv[t]);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[t][v] = fmaf(xv[t][v], g[v], acc[t][v]);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < V; ++v) s_part[(t * 256 + threadIdx.x) * V + v] = acc[t][v];
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * d.C; e += 256) {
        const int t = e / d.C, ch = e % d.C;
        float sum = 0.f;
        for (int rr = 0; rr < rows; ++rr) sum += s_part[(t * 256 + rr * cvn + ch / V) * V + ch % V];
        if (part) part[(long)blockIdx.x * 9 * d.C + e] = sum;
        else atomicAdd(d.dw + e, sum);
    }
}

static int dwbwd_validate(const ledn_dwbwd_desc& d) {
    LEDN_REQUIRE(d.dz && d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.Ho > 0 && d.Wo > 0);
    LEDN_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.group_size > 0 && d.C <= 4 * d.group_size);
    LEDN_REQUIRE(!d.ext1 || (d.H >= 2 && d.W >= 2));
    for (int g = 0; g * d.group_size < d.C; ++g) {
        LEDN_REQUIRE(d.dil[g] > 0);
        const int Hx = d.H + (d.ext1 ? 1 : 0), Wx = d.W + (d.ext1 ? 1 : 0);
        const int ph = d.pad >= 0 ? d.pad : d.dil[g] * (d.KH - 1) / 2;
        const int pw = d.pad >= 0 ? d.pad : d.dil[g] * (d.KW - 1) / 2;
        LEDN_REQUIRE(d.Ho == (Hx + 2 * ph - ((d.KH - 1) * d.dil[g] + 1)) / d.stride + 1);
        LEDN_REQUIRE(d.Wo == (Wx + 2 * pw - ((d.KW - 1) * d.dil[g] + 1)) / d.stride + 1);
    }
    return LEDN_OK;
}

int dw3x3_bwd_data_bf16(const ledn_dwbwd_desc& b, hipStream_t s);   // dwconv.hip; -1 = shape not covered

int dw8x8_bwd_data_tile(const ledn_dwbwd_desc& b, hipStream_t s);   // dwconv.hip; -1 = shape not covered
int dw8x8_bwd_weight_tile(const ledn_dwbwd_desc& b, hipStream_t s);

int dw_bwd_data_impl(const ledn_dwbwd_desc& d, hipStream_t s) {
    int rc = dwbwd_validate(d);
    if (rc != LEDN_OK) return rc;
    LEDN_REQUIRE(d.w && d.dx);
    rc = dw8x8_bwd_data_tile(d, s);
    if (rc >= 0) return rc;
    if (d.Ho == d.H && d.Wo == d.W) {
        rc = dw3x3_bwd_data_bf16(d, s);
        if (rc >= 0) return rc;
    }
    const bool v4 = d.C % 4 == 0 && d.group_size % 4 == 0;
    const long total = (long)d.N * d.H * d.W * (v4 ? d.C / 4 : d.C);
    const dim3 grid((unsigned)cdiv(total, 256));
#define LEDN_K(T)                                                                     \
    do {                                                                              \
        if (v4) LEDN_LAUNCH((dw_bwd_data_kernel<T, 4>), grid, dim3(256), 0, s, d);    \
        else LEDN_LAUNCH((dw_bwd_data_kernel<T, 1>), grid, dim3(256), 0, s, d);       \
    } while (0)
    if (d.dtype == LEDN_F32) LEDN_K(float);
    else if (d.dtype == LEDN_BF16) LEDN_K(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_K
    return check_launch();
}

int dw3x3_bwd_weight_bf16(const ledn_dwbwd_desc& d, hipStream_t s);   // stencil_bf16.hip; -1 = shape not covered
int pyr_bwd_data_bf16(const ledn_pyrbwd_desc& d, hipStream_t s);
bool pyr_tile_applies(const ledn_pyrbwd_desc& d);
int pyr_bwd_data_tile(const ledn_pyrbwd_desc& d, hipStream_t s);
int pyr_bwd_weight_bf16(const ledn_pyrbwd_desc& d, hipStream_t s);

int dw_bwd_weight_impl(const ledn_dwbwd_desc& d, hipStream_t s) {
    int rc = dwbwd_validate(d);
    if (rc != LEDN_OK) return rc;
    LEDN_REQUIRE(d.x && d.dw);
    rc = dw3x3_bwd_weight_bf16(d, s);
    if (rc >= 0) return rc;
    rc = dw8x8_bwd_weight_tile(d, s);
    if (rc >= 0) return rc;
    const bool v4 = d.C % 4 == 0 && d.group_size % 4 == 0;
    LEDN_REQUIRE((v4 ? d.C / 4 : d.C) <= 256);
    const long npix = (long)d.N * d.Ho * d.Wo;
    if (d.KH == 3 && d.KW == 3 && v4 && !d.ext1 && 256 % (d.C / 4) == 0) {
        const int rows = 256 / (d.C / 4);
        long nb = cdiv(npix, rows * 8);
        if (nb > 1024) nb = 1024;
        float* part = (nb > 32 || det()) ? ws_take(nb * 9 * d.C) : nullptr;
        if (!part && nb > 128) nb = 128;
        const dim3 g3((unsigned)nb);
        if (d.dtype == LEDN_F32) LEDN_LAUNCH((dw_bwd_weight3x3_kernel<float, 4>), g3, dim3(256), 0, s, d, part);
        else if (d.dtype == LEDN_BF16) LEDN_LAUNCH((dw_bwd_weight3x3_kernel<bf16_t, 4>), g3, dim3(256), 0, s, d, part);
        else return LEDN_EINVAL;
        if (part) return finish_partials(part, (int)nb, 9 * d.C, 1, d.dw, nullptr, nullptr, s);
        return check_launch();
    }
    if (d.KW == 8 && v4 && 256 % (d.C / 4) == 0) {     // GETB 8x8: a filter row per workgroup
        const int rows = 256 / (d.C / 4);
        long ppb = cdiv(npix * d.KH, 4096);
        ppb = cdiv(ppb < 64 ? 64 : ppb, rows) * rows;
        long nbx = cdiv(npix, ppb);
        float* part = (nbx > 8 || det()) ? ws_take(nbx * d.KH * 8 * d.C) : nullptr;
        if (!part && nbx > 64) {
            ppb = cdiv(cdiv(npix, 64), rows) * rows;
            nbx = cdiv(npix, ppb);
        }
        const dim3 g8((unsigned)nbx, (unsigned)d.KH);
        if (d.dtype == LEDN_F32) LEDN_LAUNCH((dw_bwd_weight_row_kernel<float, 4, 8>), g8, dim3(256), 0, s, d, (int)ppb, part);
        else if (d.dtype == LEDN_BF16) LEDN_LAUNCH((dw_bwd_weight_row_kernel<bf16_t, 4, 8>), g8, dim3(256), 0, s, d, (int)ppb, part);
        else return LEDN_EINVAL;
        if (part) return finish_partials(part, (int)nbx, d.KH * 8 * d.C, 1, d.dw, nullptr, nullptr, s);
        return check_launch();
    }
    // many short workgroups whose per-tap partial sums go to the workspace (summed by
    // finish_partials), else a bounded grid ending in one atomic per (tap, channel) per workgroup
    const int taps = d.KH * d.KW;
    long ppb = cdiv(npix * taps, 8192);
    if (ppb < 128) ppb = 128;
    long nbx = cdiv(npix, ppb);
    float* part = (nbx > 8 || det()) ? ws_take(nbx * taps * d.C) : nullptr;
    if (!part) {
        ppb = cdiv(npix * taps, 2048);
        if (ppb < 128) ppb = 128;
        nbx = cdiv(npix, ppb);
    }
    const dim3 grid((unsigned)nbx, (unsigned)taps);
#define LEDN_K(T)                                                                                       \
    do {                                                                                                \
        if (v4) LEDN_LAUNCH((dw_bwd_weight_kernel<T, 4>), grid, dim3(256), 0, s, d, (int)ppb, part);    \
        else LEDN_LAUNCH((dw_bwd_weight_kernel<T, 1>), grid, dim3(256), 0, s, d, (int)ppb, part);       \
    } while (0)
    if (d.dtype == LEDN_F32) LEDN_K(float);
    else if (d.dtype == LEDN_BF16) LEDN_K(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_K
    if (part) return finish_partials(part, (int)nbx, taps * d.C, 1, d.dw, nullptr, nullptr, s);
    return check_launch();
}

// ===========================================================================
// SESP pyramid backward
// ===========================================================================
template <typename T, int V>
__global__ void __launch_bounds__(256) pyr_suffix_kernel(const T* dy, T* g, long npix, int n) {
    const int cv = n / V;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * cv) return;
    const int c = (int)(idx % cv) * V;
    const long pix = idx / cv;
    float run[V];
#pragma unroll
    for (int v = 0; v < V; ++v) run[v] = 0.f;
    for (int b = 3; b >= 0; --b) {
        float t[V];
        ldv<V>(dy + pix * 4L * n + (long)b * n + c, t);
#pragma unroll
        for (int v = 0; v < V; ++v) run[v] += t[v];
        stv<V>(g + pix * 4L * n + (long)b * n + c, run);
    }
}

template <typename T, int V>
__global__ void __launch_bounds__(256) pyr_bwd_data_kernel(ledn_pyrbwd_desc d) {
    const int cv = d.n / V;
    const long total = (long)d.N * d.H * d.W * cv;
    const long idx = (long)xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, d.W, d.H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    const T* g = reinterpret_cast<const T*>(d.gsum);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    for (int b = 0; b < 4; ++b) {
        const int dl = d.dil[b];
        float gv[9][V], wv[9][V];
#pragma unroll
        for (int t = 0; t < 9; ++t) {   // nine unconditional tap loads (+ weights) in flight
            const int th = y - (t / 3 - 1) * dl, tw = x - (t % 3 - 1) * dl;
            const int ho = th / d.stride, wo = tw / d.stride;
            const bool valid = th >= 0 && tw >= 0 && th % d.stride == 0 && tw % d.stride == 0 && ho < d.Ho && wo < d.Wo;
            ldv_if<V>(g, (((long)n * d.Ho + ho) * d.Wo + wo) * 4L * d.n + (long)b * d.n + c, valid, gv[t]);
            ldv<V>(d.w + (long)(b * 9 + t) * d.n + c, wv[t]);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fmaf(gv[t][v], wv[t][v], acc[v]);
    }
    stv<V>(reinterpret_cast<T*>(d.dx) + pix * d.n + c, acc);
}

// single pass: thread (row r, channel vector cv) keeps one branch's 9 taps x V in registers
// (blockIdx.y = branch), reads gsum once per pixel.
template <typename T, int V>
__global__ void __launch_bounds__(256) pyr_bwd_weight_kernel(ledn_pyrbwd_desc d, float* part) {
    __shared__ float s_part[9 * 256 * 4];
    const int b = blockIdx.y;
    const int dl = d.dil[b];
    const int cvn = d.n / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    float acc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[t][v] = 0.f;
    if (r < rows) {
        const T* x = reinterpret_cast<const T*>(d.x);
        const T* g = reinterpret_cast<const T*>(d.gsum);
        const long npix = (long)d.N * d.Ho * d.Wo;
        const long ppb = cdiv(cdiv(npix, (long)gridDim.x), (long)rows) * rows;   // contiguous, XCD-aware range
        const long p0 = (long)xcd_block(blockIdx.x, gridDim.x) * ppb, p1 = min(npix, p0 + ppb);
        for (long p = p0 + r; p < p1; p += rows) {
            const int wo = (int)(p % d.Wo);
            const int ho = (int)((p / d.Wo) % d.Ho);
            const int n = (int)(p / ((long)d.Wo * d.Ho));
            float gv[V], xv[9][V];
            ldv<V>(g + p * 4L * d.n + (long)b * d.n + c, gv);
#pragma unroll
            for (int t = 0; t < 9; ++t) {   // nine unconditional tap loads in flight
                const int hi = ho * d.stride + (t / 3 - 1) * dl, wi = wo * d.stride + (t % 3 - 1) * dl;
                ldv_if<V>(x, (((long)n * d.H + hi) * d.W + wi) * d.n + c, hi >= 0 && hi < d.H && wi >= 0 && wi < d.W,
                          xv[t]);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[t][v] = fmaf(xv[t][v], gv[v], acc[t][v]);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < V; ++v) s_part[(t * 256 + threadIdx.x) * V + v] = acc[t][v];
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * d.n; e += 256) {
        const int t = e / d.n, ch = e % d.n;
        float sum = 0.f;
        for (int rr = 0; rr < rows; ++rr) sum += s_part[(t * 256 + rr * cvn + ch / V) * V + ch % V];
        // dw layout [4][3][3][n]; partial layout [blk][4*9*n]
        if (part) part[(long)blockIdx.x * 36 * d.n + (long)(b * 9 + t) * d.n + ch] = sum;
        else atomicAdd(d.dw + (long)(b * 9 + t) * d.n + ch, sum);
    }
}

static int pyrbwd_validate(const ledn_pyrbwd_desc& d) {
    LEDN_REQUIRE(d.gsum && d.N > 0 && d.H > 0 && d.W > 0 && d.n > 0 && (d.stride == 1 || d.stride == 2));
    LEDN_REQUIRE(d.Ho == (d.H - 1) / d.stride + 1 && d.Wo == (d.W - 1) / d.stride + 1);
    for (int b = 0; b < 4; ++b) LEDN_REQUIRE(d.dil[b] > 0);
    LEDN_REQUIRE(d.dtype == LEDN_F32 || d.dtype == LEDN_BF16);
    return LEDN_OK;
}

int pyr_bwd_data_impl(const ledn_pyrbwd_desc& d, hipStream_t s) {
    int rc = pyrbwd_validate(d);
    if (rc != LEDN_OK) return rc;
    LEDN_REQUIRE(d.dy && d.w && d.dx);
    const bool v4 = d.n % 4 == 0;
    const int cvn = v4 ? d.n / 4 : d.n;
    const long npo = (long)d.N * d.Ho * d.Wo;
    if (pyr_tile_applies(d)) return pyr_bwd_data_tile(d, s);
    if (d.dtype == LEDN_BF16 && v4 && d.stride == 1) {   // suffix sums, then the vectorised gather
        LEDN_LAUNCH((pyr_suffix_kernel<bf16_t, 4>), dim3((unsigned)cdiv(npo * cvn, 256)), dim3(256), 0, s,
                    (const bf16_t*)d.dy, (bf16_t*)d.gsum, npo, d.n);
        rc = pyr_bwd_data_bf16(d, s);
        if (rc >= 0) return rc;
    }
    const dim3 g1((unsigned)cdiv(npo * cvn, 256)), g2((unsigned)cdiv((long)d.N * d.H * d.W * cvn, 256));
#define LEDN_K(T)                                                                                          \
    do {                                                                                                   \
        if (v4) {                                                                                          \
            LEDN_LAUNCH((pyr_suffix_kernel<T, 4>), g1, dim3(256), 0, s, (const T*)d.dy, (T*)d.gsum, npo, d.n); \
            LEDN_LAUNCH((pyr_bwd_data_kernel<T, 4>), g2, dim3(256), 0, s, d);                              \
        } else {                                                                                           \
            LEDN_LAUNCH((pyr_suffix_kernel<T, 1>), g1, dim3(256), 0, s, (const T*)d.dy, (T*)d.gsum, npo, d.n); \
            LEDN_LAUNCH((pyr_bwd_data_kernel<T, 1>), g2, dim3(256), 0, s, d);                              \
        }                                                                                                  \
    } while (0)
    if (d.dtype == LEDN_F32) LEDN_K(float);
    else LEDN_K(bf16_t);
#undef LEDN_K
    return check_launch();
}

int pyr_bwd_weight_impl(const ledn_pyrbwd_desc& d, hipStream_t s) {
    int rc = pyrbwd_validate(d);
    if (rc != LEDN_OK) return rc;
    LEDN_REQUIRE(d.x && d.dw);
    rc = pyr_bwd_weight_bf16(d, s);
    if (rc >= 0) return rc;
    const bool v4 = d.n % 4 == 0;
    LEDN_REQUIRE((v4 ? d.n / 4 : d.n) <= 256);
    const long npix = (long)d.N * d.Ho * d.Wo;
    const int cvn = v4 ? d.n / 4 : d.n;
    const int rows = 256 / cvn;
    long nb = cdiv(npix, rows * 8);
    if (nb > 512) nb = 512;
    float* part = (nb > 32 || det()) ? ws_take(nb * 36 * d.n) : nullptr;
    if (!part && nb > 128) nb = 128;
    const dim3 grid((unsigned)nb, 4u);
#define LEDN_K(T)                                                                              \
    do {                                                                                       \
        if (v4) LEDN_LAUNCH((pyr_bwd_weight_kernel<T, 4>), grid, dim3(256), 0, s, d, part);    \
        else LEDN_LAUNCH((pyr_bwd_weight_kernel<T, 1>), grid, dim3(256), 0, s, d, part);       \
    } while (0)
    if (d.dtype == LEDN_F32) LEDN_K(float);
    else LEDN_K(bf16_t);
#undef LEDN_K
    if (part) return finish_partials(part, (int)nb, 36 * d.n, 1, d.dw, nullptr, nullptr, s);
    return check_launch();
}

// ===========================================================================
// bilinear backward (gather): dx[src] = sum over destinations that read src
// ===========================================================================
struct LerpB {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ LerpB lerp_coord_b(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    LerpB l;
    l.i0 = (int)src;
    if (l.i0 > in - 1) l.i0 = in - 1;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}
__device__ __forceinline__ void dst_range(int s, int in, int out, int& lo, int& hi) {
    // destinations d whose source coordinate lies in (s-1, s+1): d in ((s-0.5)/scale-0.5, (s+1.5)/scale-0.5)
    const float inv = (float)out / (float)in;
    lo = (int)floorf(((float)s - 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)s + 1.5f) * inv - 0.5f) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
    if (s == 0) lo = 0;              // clamped sources (src < 0 -> 0)
}

template <typename TY, typename TX, int V>
__global__ void __launch_bounds__(256) bilinear_bwd_kernel(const TY* dy, TX* dx, int N, int H, int W, int C,
                                                           int Ho, int Wo) {
    const int cv = C / V;
    const long total = (long)N * H * W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, W, H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    int ylo, yhi, xlo, xhi;
    dst_range(y, H, Ho, ylo, yhi);
    dst_range(x, W, Wo, xlo, xhi);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    for (int dyy = ylo; dyy <= yhi; ++dyy) {
        const LerpB ly = lerp_coord_b(dyy, H, Ho);
        const float wy = (ly.i0 == y ? ly.w0 : 0.f) + (ly.i1 == y ? ly.w1 : 0.f);
        if (wy == 0.f) continue;
        for (int dxx = xlo; dxx <= xhi; ++dxx) {
            const LerpB lx = lerp_coord_b(dxx, W, Wo);
            const float wx = (lx.i0 == x ? lx.w0 : 0.f) + (lx.i1 == x ? lx.w1 : 0.f);
            if (wx == 0.f) continue;
            float g[V];
            ldv<V>(dy + (((long)n * Ho + dyy) * Wo + dxx) * C + c, g);
            const float wgt = wy * wx;
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fmaf(wgt, g[v], acc[v]);
        }
    }
    stv<V>(dx + pix * C + c, acc);
}

// exact 2x upsampling (every level of the logit pyramid and the 1/16 -> 1/8 link): a source pixel
// receives from the 4 x 4 destinations 2s-1 .. 2s+2 only -- sixteen unconditional loads with the
// same weights and summation order as the general search above (which walks ~36 candidates with a
// branch each: 125 us for the 134 MB full-resolution logit gradient)
template <typename TY, typename TX, int V>
__global__ void __launch_bounds__(256) bilinear_bwd2x_kernel(const TY* dy, TX* dx, int N, int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W;
    const int cv = C / V;
    const long total = (long)N * H * W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, W, H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    float wy[4], wx[4];
    int yy[4], xx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dv = 2 * y - 1 + j, dh = 2 * x - 1 + j;
        const bool vy = dv >= 0 && dv < Ho, vx = dh >= 0 && dh < Wo;
        yy[j] = vy ? dv : 0;
        xx[j] = vx ? dh : 0;
        const LerpB ly = lerp_coord_b(yy[j], H, Ho), lx = lerp_coord_b(xx[j], W, Wo);
        wy[j] = vy ? (ly.i0 == y ? ly.w0 : 0.f) + (ly.i1 == y ? ly.w1 : 0.f) : 0.f;
        wx[j] = vx ? (lx.i0 == x ? lx.w0 : 0.f) + (lx.i1 == x ? lx.w1 : 0.f) : 0.f;
    }
    float g[16][V];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) ldv<V>(dy + (((long)n * Ho + yy[j]) * Wo + xx[i]) * C + c, g[j * 4 + i]);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float wgt = wy[j] * wx[i];
            if (wgt == 0.f) continue;        // as the general kernel skips them (an inf/nan there must not leak)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fmaf(wgt, g[j * 4 + i][v], acc[v]);
        }
    stv<V>(dx + pix * C + c, acc);
}

// Large upsampling ratios (the 1/64 -> 1/8 and 1/32 -> 1/8 context links: ~18 x 18 destinations per source pixel):
// the thread-per-source-pixel search above is a serial chain of ~300 dependent candidate steps (119 us for
// 16 x 16 x 16 x 128 <- 128 x 128).  Here a workgroup owns ONE source pixel: lane = (4-channel group, one of
// 256 / (C/4) destination-row slots), the slots split the candidate rows, an LDS reduction adds them up.  Same
// weights, candidates and per-row summation order as the search kernel.
template <typename TY, typename TX>
__global__ void __launch_bounds__(256) bilinear_bwd_wide_kernel(const TY* dy, TX* dx, int N, int H, int W, int C,
                                                                int Ho, int Wo) {
    constexpr int V = 4;
    __shared__ float s_red[256 * V];
    const int cv = C / V, slots = 256 / cv;
    const int cg = threadIdx.x % cv, slot = threadIdx.x / cv, c = cg * V;
    const long pix = blockIdx.x;
    const NhwcIdx ix_ = pix_split(pix, W, H);
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    int ylo, yhi, xlo, xhi;
    dst_range(y, H, Ho, ylo, yhi);
    dst_range(x, W, Wo, xlo, xhi);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    if (slot < slots) {
        for (int dyy = ylo + slot; dyy <= yhi; dyy += slots) {
            const LerpB ly = lerp_coord_b(dyy, H, Ho);
            const float wy = (ly.i0 == y ? ly.w0 : 0.f) + (ly.i1 == y ? ly.w1 : 0.f);
            if (wy == 0.f) continue;
            for (int dxx = xlo; dxx <= xhi; ++dxx) {
                const LerpB lx = lerp_coord_b(dxx, W, Wo);
                const float wx = (lx.i0 == x ? lx.w0 : 0.f) + (lx.i1 == x ? lx.w1 : 0.f);
                if (wx == 0.f) continue;
                float g[V];
                ldv<V>(dy + (((long)n * Ho + dyy) * Wo + dxx) * C + c, g);
                const float wgt = wy * wx;
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = fmaf(wgt, g[v], acc[v]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) s_red[threadIdx.x * V + v] = acc[v];
    __syncthreads();
    if (threadIdx.x < cv) {
        float t[V];
#pragma unroll
        for (int v = 0; v < V; ++v) t[v] = 0.f;
        for (int sl = 0; sl < slots; ++sl)
#pragma unroll
            for (int v = 0; v < V; ++v) t[v] += s_red[(sl * cv + threadIdx.x) * V + v];
        stv<V>(dx + pix * C + threadIdx.x * V, t);
    }
}

int bilinear_bwd_impl(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int dtype_dy,
                      int dtype_dx, hipStream_t s) {
    LEDN_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0);
    const bool v4 = C % 4 == 0;
    const bool v2 = !v4 && C % 2 == 0;
    const long total = (long)N * H * W * (v4 ? C / 4 : (v2 ? C / 2 : C));
    const dim3 grid((unsigned)cdiv(total, 256));
    const bool x2 = Ho == 2 * H && Wo == 2 * W;
    // >= 4x and few source pixels: one workgroup per source pixel (see bilinear_bwd_wide_kernel)
    const bool wide = v4 && !x2 && Ho >= 4 * H && Wo >= 4 * W && C / 4 <= 256 && (long)N * H * W <= (1L << 20) &&
                      (options().stream_fast & 2);
#define LEDN_K(TY, TX)                                                                                          \
    do {                                                                                                        \
        if (wide) LEDN_LAUNCH((bilinear_bwd_wide_kernel<TY, TX>), dim3((unsigned)((long)N * H * W)), dim3(256), 0, s, (const TY*)dy, (TX*)dx, N, H, W, C, Ho, Wo); \
        else if (x2 && v4) LEDN_LAUNCH((bilinear_bwd2x_kernel<TY, TX, 4>), grid, dim3(256), 0, s, (const TY*)dy, (TX*)dx, N, H, W, C); \
        else if (x2 && v2) LEDN_LAUNCH((bilinear_bwd2x_kernel<TY, TX, 2>), grid, dim3(256), 0, s, (const TY*)dy, (TX*)dx, N, H, W, C); \
        else if (v4) LEDN_LAUNCH((bilinear_bwd_kernel<TY, TX, 4>), grid, dim3(256), 0, s, (const TY*)dy, (TX*)dx, N, H, W, C, Ho, Wo); \
        else if (v2) LEDN_LAUNCH((bilinear_bwd_kernel<TY, TX, 2>), grid, dim3(256), 0, s, (const TY*)dy, (TX*)dx, N, H, W, C, Ho, Wo); \
        else LEDN_LAUNCH((bilinear_bwd_kernel<TY, TX, 1>), grid, dim3(256), 0, s, (const TY*)dy, (TX*)dx, N, H, W, C, Ho, Wo); \
    } while (0)
    if (dtype_dy == LEDN_F32 && dtype_dx == LEDN_F32) LEDN_K(float, float);
    else if (dtype_dy == LEDN_BF16 && dtype_dx == LEDN_BF16) LEDN_K(bf16_t, bf16_t);
    else if (dtype_dy == LEDN_F32 && dtype_dx == LEDN_BF16) LEDN_K(float, bf16_t);
    else if (dtype_dy == LEDN_BF16 && dtype_dx == LEDN_F32) LEDN_K(bf16_t, float);
    else return LEDN_EINVAL;
#undef LEDN_K
    return check_launch();
}

// ===========================================================================
// 3x3/s2 average pool backward:  dx[y,x] = add + 1/9 * sum_{taps} dy[(y+1-kh)/2, (x+1-kw)/2]
// ===========================================================================
template <typename T, int V>
__global__ void __launch_bounds__(256) avgpool3x3s2_bwd_kernel(const T* dy, const T* add, T* dx, int N, int H,
                                                               int W, int C, int Ho, int Wo) {
    const int cv = C / V;
    const long total = (long)N * H * W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, W, H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
        const int th = y + 1 - kh;
        if (th < 0 || (th & 1)) continue;
        const int ho = th >> 1;
        if (ho >= Ho) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int tw = x + 1 - kw;
            if (tw < 0 || (tw & 1)) continue;
            const int wo = tw >> 1;
            if (wo >= Wo) continue;
            float g[V];
            ldv<V>(dy + (((long)n * Ho + ho) * Wo + wo) * C + c, g);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += g[v];
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] *= (1.f / 9.f);
    if (add) {
        float a[V];
        ldv<V>(add + pix * C + c, a);
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] += a[v];
    }
    stv<V>(dx + pix * C + c, acc);
}

int avgpool3x3s2_bwd_impl(const void* dy, const void* add, void* dx, int N, int H, int W, int C, int Ho,
                          int Wo, int dtype, hipStream_t s) {
    LEDN_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0);
    LEDN_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1);
    const bool v4 = C % 4 == 0;
    const long total = (long)N * H * W * (v4 ? C / 4 : C);
    const dim3 grid((unsigned)cdiv(total, 256));
#define LEDN_K(T)                                                                                            \
    do {                                                                                                     \
        if (v4) LEDN_LAUNCH((avgpool3x3s2_bwd_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)dy, (const T*)add, (T*)dx, N, H, W, C, Ho, Wo); \
        else LEDN_LAUNCH((avgpool3x3s2_bwd_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)dy, (const T*)add, (T*)dx, N, H, W, C, Ho, Wo);    \
    } while (0)
    if (dtype == LEDN_F32) LEDN_K(float);
    else if (dtype == LEDN_BF16) LEDN_K(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_K
    return check_launch();
}

// ===========================================================================
// GETB mixing backward:  da[y,x] = 1/ws * ( sum_{y' in [y-ws/2, y+ws/2-1]} dout[y',x]
//     + [y == H-2] sum_{y' in [H-ws/2, H-1]} dout[y',x] + same along x )
// (forward window of output y' covers rows y'-p .. y'-p+ws-1, p = ws/2-1; row H is the
//  reflection of row H-2.)
// ===========================================================================
template <typename T, int V>
__global__ void __launch_bounds__(256) getb_pool_bwd_kernel(const T* dout, T* da, int N, int H, int W, int C,
                                                            int ws) {
    const int cv = C / V;
    const long total = (long)N * H * W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, W, H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    const T* base = dout + (long)n * H * W * C + c;
    const int p = ws / 2 - 1;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    // rows: source row r feeds outputs y' with y'-p <= r <= y'-p+ws-1  <=>  r-ws+1+p <= y' <= r+p
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && y != H - 2) break;
        const int r = pass ? H : y;
        for (int yo = r - ws + 1 + p; yo <= r + p; ++yo) {
            if (yo < 0 || yo >= H) continue;
            float t[V];
            ldv<V>(base + ((long)yo * W + x) * C, t);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += t[v];
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && x != W - 2) break;
        const int q = pass ? W : x;
        for (int xo = q - ws + 1 + p; xo <= q + p; ++xo) {
            if (xo < 0 || xo >= W) continue;
            float t[V];
            ldv<V>(base + ((long)y * W + xo) * C, t);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] += t[v];
        }
    }
    const float inv = 1.f / (float)ws;
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] *= inv;
    stv<V>(da + pix * C + c, acc);
}

int getb_pool_bwd_impl(const void* dout, void* da, int N, int H, int W, int C, int ws, int dtype,
                       hipStream_t s) {
    LEDN_REQUIRE(dout && da && N > 0 && H >= 2 && W >= 2 && C > 0 && ws >= 2 && C % 4 == 0);
    const long total = (long)N * H * W * (C / 4);
    const dim3 grid((unsigned)cdiv(total, 256));
    if (dtype == LEDN_F32)
        LEDN_LAUNCH((getb_pool_bwd_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)dout, (float*)da, N,
                    H, W, C, ws);
    else if (dtype == LEDN_BF16)
        LEDN_LAUNCH((getb_pool_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (const bf16_t*)dout,
                    (bf16_t*)da, N, H, W, C, ws);
    else return LEDN_EINVAL;
    return check_launch();
}

// ===========================================================================
// MFAF gate backward + combine
// ===========================================================================
// One workgroup = a 64-pixel segment of one image row.  The context-map gradients
// (sum of ds over every pixel that reads a cell) are accumulated in LDS per workgroup and
// flushed with one global atomic per touched (cell, channel): the 1x1 "global" cell would
// otherwise receive one same-address atomic per pixel.
constexpr int MFAF_SEG = 128, MFAF_SLOTS = 18;
template <typename T, int V>
__global__ void __launch_bounds__(256) mfaf_gate_bwd_kernel(ledn_mfafbwd_desc d, int ctx_sums) {
    __shared__ float s_ctx[4 * MFAF_SLOTS * 128];
    const int cvn = d.C / V;
    const int slots = 256 / cvn;
    const int pslot = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    const int y = blockIdx.x % d.H, n = blockIdx.x / d.H;
    const int x0 = blockIdx.y * MFAF_SEG, x1 = min(d.W, x0 + MFAF_SEG);
    for (int i = threadIdx.x; i < 4 * MFAF_SLOTS * d.C; i += blockDim.x) s_ctx[i] = 0.f;
    __syncthreads();
    int sy[4], sx_first[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int S = d.ctx_size[k];
        int t = (int)((float)y * ((float)S / (float)d.H));
        sy[k] = t > S - 1 ? S - 1 : t;
        t = (int)((float)x0 * ((float)S / (float)d.W));
        sx_first[k] = t > S - 1 ? S - 1 : t;
    }
    if (pslot < slots) {
        // every thread owns a CONTIGUOUS run of pixels: the context cell of each level changes
        // rarely along the run, so its gate contribution is loaded once per cell and its ds sum is
        // kept in registers and flushed to LDS once per cell (the first version issued 16 LDS
        // atomics per pixel, most of them on the same address across the 16 pixel slots)
        const int run = (x1 - x0 + slots - 1) / slots;
        const int xa = x0 + pslot * run, xb = min(x1, xa + run);
        int cur[4] = {-1, -1, -1, -1};
        float cs[4][V], acc[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < V; ++v) cs[k][v] = acc[k][v] = 0.f;
        float sc0[V], sh0[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            sc0[v] = d.scale[0][c + v];
            sh0[v] = d.shift[0][c + v];
        }
        for (int x = xa; x < xb; ++x) {
            const long pix = ((long)n * d.H + y) * d.W + x;
            float s[V], t[V];
            ldv<V>(reinterpret_cast<const T*>(d.xl) + pix * d.C + c, t);
#pragma unroll
            for (int v = 0; v < V; ++v) s[v] = t[v] * sc0[v] + sh0[v];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int S = d.ctx_size[k];
                int sx = (int)((float)x * ((float)S / (float)d.W));
                if (sx > S - 1) sx = S - 1;
                const int slot = sx - sx_first[k];
                if (slot != cur[k]) {
                    if (cur[k] >= 0 && ctx_sums) {
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            atomicAdd(&s_ctx[(k * MFAF_SLOTS + cur[k]) * d.C + c + v], acc[k][v]);
                            acc[k][v] = 0.f;
                        }
                    }
                    cur[k] = slot;
                    ldv<V>(d.ctx[k] + (((long)n * S + sy[k]) * S + sx) * d.C + c, t);
#pragma unroll
                    for (int v = 0; v < V; ++v) cs[k][v] = t[v] * d.scale[k + 1][c + v] + d.shift[k + 1][c + v];
                }
#pragma unroll
                for (int v = 0; v < V; ++v) s[v] += cs[k][v];
            }
            float xv[V], rv[V], g[V], dxv[V], drv[V], dsv[V];
            ldv<V>(reinterpret_cast<const T*>(d.x) + pix * d.C + c, xv);
            ldv<V>(reinterpret_cast<const T*>(d.r) + pix * d.C + c, rv);
            ldv<V>(reinterpret_cast<const T*>(d.dout) + pix * d.C + c, g);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float w = 1.f / (1.f + __expf(-s[v]));
                float go = g[v];
                if (d.act == LEDN_ACT_RELU) {
                    const float o = 2.f * xv[v] * w + 2.f * rv[v] * (1.f - w);
                    if (o <= 0.f) go = 0.f;
                }
                dxv[v] = 2.f * w * go;
                drv[v] = 2.f * (1.f - w) * go;
                dsv[v] = 2.f * (xv[v] - rv[v]) * go * w * (1.f - w);
            }
            stv<V>(reinterpret_cast<T*>(d.dx) + pix * d.C + c, dxv);
            stv<V>(reinterpret_cast<T*>(d.dr) + pix * d.C + c, drv);
            stv<V>(reinterpret_cast<T*>(d.ds) + pix * d.C + c, dsv);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[k][v] += dsv[v];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cur[k] >= 0 && ctx_sums) {
#pragma unroll
                for (int v = 0; v < V; ++v) atomicAdd(&s_ctx[(k * MFAF_SLOTS + cur[k]) * d.C + c + v], acc[k][v]);
            }
    }
    if (!ctx_sums) return;      // deterministic mode: mfaf_dctx_det_kernel sums ds per context cell in a fixed order
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        const int S = d.ctx_size[k];
        int sx_last = (int)((float)(x1 - 1) * ((float)S / (float)d.W));
        if (sx_last > S - 1) sx_last = S - 1;
        const int nslot = sx_last - sx_first[k] + 1;
        for (int i = threadIdx.x; i < nslot * d.C; i += blockDim.x) {
            const int sl = i / d.C, ch = i % d.C;
            const float v = s_ctx[(k * MFAF_SLOTS + sl) * d.C + ch];
            if (v != 0.f)
                atomicAdd(d.dctx[k] + (((long)n * S + sy[k]) * S + sx_first[k] + sl) * d.C + ch, v);
        }
    }
}

// Deterministic form of the context-map gradients: dctx[k][n, cy, cx, :] += sum of ds over the pixels whose nearest-
// upsampling source is that cell (the same float index rule as the gate kernels).  Workgroup = (cell, row chunk): thread =
// (4 channels, pixel slot), slots walk the chunk's rows of the cell's rectangle in a fixed order, LDS sum in slot order;
// chunk sums go to part[chunk][cell][C] and finish_partials adds the chunks up in order (one chunk: straight into dctx).
// (The first form, one workgroup per cell, spent 370 us on the 1 x 1 level: 16 workgroups walking 16384 pixels each.)
template <typename T>
__global__ void __launch_bounds__(256) mfaf_dctx_det_kernel(ledn_mfafbwd_desc d, int level, float* part) {
    __shared__ float s_acc[256 * 4];
    __shared__ int s_box[4];
    const int S = d.ctx_size[level];
    const int cell = blockIdx.x % (S * S), n = blockIdx.x / (S * S);
    const int cy = cell / S, cx = cell % S;
    const int cvn = d.C / 4, slots = 256 / cvn;
    const int cv = threadIdx.x % cvn, slot = threadIdx.x / cvn;
    // rows / columns of the cell: y with min(int(y * S / H), S - 1) == cy  (monotone: a contiguous range)
    if (threadIdx.x < 2) {
        const int L = threadIdx.x == 0 ? d.H : d.W, want = threadIdx.x == 0 ? cy : cx;
        int lo = L, hi = -1;
        for (int i = 0; i < L; ++i) {
            int t = (int)((float)i * ((float)S / (float)L));
            if (t > S - 1) t = S - 1;
            if (t == want) {
                if (i < lo) lo = i;
                hi = i;
            }
        }
        s_box[threadIdx.x * 2] = lo;
        s_box[threadIdx.x * 2 + 1] = hi;
    }
    __syncthreads();
    const int x0 = s_box[2], x1 = s_box[3];
    const int rows = s_box[1] - s_box[0] + 1;
    const int rpc = (rows + (int)gridDim.y - 1) / (int)gridDim.y;        // rows per chunk
    const int y0 = s_box[0] + (int)blockIdx.y * rpc, y1 = min(s_box[1], y0 + rpc - 1);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (slot < slots && y1 >= y0 && x1 >= x0) {
        const int bw = x1 - x0 + 1, np = (y1 - y0 + 1) * bw;
        for (int p = slot; p < np; p += slots) {
            const int y = y0 + p / bw, x = x0 + p % bw;
            float t[4];
            ldv<4>(reinterpret_cast<const T*>(d.ds) + (((long)n * d.H + y) * d.W + x) * d.C + cv * 4, t);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] += t[v];
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) s_acc[threadIdx.x * 4 + v] = acc[v];
    __syncthreads();
    if (threadIdx.x < cvn) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float t = 0.f;
            for (int sl = 0; sl < slots; ++sl) t += s_acc[(sl * cvn + threadIdx.x) * 4 + v];
            const long o = ((long)n * S * S + cell) * d.C + threadIdx.x * 4 + v;
            if (part) part[(long)blockIdx.y * ((long)d.N * S * S * d.C) + o] = t;
            else d.dctx[level][o] += t;
        }
    }
}

int mfaf_gate_bwd_impl(const ledn_mfafbwd_desc& d, hipStream_t s) {
    LEDN_REQUIRE(d.x && d.r && d.xl && d.dout && d.dx && d.dr && d.ds);
    LEDN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.C % 4 == 0 && d.C <= 128);
    for (int k = 0; k < 4; ++k) {
        LEDN_REQUIRE(d.ctx[k] && d.dctx[k] && d.ctx_size[k] > 0);
        // cells touched by one MFAF_SEG-pixel segment must fit the LDS slots
        const int seg = d.W < MFAF_SEG ? d.W : MFAF_SEG;
        LEDN_REQUIRE((long)seg * d.ctx_size[k] / d.W + 2 <= MFAF_SLOTS);
    }
    for (int k = 0; k < 5; ++k) LEDN_REQUIRE(d.scale[k] && d.shift[k]);
    const dim3 grid((unsigned)(d.N * d.H), (unsigned)cdiv(d.W, MFAF_SEG));
    const int ctx_sums = det() ? 0 : 1;
    if (d.dtype != LEDN_F32 && d.dtype != LEDN_BF16) return LEDN_EINVAL;
    // (a 16-byte-lane form with each lane's <= 4 pixels requested up front -- the forward's mfaf_gate_fast_kernel shape --
    //  measured 152 us against this kernel's 87 at 16 x 128 x 128 x 64: 198 registers, two workgroups per CU; not kept)
    if (d.dtype == LEDN_F32) LEDN_LAUNCH((mfaf_gate_bwd_kernel<float, 4>), grid, dim3(256), 0, s, d, ctx_sums);
    else LEDN_LAUNCH((mfaf_gate_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, s, d, ctx_sums);
    if (!ctx_sums) {
        for (int k = 0; k < 4; ++k) {
            const int S = d.ctx_size[k];
            // ~256 pixels per workgroup: row chunks per cell (a function of the shapes only: fixed summation order)
            long nch = ((long)(d.H / S) * (d.W / S)) / 256;
            if (nch < 1) nch = 1;
            if (nch > 64) nch = 64;
            if (nch > cdiv(d.H, S)) nch = cdiv(d.H, S);
            const long cells = (long)d.N * S * S * d.C;
            float* part = nch > 1 ? ws_take(nch * cells) : nullptr;
            if (nch > 1 && !part) return LEDN_EINVAL;
            const dim3 g2((unsigned)(d.N * S * S), (unsigned)nch);
            if (d.dtype == LEDN_F32) LEDN_LAUNCH((mfaf_dctx_det_kernel<float>), g2, dim3(256), 0, s, d, k, part);
            else LEDN_LAUNCH((mfaf_dctx_det_kernel<bf16_t>), g2, dim3(256), 0, s, d, k, part);
            if (part) {
                const int rc = finish_partials(part, (int)nch, (int)cells, 1, d.dctx[k], nullptr, nullptr, s);
                if (rc != LEDN_OK) return rc;
            }
        }
    }
    return check_launch();
}

struct PoolSet {
    const float* p[4];
    int S[4];
    int n;
};

template <typename T, int V>
__global__ void __launch_bounds__(256) mfaf_combine_kernel(T* dx, T* dr, const T* dxl, PoolSet ps, int N, int H,
                                                           int W, int C) {
    const int cv = C / V;
    const long total = (long)N * H * W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, W, H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    float a[V];
    if (dxl) ldv<V>(dxl + pix * C + c, a);
    else {
#pragma unroll
        for (int v = 0; v < V; ++v) a[v] = 0.f;
    }
    for (int k = 0; k < ps.n; ++k) {
        const int S = ps.S[k];
        if (H % S == 0 && W % S == 0) {
            // evenly divisible map (every LED-Net size that is a multiple of 128): the windows tile the map,
            // one cell per pixel -- the general search below costs ~36 integer divisions per pool and made
            // this kernel ALU-bound (127 us for a 100 MB pass)
            const int hs = H / S, ws = W / S;
            const float inv = 1.f / (float)(hs * ws);
            float t[V];
            ldv<V>(ps.p[k] + (((long)n * S + y / hs) * S + x / ws) * C + c, t);
#pragma unroll
            for (int v = 0; v < V; ++v) a[v] = fmaf(t[v], inv, a[v]);
            continue;
        }
        // adaptive windows [floor(i*H/S), ceil((i+1)*H/S)) may overlap: visit every cell containing (y,x)
        int oy0 = (y * S) / H - 1, ox0 = (x * S) / W - 1;
        for (int oy = max(oy0, 0); oy <= min(oy0 + 2, S - 1); ++oy) {
            const int h0 = (oy * H) / S, h1 = ((oy + 1) * H + S - 1) / S;
            if (y < h0 || y >= h1) continue;
            for (int ox = max(ox0, 0); ox <= min(ox0 + 2, S - 1); ++ox) {
                const int w0 = (ox * W) / S, w1 = ((ox + 1) * W + S - 1) / S;
                if (x < w0 || x >= w1) continue;
                const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
                float t[V];
                ldv<V>(ps.p[k] + (((long)n * S + oy) * S + ox) * C + c, t);
#pragma unroll
                for (int v = 0; v < V; ++v) a[v] = fmaf(t[v], inv, a[v]);
            }
        }
    }
    float u[V];
    ldv<V>(dx + pix * C + c, u);
#pragma unroll
    for (int v = 0; v < V; ++v) u[v] += a[v];
    stv<V>(dx + pix * C + c, u);
    if (dr) {
        ldv<V>(dr + pix * C + c, u);
#pragma unroll
        for (int v = 0; v < V; ++v) u[v] += a[v];
        stv<V>(dr + pix * C + c, u);
    }
}

int mfaf_bwd_combine_impl(void* dx, void* dr, const void* dxl, const float* const* dpool, const int* sizes,
                          int npool, int N, int H, int W, int C, int dtype, hipStream_t s) {
    LEDN_REQUIRE(dx && npool >= 0 && npool <= 4 && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    PoolSet ps;
    ps.n = npool;
    for (int k = 0; k < npool; ++k) {
        LEDN_REQUIRE(dpool[k] && sizes[k] > 0);
        ps.p[k] = dpool[k];
        ps.S[k] = sizes[k];
    }
    const long total = (long)N * H * W * (C / 4);
    const dim3 grid((unsigned)cdiv(total, 256));
    if (dtype == LEDN_F32)
        LEDN_LAUNCH((mfaf_combine_kernel<float, 4>), grid, dim3(256), 0, s, (float*)dx, (float*)dr,
                    (const float*)dxl, ps, N, H, W, C);
    else if (dtype == LEDN_BF16)
        LEDN_LAUNCH((mfaf_combine_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (bf16_t*)dx, (bf16_t*)dr,
                    (const bf16_t*)dxl, ps, N, H, W, C);
    else return LEDN_EINVAL;
    return check_launch();
}

}  // namespace ledn
