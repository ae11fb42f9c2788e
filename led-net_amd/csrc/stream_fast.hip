// stream_fast.hip -- the elementwise / per-channel-reduction passes of the train step at the streaming rate
// of the chip: BatchNorm apply + residual + activation (affine_act), BatchNorm backward (reduce, apply),
// channel statistics; bf16 activations, channel count a power of two in 8..512.
//
// Calibration (tools/micro/stream_cal.hip, MI355X): a plain copy-like kernel moves 33 MB in + 33 MB out in
// 10 us and 2 x 33 MB in + 33 MB out in 15 us (6.5-7 TB/s: tensors of this size live in the 256 MB
// Infinity Cache between producer and consumer), provided every lane issues 16-byte accesses and has ALL its
// loads in flight before the first use.  The generic kernels in norm_act.hip / backward.hip (8 B per lane, one
// dependent load -> compute -> store round trip per loop trip) ran the same shapes in 22 / 38 us.
//
// Shape of every kernel here: a workgroup of 256 lanes owns 256*UNR consecutive 8-channel vectors (16 B of
// bf16); 256 is a multiple of C/8, so a lane keeps ONE channel group for all its vectors and the per-channel
// coefficients sit in registers (staged through LDS once per workgroup); the UNR loads of each operand are
// issued back to back, unconditionally (tail lanes re-read vector 0), then consumed.
#include "ledn_rt.h"

namespace ledn {

constexpr int SF_MAXC = 512;

__host__ __device__ inline bool sf_pow2(int c) { return c > 0 && (c & (c - 1)) == 0; }
static inline bool sf_channels_ok(int C) { return sf_pow2(C) && C >= 8 && C <= SF_MAXC; }

__device__ __forceinline__ void unpack8(const uint4& v, float* o) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 ldraw(const void* base, long vec) {
    return reinterpret_cast<const uint4*>(base)[vec];
}

// ===========================================================================
// y = act(res_mode(x [+ xadd] * scale + shift, res))        (ledn_affine_act, all bf16)
// ===========================================================================
template <int ACT, int RES, bool HAS_XADD, int UNR>
__global__ void __launch_bounds__(256) affine_fast_kernel(ledn_affine_desc d, long nvec) {
    __shared__ float s_par[3][SF_MAXC];
    for (int c = threadIdx.x; c < d.C; c += 256) {
        s_par[0][c] = d.scale ? d.scale[c] : 1.f;
        s_par[1][c] = d.shift ? d.shift[c] : 0.f;
        s_par[2][c] = d.slope ? d.slope[c] : 0.f;
    }
    __syncthreads();
    const int c0 = (int)(threadIdx.x % (unsigned)(d.C >> 3)) * 8;
    float sc[8], sh[8], sl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = s_par[0][c0 + i];
        sh[i] = s_par[1][c0 + i];
        sl[i] = ACT == LEDN_ACT_PRELU ? s_par[2][c0 + i] : 0.f;
    }
    const long base = (long)blockIdx.x * (256 * UNR) + threadIdx.x;
    uint4 xr[UNR], ar[UNR], rr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long i = base + u * 256;
        const long j = i < nvec ? i : 0;
        xr[u] = ldraw(d.x, j);
        if (HAS_XADD) ar[u] = ldraw(d.xadd, j);
        if (RES != LEDN_RES_NONE) rr[u] = ldraw(d.res, j);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long i = base + u * 256;
        if (i >= nvec) break;
        float v[8];
        unpack8(xr[u], v);
        if (HAS_XADD) {
            float a[8];
            unpack8(ar[u], a);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += a[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + sh[k];
        if (RES != LEDN_RES_NONE) {
            float r[8];
            unpack8(rr[u], r);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = RES == LEDN_RES_ADD ? v[k] + r[k] : v[k] * r[k] + r[k];
        }
        if (ACT != LEDN_ACT_NONE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = act_apply(ACT, v[k], sl[k]);
        }
        st8(reinterpret_cast<bf16_t*>(d.y) + i * 8, v);
    }
}

// dispatch helpers: runtime (act, res_mode) -> template instance
#define SF_ACT_SWITCH(ACTV, BODY)                                                 \
    switch (ACTV) {                                                               \
        case LEDN_ACT_NONE: { constexpr int A_ = LEDN_ACT_NONE; BODY; } break;    \
        case LEDN_ACT_RELU: { constexpr int A_ = LEDN_ACT_RELU; BODY; } break;    \
        case LEDN_ACT_RELU6: { constexpr int A_ = LEDN_ACT_RELU6; BODY; } break;  \
        case LEDN_ACT_PRELU: { constexpr int A_ = LEDN_ACT_PRELU; BODY; } break;  \
        default: return -1;                                                       \
    }
#define SF_RES_SWITCH(RESV, BODY)                                                 \
    switch (RESV) {                                                               \
        case LEDN_RES_NONE: { constexpr int R_ = LEDN_RES_NONE; BODY; } break;    \
        case LEDN_RES_ADD: { constexpr int R_ = LEDN_RES_ADD; BODY; } break;      \
        case LEDN_RES_GATE: { constexpr int R_ = LEDN_RES_GATE; BODY; } break;    \
        default: return -1;                                                       \
    }

// y = act(x * scale + shift) AND the per-channel sums of y, y^2 (of the stored, bf16-rounded values): the batch statistics
// of a BatchNorm that follows (LEDHead's norm -> act -> conv on the stem maps) without a pass of their own over y -- 69 us
// for the 268 MB map at 1/2 resolution.  Shape of bn_reduce_fast_kernel: a capped grid, lane (row, channel group) owns UNR
// pixels per trip, workgroup totals -> part[block][2][C].
template <int ACT, int UNR>
__global__ void __launch_bounds__(256) affine_stats_fast_kernel(ledn_affine_desc d, float* part) {
    __shared__ float s_par[3][SF_MAXC];
    __shared__ float s_red[2][256 * 8];
    for (int c = threadIdx.x; c < d.C; c += 256) {
        s_par[0][c] = d.scale ? d.scale[c] : 1.f;
        s_par[1][c] = d.shift ? d.shift[c] : 0.f;
        s_par[2][c] = d.slope ? d.slope[c] : 0.f;
    }
    __syncthreads();
    const int cvn = d.C >> 3, rows = 256 / cvn;
    const int cg = (int)(threadIdx.x % (unsigned)cvn), r = (int)(threadIdx.x / (unsigned)cvn);
    const int c0 = cg * 8;
    float sc[8], sh[8], sl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = s_par[0][c0 + k];
        sh[k] = s_par[1][c0 + k];
        sl[k] = ACT == LEDN_ACT_PRELU ? s_par[2][c0 + k] : 0.f;
    }
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = b[k] = 0.f;
    const long stride = (long)gridDim.x * rows;
    for (long p0 = (long)blockIdx.x * rows + r; p0 < d.P; p0 += UNR * stride) {
        uint4 xr[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long p = p0 + u * stride;
            xr[u] = ldraw(d.x, (p < d.P ? p : p0) * cvn + cg);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long p = p0 + u * stride;
            if (p >= d.P) break;
            float v[8];
            unpack8(xr[u], v);
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                float y0 = v[k] * sc[k] + sh[k], y1 = v[k + 1] * sc[k + 1] + sh[k + 1];
                if (ACT != LEDN_ACT_NONE) {
                    y0 = act_apply(ACT, y0, sl[k]);
                    y1 = act_apply(ACT, y1, sl[k + 1]);
                }
                const unsigned w = (unsigned)f32_to_bf16(y0) | ((unsigned)f32_to_bf16(y1) << 16);
                o[k >> 1] = w;
                const float r0 = __uint_as_float(w << 16), r1 = __uint_as_float(w & 0xffff0000u);    // as stored
                a[k] += r0;
                a[k + 1] += r1;
                b[k] = fmaf(r0, r0, b[k]);
                b[k + 1] = fmaf(r1, r1, b[k + 1]);
            }
            reinterpret_cast<uint4*>(d.y)[p * cvn + cg] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s_red[0][threadIdx.x * 8 + k] = a[k];
        s_red[1][threadIdx.x * 8 + k] = b[k];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * d.C; o += 256) {
        const int j = o / d.C, c = o % d.C;
        const float* src = s_red[j] + (c >> 3) * 8 + (c & 7);
        float t0 = 0.f, t1 = 0.f;
        int rr2 = 0;
        for (; rr2 + 1 < rows; rr2 += 2) {
            t0 += src[rr2 * cvn * 8];
            t1 += src[(rr2 + 1) * cvn * 8];
        }
        if (rr2 < rows) t0 += src[rr2 * cvn * 8];
        part[(long)blockIdx.x * 2 * d.C + o] = t0 + t1;
    }
}

// -> LEDN_OK when handled, -1 when the generic kernel must take the call
int affine_act_fast(const ledn_affine_desc& d, hipStream_t s) {
    if (d.dtype_x != LEDN_BF16 || d.dtype_y != LEDN_BF16 || !sf_channels_ok(d.C)) return -1;
    const long nvec = d.P * d.C / 8;
    if (nvec < 4096) return -1;
    if (d.stat_sum) {
        if (d.xadd || d.res_mode != LEDN_RES_NONE || d.act == LEDN_ACT_RELU6) return -1;
        constexpr int UNR = 4;
        const int rows = 256 / (d.C >> 3);
        long nb = cdiv(d.P, (long)rows * UNR);
        static const long cap = exp_knob("LEDN_AFS_CAP", 2048);
        if (nb > cap) nb = cap;
        float* part = ws_take(nb * 2 * d.C);
        if (!part) return -1;
        const dim3 grid((unsigned)nb);
        switch (d.act) {
            case LEDN_ACT_NONE: LEDN_LAUNCH((affine_stats_fast_kernel<LEDN_ACT_NONE, UNR>), grid, dim3(256), 0, s, d, part); break;
            case LEDN_ACT_RELU: LEDN_LAUNCH((affine_stats_fast_kernel<LEDN_ACT_RELU, UNR>), grid, dim3(256), 0, s, d, part); break;
            case LEDN_ACT_PRELU: LEDN_LAUNCH((affine_stats_fast_kernel<LEDN_ACT_PRELU, UNR>), grid, dim3(256), 0, s, d, part); break;
            default: return -1;
        }
        return finish_partials(part, (int)nb, d.C, 2, d.stat_sum, d.stat_sqsum, nullptr, s);
    }
    constexpr int UNR = 4;
    const dim3 grid((unsigned)cdiv(nvec, 256 * UNR));
    if (d.xadd) {
        if (d.act != LEDN_ACT_NONE || d.res_mode != LEDN_RES_NONE) return -1;
        LEDN_LAUNCH((affine_fast_kernel<LEDN_ACT_NONE, LEDN_RES_NONE, true, UNR>), grid, dim3(256), 0, s, d, nvec);
        return check_launch();
    }
    SF_ACT_SWITCH(d.act, SF_RES_SWITCH(d.res_mode,
        LEDN_LAUNCH((affine_fast_kernel<A_, R_, false, UNR>), grid, dim3(256), 0, s, d, nvec)));
    return check_launch();
}

// ===========================================================================
// BatchNorm (+ residual / gate) + activation backward, all bf16
//   t  = res_mode(z*sc + sh, res);  g = dy * act'(t)  (gate: gv = g*res, gres = g*(z*sc+sh+1))
//   reduce: sum gv, sum gv*xhat, sum dslope          xhat = (z - mean) * invstd
//   apply : dz = sc*(gv - mean_g - xhat*mean_gx) = sc*gv + A*z + B,
//           A = -sc*mean_gx*invstd,  B = -sc*mean_g - A*mean      (no BatchNorm: A = B = 0)
// ===========================================================================
struct BnG {
    float gv[8], gres[8], dsl[8];
};
template <int ACT, int RES>
__device__ __forceinline__ void bn_g8(const float* z, const float* dy, const float* r, const float* sc,
                                      const float* sh, const float* sl, BnG& o) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float v = z[k] * sc[k] + sh[k];
        float t = v;
        if (RES == LEDN_RES_ADD) t = v + r[k];
        else if (RES == LEDN_RES_GATE) t = v * r[k] + r[k];
        const float gt = ACT == LEDN_ACT_NONE ? dy[k] : dy[k] * act_grad(ACT, t, sl[k]);
        if (ACT == LEDN_ACT_PRELU) o.dsl[k] = t <= 0.f ? dy[k] * t : 0.f;
        if (RES == LEDN_RES_GATE) {
            o.gv[k] = gt * r[k];
            o.gres[k] = gt * (v + 1.f);
        } else {
            o.gv[k] = gt;
            o.gres[k] = gt;
        }
    }
}

template <int ACT, int RES, bool HAS_DRES, bool HAS_ADD, int UNR>
__global__ void __launch_bounds__(256) bn_apply_fast_kernel(ledn_bnbwd_desc d, long nvec) {
    __shared__ float s_par[5][SF_MAXC];     // sc, sh, sl, A, B
    const float invn = (float)(1.0 / d.count);
    for (int c = threadIdx.x; c < d.C; c += 256) {
        const float sc = d.scale ? d.scale[c] : 1.f;
        s_par[0][c] = sc;
        s_par[1][c] = d.shift ? d.shift[c] : 0.f;
        s_par[2][c] = d.slope ? d.slope[c] : 0.f;
        float A = 0.f, B = 0.f;
        if (d.rows) {                        // per-row sums left by the reduce pass: add them up (fixed order)
            float tg = 0.f, tgx = 0.f, tsl = 0.f;
            for (int r = 0; r < LEDN_BNBWD_ROWS; ++r) {
                const float* row = d.rows + (long)r * 3 * d.C;
                tg += row[c];
                tgx += row[d.C + c];
                tsl += row[2 * d.C + c];
            }
            if (d.bn_mode) {
                A = -sc * (tgx * invn) * d.invstd[c];
                B = -sc * (tg * invn) - A * d.mean[c];
            }
            if (blockIdx.x == 0) {           // the totals ARE d_beta, d_gamma, d_slope: accumulated into the sinks once
                if (d.sum_g) d.sum_g[c] += tg;
                if (d.sum_gx) d.sum_gx[c] += tgx;
                if (d.dslope) d.dslope[c] += tsl;
            }
        } else if (d.bn_mode) {
            const float mg = d.sum_g[c] * invn, mgx = d.sum_gx[c] * invn;
            A = -sc * mgx * d.invstd[c];
            B = -sc * mg - A * d.mean[c];
        }
        s_par[3][c] = A;
        s_par[4][c] = B;
    }
    __syncthreads();
    const int c0 = (int)(threadIdx.x % (unsigned)(d.C >> 3)) * 8;
    float sc[8], sh[8], sl[8], A[8], B[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = s_par[0][c0 + k];
        sh[k] = s_par[1][c0 + k];
        sl[k] = ACT == LEDN_ACT_PRELU ? s_par[2][c0 + k] : 0.f;
        A[k] = s_par[3][c0 + k];
        B[k] = s_par[4][c0 + k];
    }
    const long base = (long)blockIdx.x * (256 * UNR) + threadIdx.x;
    uint4 zr[UNR], gr[UNR], rr[UNR], az[UNR], ar[UNR];
    // HAS_ADD: partial gradients of z / res from another consumer of the same forward tensor (fan-in folded in)
    const bool add_z = HAS_ADD && d.dz_add != nullptr, add_r = HAS_ADD && HAS_DRES && d.dres_add != nullptr;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long i = base + u * 256;
        const long j = i < nvec ? i : 0;
        zr[u] = ldraw(d.z, j);
        gr[u] = ldraw(d.dy, j);
        if (RES != LEDN_RES_NONE) rr[u] = ldraw(d.res, j);
        if (HAS_ADD) {
            if (add_z) az[u] = ldraw(d.dz_add, j);
            if (add_r) ar[u] = ldraw(d.dres_add, j);
        }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long i = base + u * 256;
        if (i >= nvec) break;
        float z[8], dy[8], r[8];
        unpack8(zr[u], z);
        unpack8(gr[u], dy);
        if (RES != LEDN_RES_NONE) unpack8(rr[u], r);
        BnG g;
        bn_g8<ACT, RES>(z, dy, r, sc, sh, sl, g);
        float dz[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dz[k] = fmaf(sc[k], g.gv[k], fmaf(A[k], z[k], B[k]));
        if (HAS_ADD) {
            if (add_z) {
                float a[8];
                unpack8(az[u], a);
#pragma unroll
                for (int k = 0; k < 8; ++k) dz[k] += a[k];
            }
            if (add_r) {
                float a[8];
                unpack8(ar[u], a);
#pragma unroll
                for (int k = 0; k < 8; ++k) g.gres[k] += a[k];
            }
        }
        st8(reinterpret_cast<bf16_t*>(d.dz) + i * 8, dz);
        if (HAS_DRES) st8(reinterpret_cast<bf16_t*>(d.dres) + i * 8, g.gres);
    }
}

int bn_act_bwd_apply_fast(const ledn_bnbwd_desc& d, hipStream_t s) {
    if (d.dtype_z != LEDN_BF16 || d.dtype_y != LEDN_BF16 || !sf_channels_ok(d.C)) return -1;
    const long nvec = d.P * d.C / 8;
    if (nvec < 4096) return -1;
    const bool has_dres = d.dres != nullptr;
    if (has_dres && d.res_mode == LEDN_RES_NONE) return -1;
    constexpr int UNR = 4;
    const dim3 grid((unsigned)cdiv(nvec, 256 * UNR));
    const bool has_add = d.dz_add != nullptr || d.dres_add != nullptr;
    if (has_add) {      // (one instance per (act, res) with both optional adds resolved at run time: workgroup-uniform)
        if (has_dres) {
            SF_ACT_SWITCH(d.act, SF_RES_SWITCH(d.res_mode,
                LEDN_LAUNCH((bn_apply_fast_kernel<A_, R_, R_ != LEDN_RES_NONE, true, UNR>), grid, dim3(256), 0, s, d, nvec)));
        } else {
            SF_ACT_SWITCH(d.act, SF_RES_SWITCH(d.res_mode,
                LEDN_LAUNCH((bn_apply_fast_kernel<A_, R_, false, true, UNR>), grid, dim3(256), 0, s, d, nvec)));
        }
    } else if (has_dres) {
        SF_ACT_SWITCH(d.act, SF_RES_SWITCH(d.res_mode,
            LEDN_LAUNCH((bn_apply_fast_kernel<A_, R_, R_ != LEDN_RES_NONE, false, UNR>), grid, dim3(256), 0, s, d, nvec)));
    } else {
        SF_ACT_SWITCH(d.act, SF_RES_SWITCH(d.res_mode,
            LEDN_LAUNCH((bn_apply_fast_kernel<A_, R_, false, false, UNR>), grid, dim3(256), 0, s, d, nvec)));
    }
    return check_launch();
}

// reduce: lane (row r, channel group cg) owns UNR pixels r, r + rows, ... of its workgroup's chunk (all loads
// first, ONE trip: workgroups start and retire continuously, so the loads of one overlap the arithmetic of
// another -- this kernel is as much VALU- as memory-bound: ~7-12 VALU operations per element at 39 T op/s
// are 3-5 us on a 33 MB tensor); workgroup totals -> part[block][3][C] (finish_partials adds them up).
// sum g*xhat is accumulated as sum g*(z - mean) and scaled by invstd once per workgroup.
template <int ACT, int RES, int UNR>
__global__ void __launch_bounds__(256) bn_reduce_fast_kernel(ledn_bnbwd_desc d, float* part) {
    __shared__ float s_par[5][SF_MAXC];     // sc, sh, sl, mean, invstd
    __shared__ float s_red[3][256 * 8];
    for (int c = threadIdx.x; c < d.C; c += 256) {
        s_par[0][c] = d.scale ? d.scale[c] : 1.f;
        s_par[1][c] = d.shift ? d.shift[c] : 0.f;
        s_par[2][c] = d.slope ? d.slope[c] : 0.f;
        s_par[3][c] = d.bn_mode ? d.mean[c] : 0.f;
        s_par[4][c] = d.bn_mode ? d.invstd[c] : 0.f;
    }
    __syncthreads();
    const int cvn = d.C >> 3, rows = 256 / cvn;
    const int cg = (int)(threadIdx.x % (unsigned)cvn), r = (int)(threadIdx.x / (unsigned)cvn);
    const int c0 = cg * 8;
    float sc[8], sh[8], sl[8], mean[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = s_par[0][c0 + k];
        sh[k] = s_par[1][c0 + k];
        sl[k] = ACT == LEDN_ACT_PRELU ? s_par[2][c0 + k] : 0.f;
        mean[k] = s_par[3][c0 + k];
    }
    float a[8], b[8], e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = b[k] = e[k] = 0.f;
    const long stride = (long)gridDim.x * rows;
    for (long p0 = (long)blockIdx.x * rows + r; p0 < d.P; p0 += UNR * stride) {
        uint4 zr[UNR], gr[UNR], rr[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long p = p0 + u * stride;
            const long j = (p < d.P ? p : p0) * cvn + cg;
            zr[u] = ldraw(d.z, j);
            gr[u] = ldraw(d.dy, j);
            if (RES != LEDN_RES_NONE) rr[u] = ldraw(d.res, j);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (p0 + u * stride >= d.P) break;
            float z[8], dy[8], rv[8];
            unpack8(zr[u], z);
            unpack8(gr[u], dy);
            if (RES != LEDN_RES_NONE) unpack8(rr[u], rv);
            BnG g;
            bn_g8<ACT, RES>(z, dy, rv, sc, sh, sl, g);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] += g.gv[k];
                b[k] = fmaf(g.gv[k], z[k] - mean[k], b[k]);
                if (ACT == LEDN_ACT_PRELU) e[k] += g.dsl[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s_red[0][threadIdx.x * 8 + k] = a[k];
        s_red[1][threadIdx.x * 8 + k] = b[k];
        s_red[2][threadIdx.x * 8 + k] = e[k];
    }
    __syncthreads();
    // output (kind j, channel c): sum over the workgroup's rows; all 256 lanes take part
    for (int o = threadIdx.x; o < 3 * d.C; o += 256) {
        const int j = o / d.C, c = o % d.C;
        const float* src = s_red[j] + (c >> 3) * 8 + (c & 7);
        float t0 = 0.f, t1 = 0.f;
        int rr2 = 0;
        for (; rr2 + 1 < rows; rr2 += 2) {
            t0 += src[rr2 * cvn * 8];
            t1 += src[(rr2 + 1) * cvn * 8];
        }
        if (rr2 < rows) t0 += src[rr2 * cvn * 8];
        float t = t0 + t1;
        if (j == 1) t *= s_par[4][c];
        if (d.rows) atomicAdd(d.rows + (long)(blockIdx.x % LEDN_BNBWD_ROWS) * 3 * d.C + o, t);
        else part[(long)blockIdx.x * 3 * d.C + o] = t;
    }
}

int bn_act_bwd_reduce_fast(const ledn_bnbwd_desc& d, hipStream_t s) {
    if (d.dtype_z != LEDN_BF16 || d.dtype_y != LEDN_BF16 || !sf_channels_ok(d.C)) return -1;
    if (d.P * d.C / 8 < 4096) return -1;
#ifndef LEDN_BNR_UNR
#define LEDN_BNR_UNR 4
#endif
    constexpr int UNR = LEDN_BNR_UNR;       // (A/B builds: -DLEDN_BNR_UNR=8 measured ... see EXPERIMENTS.md)
    const int rows = 256 / (d.C >> 3);
    // one trip per lane up to 1024 workgroups = ONE resident round (4 workgroups per CU by the 34 KB of LDS); whole step,
    // one box: cap 2048 13.13 ms, 1280 13.19, 1024 13.03 (three runs each), 768 13.05, 512 13.11  (LEDN_BNR_CAP: A/B knob).
    // A second register buffer (next trip's loads issued before this trip's arithmetic; plain / ReLU variants, 124 VGPRs)
    // measured 13.15 against 13.08 ms, three alternating runs: not kept.
    long nb = cdiv(d.P, (long)rows * UNR);
    static const long cap = exp_knob("LEDN_BNR_CAP", 1024);
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    float* part = d.rows ? nullptr : ws_take(nb * 3 * d.C);
    if (!part && !d.rows) return -1;
    const dim3 grid((unsigned)nb);
    SF_ACT_SWITCH(d.act, SF_RES_SWITCH(d.res_mode,
        LEDN_LAUNCH((bn_reduce_fast_kernel<A_, R_, UNR>), grid, dim3(256), 0, s, d, part)));
    if (d.rows) return check_launch();       // the apply pass adds the rows up (no summing launch)
    return finish_partials(part, (int)nb, d.C, 3, d.sum_g, d.sum_gx, d.dslope, s);
}

// ===========================================================================
// BatchNorm backward in ONE pass over memory (round 4): reduce + summing launch + apply fused into a persistent kernel.
// The two-kernel form reads g and z twice (reduce: 2 tensors, apply: 2 tensors + 1 written = 167 MB at C = 64, 262144
// pixels, bf16) and puts three launches on the critical stream.  Here every lane KEEPS its vectors of z and dy in
// registers across the reduction (<= 256 workgroups x 512 lanes x 16 vectors x 16 B = 32 MB: the register files of the
// chip hold the tensor); dy is read a second time, from the Infinity Cache its producer left it in: one launch, z read once:
//   phase 1  load z, dy [, res], per-lane sums of (g, g*(z - mean), d slope) -> workgroup sums through LDS -> float
//            atomics into the [3][C] totals (256 adders per address: ~2 us at the memory-side atomic units)
//   barrier  agent-scope release + arrival counter, ONE lane polls (relaxed, s_sleep), agent-scope acquire
//            (cdna_hip_programming.md Guideline 16); the spin is BOUNDED: a launch whose workgroups are not all resident
//            gives up, raises the timeout word and the host falls back to the two-kernel form for good
//   phase 2  A, B from the totals; dz = sc*g + A*z + B from the registers [+ fan-in addends], dres
// Residency: grid <= 256 workgroups of 512 lanes on a 256-CU chip (one per CU fits whatever else runs: no other kernel
// in this library waits on a resident peer, so every CU frees up in finite time).  Only launched from the step's main
// stream (the host side decides), never in deterministic mode (atomics) nor under SyncBN (the all-reduce sits between
// the two halves).  The CPU emulator runs workgroups one after another: no grid barrier there (entry returns LEDN_ESKIP).
// MEASURED (MI355X, tools/bn_fused_bench.py, profiles/r04_bn_fused_bench.txt): correct (tests/test_stream_fast.py, 25 cases,
// no barrier time-out) but SLOWER than the two kernels it replaces -- C = 64, 262144 pixels: 56.9 us against 50.6 us
// (eager launches; 46.9 us inside the graph), the whole step 13.59 against 13.19 ms with the 25 eligible launches switched
// over; 1024-lane workgroups (16 waves per CU, 128 registers): 69.6 us.  256 workgroups x 8 waves keep ~64 KB of loads in
// flight per CU in 4-vector chunks; the two-kernel form has 2048 short workgroups retiring continuously.  OFF by default
// (LEDN_OPT_BN_FUSED / LEDN_EXPERIMENTAL=1 LEDN_BN_FUSED=1); kept as the measured answer to "one pass instead of two".
// ===========================================================================
#ifndef LEDN_CPU_EMU
constexpr int BF_T = 512, BF_G = 256;
constexpr unsigned BF_SPIN_LIMIT = 1u << 21;        // x ~0.2 us per poll: gives up after ~0.4 s

template <int ACT, int RES, bool HAS_DRES, bool HAS_ADD, int V>
__global__ void __launch_bounds__(BF_T, 2) bn_bwd_fused_kernel(ledn_bnbwd_desc d, long nvec, float* tot, unsigned* sync) {
    // z stays ON CHIP between the two phases, in LDS: 512 lanes x V = 16 vectors x 16 B = 128 KB of the CU's 160 KB.
    // (Registers were the first choice -- 2 x 16 vectors per lane -- but every instance then spilled 0.2-2 KB per lane:
    //  the arithmetic of a vector wants ~100 registers whatever scheduling fences are placed.)  dy is read a second time
    //  from the Infinity Cache its producer left it in.
    __shared__ uint4 s_z[V * BF_T];
    __shared__ float s_par[5][128];         // sc, sh, sl, mean, invstd      (C <= 128)
    __shared__ float s_ab[2][128];          // A, B of phase 2
    __shared__ float s_redw[BF_T / 64][3 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < d.C; c += BF_T) {
        s_par[0][c] = d.scale ? d.scale[c] : 1.f;
        s_par[1][c] = d.shift ? d.shift[c] : 0.f;
        s_par[2][c] = d.slope ? d.slope[c] : 0.f;
        s_par[3][c] = d.bn_mode ? d.mean[c] : 0.f;
        s_par[4][c] = d.bn_mode ? d.invstd[c] : 0.f;
    }
    __syncthreads();
    const int cvn = d.C >> 3;               // <= 16: a wave holds 64 / cvn lanes per channel group
    const int cg = (int)(tid % (unsigned)cvn);
    const int c0 = cg * 8;
    const long stride = (long)gridDim.x * BF_T;
    const long first = (long)blockIdx.x * BF_T + tid;
    float sc[8], sh[8], sl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = s_par[0][c0 + k];
        sh[k] = s_par[1][c0 + k];
        sl[k] = ACT == LEDN_ACT_PRELU ? s_par[2][c0 + k] : 0.f;
    }
    {
        float mean[8], a[8], b[8], e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mean[k] = s_par[3][c0 + k];
            a[k] = b[k] = e[k] = 0.f;
        }
        for (int v0 = 0; v0 < V; v0 += 4) {
            uint4 zq[4], gq[4], rr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = first + (v0 + u) * stride;
                const long j = i < nvec ? i : 0;
                zq[u] = ldraw(d.z, j);
                gq[u] = ldraw(d.dy, j);
                if (RES != LEDN_RES_NONE) rr[u] = ldraw(d.res, j);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s_z[(v0 + u) * BF_T + tid] = zq[u];
                if (first + (v0 + u) * stride >= nvec) continue;
                float z[8], dy[8], rv[8];
                unpack8(zq[u], z);
                unpack8(gq[u], dy);
                if (RES != LEDN_RES_NONE) unpack8(rr[u], rv);
                BnG g;
                bn_g8<ACT, RES>(z, dy, rv, sc, sh, sl, g);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    a[k] += g.gv[k];
                    b[k] = fmaf(g.gv[k], z[k] - mean[k], b[k]);
                    if (ACT == LEDN_ACT_PRELU) e[k] += g.dsl[k];
                }
            }
        }
        // lanes of one channel group inside the wave (cvn apart), then the 8 waves through LDS
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            for (int o = cvn; o < 64; o <<= 1) {
                a[k] += __shfl_xor(a[k], o);
                b[k] += __shfl_xor(b[k], o);
                if (ACT == LEDN_ACT_PRELU) e[k] += __shfl_xor(e[k], o);
            }
        }
        if (lane < cvn) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s_redw[wave][c0 + k] = a[k];
                s_redw[wave][d.C + c0 + k] = b[k];
                s_redw[wave][2 * d.C + c0 + k] = e[k];
            }
        }
    }
    __syncthreads();
    for (int o = tid; o < 3 * d.C; o += BF_T) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < BF_T / 64; ++w) t += s_redw[w][o];
        if (o >= d.C && o < 2 * d.C) t *= s_par[4][o - d.C];
        atomicAdd(tot + o, t);
    }
    // ---- grid barrier (Guideline 16): every adding wave drains, the workgroup meets, lane 0 releases + arrives + polls
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > BF_SPIN_LIMIT) {          // not every workgroup is resident: give up loudly, never hang
                __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // ---- phase 2: totals -> A, B (as bn_apply_fast_kernel), the first workgroup hands the totals to the gradient sinks
    const float invn = (float)(1.0 / d.count);
    for (int c = tid; c < d.C; c += BF_T) {
        const float tg = __hip_atomic_load(tot + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float tgx = __hip_atomic_load(tot + d.C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float scc = s_par[0][c];
        float A = 0.f, B = 0.f;
        if (d.bn_mode) {
            A = -scc * (tgx * invn) * s_par[4][c];
            B = -scc * (tg * invn) - A * s_par[3][c];
        }
        if (blockIdx.x == 0) {
            if (d.sum_g) d.sum_g[c] += tg;
            if (d.sum_gx) d.sum_gx[c] += tgx;
            if (d.dslope) d.dslope[c] += __hip_atomic_load(tot + 2 * d.C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_ab[0][c] = A;
        s_ab[1][c] = B;
    }
    __syncthreads();
    float A[8], B[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        A[k] = s_ab[0][c0 + k];
        B[k] = s_ab[1][c0 + k];
    }
    const bool add_z = HAS_ADD && d.dz_add != nullptr, add_r = HAS_ADD && HAS_DRES && d.dres_add != nullptr;
    for (int v0 = 0; v0 < V; v0 += 4) {
        uint4 gq[4], rr[4], az[4], ar[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = first + (v0 + u) * stride;
            const long j = i < nvec ? i : 0;
            gq[u] = ldraw(d.dy, j);
            if (RES != LEDN_RES_NONE) rr[u] = ldraw(d.res, j);
            if (HAS_ADD) {
                if (add_z) az[u] = ldraw(d.dz_add, j);
                if (add_r) ar[u] = ldraw(d.dres_add, j);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = first + (v0 + u) * stride;
            if (i >= nvec) continue;
            float z[8], dy[8], rv[8];
            unpack8(s_z[(v0 + u) * BF_T + tid], z);
            unpack8(gq[u], dy);
            if (RES != LEDN_RES_NONE) unpack8(rr[u], rv);
            BnG g;
            bn_g8<ACT, RES>(z, dy, rv, sc, sh, sl, g);
            float dz[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) dz[k] = fmaf(sc[k], g.gv[k], fmaf(A[k], z[k], B[k]));
            if (HAS_ADD) {
                if (add_z) {
                    float t[8];
                    unpack8(az[u], t);
#pragma unroll
                    for (int k = 0; k < 8; ++k) dz[k] += t[k];
                }
                if (add_r) {
                    float t[8];
                    unpack8(ar[u], t);
#pragma unroll
                    for (int k = 0; k < 8; ++k) g.gres[k] += t[k];
                }
            }
            st8(reinterpret_cast<bf16_t*>(d.dz) + i * 8, dz);
            if (HAS_DRES) st8(reinterpret_cast<bf16_t*>(d.dres) + i * 8, g.gres);
        }
    }
}

static int bf_state = 0;        // 0 untested, 1 usable, -1 disabled (fewer than 256 CUs, or a barrier timed out)

// -> LEDN_OK handled | 3 (LEDN_ESKIP) not applicable: the caller runs ledn_bn_act_bwd_reduce + _apply
int bn_act_bwd_fused(const ledn_bnbwd_desc& d, hipStream_t s) {
    if (bf_state < 0 || det() || d.rows) return LEDN_ESKIP;
    if (d.dtype_z != LEDN_BF16 || d.dtype_y != LEDN_BF16 || !sf_channels_ok(d.C) || d.C > 128) return LEDN_ESKIP;
    if (d.act == LEDN_ACT_SIGMOID || d.act == LEDN_ACT_RELU6 || d.res_mode == LEDN_RES_GATE) return LEDN_ESKIP;
    const bool has_dres = d.dres != nullptr;
    if (has_dres && d.res_mode == LEDN_RES_NONE) return LEDN_ESKIP;
    if (!d.bn_mode && d.slope == nullptr && false) return LEDN_ESKIP;
    const long nvec = d.P * d.C / 8;
    static const long vmin = exp_knob("LEDN_BNF_MIN", 1L << 16);
    if (nvec < vmin || nvec > (long)BF_G * BF_T * 16) return LEDN_ESKIP;
    if (bf_state == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            prop.multiProcessorCount < BF_G) {
            bf_state = -1;
            return LEDN_ESKIP;
        }
        bf_state = 1;
    }
    const long tot_floats = 3L * d.C + 16;
    float* ws = ws_take(tot_floats);
    if (!ws) return LEDN_ESKIP;
    unsigned* sync = reinterpret_cast<unsigned*>(ws + 3 * d.C);
    const int V = nvec > (long)BF_G * BF_T * 4 ? 16 : 4;
    long G = cdiv(nvec, (long)BF_T * V);
    if (G > BF_G) return LEDN_ESKIP;
    // totals, arrival counter and timeout word start from zero EVERY launch (a memset node when captured)
    if (hipMemsetAsync(ws, 0, sizeof(float) * (size_t)tot_floats, s) != hipSuccess) return LEDN_ELAUNCH;
    const dim3 grid((unsigned)G);
    const bool has_add = d.dz_add != nullptr || d.dres_add != nullptr;
#define LEDN_BF(ACT_, RES_, DRES_)                                                                                  \
    do {                                                                                                            \
        if (V == 16) {                                                                                              \
            if (has_add) LEDN_LAUNCH((bn_bwd_fused_kernel<ACT_, RES_, DRES_, true, 16>), grid, dim3(BF_T), 0, s, d, nvec, ws, sync);  \
            else LEDN_LAUNCH((bn_bwd_fused_kernel<ACT_, RES_, DRES_, false, 16>), grid, dim3(BF_T), 0, s, d, nvec, ws, sync);         \
        } else {                                                                                                    \
            if (has_add) LEDN_LAUNCH((bn_bwd_fused_kernel<ACT_, RES_, DRES_, true, 4>), grid, dim3(BF_T), 0, s, d, nvec, ws, sync);   \
            else LEDN_LAUNCH((bn_bwd_fused_kernel<ACT_, RES_, DRES_, false, 4>), grid, dim3(BF_T), 0, s, d, nvec, ws, sync);          \
        }                                                                                                           \
    } while (0)
#define LEDN_BF_RES(ACT_)                                                                  \
    do {                                                                                   \
        if (d.res_mode == LEDN_RES_NONE) LEDN_BF(ACT_, LEDN_RES_NONE, false);              \
        else if (has_dres) LEDN_BF(ACT_, LEDN_RES_ADD, true);                              \
        else LEDN_BF(ACT_, LEDN_RES_ADD, false);                                           \
    } while (0)
    if (d.act == LEDN_ACT_NONE) LEDN_BF_RES(LEDN_ACT_NONE);
    else if (d.act == LEDN_ACT_RELU) LEDN_BF_RES(LEDN_ACT_RELU);
    else LEDN_BF_RES(LEDN_ACT_PRELU);
#undef LEDN_BF_RES
#undef LEDN_BF
    return check_launch();
}
// the timeout word of the last fused launch on this stream's workspace (host-side check after a synchronisation):
// nonzero = a barrier gave up; the fused form is then switched off for the rest of the process
int bn_act_bwd_fused_check(int C, hipStream_t s) {
    float* ws = ws_take(3L * C + 16);
    if (!ws) return 0;
    unsigned flag = 0;
    if (hipMemcpyAsync(&flag, reinterpret_cast<unsigned*>(ws + 3 * C) + 1, sizeof(flag), hipMemcpyDeviceToHost, s) != hipSuccess) return 0;
    if (hipStreamSynchronize(s) != hipSuccess) return 0;
    if (flag) bf_state = -1;
    return (int)flag;
}
#else
int bn_act_bwd_fused(const ledn_bnbwd_desc&, hipStream_t) { return LEDN_ESKIP; }
int bn_act_bwd_fused_check(int, hipStream_t) { return 0; }
#endif

// ===========================================================================
// per-channel sum / sum of squares of x [+ xadd], bf16
// ===========================================================================
template <bool HAS_XADD, int UNR>
__global__ void __launch_bounds__(256) stats_fast_kernel(const void* x, const void* xadd, long P, int C, float* part) {
    __shared__ float s_red[2][256 * 8];
    const int cvn = C >> 3, rows = 256 / cvn;
    const int cg = (int)(threadIdx.x % (unsigned)cvn), r = (int)(threadIdx.x / (unsigned)cvn);
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = b[k] = 0.f;
    const long stride = (long)gridDim.x * rows;
    for (long p0 = (long)blockIdx.x * rows + r; p0 < P; p0 += UNR * stride) {
        uint4 xr[UNR], ar[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long p = p0 + u * stride;
            const long j = (p < P ? p : p0) * cvn + cg;
            xr[u] = ldraw(x, j);
            if (HAS_XADD) ar[u] = ldraw(xadd, j);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (p0 + u * stride >= P) break;
            float v[8];
            unpack8(xr[u], v);
            if (HAS_XADD) {
                float w[8];
                unpack8(ar[u], w);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += w[k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] += v[k];
                b[k] = fmaf(v[k], v[k], b[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s_red[0][threadIdx.x * 8 + k] = a[k];
        s_red[1][threadIdx.x * 8 + k] = b[k];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * C; o += 256) {
        const int j = o / C, c = o % C;
        const float* src = s_red[j] + (c >> 3) * 8 + (c & 7);
        float t = 0.f;
        for (int rr2 = 0; rr2 < rows; ++rr2) t += src[rr2 * cvn * 8];
        part[(long)blockIdx.x * 2 * C + o] = t;
    }
}

int channel_stats_fast(const void* x, const void* xadd, long long P, int C, int dtype, float* sum, float* sqsum,
                       hipStream_t s) {
    if (dtype != LEDN_BF16 || !sf_channels_ok(C) || !sqsum) return -1;
    if (P * C / 8 < 4096) return -1;
    constexpr int UNR = 4;
    const int rows = 256 / (C >> 3);
    long nb = cdiv((long)P, (long)rows * UNR * 2);
    if (nb > 1024) nb = 1024;
    float* part = ws_take(nb * 2 * C);
    if (!part) return -1;
    const dim3 grid((unsigned)nb);
    if (xadd) LEDN_LAUNCH((stats_fast_kernel<true, UNR>), grid, dim3(256), 0, s, x, xadd, (long)P, C, part);
    else LEDN_LAUNCH((stats_fast_kernel<false, UNR>), grid, dim3(256), 0, s, x, xadd, (long)P, C, part);
    return finish_partials(part, (int)nb, C, 2, sum, sqsum, nullptr, s);
}

}  // namespace ledn
