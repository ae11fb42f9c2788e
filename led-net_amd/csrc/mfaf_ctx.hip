// mfaf_ctx.hip -- the four pooled-context MLPs of Muti_AFF (classification/model_utils.py:377-400) for all four pooling
// scales in one launch sequence (include/ledn.h: ledn_mfaf_ctx_fwd / ledn_mfaf_ctx_bwd).  Through the per-layer entry
// points each scale is a chain of ~13 tiny launches per train step (conv, statistics, finalize, affine, conv, their
// adjoints and weight gradients: 5-20 us each on maps of 1 to 256 pixels per image) -- ~1 ms of a 15 ms step for a few
// MFLOP.  Here: forward = 2 launches, backward = 4, every workgroup owns <= 256 pixels of ONE scale, thread = pixel with
// the 16 x 64 filters of its scale in LDS; the weight gradients are C x Ci outer-product sums with one thread per four
// filter elements walking the workgroup's pixels.  f32 throughout (the pooled maps are f32).
#include "ledn_rt.h"

namespace ledn {

constexpr int MC_C = 64, MC_CI = 16, MC_PX = 256;

struct ledn_mfafctx_P {
    int P[4];
};
struct McBlock {
    int s;        // scale
    int p0;       // first pixel of the workgroup inside the scale
};
__device__ __forceinline__ McBlock mc_block(const int* P) {
    int b = (int)blockIdx.x;
    McBlock r;
    r.s = 0;
    r.p0 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int nb = (P[k] + MC_PX - 1) / MC_PX;
        if (b < nb) {
            r.s = k;
            r.p0 = b * MC_PX;
            return r;
        }
        b -= nb;
    }
    r.s = -1;
    return r;
}
static int mc_blocks(const int* P) {
    int n = 0;
    for (int k = 0; k < 4; ++k) n += (P[k] + MC_PX - 1) / MC_PX;
    return n;
}

// sum over the 64 lanes of a wave, then over the workgroup's waves through LDS; lane 0 of wave 0 returns the total
__device__ __forceinline__ float mc_wave_sum(float v) { return wave_sum(v); }

// Deterministic mode (LEDN_OPT_DETERMINISTIC): the workgroups of a launch write their partial sums as rows
// part[blockIdx.x][K] instead of adding them atomically, and this kernel adds every scale's rows up in block order.
struct McDst {
    float* p[4];
};
__global__ void __launch_bounds__(256) mfaf_ctx_finish_kernel(const float* part, int K, McDst dst, ledn_mfafctx_P P) {
    const int s = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K || !dst.p[s]) return;
    int b0 = 0;
    for (int q = 0; q < s; ++q) b0 += (P.P[q] + MC_PX - 1) / MC_PX;
    const int nb = (P.P[s] + MC_PX - 1) / MC_PX;
    float t = 0.f;
    for (int b = 0; b < nb; ++b) t += part[(long)(b0 + b) * K + k];
    dst.p[s][k] += t;
}
static int mc_finish(const float* part, int K, float* d0, float* d1, float* d2, float* d3, const int* P, hipStream_t s) {
    McDst dst;
    dst.p[0] = d0; dst.p[1] = d1; dst.p[2] = d2; dst.p[3] = d3;
    ledn_mfafctx_P pp;
    for (int k = 0; k < 4; ++k) pp.P[k] = P[k];
    LEDN_LAUNCH(mfaf_ctx_finish_kernel, dim3((unsigned)cdiv(K, 256), 4u), dim3(256), 0, s, part, K, dst, pp);
    return check_launch();
}

// rows [KWT filter-gradient elements | CA bias-gradient elements] -> dw[s], db[s] (db[s] may be null)
__global__ void __launch_bounds__(256) mfaf_ctx_finish_w_kernel(const float* part, int KWT, int CA, McDst dw, McDst db,
                                                                ledn_mfafctx_P P) {
    const int s = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x, K = KWT + CA;
    if (k >= K) return;
    float* dst = k < KWT ? (dw.p[s] ? dw.p[s] + k : nullptr) : (db.p[s] ? db.p[s] + (k - KWT) : nullptr);
    if (!dst) return;
    int b0 = 0;
    for (int q = 0; q < s; ++q) b0 += (P.P[q] + MC_PX - 1) / MC_PX;
    const int nb = (P.P[s] + MC_PX - 1) / MC_PX;
    float t = 0.f;
    for (int b = 0; b < nb; ++b) t += part[(long)(b0 + b) * K + k];
    *dst += t;
}
static int mc_finish_w(const float* part, int KWT, int CA, float* const* dw, float* const* db, const int* P, hipStream_t s) {
    McDst a, b;
    ledn_mfafctx_P pp;
    for (int k = 0; k < 4; ++k) {
        a.p[k] = dw[k];
        b.p[k] = db[k];
        pp.P[k] = P[k];
    }
    LEDN_LAUNCH(mfaf_ctx_finish_w_kernel, dim3((unsigned)cdiv(KWT + CA, 256), 4u), dim3(256), 0, s, part, KWT, CA, a, b, pp);
    return check_launch();
}

// ---- forward 1: z1 = W1 x + b1, per-scale sums of z1 -------------------------------------------------
__global__ void __launch_bounds__(MC_PX) mfaf_ctx_fwd1_kernel(ledn_mfafctx_desc d, int training, float* part) {
    __shared__ float s_w[MC_CI * MC_C];
    __shared__ float s_red[4][2 * MC_CI];
    const McBlock blk = mc_block(d.P);
    if (blk.s < 0) return;
    const int s = blk.s, tid = threadIdx.x;
    for (int i = tid; i < MC_CI * MC_C; i += MC_PX) s_w[i] = d.w1[s][i];
    __syncthreads();
    const int p = blk.p0 + tid;
    const bool ok = p < d.P[s];
    float z[MC_CI];
#pragma unroll
    for (int co = 0; co < MC_CI; ++co) z[co] = d.b1[s] ? d.b1[s][co] : 0.f;
    if (ok) {
        const float4* x4 = reinterpret_cast<const float4*>(d.pooled[s] + (long)p * MC_C);
#pragma unroll 4
        for (int q = 0; q < MC_C / 4; ++q) {
            const float4 xv = x4[q];
#pragma unroll
            for (int co = 0; co < MC_CI; ++co) {
                const float* w = s_w + co * MC_C + 4 * q;
                z[co] = fmaf(xv.x, w[0], fmaf(xv.y, w[1], fmaf(xv.z, w[2], fmaf(xv.w, w[3], z[co]))));
            }
        }
        float4* o4 = reinterpret_cast<float4*>(d.z1[s] + (long)p * MC_CI);
#pragma unroll
        for (int q = 0; q < MC_CI / 4; ++q) o4[q] = make_float4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]);
    }
    if (!training) return;
    const int wv = tid >> 6;
#pragma unroll
    for (int co = 0; co < MC_CI; ++co) {
        const float v = ok ? z[co] : 0.f;
        const float a = mc_wave_sum(v), b = mc_wave_sum(v * v);
        if ((tid & 63) == 0) {
            s_red[wv][co] = a;
            s_red[wv][MC_CI + co] = b;
        }
    }
    __syncthreads();
    if (tid < 2 * MC_CI) {
        const float t = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
        if (part) part[(long)blockIdx.x * 2 * MC_CI + tid] = t;
        else atomicAdd(d.stats1 + s * 2 * MC_CI + tid, t);
    }
}

// ---- forward 2: BatchNorm (batch or running statistics) + ReLU, z2 = W2 mid + b2 ----------------------
__global__ void __launch_bounds__(MC_PX) mfaf_ctx_fwd2_kernel(ledn_mfafctx_desc d, int training, float* part) {
    __shared__ float s_w[MC_C * MC_CI];
    __shared__ float s_bn[2 * MC_CI];
    __shared__ float s_st[4][2 * MC_C];           // one row per wave, added up in wave order (no LDS atomics)
    const McBlock blk = mc_block(d.P);
    if (blk.s < 0) return;
    const int s = blk.s, tid = threadIdx.x;
    const bool tail = training && d.stats2 != nullptr;
    for (int i = tid; i < MC_C * MC_CI; i += MC_PX) s_w[i] = d.w2[s][i];
    if (tid < MC_CI) {
        float mean, var;
        if (training) {
            const double cnt = (double)d.P[s] * (d.count_scale > 0.f ? (double)d.count_scale : 1.0);
            const double m = (double)d.stats1[s * 2 * MC_CI + tid] / cnt;
            double v = (double)d.stats1[s * 2 * MC_CI + MC_CI + tid] / cnt - m * m;
            if (v < 0.0) v = 0.0;
            mean = (float)m;
            var = (float)v;
            if (blk.p0 == 0) {          // one workgroup per scale owns the running statistics
                const double unbiased = cnt > 1.0 ? v * cnt / (cnt - 1.0) : v;
                d.running_mean[s][tid] = (1.f - d.momentum) * d.running_mean[s][tid] + d.momentum * mean;
                d.running_var[s][tid] = (1.f - d.momentum) * d.running_var[s][tid] + d.momentum * (float)unbiased;
            }
        } else {
            mean = d.running_mean[s][tid];
            var = d.running_var[s][tid];
        }
        const float invstd = (float)(1.0 / sqrt((double)var + (double)d.eps));
        const float sc = d.gamma[s][tid] * invstd, sh = d.beta[s][tid] - mean * sc;
        s_bn[tid] = sc;
        s_bn[MC_CI + tid] = sh;
        if (blk.p0 == 0 && d.bn1[s]) {
            d.bn1[s][tid] = sc;
            d.bn1[s][MC_CI + tid] = sh;
            d.bn1[s][2 * MC_CI + tid] = mean;
            d.bn1[s][3 * MC_CI + tid] = invstd;
        }
    }
    __syncthreads();
    const int p = blk.p0 + tid;
    const bool ok = p < d.P[s];
    if (!ok && !tail) return;
    float mid[MC_CI];
    const float4* z4 = reinterpret_cast<const float4*>(d.z1[s] + (long)(ok ? p : 0) * MC_CI);
#pragma unroll
    for (int q = 0; q < MC_CI / 4; ++q) {
        const float4 v = z4[q];
        mid[4 * q] = fmaxf(v.x * s_bn[4 * q] + s_bn[MC_CI + 4 * q], 0.f);
        mid[4 * q + 1] = fmaxf(v.y * s_bn[4 * q + 1] + s_bn[MC_CI + 4 * q + 1], 0.f);
        mid[4 * q + 2] = fmaxf(v.z * s_bn[4 * q + 2] + s_bn[MC_CI + 4 * q + 2], 0.f);
        mid[4 * q + 3] = fmaxf(v.w * s_bn[4 * q + 3] + s_bn[MC_CI + 4 * q + 3], 0.f);
    }
    float4* o4 = reinterpret_cast<float4*>(d.z2[s] + (long)p * MC_C);
#pragma unroll 2
    for (int q = 0; q < MC_C / 4; ++q) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = 4 * q + j;
            float a = d.b2[s] ? d.b2[s][co] : 0.f;
#pragma unroll
            for (int ci = 0; ci < MC_CI; ++ci) a = fmaf(mid[ci], s_w[co * MC_CI + ci], a);
            o[j] = a;
        }
        if (ok) o4[q] = make_float4(o[0], o[1], o[2], o[3]);
        if (tail) {                       // sums of z2 for the trailing BatchNorm (whole waves take part)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = ok ? o[j] : 0.f;
                const float a = mc_wave_sum(v), b = mc_wave_sum(v * v);
                if ((tid & 63) == 0) {
                    s_st[tid >> 6][4 * q + j] = a;
                    s_st[tid >> 6][MC_C + 4 * q + j] = b;
                }
            }
        }
    }
    if (!tail) return;
    __syncthreads();
    if (tid < 2 * MC_C) {
        const float t = (s_st[0][tid] + s_st[1][tid]) + (s_st[2][tid] + s_st[3][tid]);
        if (part) part[(long)blockIdx.x * 2 * MC_C + tid] = t;
        else atomicAdd(d.stats2 + s * 2 * MC_C + tid, t);
    }
}

// ---- finalize of the trailing BatchNorms: one workgroup per scale, thread = channel ----------------------
__global__ void __launch_bounds__(MC_C) mfaf_ctx_fin2_kernel(ledn_mfafctx_desc d) {
    const int s = blockIdx.x, c = threadIdx.x;
    const double cnt = (double)d.P[s] * (d.count_scale > 0.f ? (double)d.count_scale : 1.0);
    const double m = (double)d.stats2[s * 2 * MC_C + c] / cnt;
    double v = (double)d.stats2[s * 2 * MC_C + MC_C + c] / cnt - m * m;
    if (v < 0.0) v = 0.0;
    const float invstd = (float)(1.0 / sqrt(v + (double)d.eps));
    const float sc = d.gamma2[s][c] * invstd;
    d.bn2[s][c] = sc;
    d.bn2[s][MC_C + c] = d.beta2[s][c] - (float)m * sc;
    d.bn2[s][2 * MC_C + c] = (float)m;
    d.bn2[s][3 * MC_C + c] = invstd;
    const double unbiased = cnt > 1.0 ? v * cnt / (cnt - 1.0) : v;
    d.running_mean2[s][c] = (1.f - d.momentum) * d.running_mean2[s][c] + d.momentum * (float)m;
    d.running_var2[s][c] = (1.f - d.momentum) * d.running_var2[s][c] + d.momentum * (float)unbiased;
}

// ---- backward of the trailing BatchNorms, reduce half: sums of dy and dy * xhat2 per scale ---------------
__global__ void __launch_bounds__(MC_PX) mfaf_ctx_bwdT_kernel(ledn_mfafctx_bwd_desc d, float* part) {
    __shared__ float s_bn[2 * MC_C];
    __shared__ float s_st[4][2 * MC_C];
    const McBlock blk = mc_block(d.P);
    if (blk.s < 0) return;
    const int s = blk.s, tid = threadIdx.x;
    if (tid < 2 * MC_C) s_bn[tid] = d.bn2[s][2 * MC_C + tid];        // mean | invstd
    __syncthreads();
    const int p = blk.p0 + tid;
    const bool ok = p < d.P[s];
    const float4* dy4 = reinterpret_cast<const float4*>(d.dz2[s] + (long)(ok ? p : 0) * MC_C);
    const float4* z4 = reinterpret_cast<const float4*>(d.z2[s] + (long)(ok ? p : 0) * MC_C);
#pragma unroll 2
    for (int q = 0; q < MC_C / 4; ++q) {
        const float4 gv = dy4[q], zv = z4[q];
        const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, za[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            const float g = ok ? ga[j] : 0.f;
            const float xh = (za[j] - s_bn[c]) * s_bn[MC_C + c];
            const float a = mc_wave_sum(g), b = mc_wave_sum(g * xh);
            if ((tid & 63) == 0) {
                s_st[tid >> 6][c] = a;
                s_st[tid >> 6][MC_C + c] = b;
            }
        }
    }
    __syncthreads();
    if (tid < 2 * MC_C) {
        const float t = (s_st[0][tid] + s_st[1][tid]) + (s_st[2][tid] + s_st[3][tid]);
        if (part) part[(long)blockIdx.x * 2 * MC_C + tid] = t;
        else atomicAdd(d.sums2 + s * 2 * MC_C + tid, t);
    }
}

// ---- backward 2: dmid = W2^T dz2, g = dmid * relu'(bn(z1)) (stored), per-scale sums of g and g * xhat ------
__global__ void __launch_bounds__(MC_PX) mfaf_ctx_bwd2_kernel(ledn_mfafctx_bwd_desc d, float* part) {
    __shared__ float s_w[MC_C * MC_CI];
    __shared__ float s_bn[4 * MC_CI];
    __shared__ float s_red[4][2 * MC_CI];
    __shared__ float s_t[5 * MC_C];        // trailing BatchNorm: scale, mean, invstd, sum dy / n, sum dy xhat / n
    const McBlock blk = mc_block(d.P);
    if (blk.s < 0) return;
    const int s = blk.s, tid = threadIdx.x;
    const bool tail = d.sums2 != nullptr;
    for (int i = tid; i < MC_C * MC_CI; i += MC_PX) s_w[i] = d.w2[s][i];
    if (tid < 4 * MC_CI) s_bn[tid] = d.bn1[s][tid];
    if (tail && tid < MC_C) {
        const float inv_n = 1.f / ((float)d.P[s] * (d.count_scale > 0.f ? d.count_scale : 1.f));
        const float s1 = d.sums2[s * 2 * MC_C + tid], s2 = d.sums2[s * 2 * MC_C + MC_C + tid];
        const float* loc = d.sums2_local ? d.sums2_local : d.sums2;     // this rank's own sums = its d beta / d gamma
        s_t[tid] = d.bn2[s][tid];
        s_t[MC_C + tid] = d.bn2[s][2 * MC_C + tid];
        s_t[2 * MC_C + tid] = d.bn2[s][3 * MC_C + tid];
        s_t[3 * MC_C + tid] = s1 * inv_n;
        s_t[4 * MC_C + tid] = s2 * inv_n;
        if (blk.p0 == 0) {          // (one workgroup per scale: the only adder of these entries in this launch)
            d.dbeta2[s][tid] += loc[s * 2 * MC_C + tid];
            d.dgamma2[s][tid] += loc[s * 2 * MC_C + MC_C + tid];
        }
    }
    __syncthreads();
    const int p = blk.p0 + tid;
    const bool ok = p < d.P[s];
    float g[MC_CI], xh[MC_CI];
#pragma unroll
    for (int ci = 0; ci < MC_CI; ++ci) g[ci] = xh[ci] = 0.f;
    if (ok) {
        const float4* dz4 = reinterpret_cast<const float4*>(d.dz2[s] + (long)p * MC_C);
#pragma unroll 2
        for (int q = 0; q < MC_C / 4; ++q) {
            float4 v = dz4[q];
            if (tail) {       // dz2 = scale2 * (dy - mean(dy) - xhat2 * mean(dy xhat2)), kept for the weight gradient
                const float4 zv = reinterpret_cast<const float4*>(d.z2[s] + (long)p * MC_C)[q];
                const float za[4] = {zv.x, zv.y, zv.z, zv.w};
                float dv2[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * q + j;
                    const float xh2 = (za[j] - s_t[MC_C + c]) * s_t[2 * MC_C + c];
                    dv2[j] = s_t[c] * (dv2[j] - s_t[3 * MC_C + c] - xh2 * s_t[4 * MC_C + c]);
                }
                v = make_float4(dv2[0], dv2[1], dv2[2], dv2[3]);
                reinterpret_cast<float4*>(d.dz2s[s] + (long)p * MC_C)[q] = v;
            }
            const float dv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ci = 0; ci < MC_CI; ++ci) g[ci] = fmaf(dv[j], s_w[(4 * q + j) * MC_CI + ci], g[ci]);
        }
        const float4* z4 = reinterpret_cast<const float4*>(d.z1[s] + (long)p * MC_CI);
#pragma unroll
        for (int q = 0; q < MC_CI / 4; ++q) {
            const float4 v = z4[q];
            const float zv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = 4 * q + j;
                const float y = zv[j] * s_bn[ci] + s_bn[MC_CI + ci];
                g[ci] = y > 0.f ? g[ci] : 0.f;
                xh[ci] = (zv[j] - s_bn[2 * MC_CI + ci]) * s_bn[3 * MC_CI + ci];
            }
        }
        float4* g4 = reinterpret_cast<float4*>(d.g[s] + (long)p * MC_CI);
#pragma unroll
        for (int q = 0; q < MC_CI / 4; ++q) g4[q] = make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
    }
    const int wv = tid >> 6;
#pragma unroll
    for (int ci = 0; ci < MC_CI; ++ci) {
        const float a = mc_wave_sum(g[ci]), b = mc_wave_sum(g[ci] * xh[ci]);
        if ((tid & 63) == 0) {
            s_red[wv][ci] = a;
            s_red[wv][MC_CI + ci] = b;
        }
    }
    __syncthreads();
    if (tid < 2 * MC_CI) {
        const float t = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
        if (part) part[(long)blockIdx.x * 2 * MC_CI + tid] = t;
        else atomicAdd(d.sums + s * 2 * MC_CI + tid, t);
    }
}

// ---- backward 1: dz1 = BatchNorm-backward(g) (written over g), dpooled = W1^T dz1, d gamma / d beta -----------
__global__ void __launch_bounds__(MC_PX) mfaf_ctx_bwd1_kernel(ledn_mfafctx_bwd_desc d) {
    __shared__ float s_w[MC_CI * MC_C];
    __shared__ float s_bn[4 * MC_CI];
    __shared__ float s_sum[2 * MC_CI];
    const McBlock blk = mc_block(d.P);
    if (blk.s < 0) return;
    const int s = blk.s, tid = threadIdx.x;
    for (int i = tid; i < MC_CI * MC_C; i += MC_PX) s_w[i] = d.w1[s][i];
    if (tid < 4 * MC_CI) s_bn[tid] = d.bn1[s][tid];
    if (tid < 2 * MC_CI) s_sum[tid] = d.sums[s * 2 * MC_CI + tid];
    __syncthreads();
    if (blk.p0 == 0 && tid < MC_CI) {      // parameter gradients of the BatchNorm: this rank's own sums
        const float* loc = d.sums_local ? d.sums_local : d.sums;
        d.dbeta[s][tid] += loc[s * 2 * MC_CI + tid];          // (the scale's first workgroup: the only adder in this launch)
        d.dgamma[s][tid] += loc[s * 2 * MC_CI + MC_CI + tid];
    }
    const int p = blk.p0 + tid;
    if (p >= d.P[s]) return;
    const float inv_n = 1.f / ((float)d.P[s] * (d.count_scale > 0.f ? d.count_scale : 1.f));
    float dz[MC_CI];
    float4* g4 = reinterpret_cast<float4*>(d.g[s] + (long)p * MC_CI);
    const float4* z4 = reinterpret_cast<const float4*>(d.z1[s] + (long)p * MC_CI);
#pragma unroll
    for (int q = 0; q < MC_CI / 4; ++q) {
        const float4 gv = g4[q], zv = z4[q];
        const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, za[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = 4 * q + j;
            const float xh = (za[j] - s_bn[2 * MC_CI + ci]) * s_bn[3 * MC_CI + ci];
            // scale = gamma * invstd
            dz[ci] = s_bn[ci] * (ga[j] - s_sum[ci] * inv_n - xh * s_sum[MC_CI + ci] * inv_n);
        }
        g4[q] = make_float4(dz[4 * q], dz[4 * q + 1], dz[4 * q + 2], dz[4 * q + 3]);
    }
    float4* o4 = reinterpret_cast<float4*>(d.dpooled[s] + (long)p * MC_C);
#pragma unroll 4
    for (int q = 0; q < MC_C / 4; ++q) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < MC_CI; ++ci) {
            const float* w = s_w + ci * MC_C + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(dz[ci], w[j], o[j]);
        }
        o4[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- weight gradients: dW[a][b] += sum_p A[p][a] * B[p][b]  (A: [P][CA], B: [P][CB], CA * CB = 1024), db[a] += sum_p A[p][a].
// MODE 0: A = dz2 (C), B = mid = relu(bn(z1)) recomputed (Ci)   -> dw2 [C][Ci], db2
// MODE 1: A = dz1 (held in g, Ci), B = pooled (C)               -> dw1 [Ci][C], db1
// thread = four consecutive b of one a; the workgroup's <= 256 pixels are staged in LDS in two halves.
template <int MODE>
__global__ void __launch_bounds__(MC_PX) mfaf_ctx_wgrad_kernel(ledn_mfafctx_bwd_desc d, float* part) {
    constexpr int CA = MODE == 0 ? MC_C : MC_CI, CB = MODE == 0 ? MC_CI : MC_C, HALF = 128;
    __shared__ float s_a[HALF * CA];
    __shared__ float s_b[HALF * CB];
    __shared__ float s_bn[2 * MC_CI];
    const McBlock blk = mc_block(d.P);
    if (blk.s < 0) return;
    const int s = blk.s, tid = threadIdx.x;
    if (MODE == 0 && tid < 2 * MC_CI) s_bn[tid] = d.bn1[s][tid];
    const float* A = MODE == 0 ? (d.sums2 ? d.dz2s[s] : d.dz2[s]) : d.g[s];
    const float* B = MODE == 0 ? d.z1[s] : d.pooled[s];
    const int a = (tid * 4) / CB, b0 = (tid * 4) % CB;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, accb = 0.f;
    const int np = min(MC_PX, d.P[s] - blk.p0);
    for (int h0 = 0; h0 < np; h0 += HALF) {
        const int nh = min(HALF, np - h0);
        __syncthreads();
        for (int i = tid; i < nh * CA; i += MC_PX) s_a[i] = A[(long)(blk.p0 + h0) * CA + i];
        for (int i = tid; i < nh * CB; i += MC_PX) {
            float v = B[(long)(blk.p0 + h0) * CB + i];
            if (MODE == 0) {
                const int ci = i % MC_CI;
                v = fmaxf(v * s_bn[ci] + s_bn[MC_CI + ci], 0.f);
            }
            s_b[i] = v;
        }
        __syncthreads();
        for (int q = 0; q < nh; ++q) {
            const float av = s_a[q * CA + a];
            const float4 bv = *reinterpret_cast<const float4*>(s_b + q * CB + b0);
            acc[0] = fmaf(av, bv.x, acc[0]);
            acc[1] = fmaf(av, bv.y, acc[1]);
            acc[2] = fmaf(av, bv.z, acc[2]);
            acc[3] = fmaf(av, bv.w, acc[3]);
            if (b0 == 0) accb += av;
        }
    }
    float* dw = MODE == 0 ? d.dw2[s] : d.dw1[s];
    float* db = MODE == 0 ? d.db2[s] : d.db1[s];
    if (part) {         // row = [CA x CB filter gradient | CA bias gradient]
        float* row = part + (long)blockIdx.x * (CA * CB + CA);
#pragma unroll
        for (int j = 0; j < 4; ++j) row[a * CB + b0 + j] = acc[j];
        if (b0 == 0) row[CA * CB + a] = accb;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(dw + (long)a * CB + b0 + j, acc[j]);
    if (b0 == 0 && db) atomicAdd(db + a, accb);
}

static int mc_check(const int* P, int C, int Ci) {
    LEDN_REQUIRE(C == MC_C && Ci == MC_CI);
    for (int k = 0; k < 4; ++k) LEDN_REQUIRE(P[k] > 0 && P[k] <= (1 << 24));
    return LEDN_OK;
}

int mfaf_ctx_fwd_impl(const ledn_mfafctx_desc& d, int training, hipStream_t s) {
    int rc = mc_check(d.P, d.C, d.Ci);
    if (rc != LEDN_OK) return rc;
    for (int k = 0; k < 4; ++k) {
        LEDN_REQUIRE(d.pooled[k] && d.z1[k] && d.z2[k] && d.w1[k] && d.w2[k] && d.gamma[k] && d.beta[k]);
        LEDN_REQUIRE(d.running_mean[k] && d.running_var[k]);
        LEDN_REQUIRE(!training || d.bn1[k]);
    }
    LEDN_REQUIRE(!training || d.stats1);
    if (d.stats2) {
        LEDN_REQUIRE(training);
        for (int k = 0; k < 4; ++k)
            LEDN_REQUIRE(d.gamma2[k] && d.beta2[k] && d.running_mean2[k] && d.running_var2[k] && d.bn2[k]);
    }
    const int nblk = mc_blocks(d.P);
    const dim3 grid((unsigned)nblk);
    const int ph = d.phase ? d.phase : 7;
    float* part = (training && det()) ? ws_take((long)nblk * 2 * MC_C) : nullptr;
    if (training && det() && !part) return LEDN_EINVAL;
    if (ph & 1) {
        LEDN_LAUNCH(mfaf_ctx_fwd1_kernel, grid, dim3(MC_PX), 0, s, d, training, part);
        if (part) {
            rc = mc_finish(part, 2 * MC_CI, d.stats1, d.stats1 + 2 * MC_CI, d.stats1 + 4 * MC_CI, d.stats1 + 6 * MC_CI, d.P, s);
            if (rc != LEDN_OK) return rc;
        }
    }
    if (ph & 2) {
        LEDN_LAUNCH(mfaf_ctx_fwd2_kernel, grid, dim3(MC_PX), 0, s, d, training, d.stats2 ? part : nullptr);
        if (part && d.stats2) {
            rc = mc_finish(part, 2 * MC_C, d.stats2, d.stats2 + 2 * MC_C, d.stats2 + 4 * MC_C, d.stats2 + 6 * MC_C, d.P, s);
            if (rc != LEDN_OK) return rc;
        }
    }
    if ((ph & 4) && d.stats2) LEDN_LAUNCH(mfaf_ctx_fin2_kernel, dim3(4), dim3(MC_C), 0, s, d);
    return check_launch();
}

int mfaf_ctx_bwd_impl(const ledn_mfafctx_bwd_desc& d, hipStream_t s) {
    int rc = mc_check(d.P, d.C, d.Ci);
    if (rc != LEDN_OK) return rc;
    for (int k = 0; k < 4; ++k) {
        LEDN_REQUIRE(d.pooled[k] && d.z1[k] && d.dz2[k] && d.w1[k] && d.w2[k] && d.bn1[k] && d.g[k] && d.dpooled[k]);
        LEDN_REQUIRE(d.dw1[k] && d.dw2[k] && d.dgamma[k] && d.dbeta[k]);
    }
    LEDN_REQUIRE(d.sums);
    const int nblk = mc_blocks(d.P);
    const dim3 grid((unsigned)nblk);
    const int ph = d.phase ? d.phase : 7;
    constexpr int KW = MC_C * MC_CI + MC_C;         // widest partial row: a filter gradient + its bias gradient
    float* part = det() ? ws_take((long)nblk * KW) : nullptr;
    if (det() && !part) return LEDN_EINVAL;
    if (d.sums2) {
        for (int k = 0; k < 4; ++k) LEDN_REQUIRE(d.z2[k] && d.bn2[k] && d.dz2s[k] && d.dgamma2[k] && d.dbeta2[k]);
        if (ph & 1) {
            ledn_mfafctx_bwd_desc dl = d;            // this launch accumulates the LOCAL sums
            if (d.sums2_local) dl.sums2 = const_cast<float*>(d.sums2_local);
            LEDN_LAUNCH(mfaf_ctx_bwdT_kernel, grid, dim3(MC_PX), 0, s, dl, part);
            if (part) {
                rc = mc_finish(part, 2 * MC_C, dl.sums2, dl.sums2 + 2 * MC_C, dl.sums2 + 4 * MC_C, dl.sums2 + 6 * MC_C, d.P, s);
                if (rc != LEDN_OK) return rc;
            }
        }
    }
    if (ph & 2) {
        ledn_mfafctx_bwd_desc dl = d;                // reads the global sums2, accumulates the LOCAL sums
        if (d.sums_local) dl.sums = const_cast<float*>(d.sums_local);
        LEDN_LAUNCH(mfaf_ctx_bwd2_kernel, grid, dim3(MC_PX), 0, s, dl, part);
        if (part) {
            rc = mc_finish(part, 2 * MC_CI, dl.sums, dl.sums + 2 * MC_CI, dl.sums + 4 * MC_CI, dl.sums + 6 * MC_CI, d.P, s);
            if (rc != LEDN_OK) return rc;
        }
        LEDN_LAUNCH(mfaf_ctx_wgrad_kernel<0>, grid, dim3(MC_PX), 0, s, d, part);
        if (part) {
            rc = mc_finish_w(part, MC_C * MC_CI, MC_C, d.dw2, d.db2, d.P, s);
            if (rc != LEDN_OK) return rc;
        }
    }
    if (ph & 4) {
        LEDN_LAUNCH(mfaf_ctx_bwd1_kernel, grid, dim3(MC_PX), 0, s, d);
        LEDN_LAUNCH(mfaf_ctx_wgrad_kernel<1>, grid, dim3(MC_PX), 0, s, d, part);
        if (part) {
            rc = mc_finish_w(part, MC_C * MC_CI, MC_CI, d.dw1, d.db1, d.P, s);
            if (rc != LEDN_OK) return rc;
        }
    }
    return check_launch();
}

}  // namespace ledn
