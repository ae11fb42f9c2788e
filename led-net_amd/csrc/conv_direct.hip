// conv_direct.hip -- generic dense / grouped convolution on the vector ALUs.
//
// The universal path: any Cin/Cout (3->32 stem, 32->2 heads, 64->1 / 1->64
// SEAM, 16-channel MFAF MLPs, grouped 1x1 of SESP), forward and data-gradient
// (transposed), f32 or bf16 activations, f32 accumulation.  The MFMA
// implicit-GEMM engine (conv_mfma.hip) takes the heavy 3x3 / 1x1 layers; this
// kernel is also its on-device cross-check.
//
// Mapping: one thread = one output pixel x CO_T output channels.  blockIdx.y
// selects the channel group, so every weight address is wave-uniform and the
// compiler keeps weights on the scalar path (s_load + SGPR FMA operands); the
// per-lane traffic is the NHWC input vector (16 B / lane for VEC=4).
// HBM-side: x is read once per tap from L1/L2 (3x3 halo reuse), y written once.
#include "ledn_rt.h"

namespace ledn {

template <typename TX, typename TY, int CO_T, int VEC>
__global__ void __launch_bounds__(256) conv_direct_kernel(ledn_conv_desc d, float* part) {
    // weights of this workgroup's CO_T output channels, one ci-chunk at a time:
    // s_w[(tap * cib + ci) * CO_T + j]; every lane reads the same address (LDS broadcast)
    constexpr int CIB = 32;
    __shared__ __attribute__((aligned(16))) float s_w[9 * CIB * CO_T];
    __shared__ float s_stat[4][2 * CO_T];
    const long npix = (long)d.N * d.Ho * d.Wo;
    const int cog = d.Cout / d.groups, cig = d.Cin / d.groups;
    const int co0 = blockIdx.y * CO_T;
    const int g = co0 / cog;
    const int ci0 = g * cig;
    const int col0 = co0 - g * cog;  // channel index inside the group
    const int taps = d.KH * d.KW;
    const bool one_chunk = cig <= CIB;
    const TX* x = reinterpret_cast<const TX*>(d.x);
    const TX* xadd = reinterpret_cast<const TX*>(d.xadd);
    float st1[CO_T], st2[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) st1[j] = st2[j] = 0.f;

    // grid-stride over 256-pixel batches: a bounded grid keeps the per-channel statistics
    // atomics to one per workgroup (same-address atomics serialise at ~0.3 us each)
    for (long base = (long)blockIdx.x * blockDim.x; base < npix; base += (long)gridDim.x * blockDim.x) {
        const long pix = base + threadIdx.x;
        const bool active = pix < npix;
        float acc[CO_T];
#pragma unroll
        for (int j = 0; j < CO_T; ++j) acc[j] = 0.f;
        int wo = 0, ho = 0, n = 0;
        if (active) {
            wo = (int)(pix % d.Wo);
            ho = (int)((pix / d.Wo) % d.Ho);
            n = (int)(pix / ((long)d.Wo * d.Ho));
        }
        for (int cb = 0; cb < cig; cb += CIB) {
            const int cib = min(CIB, cig - cb);
            if (!one_chunk || base == (long)blockIdx.x * blockDim.x) {
                if (!one_chunk) __syncthreads();   // previous chunk's readers are done
                // forward: W(co_global, ci_local); transposed: W(ci_global_fwd, co_local_fwd)
                for (int e = threadIdx.x; e < taps * cib * CO_T; e += blockDim.x) {
                    const int j = e % CO_T, ci = (e / CO_T) % cib, tap = e / (CO_T * cib);
                    const long wb = d.transposed ? (long)(col0 + j) * d.ws_co + (long)ci0 * d.ws_ci
                                                 : (long)(co0 + j) * d.ws_co;
                    s_w[e] = d.w[wb + (long)(cb + ci) * d.ws_ci + (long)tap * d.ws_tap];
                }
                __syncthreads();
            }
            if (!active) continue;
            for (int kh = 0; kh < d.KH; ++kh) {
                int hi;
                if (!d.transposed) {
                    hi = ho * d.stride - d.pad + kh * d.dil;
                } else {
                    const int t = ho + d.pad - kh * d.dil;
                    if (t < 0 || (t % d.stride) != 0) continue;
                    hi = t / d.stride;
                }
                if (hi < 0 || hi >= d.H) continue;
                for (int kw = 0; kw < d.KW; ++kw) {
                    int wi;
                    if (!d.transposed) {
                        wi = wo * d.stride - d.pad + kw * d.dil;
                    } else {
                        const int t = wo + d.pad - kw * d.dil;
                        if (t < 0 || (t % d.stride) != 0) continue;
                        wi = t / d.stride;
                    }
                    if (wi < 0 || wi >= d.W) continue;
                    const long xoff = (((long)n * d.H + hi) * d.W + wi) * d.Cin + ci0 + cb;
                    const float* wt = s_w + (kh * d.KW + kw) * cib * CO_T;
                    for (int ci = 0; ci < cib; ci += VEC) {
                        float xv[VEC];
                        ldv<VEC>(x + xoff + ci, xv);
                        if (xadd) {
                            float xa[VEC];
                            ldv<VEC>(xadd + xoff + ci, xa);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) xv[v] += xa[v];
                        }
                        if (d.in_scale) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                xv[v] = xv[v] * d.in_scale[ci0 + cb + ci + v] + d.in_shift[ci0 + cb + ci + v];
                        }
                        if (d.in_act == LEDN_ACT_RELU) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v) xv[v] = fmaxf(xv[v], 0.f);
                        } else if (d.in_act == LEDN_ACT_PRELU) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                xv[v] = xv[v] > 0.f ? xv[v] : xv[v] * d.in_slope[ci0 + cb + ci + v];
                        }
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const float* wr = wt + (ci + v) * CO_T;
#pragma unroll
                            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(xv[v], wr[j], acc[j]);
                        }
                    }
                }
            }
        }
        if (!active) continue;
        // ---- epilogue: affine, statistics, residual, activation, store
        float v[CO_T];
#pragma unroll
        for (int j = 0; j < CO_T; ++j) {
            const int co = co0 + j;
            const float s = d.out_scale ? d.out_scale[co] : 1.f;
            const float b = d.out_shift ? d.out_shift[co] : 0.f;
            v[j] = acc[j] * s + b;
            st1[j] += v[j];
            st2[j] = fmaf(v[j], v[j], st2[j]);
        }
        const long yoff = pix * d.Cout + co0;
        if (d.res_mode != LEDN_RES_NONE) {
            const TY* r = reinterpret_cast<const TY*>(d.res) + yoff;
#pragma unroll
            for (int j = 0; j < CO_T; ++j) {
                const float rv = ld(r + j);
                v[j] = d.res_mode == LEDN_RES_ADD ? v[j] + rv : v[j] * rv + rv;
            }
        }
        if (d.act_out != LEDN_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < CO_T; ++j)
                v[j] = act_apply(d.act_out, v[j], d.slope ? d.slope[co0 + j] : 0.f);
        }
        TY* y = reinterpret_cast<TY*>(d.y) + yoff;
        if constexpr (CO_T % 4 == 0) {
#pragma unroll
            for (int j = 0; j < CO_T; j += 4) st4(y + j, v + j);
        } else {
#pragma unroll
            for (int j = 0; j < CO_T; ++j) st(y + j, v[j]);
        }
    }
    if (d.stat_sum) {   // wave reduction -> 4 partials in LDS -> one atomic per channel per workgroup
#pragma unroll
        for (int j = 0; j < CO_T; ++j) {
            const float a1 = wave_sum(st1[j]), a2 = wave_sum(st2[j]);
            if (lane_id() == 0) {
                s_stat[threadIdx.x >> 6][j] = a1;
                s_stat[threadIdx.x >> 6][CO_T + j] = a2;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * CO_T) {
            const float t = s_stat[0][threadIdx.x] + s_stat[1][threadIdx.x] + s_stat[2][threadIdx.x] +
                            s_stat[3][threadIdx.x];
            if (part) {     // one row [sum | sum of squares] per workgroup column, added up in row order by finish_partials
                part[(long)blockIdx.x * 2 * d.Cout + (threadIdx.x < CO_T ? co0 + threadIdx.x : d.Cout + co0 + threadIdx.x - CO_T)] = t;
            } else {
                float* dst = threadIdx.x < CO_T ? d.stat_sum + co0 + threadIdx.x
                                                : d.stat_sqsum + co0 + threadIdx.x - CO_T;
                atomicAdd(dst, t);
            }
        }
    }
}

template <typename TX, typename TY, int CO_T>
static int launch_vec(const ledn_conv_desc& d, hipStream_t s) {
    const long npix = (long)d.N * d.Ho * d.Wo;
    long gx = cdiv(npix, 256);
    const long cap = 2048 / (d.Cout / CO_T) > 64 ? 2048 / (d.Cout / CO_T) : 64;   // bounded grid (see kernel)
    if (gx > cap) gx = cap;
    const dim3 grid((unsigned)gx, (unsigned)(d.Cout / CO_T));
    const int cig = d.Cin / d.groups;
    // statistics: deterministic mode (or > 16 workgroup columns) -> partial rows + ordered summing launch
    float* part = (d.stat_sum && (gx > 16 || det())) ? ws_take(gx * 2 * d.Cout) : nullptr;
    if (d.stat_sum && det() && !part) return LEDN_EINVAL;
    if (cig % 4 == 0 && d.ws_ci != 0)
        LEDN_LAUNCH((conv_direct_kernel<TX, TY, CO_T, 4>), grid, dim3(256), 0, s, d, part);
    else
        LEDN_LAUNCH((conv_direct_kernel<TX, TY, CO_T, 1>), grid, dim3(256), 0, s, d, part);
    if (part) return finish_partials(part, (int)gx, d.Cout, 2, d.stat_sum, d.stat_sqsum, nullptr, s);
    return check_launch();
}

template <typename TX, typename TY>
static int launch_cot(const ledn_conv_desc& d, hipStream_t s) {
    const int cog = d.Cout / d.groups;
    if (cog % 16 == 0) return launch_vec<TX, TY, 16>(d, s);
    if (cog % 8 == 0) return launch_vec<TX, TY, 8>(d, s);
    if (cog % 4 == 0) return launch_vec<TX, TY, 4>(d, s);
    if (cog % 2 == 0) return launch_vec<TX, TY, 2>(d, s);
    return launch_vec<TX, TY, 1>(d, s);
}

// ---------------------------------------------------------------------------
// Narrow input (Cin <= 2) -> wide output, stride 1, plain epilogue: the data gradient of the
// 32->2 heads (dz has 2 channels, dx has 32: led_head.py:47-48) and of 64->1 / 64->2 layers.
// thread = (pixel, 8 output channels): the lanes of a pixel cover its whole channel row with
// 16-byte stores (the generic kernel wrote 8-byte pieces 64 bytes apart: 0.35 ms for the 268 MB
// dx of head_x1 at 16 x 512 x 512); the taps x Cin input scalars are unconditional loads, the
// weights sit in LDS as [tap][ci][co] (two ds_read_b128 per tap and input channel).
// ---------------------------------------------------------------------------
template <typename TX, typename TY, int CIN, int K, bool TR>
__global__ void __launch_bounds__(256, 2) conv_narrowin_kernel(ledn_conv_desc d) {
    constexpr int PX = CIN <= 2 ? 4 : 2;                    // output pixels (along W) per thread
    __shared__ __attribute__((aligned(16))) float s_w[K * K * CIN * 128];
    for (int e = threadIdx.x; e < K * K * CIN * d.Cout; e += blockDim.x) {
        const int co = e % d.Cout, ci = (e / d.Cout) % CIN, tap = e / (d.Cout * CIN);
        s_w[e] = ci < d.Cin ? d.w[(long)co * d.ws_co + (long)ci * d.ws_ci + (long)tap * d.ws_tap] : 0.f;
    }
    __syncthreads();
    const int cgn = d.Cout / 8;
    const int wq = (d.Wo + PX - 1) / PX;
    const long total = (long)d.N * d.Ho * wq * cgn;
    // input patch of the thread: rows ho - lo .. ho - lo + K-1, columns wo0 - lo .. wo0 - lo + K+PX-2;
    // tap (kh, kw) reads patch row kh (forward) or K-1-kh (data gradient: hi = ho + pad - kh)
    const int lo = TR ? (K - 1) - d.pad : d.pad;
    const TX* x = reinterpret_cast<const TX*>(d.x);
    TY* y = reinterpret_cast<TY*>(d.y);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const NhwcIdx ix_ = nhwc_split(idx, cgn, wq, d.Ho);      // (32-bit divisions: ledn_rt.h)
        const int cg = ix_.cv, wo0 = ix_.x * PX, ho = ix_.y, n = ix_.n;
        float xv[K][K + PX - 1][CIN];
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int c = 0; c < K + PX - 1; ++c) {
                const int hi = ho - lo + r, wi = wo0 - lo + c;
                const bool valid = hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
                const long xoff = valid ? (((long)n * d.H + hi) * d.W + wi) * d.Cin : 0L;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float v = ld(x + xoff + (ci < d.Cin ? ci : 0));
                    xv[r][c][ci] = (valid && ci < d.Cin) ? v : 0.f;
                }
            }
        float acc[PX][8];
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    float w8[8];
                    ld8(s_w + ((kh * K + kw) * CIN + ci) * d.Cout + cg * 8, w8);
#pragma unroll
                    for (int p = 0; p < PX; ++p) {
                        const float xs = TR ? xv[K - 1 - kh][K - 1 - kw + p][ci] : xv[kh][kw + p][ci];
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[p][j] = fmaf(xs, w8[j], acc[p][j]);
                    }
                }
            }
#pragma unroll
        for (int p = 0; p < PX; ++p)
            if (wo0 + p < d.Wo) st8(y + (((long)n * d.Ho + ho) * d.Wo + wo0 + p) * d.Cout + cg * 8, acc[p]);
    }
}

static bool narrowin_ok(const ledn_conv_desc& d) {
    if (d.Cin > 2 || d.Cout % 8 || d.Cout > 128 || d.groups != 1 || d.stride != 1 || d.dil != 1) return false;
    if (d.xadd || d.in_scale || d.in_act != LEDN_ACT_NONE || d.out_scale || d.out_shift || d.stat_sum) return false;
    if (d.res_mode != LEDN_RES_NONE || d.act_out != LEDN_ACT_NONE) return false;
    return d.KH == d.KW && (d.KH == 1 || d.KH == 3);
}

template <typename TX, typename TY>
static int launch_narrowin(const ledn_conv_desc& d, hipStream_t s) {
    const long total = (long)d.N * d.Ho * cdiv(d.Wo, 4) * (d.Cout / 8);
    long nb = cdiv(total, 256);
    if (nb > 4096) nb = 4096;
    const dim3 grid((unsigned)nb);
#define LEDN_NI(CI, KK)                                                                                   \
    do {                                                                                                  \
        if (d.transposed) LEDN_LAUNCH((conv_narrowin_kernel<TX, TY, CI, KK, true>), grid, dim3(256), 0, s, d); \
        else LEDN_LAUNCH((conv_narrowin_kernel<TX, TY, CI, KK, false>), grid, dim3(256), 0, s, d);        \
    } while (0)
    if (d.Cin == 1 && d.KH == 3) LEDN_NI(1, 3);          // SEAM's conv_2 (1 -> 64): half the loads and FMAs of the 2-channel instance
    else if (d.KH == 3) LEDN_NI(2, 3);
    else LEDN_NI(2, 1);
#undef LEDN_NI
    return check_launch();
}

// ---------------------------------------------------------------------------
// 1x1 convolution with Cout <= 2 on bf16 maps (the classifiers cls_seg / aux_cls_seg 64 -> 2, led_head.py:87-98) at the
// streaming shape: lane = 8 input channels of a pixel (one 16-byte load), the 2 x 8 filter values in registers, the pixel's
// C / 8 lanes add their partial dot products with DPP steps inside the group; the group's first lane stores.  (The
// generic output-tiled kernel: 27 us for the 33.5 MB map at 16 x 128 x 128 x 64 = 1.3 TB/s.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float c11n_group_sum(float v, int cgn) {      // sum over the cgn (power of two <= 16) lanes of a pixel
#ifdef LEDN_CPU_EMU
    for (int m = 1; m < cgn; m <<= 1) v += __shfl_xor(v, m);
    return v;
#else
    const int iv0 = __float_as_int(v);
    if (cgn >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, iv0, 0xB1, 0xf, 0xf, false));                       // quad_perm [1,0,3,2]
    if (cgn >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));         // quad_perm [2,3,0,1]
    if (cgn >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));        // row_half_mirror
    if (cgn >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));       // row_mirror
    return v;
#endif
}
template <typename TY, int UNR>
__global__ void __launch_bounds__(256) conv1x1_narrow_kernel(ledn_conv_desc d) {
    const int cgn = d.Cin >> 3, rows = 256 / cgn;
    const int cg = (int)(threadIdx.x % (unsigned)cgn), r = (int)(threadIdx.x / (unsigned)cgn), c = cg * 8;
    const float* w = reinterpret_cast<const float*>(d.w);
    float w0[8], w1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        w0[i] = w[(long)(c + i) * d.ws_ci];
        w1[i] = d.Cout > 1 ? w[d.ws_co + (long)(c + i) * d.ws_ci] : 0.f;
    }
    const float b0 = d.out_shift ? d.out_shift[0] : 0.f, b1 = (d.out_shift && d.Cout > 1) ? d.out_shift[1] : 0.f;
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
    TY* y = reinterpret_cast<TY*>(d.y);
    const long npix = (long)d.N * d.H * d.W;
    const long p0 = (long)blockIdx.x * (rows * UNR) + r;
    uint4 xr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long p = p0 + (long)u * rows;
        xr[u] = *reinterpret_cast<const uint4*>(x + (p < npix ? p : 0) * d.Cin + c);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long p = p0 + (long)u * rows;
        float xv[8];
        ld8(reinterpret_cast<const bf16_t*>(&xr[u]), xv);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a0 = fmaf(xv[i], w0[i], a0);
            a1 = fmaf(xv[i], w1[i], a1);
        }
        a0 = c11n_group_sum(a0, cgn);           // (every lane of the wave takes part: no early exit above)
        a1 = c11n_group_sum(a1, cgn);
        if (cg == 0 && p < npix) {
            st(y + p * d.Cout, a0 + b0);
            if (d.Cout > 1) st(y + p * d.Cout + 1, a1 + b1);
        }
    }
}

static bool conv1x1_narrow_ok(const ledn_conv_desc& d) {
    if (!(options().stream_fast & 1) || d.dtype_x != LEDN_BF16 || (d.dtype_y != LEDN_BF16 && d.dtype_y != LEDN_F32)) return false;
    if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad != 0 || d.groups != 1 || d.transposed || d.Cout > 2) return false;
    if (d.Cin < 8 || d.Cin > 128 || (d.Cin & (d.Cin - 1))) return false;
    if (d.xadd || d.in_scale || d.in_act != LEDN_ACT_NONE || d.out_scale || d.stat_sum || d.res || d.res_mode != LEDN_RES_NONE ||
        d.act_out != LEDN_ACT_NONE)
        return false;
    return d.ws_tap == 1 && (long)d.N * d.H * d.W >= 4096 && (long)d.N * d.H * d.W * d.Cin < (1L << 31);
}

int conv_direct(const ledn_conv_desc& d, hipStream_t s) {
    if (conv1x1_narrow_ok(d)) {
        constexpr int UNR = 4;
        const int rows = 256 / (d.Cin >> 3);
        const dim3 grid((unsigned)cdiv((long)d.N * d.H * d.W, (long)rows * UNR));
        if (d.dtype_y == LEDN_F32) LEDN_LAUNCH((conv1x1_narrow_kernel<float, UNR>), grid, dim3(256), 0, s, d);
        else LEDN_LAUNCH((conv1x1_narrow_kernel<bf16_t, UNR>), grid, dim3(256), 0, s, d);
        return check_launch();
    }
    if (narrowin_ok(d)) {
        if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_F32) return launch_narrowin<float, float>(d, s);
        if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_BF16) return launch_narrowin<bf16_t, bf16_t>(d, s);
        if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_F32) return launch_narrowin<bf16_t, float>(d, s);
        if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_BF16) return launch_narrowin<float, bf16_t>(d, s);
        return LEDN_EINVAL;
    }
    if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_F32) return launch_cot<float, float>(d, s);
    if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_BF16) return launch_cot<bf16_t, bf16_t>(d, s);
    if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_F32) return launch_cot<bf16_t, float>(d, s);
    if (d.dtype_x == LEDN_F32 && d.dtype_y == LEDN_BF16) return launch_cot<float, bf16_t>(d, s);
    return LEDN_EINVAL;
}

int conv_validate(const ledn_conv_desc& d) {
    LEDN_REQUIRE(d.x && d.w && d.y);
    LEDN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Ho > 0 && d.Wo > 0 && d.Cout > 0);
    LEDN_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.dil > 0 && d.pad >= 0 && d.groups > 0);
    LEDN_REQUIRE(d.KH * d.KW <= 9);
    LEDN_REQUIRE(d.Cin % d.groups == 0 && d.Cout % d.groups == 0);
    LEDN_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr));
    LEDN_REQUIRE((d.stat_sum == nullptr) == (d.stat_sqsum == nullptr));
    LEDN_REQUIRE(d.res_mode == LEDN_RES_NONE || d.res != nullptr);
    LEDN_REQUIRE(d.act_out != LEDN_ACT_PRELU || d.slope != nullptr);
    LEDN_REQUIRE(d.in_act == LEDN_ACT_NONE || d.in_act == LEDN_ACT_RELU || (d.in_act == LEDN_ACT_PRELU && d.in_slope));
    const int kh_ext = (d.KH - 1) * d.dil + 1, kw_ext = (d.KW - 1) * d.dil + 1;
    if (!d.transposed) {
        LEDN_REQUIRE(d.Ho == (d.H + 2 * d.pad - kh_ext) / d.stride + 1);
        LEDN_REQUIRE(d.Wo == (d.W + 2 * d.pad - kw_ext) / d.stride + 1);
    } else {  // (H,W) is the forward OUTPUT size, (Ho,Wo) the forward INPUT size
        LEDN_REQUIRE(d.H == (d.Ho + 2 * d.pad - kh_ext) / d.stride + 1);
        LEDN_REQUIRE(d.W == (d.Wo + 2 * d.pad - kw_ext) / d.stride + 1);
    }
    return LEDN_OK;
}

}  // namespace ledn

// ---------------------------------------------------------------------------
// weight gradient (direct): one workgroup = one filter tap x a 64(ci) x 64(co)
// tile of dW x a chunk of output pixels; each thread owns a 4x4 register tile
// and walks the pixel chunk; partial tiles are added to dW with f32 atomics
// (256 contiguous bytes per wave-instruction along co for OIHW^T strides is not
// guaranteed -- this is the correctness path; the MFMA wgrad is the fast one).
// ---------------------------------------------------------------------------
namespace ledn {

constexpr int WG_PC = 32;   // pixels per staged chunk

// natural OIHW strides: the per-workgroup partial tiles can use dW's own linear index
static bool wgrad_natural_strides(const ledn_wgrad_desc& d) {
    const long kk = (long)d.KH * d.KW;
    return d.ws_tap == 1 && d.ws_ci == kk && d.ws_co == (long)(d.Cin / d.groups) * kk;
}

template <typename TX, typename TZ>
__global__ void __launch_bounds__(256) conv_wgrad_direct_kernel(ledn_wgrad_desc d, int pix_per_block,
                                                                int ci_tiles, int co_tiles, float* part,
                                                                long part_stride) {
    // chunk of WG_PC pixels x 64 ci / 64 co staged in LDS by coalesced, mutually independent loads
    // (the first version walked the pixels one by one per thread: 80 us for a 4096-pixel MLP layer)
    __shared__ __attribute__((aligned(16))) float s_x[WG_PC][64];
    __shared__ __attribute__((aligned(16))) float s_z[WG_PC][64];
    const int cog = d.Cout / d.groups, cig = d.Cin / d.groups;
    int t = blockIdx.y;
    const int co_tile = t % co_tiles; t /= co_tiles;
    const int ci_tile = t % ci_tiles; t /= ci_tiles;
    const int g = t;
    const int tap = blockIdx.z;
    const int kh = tap / d.KW, kw = tap % d.KW;
    const int tci = threadIdx.x / 16, tco = threadIdx.x % 16;
    const int cil = ci_tile * 64 + tci * 4;   // local (in-group) ci of this thread's tile
    const int col = co_tile * 64 + tco * 4;   // local co
    const int nci = min(4, cig - cil), nco = min(4, cog - col);
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    const TX* x = reinterpret_cast<const TX*>(d.x);
    const TX* xadd = reinterpret_cast<const TX*>(d.xadd);
    const TZ* dz = reinterpret_cast<const TZ*>(d.dz);
    const long npix = (long)d.N * d.Ho * d.Wo;
    const long p0 = (long)blockIdx.x * pix_per_block;
    const long p1 = min(npix, p0 + pix_per_block);
    for (long chunk = p0; chunk < p1; chunk += WG_PC) {   // workgroup-uniform trip count
        __syncthreads();
        for (int e = threadIdx.x; e < WG_PC * 64; e += 256) {
            const int pl = e >> 6, c = e & 63;
            const long p = chunk + pl;
            float xv = 0.f, zv = 0.f;
            if (p < p1) {
                const int wo = (int)(p % d.Wo);
                const int ho = (int)((p / d.Wo) % d.Ho);
                const int n = (int)(p / ((long)d.Wo * d.Ho));
                const int hi = ho * d.stride - d.pad + kh * d.dil;
                const int wi = wo * d.stride - d.pad + kw * d.dil;
                if (hi >= 0 && hi < d.H && wi >= 0 && wi < d.W) {
                    const int ci = ci_tile * 64 + c, co = co_tile * 64 + c;
                    if (ci < cig) {
                        const long xoff = (((long)n * d.H + hi) * d.W + wi) * d.Cin + g * cig + ci;
                        xv = ld(x + xoff);
                        if (xadd) xv += ld(xadd + xoff);
                        if (d.in_scale) xv = xv * d.in_scale[g * cig + ci] + d.in_shift[g * cig + ci];
                        if (d.in_act == LEDN_ACT_RELU) xv = fmaxf(xv, 0.f);
                        else if (d.in_act == LEDN_ACT_PRELU) xv = xv > 0.f ? xv : xv * d.in_slope[g * cig + ci];
                    }
                    if (co < cog) zv = ld(dz + p * d.Cout + g * cog + co);
                }
            }
            s_x[pl][c] = xv;
            s_z[pl][c] = zv;
        }
        __syncthreads();
#pragma unroll 8
        for (int pl = 0; pl < WG_PC; ++pl) {
            float xv[4], zv[4];
            ld4(&s_x[pl][tci * 4], xv);
            ld4(&s_z[pl][tco * 4], zv);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(xv[a], zv[b], acc[a][b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {   // (unrolled with guards: a runtime index would put acc in scratch)
            if (a >= nci || b >= nco) continue;
            if (part) {   // natural strides: [workgroup row][dW linear index], summed by finish_partials
                part[(long)blockIdx.x * part_stride + (long)(g * cog + col + b) * d.ws_co + (long)(cil + a) * d.ws_ci +
                     (long)tap * d.ws_tap] = acc[a][b];
            } else {
                atomicAdd(d.dw + (long)(g * cog + col + b) * d.ws_co + (long)(cil + a) * d.ws_ci +
                              (long)tap * d.ws_tap,
                          acc[a][b]);
            }
        }
}

// ---------------------------------------------------------------------------
// weight gradient when one side has <= 4 channels (stem 3->32, heads 32->2, SEAM 1->64,
// 1x1 classifiers 64->2): thread = (pixel slot, channel of the WIDE side), the narrow
// side x taps live in registers; lanes of a wave read 64-256 contiguous bytes of the wide
// tensor, the narrow tensor is a wave-uniform (broadcast) load.  LDS reduction over the
// pixel slots, then one atomic per dW element per workgroup.
// ---------------------------------------------------------------------------
template <typename TX, typename TZ, bool NARROW_CO, int NC, int TAPS>
__global__ void __launch_bounds__(256) conv_wgrad_narrow_kernel(ledn_wgrad_desc d, int pix_per_block, float* part,
                                                                long part_stride) {
    __shared__ float s_red[256];
    const int wide = NARROW_CO ? d.Cin : d.Cout;
    const int slots = 256 / wide;
    const int slot = threadIdx.x / wide, wc = threadIdx.x % wide;
    const bool worker = slot < slots;
    float acc[TAPS][NC];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[t][c] = 0.f;
    const TX* x = reinterpret_cast<const TX*>(d.x);
    const TZ* dz = reinterpret_cast<const TZ*>(d.dz);
    const int ncn = NARROW_CO ? d.Cout : d.Cin;   // actual narrow channel count (<= NC)
    if (worker) {
        const long npix = (long)d.N * d.Ho * d.Wo;
        const long p0 = (long)blockIdx.x * pix_per_block;
        const long p1 = min(npix, p0 + (long)pix_per_block);
        for (long p = p0 + slot; p < p1; p += slots) {
            const int wo = (int)(p % d.Wo);
            const int ho = (int)((p / d.Wo) % d.Ho);
            const int n = (int)(p / ((long)d.Wo * d.Ho));
            float zv[NC];
            float zw = 0.f;
            if (NARROW_CO) {
#pragma unroll
                for (int c = 0; c < NC; ++c) zv[c] = c < ncn ? ld(dz + p * d.Cout + c) : 0.f;
            } else {
                zw = ld(dz + p * d.Cout + wc);
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int kh = t / d.KW, kw = t % d.KW;
                const int hi = ho * d.stride - d.pad + kh * d.dil, wi = wo * d.stride - d.pad + kw * d.dil;
                if (hi < 0 || hi >= d.H || wi < 0 || wi >= d.W) continue;
                const long xoff = (((long)n * d.H + hi) * d.W + wi) * d.Cin;
                if (NARROW_CO) {
                    float xv = ld(x + xoff + wc);
                    if (d.in_scale) xv = xv * d.in_scale[wc] + d.in_shift[wc];
                    if (d.in_act == LEDN_ACT_RELU) xv = fmaxf(xv, 0.f);
                    else if (d.in_act == LEDN_ACT_PRELU) xv = xv > 0.f ? xv : xv * d.in_slope[wc];
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[t][c] = fmaf(xv, zv[c], acc[t][c]);
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (c >= ncn) break;
                        float xv = ld(x + xoff + c);
                        if (d.in_scale) xv = xv * d.in_scale[c] + d.in_shift[c];
                        if (d.in_act == LEDN_ACT_RELU) xv = fmaxf(xv, 0.f);
                        else if (d.in_act == LEDN_ACT_PRELU) xv = xv > 0.f ? xv : xv * d.in_slope[c];
                        acc[t][c] = fmaf(xv, zw, acc[t][c]);
                    }
                }
            }
        }
    }
    // reduce over the pixel slots, one (tap, narrow channel) at a time
    // (fully unrolled: a runtime index into acc[][] would push it to scratch)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c >= ncn) break;
            s_red[threadIdx.x] = worker ? acc[t][c] : 0.f;
            __syncthreads();
            if (threadIdx.x < wide) {
                float v = 0.f;
                for (int sl = 0; sl < slots; ++sl) v += s_red[sl * wide + threadIdx.x];
                const int co = NARROW_CO ? c : threadIdx.x, ci = NARROW_CO ? threadIdx.x : c;
                const long idx = (long)co * d.ws_co + (long)ci * d.ws_ci + (long)t * d.ws_tap;
                if (part) part[(long)blockIdx.x * part_stride + idx] = v;
                else atomicAdd(d.dw + idx, v);
            }
            __syncthreads();
        }
}

template <typename TX, typename TZ>
static int launch_narrow(const ledn_wgrad_desc& d, hipStream_t s) {
    const long npix = (long)d.N * d.Ho * d.Wo;
    const bool nco = d.Cout <= 4;
    const int taps = d.KH * d.KW;
    // every thread walks its pixels serially (dependent loads): many short workgroups whose partial
    // dW tiles go to the workspace (natural strides), else <= 256 workgroups ending in atomics
    const long numel = (long)d.Cout * d.Cin * taps;
    long ppb = cdiv(npix, 2048);
    if (ppb < 64) ppb = 64;
    long nb = cdiv(npix, ppb);
    float* part = (wgrad_natural_strides(d) && (nb > 8 || det())) ? ws_take(nb * numel) : nullptr;
    if (!part) {
        ppb = cdiv(npix, 256);
        if (ppb < 64) ppb = 64;
        nb = cdiv(npix, ppb);
    }
    const dim3 grid((unsigned)nb);
#define LEDN_NW(NCO, TP) \
    LEDN_LAUNCH((conv_wgrad_narrow_kernel<TX, TZ, NCO, 4, TP>), grid, dim3(256), 0, s, d, (int)ppb, part, numel)
    if (nco && taps == 9) LEDN_NW(true, 9);
    else if (nco && taps == 1) LEDN_NW(true, 1);
    else if (!nco && taps == 9) LEDN_NW(false, 9);
    else if (!nco && taps == 1) LEDN_NW(false, 1);
    else return LEDN_EINVAL;
#undef LEDN_NW
    if (part) return finish_partials(part, (int)nb, (int)numel, 1, d.dw, nullptr, nullptr, s);
    return check_launch();
}

static bool narrow_ok(const ledn_wgrad_desc& d) {
    if (d.groups != 1 || d.xadd) return false;
    const int taps = d.KH * d.KW;
    if (taps != 1 && taps != 9) return false;
    if (d.Cout <= 4 && d.Cin <= 256 && d.Cin > 4) return true;
    if (d.Cin <= 4 && d.Cout <= 256 && d.Cout > 4) return true;
    return false;
}

int channel_stats_impl(const void* x, const void* xadd, long long P, int C, int dtype, float* sum,
                       float* sqsum, hipStream_t s);

// ---------------------------------------------------------------------------
// Weight gradient of the two-class heads (Cout <= 2: head_x1 / head_x2 3x3 32->2, cls_seg 1x1 64->2,
// led_head.py:44-51), bf16, stride 1: thread = (input pixel q, 8 input channels).  x[q] is read ONCE
// (16 bytes, the producer's BatchNorm + ReLU applied in registers); the K*K x 2 gradient values
// dz[q - tap] that multiply it are 4-byte loads of neighbouring pixels (cache hits: dz is 1/16 of x);
// K*K x 2 x 8 sums live in registers over the thread's pixels, then an LDS reduction over the 64
// pixel slots of the workgroup, one partial row per workgroup, finish_partials.  (On the MFMA path
// 30 of 32 output rows were padding and the narrow dz staging was element-wise: 0.28 ms for the
// 268 MB x1; this kernel streams x once.)
// ---------------------------------------------------------------------------
template <typename TZ, int K>
__global__ void __launch_bounds__(256) conv_wgrad_cout2_kernel(ledn_wgrad_desc d, float* part) {
    constexpr int KK = K * K;
    __shared__ float s_red[256 * 16];
    const int cgn = d.Cin / 8;                       // channel groups (4 for 32, 8 for 64)
    const int slots = 256 / cgn;
    const int slot = threadIdx.x / cgn, cg = threadIdx.x % cgn;
    const int c = cg * 8;
    float acc[KK][2][8];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[t][o][i] = 0.f;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = d.in_scale ? d.in_scale[c + i] : 1.f;
        sh[i] = d.in_shift ? d.in_shift[c + i] : 0.f;
    }
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
    const TZ* dz = reinterpret_cast<const TZ*>(d.dz);
    const long npix = (long)d.N * d.H * d.W;
    const long stride = (long)gridDim.x * slots;
    if (K == 1 && slot < slots) {
        // 1x1: dz[q] multiplies x[q] -- no coordinates; four pixels per trip, all eight loads issued before the first use
        // (the loop below walked its pixels one dependent round trip at a time, with three 64-bit divisions each)
        for (long q = (long)blockIdx.x * slots + slot; q < npix; q += 4 * stride) {
            uint4 xr[4];
            float g0[4], g1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long qq = q + u * stride;
                const bool ok = qq < npix;
                const long qs = ok ? qq : q;
                xr[u] = *reinterpret_cast<const uint4*>(x + qs * d.Cin + c);
                const float a = ld(dz + qs * d.Cout), b = d.Cout > 1 ? ld(dz + qs * d.Cout + 1) : 0.f;
                g0[u] = ok ? a : 0.f;
                g1[u] = ok ? b : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float xv[8];
                ld8(reinterpret_cast<const bf16_t*>(&xr[u]), xv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xv[i] = xv[i] * sc[i] + sh[i];
                    if (d.in_act == LEDN_ACT_RELU) xv[i] = fmaxf(xv[i], 0.f);
                    acc[0][0][i] = fmaf(xv[i], g0[u], acc[0][0][i]);
                    acc[0][1][i] = fmaf(xv[i], g1[u], acc[0][1][i]);
                }
            }
        }
    } else if (slot < slots) {
        for (long q = (long)blockIdx.x * slots + slot; q < npix; q += stride) {
            const int wi = (int)(q % d.W);
            const int hi = (int)((q / d.W) % d.H);
            const int n = (int)(q / ((long)d.W * d.H));
            float xv[8];
            ld8(x + q * d.Cin + c, xv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xv[i] = xv[i] * sc[i] + sh[i];
                if (d.in_act == LEDN_ACT_RELU) xv[i] = fmaxf(xv[i], 0.f);
            }
            float g[KK][2];
#pragma unroll
            for (int t = 0; t < KK; ++t) {           // output pixel fed by x[q] through tap t
                const int ho = hi + d.pad - t / K, wo = wi + d.pad - t % K;
                const bool ok = ho >= 0 && ho < d.Ho && wo >= 0 && wo < d.Wo;
                const long zo = ok ? (((long)n * d.Ho + ho) * d.Wo + wo) * d.Cout : 0L;
                const float g0 = ld(dz + zo), g1 = d.Cout > 1 ? ld(dz + zo + 1) : 0.f;
                g[t][0] = ok ? g0 : 0.f;
                g[t][1] = ok ? g1 : 0.f;
            }
#pragma unroll
            for (int t = 0; t < KK; ++t)
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[t][o][i] = fmaf(xv[i], g[t][o], acc[t][o][i]);
        }
    }
    // reduce over the pixel slots, one tap at a time: s_red[thread][o*8 + i]
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int i = 0; i < 8; ++i) s_red[threadIdx.x * 16 + o * 8 + i] = slot < slots ? acc[t][o][i] : 0.f;
        __syncthreads();
        for (int e = threadIdx.x; e < cgn * 16; e += 256) {      // (cg, o, i)
            const int g_ = e / 16, oi = e % 16;
            float v = 0.f;
            for (int sl = 0; sl < slots; ++sl) v += s_red[(sl * cgn + g_) * 16 + oi];
            const int co = oi / 8, ci = g_ * 8 + oi % 8;
            if (co < d.Cout) {
                const long idx = (long)co * d.ws_co + (long)ci * d.ws_ci + (long)t * d.ws_tap;
                if (part) part[(long)blockIdx.x * ((long)d.Cout * d.Cin * KK) + idx] = v;
                else atomicAdd(d.dw + idx, v);
            }
        }
    }
}

static bool cout2_ok(const ledn_wgrad_desc& d) {
    if (d.Cout > 2 || d.groups != 1 || d.xadd || d.stride != 1 || d.dil != 1) return false;
    if (d.dtype_x != LEDN_BF16 || (d.Cin != 32 && d.Cin != 64 && d.Cin != 128)) return false;
    if (d.in_act != LEDN_ACT_NONE && d.in_act != LEDN_ACT_RELU) return false;
    if (!((d.KH == 3 && d.KW == 3 && d.pad == 1) || (d.KH == 1 && d.KW == 1 && d.pad == 0))) return false;
    return d.Ho == d.H && d.Wo == d.W;
}

// 1x1 only: measured 0.142 -> 0.069 ms for the two 64->2 classifiers; the 3x3 instance (144 accumulators,
// 214 VGPRs) ran SLOWER than the MFMA narrow path (0.41 vs 0.28 ms at 16 x 512 x 512) and stays there
bool conv_wgrad_cout2_supported(const ledn_wgrad_desc& d) { return cout2_ok(d) && d.KH == 1 && wgrad_natural_strides(d); }

int conv_wgrad_cout2(const ledn_wgrad_desc& d, hipStream_t s) {
    const long npix = (long)d.N * d.H * d.W;
    const int slots = 256 / (d.Cin / 8);
    long nb = cdiv(npix, (long)slots * 16);           // >= 16 pixels per thread
    if (nb > 1024) nb = 1024;
    const long numel = (long)d.Cout * d.Cin * d.KH * d.KW;
    float* part = (nb > 4 || det()) ? ws_take(nb * numel) : nullptr;
    if (!part && nb > 16) nb = 16;
    const dim3 grid((unsigned)nb);
#define LEDN_C2(TZ)                                                                                   \
    do {                                                                                              \
        if (d.KH == 3) LEDN_LAUNCH((conv_wgrad_cout2_kernel<TZ, 3>), grid, dim3(256), 0, s, d, part); \
        else LEDN_LAUNCH((conv_wgrad_cout2_kernel<TZ, 1>), grid, dim3(256), 0, s, d, part);           \
    } while (0)
    if (d.dtype_dz == LEDN_BF16) LEDN_C2(bf16_t);
    else if (d.dtype_dz == LEDN_F32) LEDN_C2(float);
    else return LEDN_EINVAL;
#undef LEDN_C2
    int rc = part ? finish_partials(part, (int)nb, (int)numel, 1, d.dw, nullptr, nullptr, s) : check_launch();
    if (rc != LEDN_OK || !d.db) return rc;
    return channel_stats_impl(d.dz, nullptr, npix, d.Cout, d.dtype_dz, d.db, nullptr, s);
}

// Weight gradient of a 1 -> Cout convolution (SEAM's 3x3 conv_2, tools/speed/ddrnet_speed.py:302-338: the edge map
// gates 64 channels): dw[co][tap] = sum_px dz[px][co] * x[px @ tap].  Thread = (pixel slot, 8 output channels):
// dz is read ONCE (16-byte loads), the K*K taps of the single input channel are scalar loads of neighbouring pixels
// (cache hits: x is 1/64 of dz), K*K x 8 sums per thread, LDS reduction over the pixel slots, one partial row per
// workgroup.  (The generic tile kernel ran this shape at 0.38 TB/s: 178 us for the 33 MB gradient.)
template <typename TX, typename TZ, int K>
__global__ void __launch_bounds__(256) conv_wgrad_cin1_kernel(ledn_wgrad_desc d, float* part) {
    constexpr int KK = K * K;
    __shared__ float s_red[256 * 8];
    const int cgn = d.Cout / 8, slots = 256 / cgn;
    const int slot = threadIdx.x / cgn, cg = threadIdx.x % cgn, c = cg * 8;
    float acc[KK][8];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
    const TX* x = reinterpret_cast<const TX*>(d.x);
    const TZ* dz = reinterpret_cast<const TZ*>(d.dz);
    const long npix = (long)d.N * d.Ho * d.Wo;
    const long stride = (long)gridDim.x * slots;
    if (slot < slots) {
        for (long q = (long)blockIdx.x * slots + slot; q < npix; q += stride) {
            const int wo = (int)(q % d.Wo), ho = (int)((q / d.Wo) % d.Ho), n = (int)(q / ((long)d.Wo * d.Ho));
            float g[8];
            ld8(dz + q * d.Cout + c, g);
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                const int hi = ho - d.pad + t / K, wi = wo - d.pad + t % K;
                const bool ok = hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
                float xv = ld(x + (ok ? ((long)n * d.H + hi) * d.W + wi : 0L));
                xv = ok ? xv : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[t][i] = fmaf(xv, g[i], acc[t][i]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) s_red[threadIdx.x * 8 + i] = slot < slots ? acc[t][i] : 0.f;
        __syncthreads();
        for (int e = threadIdx.x; e < d.Cout; e += 256) {
            float v = 0.f;
            for (int sl = 0; sl < slots; ++sl) v += s_red[(sl * cgn + e / 8) * 8 + e % 8];
            const long idx = (long)e * d.ws_co + (long)t * d.ws_tap;
            if (part) part[(long)blockIdx.x * ((long)d.Cout * KK) + idx] = v;
            else atomicAdd(d.dw + idx, v);
        }
    }
}

static bool cin1_ok(const ledn_wgrad_desc& d) {
    if (d.Cin != 1 || d.groups != 1 || d.xadd || d.stride != 1 || d.dil != 1 || d.in_scale || d.in_act != LEDN_ACT_NONE)
        return false;
    if (d.Cout % 8 || d.Cout > 256 || 256 % (d.Cout / 8)) return false;
    if (!((d.KH == 3 && d.KW == 3) || (d.KH == 1 && d.KW == 1))) return false;
    return wgrad_natural_strides(d);
}

static int conv_wgrad_cin1(const ledn_wgrad_desc& d, hipStream_t s) {
    const long npix = (long)d.N * d.Ho * d.Wo;
    const int slots = 256 / (d.Cout / 8);
    long nb = cdiv(npix, (long)slots * 8);
    if (nb > 512) nb = 512;
    const long numel = (long)d.Cout * d.KH * d.KW;
    float* part = (nb > 4 || det()) ? ws_take(nb * numel) : nullptr;
    if (!part && nb > 16) nb = 16;
    const dim3 grid((unsigned)nb);
#define LEDN_C1(TX, TZ)                                                                                       \
    do {                                                                                                      \
        if (d.KH == 3) LEDN_LAUNCH((conv_wgrad_cin1_kernel<TX, TZ, 3>), grid, dim3(256), 0, s, d, part);      \
        else LEDN_LAUNCH((conv_wgrad_cin1_kernel<TX, TZ, 1>), grid, dim3(256), 0, s, d, part);                \
    } while (0)
    if (d.dtype_x == LEDN_F32 && d.dtype_dz == LEDN_F32) LEDN_C1(float, float);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_dz == LEDN_BF16) LEDN_C1(bf16_t, bf16_t);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_dz == LEDN_F32) LEDN_C1(bf16_t, float);
    else if (d.dtype_x == LEDN_F32 && d.dtype_dz == LEDN_BF16) LEDN_C1(float, bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_C1
    int rc = part ? finish_partials(part, (int)nb, (int)numel, 1, d.dw, nullptr, nullptr, s) : check_launch();
    if (rc != LEDN_OK || !d.db) return rc;
    return channel_stats_impl(d.dz, nullptr, npix, d.Cout, d.dtype_dz, d.db, nullptr, s);
}

int conv_wgrad_direct(const ledn_wgrad_desc& d, hipStream_t s) {
    if (cin1_ok(d) && (options().stream_fast & 2)) return conv_wgrad_cin1(d, s);
    if (narrow_ok(d)) {
        int rc;
        if (d.dtype_x == LEDN_F32 && d.dtype_dz == LEDN_F32) rc = launch_narrow<float, float>(d, s);
        else if (d.dtype_x == LEDN_BF16 && d.dtype_dz == LEDN_BF16) rc = launch_narrow<bf16_t, bf16_t>(d, s);
        else if (d.dtype_x == LEDN_BF16 && d.dtype_dz == LEDN_F32) rc = launch_narrow<bf16_t, float>(d, s);
        else if (d.dtype_x == LEDN_F32 && d.dtype_dz == LEDN_BF16) rc = launch_narrow<float, bf16_t>(d, s);
        else return LEDN_EINVAL;
        if (rc != LEDN_OK || !d.db) return rc;
        return channel_stats_impl(d.dz, nullptr, (long long)d.N * d.Ho * d.Wo, d.Cout, d.dtype_dz, d.db, nullptr, s);
    }
    const int cog = d.Cout / d.groups, cig = d.Cin / d.groups;
    const int ci_tiles = (int)cdiv(cig, 64), co_tiles = (int)cdiv(cog, 64);
    const long npix = (long)d.N * d.Ho * d.Wo;
    const int tiles = ci_tiles * co_tiles * d.groups * d.KH * d.KW;
    // ~1024 workgroups of >= one WG_PC-pixel chunk; their partial tiles go to the workspace and
    // finish_partials sums them (natural strides), else <= 16 pixel ranges ending in atomics
    const long numel = (long)d.Cout * (d.Cin / d.groups) * d.KH * d.KW;
    long nbx = cdiv(1024, tiles);
    if (nbx > cdiv(npix, WG_PC)) nbx = cdiv(npix, WG_PC);
    float* part = (wgrad_natural_strides(d) && (nbx > 4 || det())) ? ws_take(nbx * numel) : nullptr;
    if (!part && nbx > 16) nbx = 16;
    long ppb = cdiv(cdiv(npix, nbx), WG_PC) * WG_PC;
    nbx = cdiv(npix, ppb);
    const dim3 grid((unsigned)nbx, (unsigned)(ci_tiles * co_tiles * d.groups), (unsigned)(d.KH * d.KW));
#define LEDN_WG(TX, TZ) \
    LEDN_LAUNCH((conv_wgrad_direct_kernel<TX, TZ>), grid, dim3(256), 0, s, d, (int)ppb, ci_tiles, co_tiles, part, numel)
    if (d.dtype_x == LEDN_F32 && d.dtype_dz == LEDN_F32) LEDN_WG(float, float);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_dz == LEDN_BF16) LEDN_WG(bf16_t, bf16_t);
    else if (d.dtype_x == LEDN_BF16 && d.dtype_dz == LEDN_F32) LEDN_WG(bf16_t, float);
    else if (d.dtype_x == LEDN_F32 && d.dtype_dz == LEDN_BF16) LEDN_WG(float, bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_WG
    int rc = part ? finish_partials(part, (int)nbx, (int)numel, 1, d.dw, nullptr, nullptr, s) : check_launch();
    if (rc != LEDN_OK) return rc;
    if (d.db) rc = channel_stats_impl(d.dz, nullptr, npix, d.Cout, d.dtype_dz, d.db, nullptr, s);
    return rc;
}

int wgrad_validate(const ledn_wgrad_desc& d) {
    LEDN_REQUIRE(d.x && d.dz && d.dw);
    LEDN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Ho > 0 && d.Wo > 0 && d.Cout > 0);
    LEDN_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.dil > 0 && d.pad >= 0 && d.groups > 0);
    LEDN_REQUIRE(d.Cin % d.groups == 0 && d.Cout % d.groups == 0);
    LEDN_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr));
    LEDN_REQUIRE(d.in_act == LEDN_ACT_NONE || d.in_act == LEDN_ACT_RELU || (d.in_act == LEDN_ACT_PRELU && d.in_slope));
    LEDN_REQUIRE(d.Ho == (d.H + 2 * d.pad - ((d.KH - 1) * d.dil + 1)) / d.stride + 1);
    LEDN_REQUIRE(d.Wo == (d.W + 2 * d.pad - ((d.KW - 1) * d.dil + 1)) / d.stride + 1);
    return LEDN_OK;
}

}  // namespace ledn
