// conv_f32.hip -- the reference-precision (f32) convolutions on the matrix cores.
//
// gfx950 has an f32-input MFMA, v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation, bitwise a chain of
// fmaf over k (MI355X_MICROARCH.md, Matrix cores), at the f32 vector rate (157 TFLOP/s dense).  The reference trains in
// fp32 unless --amp is given (tools/train.py:79-91); the f32 mode of this library is the one whose argmax masks are
// bit-exact against the CPU oracle, and until round 4 it ran on the VALU kernels of conv_direct.hip only (one thread =
// one pixel x 16 output channels, weights broadcast from LDS: 18-27 TFLOP/s forward, 4-17 TFLOP/s weight gradient).
//
//   conv_f32_mfma_kernel        forward and data gradient (ledn_conv2d, transposed = 0 / 1), implicit GEMM:
//                               M = 32 output pixels per accumulator tile, N = 32 output channels, K = taps x Cin.
//                               A fragments are 16-byte global loads of the NHWC input (lane = pixel, four consecutive
//                               input channels: four K steps), B fragments 16-byte LDS reads of the weights staged as
//                               [tap][channel octet][half][cout][4]; a wave owns PT x CT accumulator tiles, the
//                               workgroup (4 waves) 128 PT pixels x 32 CT channels.  Grouped 1x1 layers are densified
//                               in the staging (zeros outside the group), as the bf16 path does.
//   conv_wgrad_f32_mfma_kernel  weight gradient (ledn_conv2d_wgrad): K = pixels; both operands are pixel-major in
//                               memory, so a fragment is ONE coalesced dword load (lane = channel, two pixels per K
//                               step); a wave owns one 32 x 32 (cout, cin) tile for ALL taps (<= 9 x 16 accumulator
//                               registers), the four waves of a workgroup split its pixel range and add up through LDS;
//                               per-workgroup partial tiles + an ordered summing kernel (no atomics: deterministic).
//
// Same semantics as conv_direct_kernel / conv_wgrad_direct_kernel (prologue on the input, affine + statistics +
// residual + activation epilogue, stride / dilation / padding, transposed addressing of the OIHW master weights).
#include "ledn_rt.h"

namespace ledn {

#ifdef LEDN_CPU_EMU
__device__ __forceinline__ f32x16_t mfma_32x32x2_f32(float a, float b, f32x16_t c) { return emu::mfma_32x32x2_f32(a, b, c); }
#else
__device__ __forceinline__ f32x16_t mfma_32x32x2_f32(float a, float b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
#endif

constexpr int CF_CIB = 32;        // input channels per staged weight chunk

// --------------------------------------------------------------------------------------------------------------------
// forward / data gradient
// --------------------------------------------------------------------------------------------------------------------
template <int PT, int CT, bool PRO>
__global__ void __launch_bounds__(256) conv_f32_mfma_kernel(ledn_conv_desc d, float* part, int cibmax) {
    LEDN_DYN_SHARED(float, s_w);                                      // [taps][Q][2][NT][4]
    __shared__ float s_stat[4][2 * 32 * CT];
    constexpr int NT = 32 * CT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int taps = d.KH * d.KW;
    const int npix = d.N * d.Ho * d.Wo;            // (tensors below 2^31 elements: conv_f32_mfma_supported)
    const int cog = d.Cout / d.groups, cig = d.Cin / d.groups;
    const int co0 = blockIdx.y * NT;
    // pixels of this lane: tile pt covers pixels base + pt * 32 + lp
    const int base = (blockIdx.x * 4 + wave) * (32 * PT);
    int pn[PT], pho[PT], pwo[PT];
    bool pok[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int pix = base + pt * 32 + lp;
        pok[pt] = pix < npix;
        const int q = pok[pt] ? pix : 0;
        const int t = q / d.Wo;
        pwo[pt] = q - t * d.Wo;
        pn[pt] = t / d.Ho;
        pho[pt] = t - pn[pt] * d.Ho;
    }
    f32x16_t acc[PT][CT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pt][ct][r] = 0.f;
    const float* x = reinterpret_cast<const float*>(d.x);
    const float* xadd = reinterpret_cast<const float*>(d.xadd);

    // input-channel range that can meet this tile's output channels (grouped layers: the tile's groups only)
    int c_lo = 0, c_hi = d.Cin;
    if (d.groups > 1) {
        const int g_lo = co0 / cog, g_hi = min(d.Cout - 1, co0 + NT - 1) / cog;
        c_lo = g_lo * cig;
        c_hi = (g_hi + 1) * cig;
    }
    for (int cb = c_lo; cb < c_hi; cb += cibmax) {
        const int cib = min(cibmax, c_hi - cb);       // multiple of 8
        const int Q = cib >> 3;
        __syncthreads();                              // the previous chunk's fragment reads are done
        // stage W(co, ci, tap) of the chunk: s_w[tap][q][h][co][j], channel ci = cb + 8 q + 4 h + j.  Index arithmetic by
        // shifts only (NT, Q powers of two): a division per element cost as much as the matrix work of the chunk.
        {
            const int lq = Q == 4 ? 2 : (Q == 2 ? 1 : 0);
            constexpr int LNT = CT == 2 ? 6 : 5;
            const int per_tap = Q * 2 * NT * 4;
            for (int i = threadIdx.x; i < per_tap; i += 256) {
                const int j = i & 3, co = (i >> 2) & (NT - 1), h = (i >> (2 + LNT)) & 1, qq = (i >> (3 + LNT)) & ((1 << lq) - 1);
                const int cg = co0 + co, ci = cb + 8 * qq + 4 * h + j;       // global output / input channel (kernel space)
                long wb = -1;
                if (cg < d.Cout) {
                    int g = 0, cil = ci;
                    if (d.groups > 1) {
                        g = cg / cog;
                        cil = ci - g * cig;
                    }
                    if (cil >= 0 && cil < cig)
                        // forward: W(co global, ci local); transposed: W(kernel ci = forward output channel, global; co local)
                        // -- the addressing of conv_direct_kernel
                        wb = d.transposed ? (long)(cg - g * cog) * d.ws_co + (long)ci * d.ws_ci
                                          : (long)cg * d.ws_co + (long)cil * d.ws_ci;
                }
                for (int tap = 0; tap < taps; ++tap)
                    s_w[tap * per_tap + i] = wb >= 0 ? d.w[wb + (long)tap * d.ws_tap] : 0.f;
            }
        }
        __syncthreads();
        // prologue coefficients of this lane's channels (chunk-invariant over taps)
        float4 psc[PRO ? 4 : 1], psh[PRO ? 4 : 1], psl[PRO ? 4 : 1];
        if (PRO) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int c = cb + 8 * qq + 4 * lh;
                const bool ok = qq < Q;
                psc[qq] = (ok && d.in_scale) ? *reinterpret_cast<const float4*>(d.in_scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
                psh[qq] = (ok && d.in_shift) ? *reinterpret_cast<const float4*>(d.in_shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                psl[qq] = (ok && d.in_act == LEDN_ACT_PRELU) ? *reinterpret_cast<const float4*>(d.in_slope + c)
                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int kh = 0; kh < d.KH; ++kh) {
            // input row of every pixel tile for this filter row (32-bit element offsets: the launcher checks the size)
            int rowoff[PT];
            bool rowok[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                int hi;
                bool v = pok[pt];
                if (!d.transposed) {
                    hi = pho[pt] * d.stride - d.pad + kh * d.dil;
                } else {
                    const int th = pho[pt] + d.pad - kh * d.dil;
                    if (d.stride == 1) hi = th;
                    else if (d.stride == 2) { v = v && !(th & 1); hi = th >> 1; }
                    else { v = v && th >= 0 && (th % d.stride) == 0; hi = th >= 0 ? th / d.stride : -1; }
                }
                rowok[pt] = v && hi >= 0 && hi < d.H;
                rowoff[pt] = (pn[pt] * d.H + hi) * d.W;
            }
            for (int kw = 0; kw < d.KW; ++kw) {
                const int tap = kh * d.KW + kw;
                int off[PT];
                bool val[PT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    int wi;
                    bool v = rowok[pt];
                    if (!d.transposed) {
                        wi = pwo[pt] * d.stride - d.pad + kw * d.dil;
                    } else {
                        const int tw = pwo[pt] + d.pad - kw * d.dil;
                        if (d.stride == 1) wi = tw;
                        else if (d.stride == 2) { v = v && !(tw & 1); wi = tw >> 1; }
                        else { v = v && tw >= 0 && (tw % d.stride) == 0; wi = tw >= 0 ? tw / d.stride : -1; }
                    }
                    v = v && wi >= 0 && wi < d.W;
                    val[pt] = v;
                    off[pt] = v ? (rowoff[pt] + wi) * d.Cin + cb + 4 * lh : 0;
                }
                const float* wt = s_w + tap * Q * 2 * NT * 4 + (lh * NT + lp) * 4;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    if (qq < Q) {
                        float4 a[PT];
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt) {
                            a[pt] = *reinterpret_cast<const float4*>(x + off[pt] + 8 * qq);
                            if (PRO) {
                                if (xadd) {
                                    const float4 u = *reinterpret_cast<const float4*>(xadd + off[pt] + 8 * qq);
                                    a[pt].x += u.x; a[pt].y += u.y; a[pt].z += u.z; a[pt].w += u.w;
                                }
                                a[pt].x = a[pt].x * psc[qq].x + psh[qq].x;
                                a[pt].y = a[pt].y * psc[qq].y + psh[qq].y;
                                a[pt].z = a[pt].z * psc[qq].z + psh[qq].z;
                                a[pt].w = a[pt].w * psc[qq].w + psh[qq].w;
                                if (d.in_act == LEDN_ACT_RELU) {
                                    a[pt].x = fmaxf(a[pt].x, 0.f); a[pt].y = fmaxf(a[pt].y, 0.f);
                                    a[pt].z = fmaxf(a[pt].z, 0.f); a[pt].w = fmaxf(a[pt].w, 0.f);
                                } else if (d.in_act == LEDN_ACT_PRELU) {
                                    a[pt].x = a[pt].x > 0.f ? a[pt].x : a[pt].x * psl[qq].x;
                                    a[pt].y = a[pt].y > 0.f ? a[pt].y : a[pt].y * psl[qq].y;
                                    a[pt].z = a[pt].z > 0.f ? a[pt].z : a[pt].z * psl[qq].z;
                                    a[pt].w = a[pt].w > 0.f ? a[pt].w : a[pt].w * psl[qq].w;
                                }
                            }
                            if (!val[pt]) a[pt] = make_float4(0.f, 0.f, 0.f, 0.f);      // zero padding AFTER the prologue
                        }
                        float4 b[CT];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            b[ct] = *reinterpret_cast<const float4*>(wt + qq * 2 * NT * 4 + ct * 32 * 4);
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) {
                                acc[pt][ct] = mfma_32x32x2_f32(a[pt].x, b[ct].x, acc[pt][ct]);
                                acc[pt][ct] = mfma_32x32x2_f32(a[pt].y, b[ct].y, acc[pt][ct]);
                                acc[pt][ct] = mfma_32x32x2_f32(a[pt].z, b[ct].z, acc[pt][ct]);
                                acc[pt][ct] = mfma_32x32x2_f32(a[pt].w, b[ct].w, acc[pt][ct]);
                            }
                    }
                }
            }
        }
    }
    // ---- epilogue: lane = output channel (column), registers = 16 pixels (rows (r&3) + 8 (r>>2) + 4 lh)
    float* y = reinterpret_cast<float*>(d.y);
    const float* res = reinterpret_cast<const float*>(d.res);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int co = co0 + ct * 32 + lp;
        const bool cok = co < d.Cout;
        const float sc = (cok && d.out_scale) ? d.out_scale[co] : 1.f;
        const float sh = (cok && d.out_shift) ? d.out_shift[co] : 0.f;
        const float sl = (cok && d.slope) ? d.slope[co] : 0.f;
        float st1 = 0.f, st2 = 0.f;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int pix = base + pt * 32 + row;
                if (pix < npix && cok) {
                    float v = acc[pt][ct][r] * sc + sh;
                    st1 += v;
                    st2 = fmaf(v, v, st2);
                    if (d.res_mode != LEDN_RES_NONE) {
                        const float rv = res[(long)pix * d.Cout + co];
                        v = d.res_mode == LEDN_RES_ADD ? v + rv : v * rv + rv;
                    }
                    if (d.act_out != LEDN_ACT_NONE) v = act_apply(d.act_out, v, sl);
                    y[(long)pix * d.Cout + co] = v;
                }
            }
        }
        if (d.stat_sum) {
            st1 += __shfl_xor(st1, 32);
            st2 += __shfl_xor(st2, 32);
            if (lh == 0) {
                s_stat[wave][ct * 32 + lp] = st1;
                s_stat[wave][32 * CT + ct * 32 + lp] = st2;
            }
        }
    }
    if (d.stat_sum) {
        __syncthreads();
        if (threadIdx.x < 2 * NT) {
            const int j = threadIdx.x / NT, c = threadIdx.x % NT;
            const float t = (s_stat[0][threadIdx.x] + s_stat[1][threadIdx.x]) + (s_stat[2][threadIdx.x] + s_stat[3][threadIdx.x]);
            if (co0 + c < d.Cout) {
                if (part) part[(long)blockIdx.x * 2 * d.Cout + j * d.Cout + co0 + c] = t;
                else atomicAdd((j ? d.stat_sqsum : d.stat_sum) + co0 + c, t);
            }
        }
    }
}

bool conv_f32_mfma_supported(const ledn_conv_desc& d) {
    if (d.dtype_x != LEDN_F32 || d.dtype_y != LEDN_F32) return false;
    if (d.Cin % 8 || d.Cout < 16) return false;
    const int cig = d.Cin / d.groups;
    if (d.groups > 1 && (cig % 8)) return false;          // chunk bounds are multiples of 8 channels
    if (d.KH * d.KW > 9) return false;
    if ((long)d.N * d.Ho * d.Wo < 2048) return false;     // tiny maps: the direct kernel's launch is cheaper
    if ((long)d.N * d.H * d.W * d.Cin >= (1L << 31) || (long)d.N * d.Ho * d.Wo * d.Cout >= (1L << 31)) return false;   // 32-bit offsets
    return true;
}

int conv_f32_mfma(const ledn_conv_desc& d, hipStream_t s) {
    const long npix = (long)d.N * d.Ho * d.Wo;
    const int taps = d.KH * d.KW;
    const bool pro = d.in_scale || d.in_act != LEDN_ACT_NONE || d.xadd;
    const int CT = d.Cout > 32 ? 2 : 1;
    // two pixel tiles per wave when that still leaves >= 2 workgroups per CU
    const int PT = (npix / 256) * cdiv(d.Cout, 32 * CT) >= 512 ? 2 : 1;
    const long gx = cdiv(npix, 128L * PT);
    const dim3 grid((unsigned)gx, (unsigned)cdiv(d.Cout, 32 * CT));
    // weight chunk: 32 input channels, 16 where the 32-channel chunk of all taps would not fit 64 KB of LDS (3x3, 64 columns)
    int cib = (size_t)taps * CF_CIB * 32 * CT * sizeof(float) <= 65536 ? CF_CIB : CF_CIB / 2;
    if (d.groups == 1 && d.Cin < cib) cib = d.Cin;
    const size_t lds = (size_t)taps * cib * 32 * CT * sizeof(float);
    float* part = nullptr;
    if (d.stat_sum) {
        part = (gx > 16 || det()) ? ws_take(gx * 2 * d.Cout) : nullptr;
        if (det() && !part) return LEDN_EINVAL;
    }
#define LEDN_CF(PT_, CT_)                                                                                          \
    do {                                                                                                           \
        if (pro) LEDN_LAUNCH((conv_f32_mfma_kernel<PT_, CT_, true>), grid, dim3(256), lds, s, d, part, cib);       \
        else LEDN_LAUNCH((conv_f32_mfma_kernel<PT_, CT_, false>), grid, dim3(256), lds, s, d, part, cib);          \
    } while (0)
    if (PT == 2 && CT == 2) LEDN_CF(2, 2);
    else if (PT == 2) LEDN_CF(2, 1);
    else if (CT == 2) LEDN_CF(1, 2);
    else LEDN_CF(1, 1);
#undef LEDN_CF
    if (part) return finish_partials(part, (int)gx, d.Cout, 2, d.stat_sum, d.stat_sqsum, nullptr, s);
    return check_launch();
}

// --------------------------------------------------------------------------------------------------------------------
// weight gradient
// --------------------------------------------------------------------------------------------------------------------
// grid: x = pixel ranges, y = (cout tile, cin tile) pairs over the DENSE channel ranges.  part[(bx * pairs + pair) * taps
// + tap][1024]: element r * 64 + lane = D[row = (r&3) + 8 (r>>2) + 4 (lane>>5)][col = lane & 31], rows = cout, cols = cin.
template <int TAPS>
__global__ void __launch_bounds__(256, 2) conv_wgrad_f32_mfma_kernel(ledn_wgrad_desc d, int ppb, int ci_tiles, float* part) {
    __shared__ float s_red[3][16 * 64];           // waves 1..3 hand their accumulators over, one tap at a time
    constexpr int U = TAPS == 9 ? 2 : 8;          // pixel pairs (K steps) per pipeline stage: U x (1 + TAPS) loads in flight per lane
                                                  // (3x3: two stages of 20 registers next to the 144 accumulators = two waves per SIMD)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lc = lane & 31, lh = lane >> 5;
    const int pair = blockIdx.y, cot = pair / ci_tiles, cit = pair % ci_tiles;
    const int co = cot * 32 + lc, ci = cit * 32 + lc;
    const bool co_ok = co < d.Cout, ci_ok = ci < d.Cin;
    const int npix = d.N * d.Ho * d.Wo;           // (< 2^31: checked by the launcher)
    const int p_begin = blockIdx.x * ppb, p_end = min(npix, p_begin + ppb);
    const float* x = reinterpret_cast<const float*>(d.x);
    const float* xadd = reinterpret_cast<const float*>(d.xadd);
    const float* dz = reinterpret_cast<const float*>(d.dz);
    const float psc = (ci_ok && d.in_scale) ? d.in_scale[ci] : 1.f;
    const float psh = (ci_ok && d.in_shift) ? d.in_shift[ci] : 0.f;
    const float psl = (ci_ok && d.in_act == LEDN_ACT_PRELU) ? d.in_slope[ci] : 0.f;
    const bool pro = d.in_scale || d.in_act != LEDN_ACT_NONE || d.xadd;
    f32x16_t acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // wave w walks the pixel pairs w, w + 4, ... of the workgroup's range; stage = U consecutive pairs of the wave.
    // Coordinates advance incrementally (no division in the loop): this lane's pixel = p0 + 8 u + lh.
    float a_cur[U], b_cur[U][TAPS], a_nxt[U], b_nxt[U][TAPS];
    constexpr int KS = TAPS == 9 ? 3 : 1;         // square filters: 3 x 3 or 1 x 1
    auto fetch = [&](int p0, float (&av)[U], float (&bv)[U][TAPS]) {
        int pix = p0 + lh;
        int t = pix / d.Wo;
        int wo = pix - t * d.Wo;
        int n = t / d.Ho;
        int ho = t - n * d.Ho;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool pok = pix < p_end;
            av[u] = (pok && co_ok) ? dz[pix * d.Cout + co] : 0.f;
            // 32-bit element offsets (the launcher checks the tensor sizes); per tap: one add and two mask tests
            const int h0 = ho * d.stride - d.pad, w0 = wo * d.stride - d.pad;
            const int base = ((n * d.H + h0) * d.W + w0) * d.Cin + ci;
            bool rok[KS], cok[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                rok[k] = (unsigned)(h0 + k * d.dil) < (unsigned)d.H;
                cok[k] = (unsigned)(w0 + k * d.dil) < (unsigned)d.W;
            }
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) {
                const int kh = tp / KS, kw = tp % KS;
                const bool v = pok && ci_ok && rok[kh] && cok[kw];
                const int off = v ? base + (kh * d.dil * d.W + kw * d.dil) * d.Cin : 0;
                float xv = x[off];
                if (pro) {
                    if (xadd) xv += xadd[off];
                    xv = xv * psc + psh;
                    if (d.in_act == LEDN_ACT_RELU) xv = fmaxf(xv, 0.f);
                    else if (d.in_act == LEDN_ACT_PRELU) xv = xv > 0.f ? xv : xv * psl;
                }
                bv[u][tp] = v ? xv : 0.f;
            }
            pix += 8;
            wo += 8;
            while (wo >= d.Wo) {
                wo -= d.Wo;
                if (++ho >= d.Ho) {
                    ho = 0;
                    ++n;
                }
            }
        }
    };
    int p0 = p_begin + 2 * wave;
    if (p0 < p_end) fetch(p0, a_cur, b_cur);
    while (p0 < p_end) {
        const int pn = p0 + 8 * U;
        if (pn < p_end) fetch(pn, a_nxt, b_nxt);           // the next stage's loads fly under this stage's products
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] = mfma_32x32x2_f32(a_cur[u], b_cur[u][t], acc[t]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a_cur[u] = a_nxt[u];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) b_cur[u][t] = b_nxt[u][t];
        }
        p0 = pn;
    }
    // waves 1..3 -> wave 0, tap by tap, in wave order (fixed summation order)
    float* out = part + ((long)blockIdx.x * gridDim.y + pair) * TAPS * 1024;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[wave - 1][r * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[t * 1024 + r * 64 + lane] = ((acc[t][r] + s_red[0][r * 64 + lane]) + s_red[1][r * 64 + lane]) + s_red[2][r * 64 + lane];
        }
    }
}

// dW(co, ci_local, tap) += sum over the pixel-range rows of the partial tiles (row order: deterministic)
__global__ void __launch_bounds__(256) conv_wgrad_f32_finish_kernel(ledn_wgrad_desc d, const float* part, int nbx, int pairs,
                                                                    int ci_tiles, int taps) {
    const long total = (long)pairs * taps * 1024;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx % 1024), tap = (int)((idx / 1024) % taps), pair = (int)(idx / (1024L * taps));
    const int r = e >> 6, lane = e & 63;
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    const int co = (pair / ci_tiles) * 32 + row, ci = (pair % ci_tiles) * 32 + col;
    if (co >= d.Cout || ci >= d.Cin) return;
    const int cog = d.Cout / d.groups, cig = d.Cin / d.groups;
    const int g = co / cog, cil = ci - g * cig;
    if (cil < 0 || cil >= cig) return;
    float t = 0.f;
    for (int b = 0; b < nbx; ++b) t += part[((long)b * pairs + pair) * taps * 1024 + tap * 1024 + e];
    d.dw[(long)co * d.ws_co + (long)cil * d.ws_ci + (long)tap * d.ws_tap] += t;
}

static bool wgrad_f32_taps_ok(int taps) { return taps == 1 || taps == 9; }

bool conv_wgrad_f32_mfma_supported(const ledn_wgrad_desc& d) {
    if (d.dtype_x != LEDN_F32 || d.dtype_dz != LEDN_F32) return false;
    if (!wgrad_f32_taps_ok(d.KH * d.KW)) return false;
    if (d.groups > 1 && d.KH != 1) return false;
    // (any input width: lanes beyond Cin load zeros -- the 3 -> 32 stem included: 3.2 -> 1.6 ms at 16 x 1024^2; the two-class
    //  heads' 32 -> 2 layers measured SLOWER here than on the VALU kernel, 4.4 vs 2.3 ms: 30 of 32 tile rows empty)
    if (d.Cout < 16 || d.Cin < 3) return false;      // (1 -> 64, SEAM: conv_wgrad_cin1's 20 us against 300 us here)
    if ((long)d.N * d.Ho * d.Wo < 2048) return false;
    if ((long)d.N * d.H * d.W * d.Cin >= (1L << 31) || (long)d.N * d.Ho * d.Wo * d.Cout >= (1L << 31)) return false;   // 32-bit offsets
    if (d.KH != d.KW) return false;
    return true;
}

int channel_stats_impl(const void* x, const void* xadd, long long P, int C, int dtype, float* sum, float* sqsum,
                       hipStream_t s);

int conv_wgrad_f32_mfma(const ledn_wgrad_desc& d, hipStream_t s) {
    const int taps = d.KH * d.KW;
    const long npix = (long)d.N * d.Ho * d.Wo;
    const int co_tiles = (int)cdiv(d.Cout, 32), ci_tiles = (int)cdiv(d.Cin, 32);
    const int pairs = co_tiles * ci_tiles;
    long nbx = cdiv(1024, pairs);
    if (nbx > cdiv(npix, 256)) nbx = cdiv(npix, 256);       // >= 256 pixels (32 K steps per wave) per workgroup
    if (nbx < 1) nbx = 1;
    long ppb = cdiv(cdiv(npix, nbx), 8) * 8;
    nbx = cdiv(npix, ppb);
    float* part = ws_take(nbx * pairs * taps * 1024);
    if (!part) return -1;                                    // no workspace (or too small): the caller's VALU path
    const dim3 grid((unsigned)nbx, (unsigned)pairs);
    if (taps == 9) LEDN_LAUNCH((conv_wgrad_f32_mfma_kernel<9>), grid, dim3(256), 0, s, d, (int)ppb, ci_tiles, part);
    else LEDN_LAUNCH((conv_wgrad_f32_mfma_kernel<1>), grid, dim3(256), 0, s, d, (int)ppb, ci_tiles, part);
    const long total = (long)pairs * taps * 1024;
    LEDN_LAUNCH(conv_wgrad_f32_finish_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, d, part, (int)nbx, pairs,
                ci_tiles, taps);
    int rc = check_launch();
    if (rc != LEDN_OK) return rc;
    if (d.db) rc = channel_stats_impl(d.dz, nullptr, npix, d.Cout, d.dtype_dz, d.db, nullptr, s);
    return rc;
}

}  // namespace ledn
