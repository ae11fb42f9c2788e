// conv1x1.hip -- 1x1 stride-1 convolution (forward and data gradient) as a register-direct streaming GEMM on the
// CDNA4 matrix cores (gfx950, v_mfma_f32_16x16x32_bf16).
//
// A 1x1 convolution over NHWC storage is a plain [pixels][Cin] x [Cin][Cout] product: no halo, every input row is
// read once and every output row written once -- the kernel is bound by HBM, not by the matrix cores (64 -> 64:
// 256 B per pixel against 8 KFLOP).  conv_mfma_kernel (conv_mfma.hip) runs such layers through its 3x3 machinery
// (LDS patch, workgroup barriers, per-tile prologue chain of weights -> patch -> first product); here nothing goes
// through LDS at all:
//   * B operand = pixels.  Lane l of a wave owns pixel (l & 15) of a 16-pixel group and the 8 consecutive input
//     channels 32 ks + 8 (l >> 4) .. + 7 of k-step ks: ONE 16-byte global load per fragment, straight into the MFMA
//     register layout; a wave instruction covers 16 pixels x 64 contiguous bytes (whole rows when Cin = 32).
//   * A operand = weights, resident in registers for the lifetime of the wave (64 x 64 bf16 = 32 VGPRs per lane;
//     grouped convolutions keep only the 32 x 32 diagonal blocks that are not zero).
//   * The output channels are PERMUTED inside the A fragments (a free row permutation of the weight matrix): row
//     4q + i of M-tile mt is channel 32 (mt >> 1) + 8 q + 4 (mt & 1) + i, so that lane (pixel, q) ends up with the
//     8 CONSECUTIVE channels 32 p + 8 q .. + 7 of its pixel in the accumulators of tiles 2p, 2p + 1: one 16-byte
//     store per lane and pair, the four lanes of a pixel write its 64-byte row piece.  No LDS transposition.
//   * Waves are autonomous: no barrier, no LDS in the main loop; each walks pixel-group pairs with a grid stride
//     (neighbouring waves touch neighbouring memory) and fetches the fragments of its NEXT iteration before the matrix
//     instructions of the current one, so ~3 waves per SIMD keep > 48 KB of loads in flight per CU.
//   * Epilogue flavours: raw, raw + per-channel statistics (training forward: one partial row per workgroup, the
//     deferred-rows protocol of finish_partials), raw + addend (gradient fan-in of the training backward); the bias is
//     the accumulator's initial value.
#include "regconv.h"

namespace ledn {

struct C11Args {
    const bf16_t* x;       // [P][Cin]
    const bf16_t* wp;      // [Cout][Cin] bf16 (ledn_pack_conv_weights, either mode: rows = this kernel's outputs)
    bf16_t* y;             // [P][Cout]
    const bf16_t* res;     // optional addend [P][Cout] (EPI_ACC)
    const float* bias;     // optional [Cout]
    const float* out_scale;// C11_FULL: y = act(res_mode(z * out_scale + bias, res)); NULL = 1
    const float* slope;    // [Cout] when act_out == PRELU
    int act_out, res_mode;
    const float* in_scale; // optional input prologue pre(x) = act_in(x * in_scale + in_shift) (the producer's BatchNorm +
    const float* in_shift; // activation folded into this convolution), applied to the fragments in registers
    const float* in_slope;
    int in_act;
    float* part;           // statistics: per-workgroup rows [gridDim.x][2][Cout], or NULL -> atomics
    float* stat_sum;
    float* stat_sqsum;
    long P;
    long iters;            // ceil(P / (16 G))
    int Cin, Cout;
};

constexpr int C11_RAW = 0, C11_STATS = 1, C11_ACC = 2, C11_FULL = 3;   // FULL: scale / shift, residual, activation (inference)

// NKS: k-steps of 32 input channels; NMT: M-tiles of 16 output channels; DIAG: only the 32 x 32 diagonal blocks of
// the weight matrix are non-zero (grouped convolution, Cin == Cout); G: 16-pixel groups per iteration
template <int NKS, int NMT, bool DIAG, int G, int EPI, int OCC, bool PRO = false>
__global__ void __launch_bounds__(256, OCC) conv1x1_mfma_kernel(C11Args a) {
    constexpr int NP = (NMT + 1) / 2;                        // 32-channel pairs of M-tiles (the last may be half)
    constexpr int NWF = DIAG ? NMT : NMT * NKS;              // resident weight fragments
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int Cin = a.Cin, Cout = a.Cout;

    // ---- weight fragments: A[row r = lane & 15][k = 8 (lane >> 4) + j] of (mt, ks) = W[channel(mt, r)][32 ks + 8 q + j]
    bf16x8_t wf[NWF];
#pragma unroll
    for (int f = 0; f < NWF; ++f) {
        const int mt = DIAG ? f : f / NKS, ks = DIAG ? (f >> 1) : f % NKS;
        const int co = c11_channel<NMT>(mt, pl >> 2, pl & 3);
        const int ci = 32 * ks + 8 * q;
        const bool ok = ci < Cin && co < Cout;
        uint4 v = *reinterpret_cast<const uint4*>(a.wp + (ok ? (long)co * Cin + ci : 0L));
        if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
        wf[f] = __builtin_bit_cast(bf16x8_t, v);
    }
    // ---- bias = initial accumulator (rows 4q + i of tile mt)
    float binit[NMT][4];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c11_channel<NMT>(mt, q, i);
            binit[mt][i] = (a.bias && c < Cout) ? a.bias[c] : 0.f;
        }
    // input prologue coefficients of this lane's 8 channels per k-step
    constexpr int NPR = PRO ? NKS : 1;
    float psc[NPR][8], psh[NPR][8], png[NPR][8];
    if constexpr (PRO) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 32 * ks + 8 * q + i;
                const bool ok = c < Cin;
                psc[ks][i] = (ok && a.in_scale) ? a.in_scale[c] : 1.f;
                psh[ks][i] = (ok && a.in_shift) ? a.in_shift[c] : 0.f;
                png[ks][i] = a.in_act == LEDN_ACT_PRELU ? (ok ? a.in_slope[c] : 0.f) : (a.in_act == LEDN_ACT_NONE ? 1.f : 0.f);
            }
    }
    constexpr int NST = EPI == C11_STATS ? NMT * 4 : 1;
    float st1[NST], st2[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) st1[i] = st2[i] = 0.f;

    // C11_FULL: per-channel (scale, negative slope) in LDS, read per use (the bias is already the accumulator's start
    // value -- scaled below: y = z * scale + shift = (z + shift / scale) * scale would round differently, so the shift is
    // added after the scale instead and binit stays zero)
    __shared__ float s_par[EPI == C11_FULL ? 3 * NMT * 16 : 1];
    if constexpr (EPI == C11_FULL) {
        for (int i = tid; i < NMT * 16; i += 256) {
            const bool ok = i < Cout;
            s_par[i] = (ok && a.out_scale) ? a.out_scale[i] : 1.f;
            s_par[NMT * 16 + i] = (ok && a.bias) ? a.bias[i] : 0.f;
            s_par[2 * NMT * 16 + i] = a.act_out == LEDN_ACT_PRELU ? (ok ? a.slope[i] : 0.f) : (a.act_out == LEDN_ACT_NONE ? 1.f : 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) binit[mt][i] = 0.f;
    }
    const float act_hi = a.act_out == LEDN_ACT_RELU6 ? 6.f : 3.0e38f;
    const long nwaves = (long)gridDim.x * 4;
    long it = (long)blockIdx.x * 4 + wid;
    const bool kok[2] = {true, true};
    (void)kok;

    // fragments of one iteration: G groups x NKS k-steps (one 16-byte load each; out-of-range pieces read the tensor
    // base and are zeroed)
    auto fetch = [&](long iter, bf16x8_t (&bf)[G][NKS]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long pix = (iter * G + g) * 16 + pl;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int ci = 32 * ks + 8 * q;
                const bool ok = pix < a.P && ci < Cin;
                uint4 v = *reinterpret_cast<const uint4*>(a.x + (ok ? pix * Cin + ci : 0L));
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                bf[g][ks] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
    };

    bf16x8_t bcur[G][NKS], bnext[G][NKS];
    if (it < a.iters) fetch(it, bcur);
    while (it < a.iters) {
        const long nit = it + nwaves;
        if (nit < a.iters) fetch(nit, bnext);
        // addend pieces of this iteration (EPI_ACC): in flight during the matrix instructions
        uint4 radd[G][NP];
        if (EPI == C11_ACC || (EPI == C11_FULL && a.res_mode != LEDN_RES_NONE)) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const long pix = (it * G + g) * 16 + pl;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const bool ok = pix < a.P && 32 * p + 8 * q < Cout;
                    radd[g][p] = *reinterpret_cast<const uint4*>(a.res + (ok ? pix * Cout + 32 * p + 8 * q : 0L));
                }
            }
        }
        if constexpr (PRO) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool pok = (it * G + g) * 16 + pl < a.P;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const bool ok = pok && 32 * ks + 8 * q < Cin;
                    const bf16x8_t t = c11_prologue(bcur[g][ks], psc[ks], psh[ks], png[ks]);
                    bcur[g][ks] = ok ? t : __builtin_bit_cast(bf16x8_t, make_uint4(0u, 0u, 0u, 0u));
                }
            }
        }
        sched_fence();
        f32x4_t acc[G][NMT];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[g][mt][i] = binit[mt][i];
#pragma unroll
        for (int f = 0; f < NWF; ++f) {
            const int mt = DIAG ? f : f / NKS, ks = DIAG ? (f >> 1) : f % NKS;
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][mt] = mfma_16x16x32_bf16(wf[f], bcur[g][ks], acc[g][mt]);
        }
        // ---- epilogue: lane (pixel pl, q) holds channels 32 p + 8 q .. + 7 in acc[2p], acc[2p + 1]
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long pix = (it * G + g) * 16 + pl;
            const bool pok = pix < a.P;
            if constexpr (EPI == C11_STATS) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = pok ? acc[g][mt][i] : 0.f;
                        st1[mt * 4 + i] += v;
                        st2[mt * 4 + i] = fmaf(v, v, st2[mt * 4 + i]);
                    }
            }
            c11_for<NP>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if constexpr (2 * p + 1 < NMT) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = acc[g][2 * p][i];
                        v[4 + i] = acc[g][2 * p + 1][i];
                    }
                    if constexpr (EPI == C11_ACC) {
                        float r[8];
                        ld8(reinterpret_cast<const bf16_t*>(&radd[g][p]), r);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = bf16_to_f32(f32_to_bf16(v[i])) + r[i];   // as conv_mfma: bf16(z) + addend
                    }
                    if constexpr (EPI == C11_FULL) {
                        const int c0 = 32 * p + 8 * q;
                        float sc[8], sh[8], ng[8];
                        ld8(s_par + c0, sc);
                        ld8(s_par + NMT * 16 + c0, sh);
                        ld8(s_par + 2 * NMT * 16 + c0, ng);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[i] + sh[i];
                        if (a.res_mode != LEDN_RES_NONE) {
                            float r[8];
                            ld8(reinterpret_cast<const bf16_t*>(&radd[g][p]), r);
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = a.res_mode == LEDN_RES_ADD ? v[i] + r[i] : v[i] * r[i] + r[i];
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = fminf(fmaxf(v[i], 0.f) + ng[i] * fminf(v[i], 0.f), act_hi);
                    }
                    if (pok && 32 * p + 8 * q < Cout) st8(a.y + pix * Cout + 32 * p + 8 * q, v);
                } else {                                     // unpaired tile: channels 16 mt + 4 q .. + 3, 8-byte store
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[g][2 * p][i];
                    const int c = 32 * p + 4 * q;
                    if constexpr (EPI == C11_ACC) {
                        float r[4];
                        if (pok) {
                            ld4(a.res + pix * Cout + c, r);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = bf16_to_f32(f32_to_bf16(v[i])) + r[i];
                        }
                    }
                    if constexpr (EPI == C11_FULL) {
                        float sc[4], sh[4], ng[4];
                        ld4(s_par + c, sc);
                        ld4(s_par + NMT * 16 + c, sh);
                        ld4(s_par + 2 * NMT * 16 + c, ng);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] * sc[i] + sh[i];
                        if (a.res_mode != LEDN_RES_NONE && pok) {
                            float r[4];
                            ld4(a.res + pix * Cout + c, r);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = a.res_mode == LEDN_RES_ADD ? v[i] + r[i] : v[i] * r[i] + r[i];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fminf(fmaxf(v[i], 0.f) + ng[i] * fminf(v[i], 0.f), act_hi);
                    }
                    if (pok) st4(a.y + pix * Cout + c, v);
                }
            });
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) bcur[g][ks] = bnext[g][ks];
        it = nit;
    }

    if constexpr (EPI == C11_STATS) {
        // per-wave totals -> LDS -> one row per workgroup.  16 values per reduction call (4 M-tiles); lanes with equal
        // (lane >> 4) hold the same channels of different pixels.
        __shared__ float s_st[4][2][NMT * 16];
        constexpr int CH = NMT >= 4 ? 16 : NMT * 4;          // values per reduction call
#pragma unroll
        for (int c0 = 0; c0 < NMT * 4; c0 += CH) {
            const float t1 = c11_reduce16<CH>(st1 + c0, lane), t2 = c11_reduce16<CH>(st2 + c0, lane);
            const int vi = c0 + c11_red_index<CH>(lane);     // value index = mt * 4 + i
            const int ch = c11_channel<NMT>(vi >> 2, q, vi & 3);
            if ((lane & 15) < CH) {                          // (with CH < 16 the other lanes hold copies of the same totals)
                s_st[wid][0][ch] = t1;
                s_st[wid][1][ch] = t2;
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * Cout; i += 256) {
            const int j = i / Cout, c = i % Cout;
            const float t = s_st[0][j][c] + s_st[1][j][c] + s_st[2][j][c] + s_st[3][j][c];
            if (a.part) a.part[(long)blockIdx.x * 2 * Cout + (long)j * Cout + c] = t;
            else atomicAdd((j ? a.stat_sqsum : a.stat_sum) + c, t);
        }
    }
}

template <int NKS, int NMT, bool DIAG, int EPI>
static int c11_launch(C11Args a, hipStream_t s) {
    constexpr int G = NMT >= 8 ? 1 : 2;
    constexpr int OCC = (EPI == C11_STATS || NMT * NKS > 8) ? 2 : 3;       // waves per SIMD the register budget allows
    a.iters = cdiv(a.P, 16 * G);
    long nb = cdiv(a.iters, 4);
    const long cap = (long)options().conv_workgroups * OCC;                  // default 512 x OCC: two rounds of resident workgroups
    if (nb > cap) nb = cap;
    a.part = (EPI == C11_STATS && a.stat_sum && (nb > 16 || det())) ? ws_take(nb * 2 * a.Cout) : nullptr;
    const bool pro = a.in_scale || a.in_act != LEDN_ACT_NONE;
    if constexpr (NKS <= 2) {        // (prologue coefficients: 24 VGPRs per k-step -- offered for Cin <= 64)
        if (pro) LEDN_LAUNCH((conv1x1_mfma_kernel<NKS, NMT, DIAG, G, EPI, 2, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
        else LEDN_LAUNCH((conv1x1_mfma_kernel<NKS, NMT, DIAG, G, EPI, OCC>), dim3((unsigned)nb), dim3(256), 0, s, a);
    } else {
        if (pro) return LEDN_EINVAL;
        LEDN_LAUNCH((conv1x1_mfma_kernel<NKS, NMT, DIAG, G, EPI, OCC>), dim3((unsigned)nb), dim3(256), 0, s, a);
    }
    if (a.part) return finish_partials(a.part, (int)nb, a.Cout, 2, a.stat_sum, a.stat_sqsum, nullptr, s);
    return check_launch();
}

template <int NKS, int NMT, bool DIAG>
static int c11_epi(const C11Args& a, int epi, hipStream_t s) {
    switch (epi) {
        case C11_STATS: return c11_launch<NKS, NMT, DIAG, C11_STATS>(a, s);
        case C11_ACC: return c11_launch<NKS, NMT, DIAG, C11_ACC>(a, s);
        case C11_FULL: return c11_launch<NKS, NMT, DIAG, C11_FULL>(a, s);
        default: return c11_launch<NKS, NMT, DIAG, C11_RAW>(a, s);
    }
}

static bool c11_diag(const ledn_conv_desc& d) {
    if (d.groups <= 1 || d.Cin != d.Cout) return false;
    const int cg = d.Cin / d.groups;
    return cg <= 32 && 32 % cg == 0 && d.Cin % 32 == 0;    // every group lies inside one 32 x 32 diagonal block
}

// shapes the register-resident weight fragments fit (<= 16 fragments = 64 VGPRs)
bool conv1x1_reg_supported(const ledn_conv_desc& d) {
    if (!(options().stream_fast & 16)) return false;
    if (!d.w_bf16 || d.dtype_x != LEDN_BF16 || d.dtype_y != LEDN_BF16) return false;
    if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad != 0 || d.dil != 1 || d.xadd) return false;
    if (d.in_scale || d.in_shift || d.in_act != LEDN_ACT_NONE) {       // input prologue: Cin <= 64, none / ReLU / PReLU
        if (d.Cin > 64 || (d.in_scale == nullptr) != (d.in_shift == nullptr)) return false;
        if (d.in_act != LEDN_ACT_NONE && d.in_act != LEDN_ACT_RELU && !(d.in_act == LEDN_ACT_PRELU && d.in_slope)) return false;
    }
    const bool full = d.out_scale || d.act_out != LEDN_ACT_NONE || d.res_mode == LEDN_RES_GATE ||
                      (d.res_mode == LEDN_RES_ADD && d.out_shift);
    if (full) {        // inference epilogue: scale / shift, residual, activation; no statistics
        if (d.stat_sum || d.act_out == LEDN_ACT_SIGMOID || (d.act_out == LEDN_ACT_PRELU && !d.slope)) return false;
        if (d.res_mode != LEDN_RES_NONE && !d.res) return false;
    } else if (d.res_mode != LEDN_RES_NONE && !(d.res_mode == LEDN_RES_ADD && d.res && !d.stat_sum)) {
        return false;
    }
    if (d.Cin % 16 || d.Cout % 16 || d.Cin > 128 || d.Cout > 128) return false;
    const int nks = (d.Cin + 31) / 32, nmt = d.Cout / 16;
    if (!(nks == 1 || nks == 2 || nks == 4) || !(nmt == 1 || nmt == 2 || nmt == 4 || nmt == 8)) return false;
    const int nwf = c11_diag(d) ? nmt : nmt * nks;
    return nwf <= 16;
}

int conv1x1_reg(const ledn_conv_desc& d, hipStream_t s) {
    C11Args a;
    a.x = (const bf16_t*)d.x; a.wp = (const bf16_t*)d.w_bf16; a.y = (bf16_t*)d.y;
    a.res = d.res_mode == LEDN_RES_ADD ? (const bf16_t*)d.res : nullptr;
    a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.in_slope = d.in_slope; a.in_act = d.in_act;
    a.bias = d.out_shift; a.part = nullptr; a.stat_sum = d.stat_sum; a.stat_sqsum = d.stat_sqsum;
    a.out_scale = d.out_scale; a.slope = d.slope; a.act_out = d.act_out; a.res_mode = d.res_mode;
    a.P = (long)d.N * d.H * d.W; a.iters = 0; a.Cin = d.Cin; a.Cout = d.Cout;
    const bool full = d.out_scale || d.act_out != LEDN_ACT_NONE || d.res_mode == LEDN_RES_GATE ||
                      (d.res_mode == LEDN_RES_ADD && d.out_shift);
    if (full) a.res = (const bf16_t*)d.res;
    const int epi = full ? C11_FULL : (a.res ? C11_ACC : (d.stat_sum ? C11_STATS : C11_RAW));
    const int nks = (d.Cin + 31) / 32, nmt = d.Cout / 16;
    const bool diag = c11_diag(d);
#define LEDN_C11(NKS_, NMT_)                                                     \
    if (nks == NKS_ && nmt == NMT_) {                                            \
        if constexpr (NKS_ * 2 == NMT_) {                                        \
            if (diag) return c11_epi<NKS_, NMT_, true>(a, epi, s);               \
        }                                                                        \
        if constexpr (NKS_ * NMT_ <= 16) return c11_epi<NKS_, NMT_, false>(a, epi, s); \
    }
    LEDN_C11(1, 1) LEDN_C11(1, 2) LEDN_C11(1, 4) LEDN_C11(1, 8)
    LEDN_C11(2, 1) LEDN_C11(2, 2) LEDN_C11(2, 4) LEDN_C11(2, 8)
    LEDN_C11(4, 1) LEDN_C11(4, 2) LEDN_C11(4, 4) LEDN_C11(4, 8)
#undef LEDN_C11
    return LEDN_EINVAL;
}

}  // namespace ledn
