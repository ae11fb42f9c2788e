// conv3x3.hip -- 3x3 stride-1 pad-1 convolution (forward and data gradient) for 32 input channels as a
// register-direct, wave-autonomous implicit GEMM on the CDNA4 matrix cores (gfx950, v_mfma_f32_16x16x32_bf16).
//
// conv_mfma_kernel (conv_mfma.hip) stages a patch in LDS for four waves: fetch -> commit -> barrier -> fragments +
// matrix instructions -> barrier -> stores is ONE serial chain per workgroup (profiles/r03_conv_phase_costs.txt: no
// phase owns the time, none overlaps another, the matrix pipe is busy 9-18 % of a launch).  Here a wave owns a column
// strip of 16 G output pixels and walks DOWN the rows of its segment; nothing goes through LDS and there is no
// barrier in the main loop:
//   * B operand = pixels, K = 32 input channels per matrix instruction: lane l owns pixel (l & 15) of a 16-pixel group
//     and channels 8 (l >> 4) .. + 7 -- one 16-byte global load per (input row, group, 32-channel chunk) is the whole
//     fragment, four lanes read a pixel's 64 contiguous bytes, a wave instruction 16 pixels x 64 B.
//   * The kw = 0 / 2 taps are the SAME fragment shifted by one pixel: DPP row_shr:1 / row_shl:1 move it by one lane inside
//     the 16-lane rows (= pixels of equal channel octet); the lane that has no source keeps `old` = the halo pixel, which
//     a second load instruction puts into lanes 0 (pixel x0 - 1) and 15 (pixel x0 + 16).  8 v_mov_dpp per fragment
//     instead of two more loads.
//   * One input row feeds the three output rows r - 1, r, r + 1 (kh = 2, 1, 0): three rolling accumulator sets; after
//     input row r the set of output row r - 1 is complete and goes through the epilogue.  The row loop is unrolled by
//     three so that every register index is static.
//   * A operand = weights, resident in registers for the lifetime of the wave: 9 taps x 2 M-tiles (32 output channels per
//     workgroup slice, blockIdx.y) x Cin / 32 chunks = 18 (36) fragments of 4 VGPRs.  Output channels permuted inside the
//     fragments as in conv1x1.hip: lane (pixel, q) ends up with channels 8 q .. + 7 -> one 16-byte store.
//   * The loads of input row r + 3 are issued when row r is consumed (three rows = 6 KB per wave in flight); the first
//     three rows of a wave's first task are requested before its weights.
//   * Segments of RS output rows re-read 2 halo rows (L2 hits of the neighbouring segment's rows).
// Epilogue flavours as conv1x1.hip: raw, raw + per-channel statistics, raw + addend (gradient fan-in), full (scale /
// shift, residual add | gate, activation: inference).
#include <cstdlib>
#include "regconv.h"

namespace ledn {

struct C33Args {
    const bf16_t* x;       // [N][H][W][Cin]
    const bf16_t* wp;      // [9][Cout][Cin] bf16 (ledn_pack_conv_weights, either mode: rows = this kernel's outputs)
    bf16_t* y;             // [N][H][W][Cout]
    const bf16_t* res;     // addend (C33_ACC) / residual (C33_FULL)
    const float* bias;
    const float* out_scale;
    const float* slope;
    int act_out, res_mode;
    const float* in_scale; // C33_NARROW with PRO: input prologue pre(x) = act_in(x * in_scale + in_shift) (the producer's
    const float* in_shift; // BatchNorm + activation folded into this convolution; the zero padding is applied AFTER it)
    const float* in_slope;
    int in_act, y_f32;
    float* part;           // statistics: per-workgroup rows [gridDim.y][gridDim.x][2][32] -> see c33_launch
    float* stat_sum;
    float* stat_sqsum;
    int N, H, W, Cin, Cout;
    int strips, segs, RS;  // column strips of 16 G pixels, row segments of RS rows per image
    long tasks;            // N * segs * strips
};

constexpr int C33_RAW = 0, C33_STATS = 1, C33_ACC = 2, C33_FULL = 3;
constexpr int C33_NARROW = 4;   // Cout <= 4 (the two-class heads): one M-tile, rows 0..3 of lane group q = 0, scalar stores, f32 | bf16

// v_mov_b32_dpp row_shr:1 / row_shl:1 with bound_ctrl off: lane i of a 16-lane row takes src of lane i -/+ 1; the lane
// without a source keeps `old`
__device__ __forceinline__ unsigned c33_row_shift(unsigned old, unsigned src, bool right) {
#ifdef LEDN_CPU_EMU
    const int lane = lane_id(), pl = lane & 15;
    const bool has = right ? pl > 0 : pl < 15;
    const unsigned got = __shfl(src, has ? (right ? lane - 1 : lane + 1) : lane);
    return has ? got : old;
#else
    return right ? (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x111, 0xf, 0xf, false)
                 : (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x101, 0xf, 0xf, false);
#endif
}
// row_ror:1 (right: lane i takes lane i - 1, lane 0 takes lane 15) / row_ror:15 (left: lane 15 takes lane 0)
__device__ __forceinline__ unsigned c33_row_rot(unsigned src, bool right) {
#ifdef LEDN_CPU_EMU
    const int lane = lane_id(), pl = lane & 15;
    return __shfl(src, (lane & ~15) | ((pl + (right ? 15 : 1)) & 15));
#else
    return right ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)src, 0x121, 0xf, 0xf, false)
                 : (unsigned)__builtin_amdgcn_update_dpp(0, (int)src, 0x12f, 0xf, 0xf, false);
#endif
}
__device__ __forceinline__ uint4 c33_rotate(uint4 v, bool right) {
    return make_uint4(c33_row_rot(v.x, right), c33_row_rot(v.y, right), c33_row_rot(v.z, right), c33_row_rot(v.w, right));
}
__device__ __forceinline__ bf16x8_t c33_shift(uint4 halo, uint4 main, bool right) {
    return __builtin_bit_cast(bf16x8_t, make_uint4(c33_row_shift(halo.x, main.x, right), c33_row_shift(halo.y, main.y, right),
                                                   c33_row_shift(halo.z, main.z, right), c33_row_shift(halo.w, main.w, right)));
}

// NKC: 32-channel chunks of the input; G: 16-pixel groups side by side
template <int NKC, int G, int EPI, int OCC, int NMT = 2, bool PRO = false>
__global__ void __launch_bounds__(256, OCC) conv3x3_reg_kernel(C33Args a) {
    static_assert((EPI == C33_NARROW) == (NMT == 1), "one M-tile = the narrow heads");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int Cin = a.Cin, Cout = a.Cout, H = a.H, W = a.W;
    const int co0 = blockIdx.y * 32;

    // this lane's 8 output channels co0 + 8 q .. + 7: acc[0][i] = channel 8 q + i, acc[1][i] = 8 q + 4 + i
    const int cl = 8 * q, cg = co0 + cl;
    const bool c_ok = cg < Cout;
    // C33_FULL: per-channel (scale, shift, negative slope) of the slice in LDS, read per use; the other flavours add the
    // bias (rare in front of a BatchNorm) in the epilogue straight from memory -- no registers held across the row loop
    __shared__ float s_par[EPI == C33_FULL ? 96 : 1];
    if constexpr (EPI == C33_FULL) {
        if (tid < 32) {
            const bool ok = co0 + tid < Cout;
            s_par[tid] = (a.out_scale && ok) ? a.out_scale[co0 + tid] : 1.f;
            s_par[32 + tid] = (a.bias && ok) ? a.bias[co0 + tid] : 0.f;
            s_par[64 + tid] = a.act_out == LEDN_ACT_PRELU ? (ok ? a.slope[co0 + tid] : 0.f) : (a.act_out == LEDN_ACT_NONE ? 1.f : 0.f);
        }
        __syncthreads();
    }
    const float act_hi = a.act_out == LEDN_ACT_RELU6 ? 6.f : 3.0e38f;
    constexpr int NST = EPI == C33_STATS ? 8 : 1;
    float st1[NST], st2[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) st1[i] = st2[i] = 0.f;

    // input prologue coefficients of this lane's 8 input channels per chunk
    constexpr int NPR = PRO ? NKC : 1;
    float psc[NPR][8], psh[NPR][8], png[NPR][8];
    if constexpr (PRO) {
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 32 * kc + 8 * q + i;
                const bool ok = c < Cin;
                psc[kc][i] = (ok && a.in_scale) ? a.in_scale[c] : 1.f;
                psh[kc][i] = (ok && a.in_shift) ? a.in_shift[c] : 0.f;
                png[kc][i] = a.in_act == LEDN_ACT_PRELU ? (ok ? a.in_slope[c] : 0.f) : (a.in_act == LEDN_ACT_NONE ? 1.f : 0.f);
            }
    }
    const long nwaves = (long)gridDim.x * 4;
    struct Task { int n, x0, r0, r1; };
    auto geom = [&](long task) {
        Task t;
        const int strip = (int)(task % a.strips);
        const int seg = (int)((task / a.strips) % a.segs);
        t.n = (int)(task / ((long)a.strips * a.segs));
        t.x0 = strip * 16 * G;
        t.r0 = seg * a.RS;
        t.r1 = min(t.r0 + a.RS, H);
        return t;
    };
    // raw loads of one input row: [g][kc] = this lane's pixel of group g; [G][kc] = the strip's outer halo (lane 0:
    // pixel x0 - 1, lane 15: pixel x0 + 16 G); the halo between two groups comes from the neighbouring group's
    // fragment (row rotation)
    auto fetch = [&](const Task& t, int ir, uint4 (&rw)[G + 1][NKC]) {
        const bf16_t* xn = a.x + (long)t.n * H * W * Cin;
        const bool rok = ir >= 0 && ir < H;
        const int hx = pl == 0 ? t.x0 - 1 : t.x0 + 16 * G;                // (lanes 1..14: never used)
        const bool hok = rok && (pl == 0 || pl == 15) && hx >= 0 && hx < W;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int ci = 32 * kc + 8 * q;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int px = t.x0 + 16 * g + pl;
                const bool mok = rok && px < W;
                uint4 m = *reinterpret_cast<const uint4*>(xn + (mok ? ((long)ir * W + px) * Cin + ci : 0L));
                if (!mok) m = make_uint4(0u, 0u, 0u, 0u);
                rw[g][kc] = m;
            }
            uint4 h = *reinterpret_cast<const uint4*>(xn + (hok ? ((long)ir * W + hx) * Cin + ci : 0L));
            if (!hok) h = make_uint4(0u, 0u, 0u, 0u);
            rw[G][kc] = h;
        }
    };

    // the first three input rows of the first task are requested BEFORE the weights (18 / 36 fragments from L2): the
    // HBM round trip of the activations and the L2 round trips of the weights overlap instead of adding up
    const long first = (long)blockIdx.x * 4 + wid;
    long task = first;
    uint4 raw[3][G + 1][NKC];
    if (task < a.tasks) {
        const Task t = geom(task);
        fetch(t, t.r0 - 1, raw[0]);
        fetch(t, t.r0, raw[1]);
        fetch(t, t.r0 + 1, raw[2]);
    }
    // ---- weight fragments: A[row r = lane & 15][k = 8 q + j] of (tap, kc, mt) = W[tap][co0 + channel(mt, r)][32 kc + 8 q + j]
    bf16x8_t wf[9][NKC][NMT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int co = co0 + c11_channel<NMT>(mt, pl >> 2, pl & 3);
                const int ci = 32 * kc + 8 * q;
                const bool ok = co < Cout && ci < Cin;
                uint4 v = *reinterpret_cast<const uint4*>(a.wp + (ok ? ((long)t * Cout + co) * Cin + ci : 0L));
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                wf[t][kc][mt] = __builtin_bit_cast(bf16x8_t, v);
            }

    for (; task < a.tasks; task += nwaves) {
        const Task tk = geom(task);
        const int n = tk.n, x0 = tk.x0, r0 = tk.r0, r1 = tk.r1;
        if (task != first) {
            fetch(tk, r0 - 1, raw[0]);
            fetch(tk, r0, raw[1]);
            fetch(tk, r0 + 1, raw[2]);
        }
        f32x4_t acc[3][G][NMT];

        // epilogue of output row `o` (accumulator slot s)
        auto finish = [&](int o, f32x4_t (&ac)[G][NMT]) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int px = x0 + 16 * g + pl;
                if constexpr (EPI == C33_NARROW) {                  // lanes q = 0: channels 0 .. Cout - 1 of pixel px
                    if (q == 0 && px < W) {
                        const long o0 = (((long)n * H + o) * W + px) * Cout;
                        float vv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) vv[i] = ac[g][0][i] + ((a.bias && i < Cout) ? a.bias[i] : 0.f);
                        if (Cout == 2) {       // the two-class heads: ONE store per pixel (sub-dword stores are partial writes)
                            if (a.y_f32) *reinterpret_cast<float2*>(reinterpret_cast<float*>(a.y) + o0) = make_float2(vv[0], vv[1]);
                            else *reinterpret_cast<unsigned*>(a.y + o0) = (unsigned)f32_to_bf16(vv[0]) | ((unsigned)f32_to_bf16(vv[1]) << 16);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < Cout) {
                                    if (a.y_f32) reinterpret_cast<float*>(a.y)[o0 + i] = vv[i];
                                    else st(a.y + o0 + i, vv[i]);
                                }
                        }
                    }
                    continue;
                }
                const bool pok = px < W && c_ok;
                const long off = (((long)n * H + o) * W + px) * Cout + cg;
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = ac[g][0][i];
                    v[4 + i] = ac[g][NMT - 1][i];
                }
                if (EPI != C33_FULL && a.bias) {                    // (wave-uniform)
                    float bb[8];
                    ld8(a.bias + (c_ok ? cg : 0), bb);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += bb[i];
                }
                if constexpr (EPI == C33_STATS) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float vm = px < W ? v[i] : 0.f;
                        st1[i] += vm;
                        st2[i] = fmaf(vm, vm, st2[i]);
                    }
                }
                if constexpr (EPI == C33_ACC) {
                    float r[8];
                    const uint4 rv = *reinterpret_cast<const uint4*>(a.res + (pok ? off : 0L));
                    ld8(reinterpret_cast<const bf16_t*>(&rv), r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = bf16_to_f32(f32_to_bf16(v[i])) + r[i];   // as conv_mfma: bf16(z) + addend
                }
                if constexpr (EPI == C33_FULL) {
                    float fsc[8], fsh[8], fng[8];
                    ld8(s_par + cl, fsc);
                    ld8(s_par + 32 + cl, fsh);
                    ld8(s_par + 64 + cl, fng);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = v[i] * fsc[i] + fsh[i];
                    if (a.res_mode != LEDN_RES_NONE) {
                        float r[8];
                        const uint4 rv = *reinterpret_cast<const uint4*>(a.res + (pok ? off : 0L));
                        ld8(reinterpret_cast<const bf16_t*>(&rv), r);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = a.res_mode == LEDN_RES_ADD ? v[i] + r[i] : v[i] * r[i] + r[i];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fminf(fmaxf(v[i], 0.f) + fng[i] * fminf(v[i], 0.f), act_hi);
                }
                if (pok) st8(a.y + off, v);
            }
        };

        // input row ir = r0 - 1 + 3 b + j feeds output rows ir + 1 (kh = 0, slot j: its FIRST contribution), ir (kh = 1,
        // slot (j + 2) % 3) and ir - 1 (kh = 2, slot (j + 1) % 3: complete afterwards)
        for (int base = r0 - 1; base <= r1; base += 3) {
            c11_for<3>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int ir = base + j;
                if (ir > r1) return;                                 // wave-uniform
                bf16x8_t bf[3][G][NKC];                              // [kw]
                uint4 rm[G + 1][NKC];                                // this row's pixels (after the input prologue)
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
                    for (int g = 0; g <= G; ++g) rm[g][kc] = raw[j][g][kc];
                if constexpr (PRO) {                                 // pre(x) on the loaded pixels; padding stays zero
                    const bool rok = ir >= 0 && ir < H;
                    const int hx = pl == 0 ? x0 - 1 : x0 + 16 * G;
                    const bool hok = rok && (pl == 0 || pl == 15) && hx >= 0 && hx < W;
                    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
                        for (int g = 0; g <= G; ++g) {
                            const bool ok = g == G ? hok : (rok && x0 + 16 * g + pl < W);
                            uint4 t = __builtin_bit_cast(uint4, c11_prologue(__builtin_bit_cast(bf16x8_t, rm[g][kc]), psc[kc], psh[kc], png[kc]));
                            if (!ok) t = zero;
                            rm[g][kc] = t;
                        }
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        const uint4 m = rm[g][kc];
                        // left neighbour of pixel 0 / right neighbour of pixel 15: the outer halo, or lane 15 / lane 0 of
                        // the neighbouring group's fragment (rotated into place)
                        const uint4 hl = g == 0 ? rm[G][kc] : c33_rotate(rm[g > 0 ? g - 1 : 0][kc], true);
                        const uint4 hr = g == G - 1 ? rm[G][kc] : c33_rotate(rm[g < G - 1 ? g + 1 : g][kc], false);
                        bf[0][g][kc] = c33_shift(hl, m, true);
                        bf[1][g][kc] = __builtin_bit_cast(bf16x8_t, m);
                        bf[2][g][kc] = c33_shift(hr, m, false);
                    }
                if (ir + 3 <= r1) fetch(tk, ir + 3, raw[j]);
                sched_fence();
                constexpr int s0 = j, s1 = (j + 2) % 3, s2 = (j + 1) % 3;
                if (ir + 1 < r1) {                                   // kh = 0 -> output row ir + 1 starts here
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[s0][g][0][i] = 0.f;
                            acc[s0][g][NMT - 1][i] = 0.f;
                        }
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
                            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                                for (int g = 0; g < G; ++g)
                                    acc[s0][g][mt] = mfma_16x16x32_bf16(wf[kw][kc][mt], bf[kw][g][kc], acc[s0][g][mt]);
                }
                if (ir >= r0 && ir < r1) {                           // kh = 1 -> output row ir
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
                            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                                for (int g = 0; g < G; ++g)
                                    acc[s1][g][mt] = mfma_16x16x32_bf16(wf[3 + kw][kc][mt], bf[kw][g][kc], acc[s1][g][mt]);
                }
                if (ir - 1 >= r0) {                                  // kh = 2 -> output row ir - 1, complete
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
                            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                                for (int g = 0; g < G; ++g)
                                    acc[s2][g][mt] = mfma_16x16x32_bf16(wf[6 + kw][kc][mt], bf[kw][g][kc], acc[s2][g][mt]);
                    finish(ir - 1, acc[s2]);
                }
            });
        }
    }

    if constexpr (EPI == C33_STATS) {
        // per-wave totals -> LDS -> one row of this slice's 32 channels per workgroup (lanes with equal q hold the same
        // channels of different pixels)
        __shared__ float s_st[4][2][32];
        const float t1 = c11_reduce16<8>(st1, lane), t2 = c11_reduce16<8>(st2, lane);
        const int vi = c11_red_index<8>(lane);
        if ((lane & 15) < 8) {
            s_st[wid][0][cl + vi] = t1;
            s_st[wid][1][cl + vi] = t2;
        }
        __syncthreads();
        if (tid < 64) {
            const int jj = tid >> 5, c = tid & 31;
            const float t = (s_st[0][jj][c] + s_st[1][jj][c]) + (s_st[2][jj][c] + s_st[3][jj][c]);
            if (co0 + c < Cout) {
                if (a.part) a.part[(long)blockIdx.x * 2 * Cout + (long)jj * Cout + co0 + c] = t;
                else atomicAdd((jj ? a.stat_sqsum : a.stat_sum) + co0 + c, t);
            }
        }
    }
}

// rows per segment: the longest of 32 / 16 / 8 that still yields one task per resident wave (2 per SIMD); a task
// re-reads 2 halo rows and starts with an exposed load round trip, so fewer, longer tasks win once the chip is full
static int c33_rows(long columns, int H) {
    static const int forced = (int)exp_knob("LEDN_C33_RS", 0);      // (A/B knob)
    if (forced > 0) return forced;
    if (columns * cdiv(H, 32) >= 2048) return 32;
    if (columns * cdiv(H, 16) >= 1024) return 16;
    return 8;
}

template <int NKC, int G, int EPI>
static int c33_launch(C33Args a, hipStream_t s) {
    constexpr int OCC = NKC == 1 ? 2 : 1;
    a.strips = (int)cdiv(a.W, 16 * G);
    const int slices = a.Cout / 32;
    // segments of 16 rows (12.5 % halo rows); smaller maps: as many 8-row segments as fill the chip
    a.RS = c33_rows((long)a.N * a.strips * slices, a.H);
    a.segs = (int)cdiv(a.H, a.RS);
    a.tasks = (long)a.N * a.segs * a.strips;
    long nb = cdiv(a.tasks, 4);
    const long cap = cdiv((long)options().conv_workgroups * OCC, slices);
    if (nb > cap) nb = cap;
    a.part = (EPI == C33_STATS && a.stat_sum && (nb > 16 || det())) ? ws_take(nb * 2 * a.Cout) : nullptr;
    LEDN_LAUNCH((conv3x3_reg_kernel<NKC, G, EPI, OCC>), dim3((unsigned)nb, (unsigned)slices), dim3(256), 0, s, a);
    if (a.part) return finish_partials(a.part, (int)nb, a.Cout, 2, a.stat_sum, a.stat_sqsum, nullptr, s);
    return check_launch();
}

template <int NKC, int G>
static int c33_epi(const C33Args& a, int epi, hipStream_t s) {
    switch (epi) {
        case C33_STATS: return c33_launch<NKC, G, C33_STATS>(a, s);
        case C33_ACC: return c33_launch<NKC, G, C33_ACC>(a, s);
        case C33_FULL: return c33_launch<NKC, G, C33_FULL>(a, s);
        default: return c33_launch<NKC, G, C33_RAW>(a, s);
    }
}

template <int EPI>
static int c33_ring64_launch(C33Args a, hipStream_t s);     // (64 input channels: below)

static bool c33_full(const ledn_conv_desc& d) {
    return d.out_scale || d.act_out != LEDN_ACT_NONE || d.res_mode == LEDN_RES_GATE || (d.res_mode == LEDN_RES_ADD && d.out_shift);
}

// the two-class heads (led_head.py:44-51: norm -> act -> 3x3 conv 32 -> 2, f32 logits): Cout <= 4, optional input
// prologue (the BatchNorm + ReLU in front), optional bias, no statistics / residual / activation behind
static bool c33_narrow(const ledn_conv_desc& d) {
    // opt-in (LEDN_OPT_STREAM_FAST bit 7): measured 145-155 us against conv_mfma_kernel's narrow epilogue at 123-132 us
    // (16 x 512 x 512, prologue + bias, r03) -- correct and tested, not the default
    if (!(options().stream_fast & 128)) return false;
    if (d.Cout > 4 || d.Cin != 32 || d.transposed) return false;
    if (d.dtype_y != LEDN_BF16 && d.dtype_y != LEDN_F32) return false;
    if ((d.in_scale == nullptr) != (d.in_shift == nullptr)) return false;
    if (d.in_act != LEDN_ACT_NONE && d.in_act != LEDN_ACT_RELU && !(d.in_act == LEDN_ACT_PRELU && d.in_slope)) return false;
    return !d.out_scale && !d.stat_sum && d.res_mode == LEDN_RES_NONE && d.act_out == LEDN_ACT_NONE;
}

// 3x3, stride 1, pad 1 (forward, or the data gradient of such a layer: the same correlation with the flipped /
// transposed weight pack), 32 or 64 input channels, output channels a multiple of 32
bool conv3x3_reg_supported(const ledn_conv_desc& d) {
    if (!(options().stream_fast & 64)) return false;
    if (!d.w_bf16 || d.dtype_x != LEDN_BF16) return false;
    if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1 || d.groups != 1 || d.xadd) return false;
    if (d.Ho != d.H || d.Wo != d.W) return false;
    if (d.dtype_y != LEDN_BF16 && !c33_narrow(d)) return false;
    if (c33_narrow(d)) return true;
    if (d.in_scale || d.in_shift || d.in_act != LEDN_ACT_NONE) return false;
    // 64 input channels (36 weight fragments = 144 VGPRs): in the register form one wave per SIMD, 60 us against
    // conv_mfma_kernel's 37 us at 16 x 128 x 128 (r03) -- offered only as conv3x3_ring64_kernel (bit 8)
    if (d.Cin == 64 && !(options().stream_fast & 256)) return false;
    if ((d.Cin != 32 && d.Cin != 64) || d.Cout % 32 || d.Cout > 128) return false;
    if (d.Ho != d.H || d.Wo != d.W) return false;
    if (c33_full(d)) {
        if (d.stat_sum || d.act_out == LEDN_ACT_SIGMOID || (d.act_out == LEDN_ACT_PRELU && !d.slope)) return false;
        if (d.res_mode != LEDN_RES_NONE && !d.res) return false;
    } else if (d.res_mode != LEDN_RES_NONE && !(d.res_mode == LEDN_RES_ADD && d.res && !d.stat_sum)) {
        return false;
    }
    return true;
}

int conv3x3_reg(const ledn_conv_desc& d, hipStream_t s) {
    C33Args a;
    a.x = (const bf16_t*)d.x; a.wp = (const bf16_t*)d.w_bf16; a.y = (bf16_t*)d.y;
    a.res = (const bf16_t*)d.res; a.bias = d.out_shift; a.out_scale = d.out_scale; a.slope = d.slope;
    a.act_out = d.act_out; a.res_mode = d.res_mode;
    a.part = nullptr; a.stat_sum = d.stat_sum; a.stat_sqsum = d.stat_sqsum;
    a.N = d.N; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Cout = d.Cout;
    a.strips = a.segs = a.RS = 0; a.tasks = 0;
    a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.in_slope = d.in_slope; a.in_act = d.in_act;
    a.y_f32 = d.dtype_y == LEDN_F32;
    if (c33_narrow(d)) {
        constexpr int G = 2;
        a.strips = (int)cdiv(a.W, 16 * G);
        a.RS = c33_rows((long)a.N * a.strips, a.H);
        a.segs = (int)cdiv(a.H, a.RS);
        a.tasks = (long)a.N * a.segs * a.strips;
        long nb = cdiv(a.tasks, 4);
        const bool pro = d.in_scale || d.in_act != LEDN_ACT_NONE;
        const long cap = (long)options().conv_workgroups * (pro ? 2 : 3);
        if (nb > cap) nb = cap;
        if (pro)
            LEDN_LAUNCH((conv3x3_reg_kernel<1, G, C33_NARROW, 2, 1, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
        else
            LEDN_LAUNCH((conv3x3_reg_kernel<1, G, C33_NARROW, 3, 1, false>), dim3((unsigned)nb), dim3(256), 0, s, a);
        return check_launch();
    }
    const int epi = c33_full(d) ? C33_FULL : (d.res_mode == LEDN_RES_ADD ? C33_ACC : (d.stat_sum ? C33_STATS : C33_RAW));
    if (d.Cin == 64) {
        switch (epi) {
            case C33_STATS: return c33_ring64_launch<C33_STATS>(a, s);
            case C33_ACC: return c33_ring64_launch<C33_ACC>(a, s);
            case C33_FULL: return c33_ring64_launch<C33_FULL>(a, s);
            default: return c33_ring64_launch<C33_RAW>(a, s);
        }
    }
    return c33_epi<1, 2>(a, epi, s);
}

// ---------------------------------------------------------------------------
// The same convolution for 64 INPUT channels (36 weight fragments = 144 VGPRs): the input rows of a wave's 16-pixel strip go
// through a wave-private LDS ring instead of registers -- one row of 18 pixels x 128 B per iteration (16-byte pieces, next
// row's loads in flight), the B fragment of tap (kh, kw) and chunk kc is ONE ds_read_b128 at pixel offset kw of ring row
// kh (no DPP, no halo registers, no register prefetch ring) -- which is what lets the weights stay resident next to two
// waves per SIMD.  Raw / + statistics / + addend / full epilogues as conv3x3_reg_kernel; 32 output channels per workgroup
// slice.  Opt-in (LEDN_OPT_STREAM_FAST bit 8): measured 35-41 us against conv_mfma_kernel's 31-37 us at 16 x 128 x 128
// (r03; neither two rows in flight nor the occupancy moved it) -- correct and tested, not the default.
// ---------------------------------------------------------------------------
constexpr int R64_PW = 18, R64_PIXB = 144;      // ring row: pixels x0 - 1 .. x0 + 16, 128 B + 16 pad per pixel

template <int EPI>
__global__ void __launch_bounds__(256, 2) conv3x3_ring64_kernel(C33Args a) {
    constexpr int NKC = 2, NMT = 2;
    __shared__ __attribute__((aligned(16))) unsigned char s_ring[4][3][R64_PW * R64_PIXB];
    __shared__ float s_par[EPI == C33_FULL ? 96 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int Cout = a.Cout, H = a.H, W = a.W;
    const int co0 = blockIdx.y * 32;
    unsigned char* ring = &s_ring[wid][0][0];
    if constexpr (EPI == C33_FULL) {
        if (tid < 32) {
            const bool ok = co0 + tid < Cout;
            s_par[tid] = (a.out_scale && ok) ? a.out_scale[co0 + tid] : 1.f;
            s_par[32 + tid] = (a.bias && ok) ? a.bias[co0 + tid] : 0.f;
            s_par[64 + tid] = a.act_out == LEDN_ACT_PRELU ? (ok ? a.slope[co0 + tid] : 0.f) : (a.act_out == LEDN_ACT_NONE ? 1.f : 0.f);
        }
        __syncthreads();
    }
    const int cl = 8 * q, cg = co0 + cl;
    const bool c_ok = cg < Cout;
    const float act_hi = a.act_out == LEDN_ACT_RELU6 ? 6.f : 3.0e38f;
    constexpr int NST = EPI == C33_STATS ? 8 : 1;
    float st1[NST], st2[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) st1[i] = st2[i] = 0.f;
    bf16x8_t wf[9][NKC][NMT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int co = co0 + c11_channel<NMT>(mt, pl >> 2, pl & 3);
                const bool ok = co < Cout;
                uint4 v = *reinterpret_cast<const uint4*>(a.wp + (ok ? ((long)t * Cout + co) * 64 + 32 * kc + 8 * q : 0L));
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                wf[t][kc][mt] = __builtin_bit_cast(bf16x8_t, v);
            }
    // load role: piece e = lane + 64 t (t = 0..2) of a ring row (18 pixels x 8 pieces = 144): pixel e >> 3, piece e & 7
    const long nwaves = (long)gridDim.x * 4;
    for (long task = (long)blockIdx.x * 4 + wid; task < a.tasks; task += nwaves) {
        const int strip = (int)(task % a.strips);
        const int seg = (int)((task / a.strips) % a.segs);
        const int n = (int)(task / ((long)a.strips * a.segs));
        const int x0 = strip * 16, r0 = seg * a.RS, r1 = min(r0 + a.RS, H);
        const bf16_t* xn = a.x + (long)n * H * W * 64;
        auto fetch_x = [&](int ir, uint4 (&rw)[3]) {
            const bool rok = ir >= 0 && ir < H;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int e = lane + 64 * t, px = x0 - 1 + (e >> 3);
                const bool ok = rok && e < R64_PW * 8 && px >= 0 && px < W;
                uint4 v = *reinterpret_cast<const uint4*>(xn + (ok ? ((long)ir * W + px) * 64 + 8 * (e & 7) : 0L));
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                rw[t] = v;
            }
        };
        auto commit_x = [&](int slot, const uint4 (&rw)[3]) {
            unsigned char* row = ring + slot * (R64_PW * R64_PIXB);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int e = lane + 64 * t;
                if (e < R64_PW * 8) *reinterpret_cast<uint4*>(row + (e >> 3) * R64_PIXB + (e & 7) * 16) = rw[t];
            }
        };
        auto finish = [&](int o, f32x4_t (&ac)[NMT]) {
            const int px = x0 + pl;
            const bool pok = px < W && c_ok;
            const long off = (((long)n * H + o) * W + px) * Cout + cg;
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = ac[0][i];
                v[4 + i] = ac[1][i];
            }
            if (EPI != C33_FULL && a.bias) {
                float bb[8];
                ld8(a.bias + (c_ok ? cg : 0), bb);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += bb[i];
            }
            if constexpr (EPI == C33_STATS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float vm = px < W ? v[i] : 0.f;
                    st1[i] += vm;
                    st2[i] = fmaf(vm, vm, st2[i]);
                }
            }
            if constexpr (EPI == C33_ACC) {
                float r[8];
                const uint4 rv = *reinterpret_cast<const uint4*>(a.res + (pok ? off : 0L));
                ld8(reinterpret_cast<const bf16_t*>(&rv), r);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = bf16_to_f32(f32_to_bf16(v[i])) + r[i];
            }
            if constexpr (EPI == C33_FULL) {
                float fsc[8], fsh[8], fng[8];
                ld8(s_par + cl, fsc);
                ld8(s_par + 32 + cl, fsh);
                ld8(s_par + 64 + cl, fng);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * fsc[i] + fsh[i];
                if (a.res_mode != LEDN_RES_NONE) {
                    float r[8];
                    const uint4 rv = *reinterpret_cast<const uint4*>(a.res + (pok ? off : 0L));
                    ld8(reinterpret_cast<const bf16_t*>(&rv), r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = a.res_mode == LEDN_RES_ADD ? v[i] + r[i] : v[i] * r[i] + r[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fminf(fmaxf(v[i], 0.f) + fng[i] * fminf(v[i], 0.f), act_hi);
            }
            if (pok) st8(a.y + off, v);
        };
        // ring slot of image row ir: (ir - (r0 - 1)) % 3; output row o needs rows o - 1, o, o + 1
        uint4 xr[3];                                             // (two rows in flight instead of one measured the same)
        fetch_x(r0 - 1, xr);
        commit_x(0, xr);
        fetch_x(r0, xr);
        commit_x(1, xr);
        fetch_x(r0 + 1, xr);
        int slot_new = 2;
        for (int o = r0; o < r1; ++o) {
            wave_sync();
            commit_x(slot_new, xr);
            if (o + 1 < r1) fetch_x(o + 2, xr);
            wave_sync();
            f32x4_t acc[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int s_top = slot_new == 2 ? 0 : slot_new + 1;
            // 18 product steps (kh, kw, kc): the fragment of step i + 1 (one ds_read_b128) is requested BEFORE the two matrix
            // instructions of step i -- read right in front of its use, every step paid the LDS round trip
            const unsigned char* rows[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                int sl = s_top + kh;
                sl = sl >= 3 ? sl - 3 : sl;
                rows[kh] = ring + sl * (R64_PW * R64_PIXB) + pl * R64_PIXB + 16 * q;
            }
            auto frag = [&](int i) {                             // i = (kh * 3 + kw) * 2 + kc: ring pixel pl + kw, channels 32 kc + 8 q ..
                const int t = i >> 1, kc = i & 1, kh = t / 3, kw = t - 3 * kh;
                return *reinterpret_cast<const bf16x8_t*>(rows[kh] + kw * R64_PIXB + kc * 64);
            };
            bf16x8_t b_next = frag(0);
            c11_for<18>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const bf16x8_t b = b_next;
                if constexpr (i + 1 < 18) b_next = frag(i + 1);
                sched_fence();
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) acc[mt] = mfma_16x16x32_bf16(wf[i >> 1][i & 1][mt], b, acc[mt]);
                sched_fence();
            });
            finish(o, acc);
            slot_new = slot_new == 2 ? 0 : slot_new + 1;
        }
        wave_sync();
    }
    if constexpr (EPI == C33_STATS) {
        __shared__ float s_st[4][2][32];
        const float t1 = c11_reduce16<8>(st1, lane), t2 = c11_reduce16<8>(st2, lane);
        const int vi = c11_red_index<8>(lane);
        if ((lane & 15) < 8) {
            s_st[wid][0][cl + vi] = t1;
            s_st[wid][1][cl + vi] = t2;
        }
        __syncthreads();
        if (tid < 64) {
            const int jj = tid >> 5, c = tid & 31;
            const float t = (s_st[0][jj][c] + s_st[1][jj][c]) + (s_st[2][jj][c] + s_st[3][jj][c]);
            if (co0 + c < Cout) {
                if (a.part) a.part[(long)blockIdx.x * 2 * Cout + (long)jj * Cout + co0 + c] = t;
                else atomicAdd((jj ? a.stat_sqsum : a.stat_sum) + co0 + c, t);
            }
        }
    }
}

template <int EPI>
static int c33_ring64_launch(C33Args a, hipStream_t s) {
    a.strips = (int)cdiv(a.W, 16);
    const int slices = a.Cout / 32;
    a.RS = c33_rows((long)a.N * a.strips * slices, a.H);
    a.segs = (int)cdiv(a.H, a.RS);
    a.tasks = (long)a.N * a.segs * a.strips;
    long nb = cdiv(a.tasks, 4);
    const long cap = cdiv((long)options().conv_workgroups * 2, slices);
    if (nb > cap) nb = cap;
    a.part = (EPI == C33_STATS && a.stat_sum && (nb > 16 || det())) ? ws_take(nb * 2 * a.Cout) : nullptr;
    LEDN_LAUNCH((conv3x3_ring64_kernel<EPI>), dim3((unsigned)nb, (unsigned)slices), dim3(256), 0, s, a);
    if (a.part) return finish_partials(a.part, (int)nb, a.Cout, 2, a.stat_sum, a.stat_sqsum, nullptr, s);
    return check_launch();
}

// ---------------------------------------------------------------------------
// 3x3 convolution with TWO input channels and 32 k output channels (the data gradient of LEDHead's 32 -> 2 heads,
// led_head.py:47-48: dz [., 2] -> dx [., 32], 268 MB written at 16 x 512 x 512): the whole 3 x 3 x 2 neighbourhood of a
// pixel is ONE K = 32 fragment (k = 2 tap + ci, 18 used), so a 16-pixel group costs one matrix instruction per 16
// output channels instead of 576 multiply-adds per pixel on the vector units (conv_narrowin_kernel: 145 us, VALU-bound).
// Lane (pixel pl, q) gathers taps 4 q .. 4 q + 3 with four 4-byte loads (L1 / L2 hits: the input is 1 / 16 of the
// output); weights (f32, any strides) become resident bf16 A fragments; stores as conv1x1.hip.  Iteration = one image
// row x strip of 16 G pixels (row and strip are wave-uniform: no per-lane division), grid-stride, next iteration's
// gathers in flight during the matrix instructions and stores.
// ---------------------------------------------------------------------------
struct NinArgs {
    const bf16_t* x;
    const float* w;
    bf16_t* y;
    long long ws_co, ws_ci, ws_tap;
    int N, H, W, Cin, Cout, pad, transposed;
    int strips;
    long iters;            // N * H * strips
};

template <int NMT, int G>
__global__ void __launch_bounds__(256, 3) conv3x3_narrowin_mfma_kernel(NinArgs a) {
    constexpr int NP = NMT / 2;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int H = a.H, W = a.W, Cout = a.Cout;
    // A[row = channel(mt, lane & 15)][k = 8 q + j], k = 2 tap + ci
    bf16x8_t wf[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int co = c11_channel<NMT>(mt, pl >> 2, pl & 3);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * q + j, tap = k >> 1, ci = k & 1;
            const bool ok = tap < 9 && ci < a.Cin && co < Cout;
            v[j] = ok ? a.w[(long)co * a.ws_co + (long)ci * a.ws_ci + (long)tap * a.ws_tap] : 0.f;
        }
        unsigned pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = (unsigned)f32_to_bf16(v[2 * j]) | ((unsigned)f32_to_bf16(v[2 * j + 1]) << 16);
        wf[mt] = __builtin_bit_cast(bf16x8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
    }
    // this lane's four taps: offsets (dh, dw) of the input pixel against the output pixel
    int dh[4], dw[4];
    bool tv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tap = 4 * q + j, kh = tap / 3, kw = tap % 3;
        tv[j] = tap < 9;
        dh[j] = a.transposed ? a.pad - kh : kh - a.pad;           // data gradient: the output pixel fed through tap (kh, kw)
        dw[j] = a.transposed ? a.pad - kw : kw - a.pad;
    }
    const bool two = a.Cin > 1;
    auto gather = [&](long it, unsigned (&bf)[G][4]) {
        const long row = it / a.strips;                           // n * H + h
        const int strip = (int)(it % a.strips);
        const int h = (int)(row % H);
        const long rbase = (row - h) * W;                          // pixel index of (n, 0, 0)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int wo = (strip * G + g) * 16 + pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hi = h + dh[j], wi = wo + dw[j];
                const bool ok = tv[j] && hi >= 0 && hi < H && wi >= 0 && wi < W && wo < W;
                const bf16_t* p = a.x + (ok ? (rbase + (long)hi * W + wi) * a.Cin : 0L);
                unsigned v = two ? *reinterpret_cast<const unsigned*>(p) : (unsigned)*reinterpret_cast<const unsigned short*>(p);
                bf[g][j] = ok ? v : 0u;
            }
        }
    };
    const long nwaves = (long)gridDim.x * 4;
    long it = (long)blockIdx.x * 4 + wid;
    unsigned bcur[G][4], bnext[G][4];
    if (it < a.iters) gather(it, bcur);
    while (it < a.iters) {
        const long nit = it + nwaves;
        if (nit < a.iters) gather(nit, bnext);
        sched_fence();
        const long row = it / a.strips;
        const int strip = (int)(it % a.strips);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bf16x8_t b = __builtin_bit_cast(bf16x8_t, make_uint4(bcur[g][0], bcur[g][1], bcur[g][2], bcur[g][3]));
            f32x4_t acc[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                acc[mt] = mfma_16x16x32_bf16(wf[mt], b, acc[mt]);
            }
            const int wo = (strip * G + g) * 16 + pl;
            const long pix = row * W + wo;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[2 * p][i];
                    v[4 + i] = acc[2 * p + 1][i];
                }
                if (wo < W && 32 * p + 8 * q < Cout) st8(a.y + pix * Cout + 32 * p + 8 * q, v);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) bcur[g][j] = bnext[g][j];
        it = nit;
    }
}

bool conv3x3_narrowin_mfma_supported(const ledn_conv_desc& d) {
    if (!(options().stream_fast & 64)) return false;
    if (d.dtype_x != LEDN_BF16 || d.dtype_y != LEDN_BF16 || !d.w) return false;
    if (d.Cin > 2 || d.Cout % 32 || d.Cout > 128 || d.groups != 1 || d.stride != 1 || d.dil != 1) return false;
    if (d.KH != 3 || d.KW != 3 || d.pad != 1 || d.Ho != d.H || d.Wo != d.W) return false;
    if (d.xadd || d.in_scale || d.in_act != LEDN_ACT_NONE || d.out_scale || d.out_shift || d.stat_sum) return false;
    return d.res_mode == LEDN_RES_NONE && d.act_out == LEDN_ACT_NONE;
}

int conv3x3_narrowin_mfma(const ledn_conv_desc& d, hipStream_t s) {
    constexpr int G = 2;
    NinArgs a;
    a.x = (const bf16_t*)d.x; a.w = d.w; a.y = (bf16_t*)d.y;
    a.ws_co = d.ws_co; a.ws_ci = d.ws_ci; a.ws_tap = d.ws_tap;
    a.N = d.N; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Cout = d.Cout; a.pad = d.pad; a.transposed = d.transposed ? 1 : 0;
    a.strips = (int)cdiv(d.W, 16 * G);
    a.iters = (long)d.N * d.H * a.strips;
    long nb = cdiv(a.iters, 4);
    const long cap = (long)options().conv_workgroups * 3;
    if (nb > cap) nb = cap;
    const int nmt = d.Cout / 16;
    if (nmt == 2) LEDN_LAUNCH((conv3x3_narrowin_mfma_kernel<2, G>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else if (nmt == 4) LEDN_LAUNCH((conv3x3_narrowin_mfma_kernel<4, G>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else if (nmt == 8) LEDN_LAUNCH((conv3x3_narrowin_mfma_kernel<8, G>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else return LEDN_EINVAL;
    return check_launch();
}

// ---------------------------------------------------------------------------
// The first stem convolution (3x3, stride 2, pad 1, 3 -> 32; ddrnet.py:123-130) and its weight gradient straight from
// the planar input batch (no [pixels][32] patch matrix: 268 MB at 16 x 1024^2, written once and read twice per step by
// ledn_im2col_stem_planar + the 1x1 GEMM and its weight gradient).
//
// Shared gather ("patch rows").  Patch element k = (kh 3 + kw) 3 + c of 8 CONSECUTIVE output pixels is 8 bytes at
// stride 2 of one image row: one 16-byte window.  Lane (k = lane & 15 (+ 16 mt), q = lane >> 4) loads the 4-byte-
// aligned 20 bytes around it (dwordx4 + dword), picks its bytes with 64-bit shifts, normalises (SegDataPreProcessor:
// channel map, x scale + shift, batch padding = pad_val in the normalised domain, zero outside the image) and holds
// A[k][pixel 8 q .. 8 q + 7] as bf16.  The first version gathered single bytes (8 loads per fragment): both kernels
// then ran at the issue rate of byte loads -- 2.1 M load instructions x ~64 cycles = 210 us each, whatever else they
// did.  Other input types (f32 / bf16: tests, pre-normalised inputs) keep the scalar loads.
//   weight gradient: dW^T[k][co] = sum over pixels A[k][px] * dz[px][co] -- A is this fragment as it is; B = dz^T through a
//     wave-private LDS tile and ds_read_b64_tr_b16 (4 consecutive PIXELS of a lane's channel per read);
//   forward: z[px][co] = sum over k W[co][k] * patch[px][k] needs the 8 consecutive k of ONE pixel per lane: the same
//     rows are written to a wave-private LDS tile [k][pixel] and read back transposed by ds_read_b64_tr_b16.
// No workgroup barrier in either main loop; next iteration's windows are in flight during the current one.
// ---------------------------------------------------------------------------
struct StemRArgs {
    const void* x;
    const bf16_t* wp;            // [32 cout][32 k] bf16
    bf16_t* y;
    const float *in_scale, *in_shift, *out_scale, *out_shift;
    const int* map;
    const int* valid_hw;
    float* stat_sum;
    float* stat_sqsum;
    float* part;
    int N, H, W, Ho, Wo, act_out, strips;
    long iters;
    float pad_val;
};

__device__ __forceinline__ float stem_ld(const unsigned char* p) { return (float)*p; }
__device__ __forceinline__ float stem_ld(const float* p) { return *p; }
__device__ __forceinline__ float stem_ld(const bf16_t* p) { return bf16_to_f32(p->v); }

constexpr int ST_PIXB = 80;     // LDS bytes per 32-element bf16 row of the wave-private tiles (64 + 16)

// raw patch-row values of one lane and iteration: 8 pixels of element k (before normalisation)
template <typename TX> struct StemRaw { float v[8]; };
template <> struct StemRaw<unsigned char> { unsigned w[5]; int d; };

// the lane's constants for patch element k
struct StemK {
    long plane;      // element offset of (channel map[c], row kh, column kw) against (row 2 o - 1, column -1) of plane 0
    int kh, kw;
    bool valid;
    float sc, sh;
};
__device__ __forceinline__ StemK stem_k(int k, const int* map, const float* in_scale, const float* in_shift, int H, int W) {
    StemK r;
    const int tap = k / 3, c = k - 3 * tap;
    r.valid = k < 27;
    r.kh = r.valid ? tap / 3 : 0;
    r.kw = r.valid ? tap % 3 : 0;
    const int cs = map ? map[c] : c;
    r.plane = (long)cs * H * W + (long)r.kh * W + r.kw;
    r.sc = r.valid ? (in_scale ? in_scale[c] : 1.f) : 0.f;      // (k = 27..31: the fragment's padding rows are zero)
    r.sh = r.valid ? (in_shift ? in_shift[c] : 0.f) : 0.f;
    return r;
}

// loads for pixels px0 .. px0 + 7 of output row o of image n (xn = image base, img = elements per image)
template <typename TX>
__device__ __forceinline__ void stem_fetch(StemRaw<TX>& r, const TX* xn, long front, long rest, const StemK& K, int o, int px0, int H, int W,
                                           int Wo) {
    const int hi = 2 * o - 1 + K.kh;
    const bool rok = K.valid && hi >= 0 && hi < H;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int px = px0 + j, wi = 2 * px - 1 + K.kw;
        const bool ok = rok && px < Wo && wi >= 0 && wi < W;
        r.v[j] = stem_ld(xn + (ok ? (long)(2 * o - 1) * W + (2 * px - 1) + K.plane : 0L));
    }
}
template <>
__device__ __forceinline__ void stem_fetch<unsigned char>(StemRaw<unsigned char>& r, const unsigned char* xn, long front, long rest,
                                                          const StemK& K, int o, int px0, int H, int W, int Wo) {
    // xn = this image, front / rest = bytes from the START OF THE TENSOR (4-byte aligned) to xn / from xn to its END.  The window is aligned on the address and may reach
    // into the neighbouring images; it is clamped only at the two ends of the tensor: below at the first aligned address
    // inside it (the element in front of the very first byte is column -1: masked), above so that it ends at the tensor's
    // end rounded up to 4 bytes (an aligned dword that holds a valid byte cannot cross a page).  The extraction follows
    // with d = offset of the first wanted byte in the window.
    const int hi = 2 * o - 1 + K.kh;
    const bool rok = K.valid && hi >= 0 && hi < H && px0 < Wo;
    const long start = rok ? (long)(2 * o - 1) * W + (2 * px0 - 1) + K.plane : 0L;
    const unsigned long long ax = (unsigned long long)xn;
    const long mis = (long)(ax & 3ull);                         // xn = aligned address + mis
    long base = ((start + mis) & ~3L) - mis;                    // element index (from xn) of an aligned address <= start
    const long last = ((rest + mis + 3) & ~3L) - mis - 20;
    base = base < -front ? -front : (base > last ? last : base);
    r.d = (int)(start - base);
    struct __attribute__((packed, aligned(4))) Win { unsigned w[5]; };
    const Win wv = *reinterpret_cast<const Win*>(xn + base);
#pragma unroll
    for (int i = 0; i < 5; ++i) r.w[i] = wv.w[i];
}

// -> the 8 values as floats (raw domain)
template <typename TX>
__device__ __forceinline__ void stem_bytes(const StemRaw<TX>& r, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = r.v[j];
}
// ({hi, lo} >> 8 s)[31:0], s = 0..3 (v_alignbyte_b32)
__device__ __forceinline__ unsigned stem_alignbyte(unsigned hi, unsigned lo, int s) {
#ifdef LEDN_CPU_EMU
    return (unsigned)((((unsigned long long)hi << 32) | lo) >> (8 * (s & 3)));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, (unsigned)s);
#endif
}
template <>
__device__ __forceinline__ void stem_bytes<unsigned char>(const StemRaw<unsigned char>& r, float (&f)[8]) {
    unsigned w[6] = {r.w[0], r.w[1], r.w[2], r.w[3], r.w[4], 0u};
    const int s = r.d & 3;
    const int dd = r.d >> 2;                        // whole dwords the window was clamped by (0 almost always)
    if (dd != 0) {                                  // rare: first / last bytes of the image
        unsigned t[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            unsigned v = 0u;
#pragma unroll
            for (int k2 = 0; k2 < 5; ++k2) v = (i + dd == k2) ? r.w[k2] : v;
            t[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = t[i];
    }
    // the 16 bytes from offset s as four aligned dwords: elements 2 i, 2 i + 1 are bytes 0 and 2 of dword i
    // (v_alignbyte_b32 + v_cvt_f32_ubyte0 / 2: 12 instructions for the 8 values; the first version shifted 64-bit pairs)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned al = stem_alignbyte(w[i + 1], w[i], s);
        f[2 * i] = (float)(al & 0xffu);
        f[2 * i + 1] = (float)((al >> 16) & 0xffu);
    }
}

// normalise, pad, mask -> bf16 x 8 (A[k][pixel px0 .. px0 + 7]).  Interior fragments (every tap of the 8 pixels inside the
// image and inside the valid extent: all but the border strips / rows) skip the per-element tests -- the patch-row
// arithmetic was the bound of both stem kernels (compiled out: 134 -> 89 us, tools/gpu_exp_sw.sh).
template <typename TX>
__device__ __forceinline__ uint4 stem_row(const StemRaw<TX>& r, const StemK& K, int o, int px0, int H, int W, int Wo, int vh, int vw,
                                          float pad_val) {
    float f[8];
    stem_bytes<TX>(r, f);
    const int hi = 2 * o - 1 + K.kh;
    const bool rok = K.valid && hi >= 0 && hi < H;
    const int wi0 = 2 * px0 - 1 + K.kw, wi7 = wi0 + 14;
    unsigned short e[8];
    // (the lanes of the five padding elements k = 27..31 take the fast path too -- their scale and shift are zeroed by the
    //  caller: with them on the slow path EVERY wave ran both branches)
    if (!K.valid || (rok && hi < vh && wi0 >= 0 && wi7 < W && wi7 < vw && px0 + 7 < Wo)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = f32_to_bf16(f[j] * K.sc + K.sh);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int px = px0 + j, wi = wi0 + 2 * j;
            const bool ok = rok && px < Wo && wi >= 0 && wi < W;
            float v = f[j] * K.sc + K.sh;
            v = (hi < vh && wi < vw) ? v : pad_val;              // batch padding (stack_batch), normalised domain
            e[j] = ok ? f32_to_bf16(v) : (unsigned short)0;       // the convolution's own zero padding
        }
    }
    return make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                      e[6] | ((unsigned)e[7] << 16));
}

template <typename TX, bool FULL>
__global__ void __launch_bounds__(256, 4) stem_conv_reg_kernel(StemRArgs a) {
    constexpr int NMT = 2;
    __shared__ __attribute__((aligned(16))) unsigned char s_t[4][32 * ST_PIXB];     // per wave: [k][32 pixels] bf16
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
    const long img = 3L * H * W;
    const TX* x = reinterpret_cast<const TX*>(a.x);
    unsigned char* st = s_t[wid];
    bf16x8_t wf[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int co = c11_channel<NMT>(mt, pl >> 2, pl & 3);
        wf[mt] = *reinterpret_cast<const bf16x8_t*>(a.wp + co * 32 + 8 * q);
    }
    StemK K[2];                                              // gather role: patch elements k = pl and 16 + pl
#pragma unroll
    for (int t = 0; t < 2; ++t) K[t] = stem_k(16 * t + pl, a.map, a.in_scale, a.in_shift, H, W);
    const int cl = 8 * q;
    float osc[FULL ? 8 : 1], osh[FULL ? 8 : 1];
    if constexpr (FULL) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            osc[i] = a.out_scale ? a.out_scale[cl + i] : 1.f;
            osh[i] = a.out_shift ? a.out_shift[cl + i] : 0.f;
        }
    }
    const float hi_clip = a.act_out == LEDN_ACT_RELU6 ? 6.f : 3.0e38f;
    constexpr int NST = FULL ? 1 : 8;
    float st1[NST], st2[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) st1[i] = st2[i] = 0.f;

    auto gather = [&](long it, StemRaw<TX> (&rw)[2]) {
        const int row = (int)(it / a.strips), strip = (int)(it % a.strips);
        const int n = row / Ho, o = row - n * Ho;
#pragma unroll
        for (int t = 0; t < 2; ++t) stem_fetch<TX>(rw[t], x + (long)n * img, (long)n * img, (long)(a.N - n) * img, K[t], o, strip * 32 + 8 * q, H, W, Wo);
    };
    const long nwaves = (long)gridDim.x * 4;
    long it = (long)blockIdx.x * 4 + wid;
    // (two iterations of windows in flight instead of one measured the same: 122 vs 119 us)
    StemRaw<TX> rcur[2], rnext[2];
    if (it < a.iters) gather(it, rcur);
    while (it < a.iters) {
        const long nit = it + nwaves;
        if (nit < a.iters) gather(nit, rnext);
        sched_fence();
        const int row = (int)(it / a.strips), strip = (int)(it % a.strips);
        const int n = row / Ho, o = row - n * Ho;
        const int vh = a.valid_hw ? a.valid_hw[2 * n] : H, vw = a.valid_hw ? a.valid_hw[2 * n + 1] : W;
#pragma unroll
        for (int t = 0; t < 2; ++t)                            // row k = 16 t + pl, pixels 8 q .. 8 q + 7
            *reinterpret_cast<uint4*>(st + (16 * t + pl) * ST_PIXB + q * 16) =
                stem_row<TX>(rcur[t], K[t], o, strip * 32 + 8 * q, H, W, Wo, vh, vw, a.pad_val);
        wave_sync();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // lane 4 r + p supplies row k = 8 q + r (+ 4), pixels 16 g + 4 p ..: lane i receives k = 8 q .. 8 q + 7 of pixel 16 g + i
            const unsigned char* tp = st + (8 * q + (pl >> 2)) * ST_PIXB + (16 * g + 4 * (pl & 3)) * 2;
            const bf16x4_t lo = lds_read_tr16(tp), hi4 = lds_read_tr16(tp + 4 * ST_PIXB);
            bf16x8_t b;
            b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
            b[4] = hi4[0]; b[5] = hi4[1]; b[6] = hi4[2]; b[7] = hi4[3];
            f32x4_t acc[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                acc[mt] = mfma_16x16x32_bf16(wf[mt], b, acc[mt]);
            }
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = acc[0][i];
                v[4 + i] = acc[1][i];
            }
            const int xo = strip * 32 + 16 * g + pl;
            const bool pok = xo < Wo;
            if constexpr (FULL) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = v[i] * osc[i] + osh[i];
                    if (a.act_out != LEDN_ACT_NONE) v[i] = fminf(fmaxf(v[i], 0.f), hi_clip);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float vm = pok ? v[i] : 0.f;
                    st1[i] += vm;
                    st2[i] = fmaf(vm, vm, st2[i]);
                }
            }
            if (pok) st8(a.y + ((long)row * Wo + xo) * 32 + cl, v);
        }
        wave_sync();                                             // the tile is rewritten by the next iteration
#pragma unroll
        for (int t = 0; t < 2; ++t) rcur[t] = rnext[t];
        it = nit;
    }
    if constexpr (!FULL) {
        if (!a.stat_sum) return;
        __shared__ float s_st[4][2][32];
        const float t1 = c11_reduce16<8>(st1, lane), t2 = c11_reduce16<8>(st2, lane);
        const int vi = c11_red_index<8>(lane);
        if ((lane & 15) < 8) {
            s_st[wid][0][cl + vi] = t1;
            s_st[wid][1][cl + vi] = t2;
        }
        __syncthreads();
        if (tid < 64) {
            const int jj = tid >> 5, c = tid & 31;
            const float t = (s_st[0][jj][c] + s_st[1][jj][c]) + (s_st[2][jj][c] + s_st[3][jj][c]);
            if (a.part) a.part[(long)blockIdx.x * 64 + jj * 32 + c] = t;
            else atomicAdd((jj ? a.stat_sqsum : a.stat_sum) + c, t);
        }
    }
}

bool stem_conv_reg_enabled() { return (options().stream_fast & 64) != 0; }

int stem_conv_reg_impl(const void* x, int dtype_x, const void* wp, void* y, int N, int H, int W, int Ho, int Wo,
                       const float* in_scale, const float* in_shift, const int* map, const int* valid_hw, float pad_val,
                       const float* out_scale, const float* out_shift, int act_out, float* stat_sum, float* stat_sqsum,
                       hipStream_t s) {
    const bool full = out_scale || out_shift || act_out != LEDN_ACT_NONE;
    LEDN_REQUIRE(3L * H * W >= 28 && ((unsigned long long)x & 3ull) == 0);
    StemRArgs a;
    a.x = x; a.wp = (const bf16_t*)wp; a.y = (bf16_t*)y;
    a.in_scale = in_scale; a.in_shift = in_shift; a.out_scale = out_scale; a.out_shift = out_shift;
    a.map = map; a.valid_hw = valid_hw; a.stat_sum = stat_sum; a.stat_sqsum = stat_sqsum;
    a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.act_out = act_out; a.pad_val = pad_val;
    a.strips = (int)cdiv(Wo, 32);
    a.iters = (long)N * Ho * a.strips;
    long nb = cdiv(a.iters, 4);
    const long cap = (long)options().conv_workgroups * 4;
    if (nb > cap) nb = cap;
    a.part = (stat_sum && (nb > 16 || det())) ? ws_take(nb * 64) : nullptr;
#define LEDN_STEMR(TX)                                                                                       \
    do {                                                                                                     \
        if (full) LEDN_LAUNCH((stem_conv_reg_kernel<TX, true>), dim3((unsigned)nb), dim3(256), 0, s, a);     \
        else LEDN_LAUNCH((stem_conv_reg_kernel<TX, false>), dim3((unsigned)nb), dim3(256), 0, s, a);         \
    } while (0)
    if (dtype_x == LEDN_U8) LEDN_STEMR(unsigned char);
    else if (dtype_x == LEDN_F32) LEDN_STEMR(float);
    else if (dtype_x == LEDN_BF16) LEDN_STEMR(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_STEMR
    if (a.part) return finish_partials(a.part, (int)nb, 32, 2, stat_sum, stat_sqsum, nullptr, s);
    return check_launch();
}

// ---------------------------------------------------------------------------
// Weight gradient of the stem convolution (see above): dW[co][c][kh][kw] = sum over output pixels of
// dz[px][co] * pre(x)[c][2 ho - 1 + kh][2 wo - 1 + kw].  4 matrix instructions per 32 pixels accumulate the whole
// 32 x 32 tile dW^T[k][co] in 16 registers for the lifetime of the wave; one 864-float partial row per workgroup (OIHW
// order), finish_partials adds them into dW.
// ---------------------------------------------------------------------------
struct StemWArgs {
    const void* x;
    const bf16_t* dz;            // [N][Ho][Wo][32]  (BNP: the gradient of the BatchNorm + activation OUTPUT)
    const bf16_t* bz;            // BNP: the convolution's output z, same shape
    const float *b_scale, *b_shift, *b_mean, *b_invstd, *b_sum_g, *b_sum_gx;   // BNP: ledn_bnbwd_desc of the BatchNorm behind the conv
    float b_inv_count;
    int b_mode, b_act;
    float* part;                 // [gridDim.x][864]
    const float *in_scale, *in_shift;
    const int* map;
    const int* valid_hw;
    int N, H, W, Ho, Wo, strips;
    long iters;
    float pad_val;
};

#ifndef LEDN_SW_EXP
#define LEDN_SW_EXP 0      // cost experiments (tools/gpu_exp_sw.sh; uint8 inputs only): 1 no matrix instructions, 2 no LDS tile, 3 no patch-row arithmetic
#endif
// BNP: dz is not read but formed piece by piece from z and the gradient dy of y = act(BatchNorm(z)) -- the apply half of the
// BatchNorm backward (dz = scale g + A z + B, g = dy act'(.): stream_fast.hip) as this kernel's operand prologue.  The
// stem's input needs no gradient, so this weight gradient is the ONLY reader of dz: the apply pass (268 MB read twice, 268 MB
// written, 130 us at 16 x 512 x 512 x 32) and this kernel's read of its output shrink to one more 268 MB read here.
template <typename TX, bool BNP>
__global__ void __launch_bounds__(256, 3) stem_wgrad_reg_kernel(StemWArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_z[4][32 * ST_PIXB];     // per wave: [pixel][32 channels] bf16
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
    const long img = 3L * H * W;
    const TX* x = reinterpret_cast<const TX*>(a.x);
    unsigned char* sz = s_z[wid];
    StemK K[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) K[mt] = stem_k(16 * mt + m16, a.map, a.in_scale, a.in_shift, H, W);
    f32x4_t acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // BNP: coefficients of this lane's 8 channels 8 (lane & 3) .. (both pieces e = lane, lane + 64 share them)
    float bsc[BNP ? 8 : 1], bsh[BNP ? 8 : 1], bca[BNP ? 8 : 1], bcb[BNP ? 8 : 1];
    if constexpr (BNP) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = 8 * (lane & 3) + k;
            const float sc = a.b_scale ? a.b_scale[c] : 1.f;
            float ca = 0.f, cb = 0.f;
            if (a.b_mode) {           // dz = sc g + A z + B,  A = -sc mean_gx invstd,  B = -sc mean_g - A mean
                ca = -sc * (a.b_sum_gx[c] * a.b_inv_count) * a.b_invstd[c];
                cb = -sc * (a.b_sum_g[c] * a.b_inv_count) - ca * a.b_mean[c];
            }
            bsc[k] = sc;
            bsh[k] = a.b_shift ? a.b_shift[c] : 0.f;
            bca[k] = ca;
            bcb[k] = cb;
        }
    }
    constexpr int NZ = BNP ? 4 : 2;

    auto gather = [&](long it, StemRaw<TX> (&ra)[2], uint4 (&rz)[NZ]) {
        const int row = (int)(it / a.strips), strip = (int)(it % a.strips);
        const int n = row / Ho, o = row - n * Ho;
        const bf16_t* zr = a.dz + (long)row * Wo * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) stem_fetch<TX>(ra[mt], x + (long)n * img, (long)n * img, (long)(a.N - n) * img, K[mt], o, strip * 32 + 8 * q, H, W, Wo);
#pragma unroll
        for (int t = 0; t < 2; ++t) {                            // piece e = lane + 64 t: pixel e >> 2, channels 8 (e & 3) ..
            const int e = lane + 64 * t, px = strip * 32 + (e >> 2);
            const bool pok = px < Wo;
            uint4 v = *reinterpret_cast<const uint4*>(zr + (pok ? (long)px * 32 + 8 * (e & 3) : 0L));
            if (!pok) v = make_uint4(0u, 0u, 0u, 0u);
            rz[t] = v;
            if constexpr (BNP) {
                const bf16_t* zz = a.bz + (long)row * Wo * 32;
                rz[2 + t] = *reinterpret_cast<const uint4*>(zz + (pok ? (long)px * 32 + 8 * (e & 3) : 0L));
            }
        }
    };
    // BNP: the piece of dz from the piece of dy (zero beyond the row end: stays zero) and the piece of z
    auto bn_piece = [&](const uint4& dyr, const uint4& zr) {
        const unsigned dv[4] = {dyr.x, dyr.y, dyr.z, dyr.w}, zv[4] = {zr.x, zr.y, zr.z, zr.w};
        unsigned o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r2[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int k = 2 * i + hh;
                const float z = __uint_as_float(hh ? (zv[i] & 0xffff0000u) : (zv[i] << 16));
                const float dy = __uint_as_float(hh ? (dv[i] & 0xffff0000u) : (dv[i] << 16));
                const float v = fmaf(z, bsc[k], bsh[k]);
                const float g = (a.b_act == LEDN_ACT_RELU && !(v > 0.f)) ? 0.f : dy;
                r2[hh] = fmaf(g, bsc[k], fmaf(z, bca[k], bcb[k]));
            }
            o[i] = (unsigned)f32_to_bf16(r2[0]) | ((unsigned)f32_to_bf16(r2[1]) << 16);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    };
    const long nwaves = (long)gridDim.x * 4;
    long it = (long)blockIdx.x * 4 + wid;
    StemRaw<TX> acur[2], anext[2];
    uint4 zcur[NZ], znext[NZ];
    if (it < a.iters) gather(it, acur, zcur);
    while (it < a.iters) {
        const long nit = it + nwaves;
        if (nit < a.iters) gather(nit, anext, znext);
        sched_fence();
        const bool exp_lds = LEDN_SW_EXP != 2 || a.N < 0;
        if (exp_lds) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = lane + 64 * t;
            uint4 piece = zcur[t];
            if constexpr (BNP) {
                const int px = (int)(it % a.strips) * 32 + (e >> 2);
                piece = px < Wo ? bn_piece(zcur[t], zcur[2 + t]) : make_uint4(0u, 0u, 0u, 0u);
            }
            *reinterpret_cast<uint4*>(sz + (e >> 2) * ST_PIXB + (e & 3) * 16) = piece;
        }
        wave_sync();
        }
        const int row = (int)(it / a.strips), strip = (int)(it % a.strips);
        const int n = row / Ho, o = row - n * Ho;
        const int vh = a.valid_hw ? a.valid_hw[2 * n] : H, vw = a.valid_hw ? a.valid_hw[2 * n + 1] : W;
        bf16x8_t af[2], bfr[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {                         // lane 4 r + p supplies pixel 8 q + r (+ 4), channels 16 nt + 4 p ..
            if (exp_lds) {
            const unsigned char* zp = sz + (8 * q + (m16 >> 2)) * ST_PIXB + (16 * nt + 4 * (m16 & 3)) * 2;
            const bf16x4_t lo = lds_read_tr16(zp), hi4 = lds_read_tr16(zp + 4 * ST_PIXB);
            bfr[nt][0] = lo[0]; bfr[nt][1] = lo[1]; bfr[nt][2] = lo[2]; bfr[nt][3] = lo[3];
            bfr[nt][4] = hi4[0]; bfr[nt][5] = hi4[1]; bfr[nt][6] = hi4[2]; bfr[nt][7] = hi4[3];
            } else {
                bfr[nt] = __builtin_bit_cast(bf16x8_t, zcur[nt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (LEDN_SW_EXP == 4 || LEDN_SW_EXP == 5) {               // 4: byte extraction only, 5: + conversion to bf16
                float f8[8];
                stem_bytes<TX>(acur[mt], f8);
                if (LEDN_SW_EXP == 4) {
                    af[mt] = __builtin_bit_cast(bf16x8_t, make_uint4(__float_as_uint(f8[0]) + __float_as_uint(f8[1]), __float_as_uint(f8[2]) + __float_as_uint(f8[3]),
                                                                     __float_as_uint(f8[4]) + __float_as_uint(f8[5]), __float_as_uint(f8[6]) + __float_as_uint(f8[7])));
                } else {
                    float o8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o8[j] = f8[j] * K[mt].sc + K[mt].sh;
                    bf16_t tmp[8];
                    st8(tmp, o8);
                    uint4 rawv;
                    __builtin_memcpy(&rawv, tmp, 16);
                    af[mt] = __builtin_bit_cast(bf16x8_t, rawv);
                }
            } else if (LEDN_SW_EXP != 3 || a.N < 0)
                af[mt] = __builtin_bit_cast(bf16x8_t, stem_row<TX>(acur[mt], K[mt], o, strip * 32 + 8 * q, H, W, Wo, vh, vw, a.pad_val));
            else if (LEDN_SW_EXP == 3) {
                uint4 rawv;
                __builtin_memcpy(&rawv, &acur[mt], 16);
                af[mt] = __builtin_bit_cast(bf16x8_t, rawv);
            }
        }
        if (LEDN_SW_EXP != 1 || a.N < 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt]);
        } else {
            acc[0][0][0] += __builtin_bit_cast(uint4, af[0]).x + __builtin_bit_cast(uint4, af[1]).y + __builtin_bit_cast(uint4, bfr[0]).x + __builtin_bit_cast(uint4, bfr[1]).y;
        }
        if (exp_lds) wave_sync();                                // the tile is rewritten by the next iteration
#pragma unroll
        for (int t = 0; t < 2; ++t) acur[t] = anext[t];
#pragma unroll
        for (int t = 0; t < NZ; ++t) zcur[t] = znext[t];
        it = nit;
    }
    // acc[mt][nt][i] = dW^T[k = 16 mt + 4 q + i][co = 16 nt + m16]: the four waves meet in LDS, one OIHW row per workgroup
    __shared__ float s_red[4][32][33];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) s_red[wid][16 * mt + 4 * q + i][16 * nt + m16] = acc[mt][nt][i];
    __syncthreads();
    for (int e = tid; e < 27 * 32; e += 256) {
        const int k = e / 32, co = e % 32;
        const float t = (s_red[0][k][co] + s_red[1][k][co]) + (s_red[2][k][co] + s_red[3][k][co]);
        const int tap = k / 3, c = k - 3 * tap;                 // k = (kh 3 + kw) 3 + c  ->  OIHW [co][c][kh][kw]
        a.part[(long)blockIdx.x * 864 + co * 27 + c * 9 + tap] = t;
    }
}

int stem_conv_wgrad_impl(const void* x, int dtype_x, const void* dz, float* dw, int N, int H, int W, int C, int Ho, int Wo,
                         int Cout, const float* in_scale, const float* in_shift, const int* map, const int* valid_hw,
                         float pad_val, const ledn_bnbwd_desc* bn, hipStream_t s) {
    if (bn) {       // dz = the apply half of this BatchNorm backward, formed inside the kernel (ledn_stem_conv_wgrad_bn)
        LEDN_REQUIRE(!dz && bn->z && bn->dy && bn->C == 32 && bn->P == (long long)N * Ho * Wo && bn->count > 0);
        LEDN_REQUIRE(bn->dtype_z == LEDN_BF16 && bn->dtype_y == LEDN_BF16 && !bn->res && bn->res_mode == LEDN_RES_NONE);
        LEDN_REQUIRE((bn->act == LEDN_ACT_NONE || bn->act == LEDN_ACT_RELU) && !bn->dz_add && !bn->dres && !bn->rows);
        LEDN_REQUIRE(!bn->bn_mode || (bn->mean && bn->invstd && bn->sum_g && bn->sum_gx));
        dz = bn->dy;
    }
    LEDN_REQUIRE(x && dz && dw && N > 0 && H > 0 && W > 0 && C == 3 && Cout == 32);
    LEDN_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1);
    LEDN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
    LEDN_REQUIRE(3L * H * W >= 28 && ((unsigned long long)x & 3ull) == 0);
    StemWArgs a;
    a.x = x; a.dz = (const bf16_t*)dz; a.in_scale = in_scale; a.in_shift = in_shift; a.map = map; a.valid_hw = valid_hw;
    a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.pad_val = pad_val;
    a.strips = (int)cdiv(Wo, 32);
    a.iters = (long)N * Ho * a.strips;
    long nb = cdiv(a.iters, 4 * 8);                              // >= 8 strips per wave
    if (nb < 1) nb = 1;
    const long cap = (long)options().conv_workgroups * 2;
    if (nb > cap) nb = cap;
    a.part = ws_take(nb * 864);
    if (!a.part) return LEDN_EINVAL;        // the kernel writes its partial tiles unconditionally: no workspace, no launch
    a.bz = nullptr;
    a.b_scale = a.b_shift = a.b_mean = a.b_invstd = a.b_sum_g = a.b_sum_gx = nullptr;
    a.b_inv_count = 0.f;
    a.b_mode = a.b_act = 0;
    if (bn) {
        a.bz = (const bf16_t*)bn->z;
        a.b_scale = bn->scale; a.b_shift = bn->shift; a.b_mean = bn->mean; a.b_invstd = bn->invstd;
        a.b_sum_g = bn->sum_g; a.b_sum_gx = bn->sum_gx;
        a.b_inv_count = (float)(1.0 / bn->count);
        a.b_mode = bn->bn_mode; a.b_act = bn->act;
        if (dtype_x == LEDN_U8) LEDN_LAUNCH((stem_wgrad_reg_kernel<unsigned char, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
        else if (dtype_x == LEDN_F32) LEDN_LAUNCH((stem_wgrad_reg_kernel<float, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
        else if (dtype_x == LEDN_BF16) LEDN_LAUNCH((stem_wgrad_reg_kernel<bf16_t, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
        else return LEDN_EINVAL;
        return finish_partials(a.part, (int)nb, 864, 1, dw, nullptr, nullptr, s);
    }
    if (dtype_x == LEDN_U8) LEDN_LAUNCH((stem_wgrad_reg_kernel<unsigned char, false>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else if (dtype_x == LEDN_F32) LEDN_LAUNCH((stem_wgrad_reg_kernel<float, false>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else if (dtype_x == LEDN_BF16) LEDN_LAUNCH((stem_wgrad_reg_kernel<bf16_t, false>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else return LEDN_EINVAL;
    return finish_partials(a.part, (int)nb, 864, 1, dw, nullptr, nullptr, s);
}

// ---------------------------------------------------------------------------
// Weight gradient of the two-class heads' 3x3 convolution (32 -> 2; led_head.py:44-51) -- and of any 3x3 stride-1 layer
// with 32 inputs and <= 16 outputs --, wave-autonomous:
//   dW[co][ci][kh][kw] = sum over pixels of dz[px][co] * pre(x)[px + (kh - 1, kw - 1)][ci],   pre = BatchNorm + ReLU folded in.
// Per tap a product D[ci][co] with K = pixels.  A = x^T needs 8 consecutive PIXELS of a lane's channel: the wave keeps
// the three input rows of its 32-pixel strip in a private LDS ring ([pixel][32 ch], prologue applied while writing, zero
// outside the image) and ds_read_b64_tr_b16 hands out the transposed fragments -- the kw / kh shifts are address
// offsets.  B = dz: for two channels the 8 pixels of a lane are 32 contiguous bytes (two 16-byte loads, the lane picks its
// channel).  36 transposing reads + 18 matrix instructions per 32 pixels; 9 x 2 accumulator tiles (72 VGPRs) live for the
// whole wave; one new input row per iteration, requested one iteration ahead; no workgroup barrier in the loop.
// conv_wgrad_mfma_kernel ran this layer as a 32 x 32 output tile (30 of 32 columns padding) with an element-wise dz
// staging: 195 us for the 268 MB of x at 16 x 512^2.
// ---------------------------------------------------------------------------
struct WnArgs {
    const bf16_t* x;
    const bf16_t* dz;
    float* part;               // [gridDim.x][Cout * 288]
    const float *in_scale, *in_shift, *in_slope;
    int in_act;
    int N, H, W, Cout;
    int strips, segs, RS;
    long tasks;
};

constexpr int WN_PW = 34;      // pixels of a ring row: x0 - 1 .. x0 + 32

template <bool PRO>     // (the prologue's 24 coefficient registers: 196 VGPRs, two waves per SIMD; without: 168 at three)
__global__ void __launch_bounds__(256, PRO ? 2 : 3) conv3x3_wgrad_narrow_kernel(WnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_x[4][3][WN_PW * ST_PIXB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int H = a.H, W = a.W, Cout = a.Cout;
    unsigned char* ring = &s_x[wid][0][0];
    // load role: piece e = lane + 64 t (t = 0..2) of a ring row: pixel e >> 2, channel octet e & 3 = lane & 3
    const int oct = lane & 3;
    float psc[8], psh[8], png[8];
    if constexpr (PRO) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = 8 * oct + i;
            psc[i] = a.in_scale ? a.in_scale[c] : 1.f;
            psh[i] = a.in_shift ? a.in_shift[c] : 0.f;
            png[i] = a.in_act == LEDN_ACT_PRELU ? a.in_slope[c] : (a.in_act == LEDN_ACT_NONE ? 1.f : 0.f);
        }
    }
    f32x4_t acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[t][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const long nwaves = (long)gridDim.x * 4;
    for (long task = (long)blockIdx.x * 4 + wid; task < a.tasks; task += nwaves) {
        const int strip = (int)(task % a.strips);
        const int seg = (int)((task / a.strips) % a.segs);
        const int n = (int)(task / ((long)a.strips * a.segs));
        const int x0 = strip * 32, r0 = seg * a.RS, r1 = min(r0 + a.RS, H);
        const bf16_t* xn = a.x + (long)n * H * W * 32;
        const bf16_t* zn = a.dz + (long)n * H * W * Cout;

        auto fetch_x = [&](int ir, uint4 (&rw)[3]) {
            const bool rok = ir >= 0 && ir < H;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int e = lane + 64 * t, px = x0 - 1 + (e >> 2);
                const bool ok = rok && e < WN_PW * 4 && px >= 0 && px < W;
                uint4 v = *reinterpret_cast<const uint4*>(xn + (ok ? ((long)ir * W + px) * 32 + 8 * oct : 0L));
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                rw[t] = v;
            }
        };
        auto commit_x = [&](int ir, int slot, const uint4 (&rw)[3]) {
            const bool rok = ir >= 0 && ir < H;
            unsigned char* row = ring + slot * (WN_PW * ST_PIXB);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int e = lane + 64 * t, px = x0 - 1 + (e >> 2);
                if (e >= WN_PW * 4) continue;
                uint4 v = rw[t];
                if constexpr (PRO) {
                    const bool ok = rok && px >= 0 && px < W;
                    v = __builtin_bit_cast(uint4, c11_prologue(__builtin_bit_cast(bf16x8_t, v), psc, psh, png));
                    if (!ok) v = make_uint4(0u, 0u, 0u, 0u);             // padding is zero AFTER the activation
                }
                *reinterpret_cast<uint4*>(row + (e >> 2) * ST_PIXB + oct * 16) = v;
            }
        };
        // B fragment of output row o: lane (co = m16, q) holds dz[(o, x0 + 8 q + j)][co], j = 0..7
        auto fetch_z = [&](int o, uint4 (&rz)[2]) {
            const bool live = m16 < Cout && o < r1;
            const int px0 = x0 + 8 * q;
            if (Cout == 2) {                                             // 8 pixels x 2 channels = 32 contiguous bytes
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = live && px0 + 4 * h < W;             // (W % 4 == 0: whole pieces)
                    uint4 v = *reinterpret_cast<const uint4*>(zn + (ok ? ((long)o * W + px0 + 4 * h) * 2 : 0L));
                    if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                    rz[h] = v;
                }
            } else {
                unsigned short e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = live && px0 + j < W;
                    const unsigned short v = zn[ok ? ((long)o * W + px0 + j) * Cout + m16 : 0L].v;
                    e[j] = ok ? v : (unsigned short)0;
                }
                rz[0] = make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                                   e[6] | ((unsigned)e[7] << 16));
                rz[1] = make_uint4(0u, 0u, 0u, 0u);
            }
        };
        auto frag_z = [&](const uint4 (&rz)[2]) {
            if (Cout != 2) return __builtin_bit_cast(bf16x8_t, rz[0]);
            // dwords = pixels (lo half channel 0, hi half channel 1): pick this lane's channel of the 8 pixels
            const unsigned w[8] = {rz[0].x, rz[0].y, rz[0].z, rz[0].w, rz[1].x, rz[1].y, rz[1].z, rz[1].w};
            unsigned o4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned lo = m16 ? (w[2 * i] >> 16) : (w[2 * i] & 0xffffu);
                const unsigned hi = m16 ? (w[2 * i + 1] >> 16) : (w[2 * i + 1] & 0xffffu);
                o4[i] = lo | (hi << 16);
            }
            return __builtin_bit_cast(bf16x8_t, make_uint4(o4[0], o4[1], o4[2], o4[3]));
        };

        // ring slot of image row ir: (ir - (r0 - 1)) % 3
        uint4 xr[3], zr[2], zr_next[2];
        fetch_x(r0 - 1, xr);
        commit_x(r0 - 1, 0, xr);
        fetch_x(r0, xr);
        commit_x(r0, 1, xr);
        fetch_x(r0 + 1, xr);                                            // row r0 + 1: committed in the first iteration
        fetch_z(r0, zr);
        int slot_new = 2;                                                // slot the row o + 1 goes to
        for (int o = r0; o < r1; ++o) {
            wave_sync();                                                 // the previous iteration's reads of slot_new are done
            commit_x(o + 1, slot_new, xr);
            if (o + 1 < r1) {
                fetch_x(o + 2, xr);
                fetch_z(o + 1, zr_next);
            }
            wave_sync();
            const bf16x8_t b = frag_z(zr);
            const int s_top = slot_new == 2 ? 0 : slot_new + 1;          // slot of row o - 1 (the oldest)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                int sl = s_top + kh;
                sl = sl >= 3 ? sl - 3 : sl;
                const unsigned char* row = ring + sl * (WN_PW * ST_PIXB);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        // ring pixel index of (output pixel x0 + 8 q + j, tap kw) = 8 q + j + kw; lane 4 r + p supplies
                        // pixel 8 q + kw + r (+ 4), channels 16 mt + 4 p ..
                        const unsigned char* ap = row + (8 * q + kw + (m16 >> 2)) * ST_PIXB + (16 * mt + 4 * (m16 & 3)) * 2;
                        const bf16x4_t lo = lds_read_tr16(ap), hi4 = lds_read_tr16(ap + 4 * ST_PIXB);
                        bf16x8_t af;
                        af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                        af[4] = hi4[0]; af[5] = hi4[1]; af[6] = hi4[2]; af[7] = hi4[3];
                        acc[kh * 3 + kw][mt] = mfma_16x16x32_bf16(af, b, acc[kh * 3 + kw][mt]);
                    }
            }
            zr[0] = zr_next[0];
            zr[1] = zr_next[1];
            slot_new = slot_new == 2 ? 0 : slot_new + 1;
        }
        wave_sync();                                                     // the next task rewrites the ring
    }
    // acc[t][mt][i] = dW[co = m16][ci = 16 mt + 4 q + i][tap t]: the four waves meet in LDS, one OIHW row per workgroup
    __syncthreads();
    float* s_red = reinterpret_cast<float*>(&s_x[0][0][0]);              // [4][Cout <= 16][288]: 4 x 16 x 288 x 4 B > ring?  only Cout rows used
    if (m16 < Cout) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) s_red[(wid * Cout + m16) * 288 + (16 * mt + 4 * q + i) * 9 + t] = acc[t][mt][i];
    }
    __syncthreads();
    const int nel = Cout * 288;
    for (int e = tid; e < nel; e += 256)
        a.part[(long)blockIdx.x * nel + e] = (s_red[e] + s_red[nel + e]) + (s_red[2 * nel + e] + s_red[3 * nel + e]);
}

bool conv_wgrad_narrow_reg_supported(const ledn_wgrad_desc& d) {
    if (!(options().stream_fast & 64)) return false;
    if (d.dtype_x != LEDN_BF16 || d.dtype_dz != LEDN_BF16 || d.xadd || d.groups != 1) return false;
    if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1 || d.Ho != d.H || d.Wo != d.W) return false;
    if (d.Cin != 32 || d.Cout > 7) return false;                         // (4 x Cout x 288 floats of reduction space inside the ring)
    if (d.Cout == 2 && d.W % 4) return false;
    if ((d.in_scale == nullptr) != (d.in_shift == nullptr)) return false;
    if (d.in_act != LEDN_ACT_NONE && d.in_act != LEDN_ACT_RELU && !(d.in_act == LEDN_ACT_PRELU && d.in_slope)) return false;
    return d.ws_tap == 1 && d.ws_ci == 9 && d.ws_co == 288;              // natural OIHW: the partial rows use dW's own index
}

int conv_wgrad_narrow_reg(const ledn_wgrad_desc& d, hipStream_t s) {
    WnArgs a;
    a.x = (const bf16_t*)d.x; a.dz = (const bf16_t*)d.dz;
    a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.in_slope = d.in_slope; a.in_act = d.in_act;
    a.N = d.N; a.H = d.H; a.W = d.W; a.Cout = d.Cout;
    a.strips = (int)cdiv(d.W, 32);
    a.RS = c33_rows((long)d.N * a.strips, d.H);
    a.segs = (int)cdiv(d.H, a.RS);
    a.tasks = (long)d.N * a.segs * a.strips;
    long nb = cdiv(a.tasks, 4);
    const long cap = (long)options().conv_workgroups * 2;
    if (nb > cap) nb = cap;
    const int nel = d.Cout * 288;
    a.part = ws_take(nb * nel);
    if (!a.part) return LEDN_EINVAL;
    if (d.in_scale || d.in_act != LEDN_ACT_NONE) LEDN_LAUNCH((conv3x3_wgrad_narrow_kernel<true>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else LEDN_LAUNCH((conv3x3_wgrad_narrow_kernel<false>), dim3((unsigned)nb), dim3(256), 0, s, a);
    return finish_partials(a.part, (int)nb, nel, 1, d.dw, nullptr, nullptr, s);
}

// ---------------------------------------------------------------------------
// Weight gradient of the 1x1 stride-1 convolutions (Cin, Cout <= 128), wave-autonomous: dW[co][ci] = sum over pixels of
// dz[px][co] * x[px][ci] -- K = pixels, so BOTH operands are needed pixel-major per lane.  A wave takes 32 consecutive
// pixels per iteration: their x rows (32 x Cin) and dz rows (32 x Cout) go into a wave-private LDS tile with 16-byte
// loads / writes (next iteration's loads in flight), ds_read_b64_tr_b16 hands out A[ci][8 pixels] and B[8 pixels][co], and
// NM x NN matrix instructions (16 x 16 x 32) accumulate the whole [Cin][Cout] product in registers for the lifetime of
// the wave.  No workgroup barrier in the loop.  conv_wgrad_mfma_kernel<1, 1> stages 8 x 32-pixel tiles for four waves
// behind two barriers per tile, one 32 x 32 (ci, co) pair per workgroup -- every pair re-reads x and dz -- and runs
// these layers at 1.3-2.9 TB/s.  The partial tiles have that kernel's layout ([workgroup][pair][32 co][32 ci]), so its
// summing kernels (conv_wgrad_finish_kernel / _finish_multi) serve both.  DIAG: only the pairs on the 32 x 32 block
// diagonal (grouped convolutions whose groups lie inside them) are formed.
// ---------------------------------------------------------------------------
struct W11Args {
    const bf16_t* x;           // [P][Cin]
    const bf16_t* dz;          // [P][Cout]
    float* part;               // [gridDim.x][pairs][1024]
    long P, iters;             // iters = ceil(P / 32)
    int Cin, Cout, ci_tiles;
};

template <int NM, int NN, bool DIAG>
__global__ void __launch_bounds__(256, 2) conv1x1_wgrad_reg_kernel(W11Args a) {
    constexpr int XB = NM * 32 + 16, ZB = NN * 32 + 16;        // LDS bytes per pixel row of the x / dz tiles
    constexpr int NA = DIAG ? 2 : NN;                          // n-tiles kept per m-tile
    constexpr int TB = 32 * (XB + ZB) > 4096 ? 32 * (XB + ZB) : 4096;   // (>= 4 KB per wave: the final reduction reuses it as [4][1024] floats)
    __shared__ __attribute__((aligned(16))) unsigned char s_t[4][TB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    unsigned char* sx = s_t[wid];
    unsigned char* sz = sx + 32 * XB;
    f32x4_t acc[NM][NA];
#pragma unroll
    for (int mt = 0; mt < NM; ++mt)
#pragma unroll
        for (int j = 0; j < NA; ++j) acc[mt][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PX = NM * 2, PZ = NN * 2;                    // 16-byte pieces per pixel
    constexpr int LX = (32 * PX + 63) / 64, LZ = (32 * PZ + 63) / 64;
    auto fetch = [&](long it, uint4 (&rx)[LX], uint4 (&rz)[LZ]) {
        const long p0 = it * 32;
#pragma unroll
        for (int t = 0; t < LX; ++t) {
            const int e = lane + 64 * t, px = e / PX, pc = e % PX;
            const bool ok = e < 32 * PX && p0 + px < a.P && pc * 8 < a.Cin;
            uint4 v = *reinterpret_cast<const uint4*>(a.x + (ok ? (p0 + px) * a.Cin + pc * 8 : 0L));
            if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
            rx[t] = v;
        }
#pragma unroll
        for (int t = 0; t < LZ; ++t) {
            const int e = lane + 64 * t, px = e / PZ, pc = e % PZ;
            const bool ok = e < 32 * PZ && p0 + px < a.P && pc * 8 < a.Cout;
            uint4 v = *reinterpret_cast<const uint4*>(a.dz + (ok ? (p0 + px) * a.Cout + pc * 8 : 0L));
            if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
            rz[t] = v;
        }
    };
    const long nwaves = (long)gridDim.x * 4;
    long it = (long)blockIdx.x * 4 + wid;
    uint4 xcur[LX], zcur[LZ], xnext[LX], znext[LZ];
    if (it < a.iters) fetch(it, xcur, zcur);
    while (it < a.iters) {
        const long nit = it + nwaves;
        if (nit < a.iters) fetch(nit, xnext, znext);
        sched_fence();
#pragma unroll
        for (int t = 0; t < LX; ++t) {
            const int e = lane + 64 * t;
            if (e < 32 * PX) *reinterpret_cast<uint4*>(sx + (e / PX) * XB + (e % PX) * 16) = xcur[t];
        }
#pragma unroll
        for (int t = 0; t < LZ; ++t) {
            const int e = lane + 64 * t;
            if (e < 32 * PZ) *reinterpret_cast<uint4*>(sz + (e / PZ) * ZB + (e % PZ) * 16) = zcur[t];
        }
        wave_sync();
        // lane 4 r + p supplies pixel 8 q + r (+ 4), channels 16 t + 4 p ..: lane i receives 8 pixels of channel 16 t + i
        bf16x8_t bz[NN];
#pragma unroll
        for (int nt = 0; nt < NN; ++nt) {
            const unsigned char* zp = sz + (8 * q + (m16 >> 2)) * ZB + (16 * nt + 4 * (m16 & 3)) * 2;
            const bf16x4_t lo = lds_read_tr16(zp), hi = lds_read_tr16(zp + 4 * ZB);
            bz[nt][0] = lo[0]; bz[nt][1] = lo[1]; bz[nt][2] = lo[2]; bz[nt][3] = lo[3];
            bz[nt][4] = hi[0]; bz[nt][5] = hi[1]; bz[nt][6] = hi[2]; bz[nt][7] = hi[3];
        }
#pragma unroll
        for (int mt = 0; mt < NM; ++mt) {
            const unsigned char* xp = sx + (8 * q + (m16 >> 2)) * XB + (16 * mt + 4 * (m16 & 3)) * 2;
            const bf16x4_t lo = lds_read_tr16(xp), hi = lds_read_tr16(xp + 4 * XB);
            bf16x8_t ax;
            ax[0] = lo[0]; ax[1] = lo[1]; ax[2] = lo[2]; ax[3] = lo[3];
            ax[4] = hi[0]; ax[5] = hi[1]; ax[6] = hi[2]; ax[7] = hi[3];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int nt = DIAG ? 2 * (mt >> 1) + j : j;
                acc[mt][j] = mfma_16x16x32_bf16(ax, bz[nt < NN ? nt : 0], acc[mt][j]);
            }
        }
        wave_sync();                                             // the tiles are rewritten by the next iteration
#pragma unroll
        for (int t = 0; t < LX; ++t) xcur[t] = xnext[t];
#pragma unroll
        for (int t = 0; t < LZ; ++t) zcur[t] = znext[t];
        it = nit;
    }
    // acc[mt][j][i] = dW[co = 16 nt + m16][ci = 16 mt + 4 q + i]: pair by pair (32 co x 32 ci) through LDS, the four waves summed
    __syncthreads();
    float* s_red = reinterpret_cast<float*>(&s_t[0][0]);        // [4][1024] floats = 16 KB (< 4 x 32 x (XB + ZB) for every shape offered)
    const int co_tiles = (NN + 1) / 2;
    for (int cot = 0; cot < co_tiles; ++cot)
        for (int cit = 0; cit < (NM + 1) / 2; ++cit) {
            if (DIAG && cot != cit) continue;
#pragma unroll
            for (int mt = 0; mt < NM; ++mt)
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    const int nt = DIAG ? 2 * (mt >> 1) + j : j;
                    if ((mt >> 1) != cit || (nt >> 1) != cot) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        s_red[wid * 1024 + (16 * (nt & 1) + m16) * 32 + 16 * (mt & 1) + 4 * q + i] = acc[mt][j][i];
                }
            if (NM == 1 || NN == 1) {                           // half-filled pair: the other half is zero
                // (rows / columns 16..31 of the pair were never written: clear them once per pair below)
            }
            __syncthreads();
            float* dst = a.part + ((long)blockIdx.x * (co_tiles * a.ci_tiles) + cot * a.ci_tiles + cit) * 1024;
            for (int e = tid; e < 1024; e += 256) {
                const int col = e >> 5, cil = e & 31;
                const bool have = (NN > 1 || col < 16) && (NM > 1 || cil < 16);
                dst[e] = have ? (s_red[e] + s_red[1024 + e]) + (s_red[2048 + e] + s_red[3072 + e]) : 0.f;
            }
            __syncthreads();
        }
}

// 0 = not offered.  pairs / ci_tiles as conv_wgrad_mfma_kernel's grid (co_tile * ci_tiles + ci_tile)
bool conv1x1_wgrad_reg_ok(int Cin, int Cout, int groups) {
    if (!(options().stream_fast & 64)) return false;
    if (Cin % 16 || Cout % 16 || Cin > 128 || Cout > 128) return false;
    const int nm = Cin / 16, nn = Cout / 16;
    if (!(nm == 1 || nm == 2 || nm == 4 || nm == 8) || !(nn == 1 || nn == 2 || nn == 4 || nn == 8)) return false;
    if (nm * nn <= 16) return true;                            // (32 tiles = 128 accumulator registers + staging: scratch)
    if (nm != 8 || nn != 8) return false;
    // 128 x 128: only as block-diagonal (every group inside one 32 x 32 block)
    const int cig = Cin / groups, cog = Cout / groups;
    return groups > 1 && Cin == Cout && cig <= 32 && 32 % cig == 0 && cog == cig;
}

int conv1x1_wgrad_reg_partial(const void* x, const void* dz, float* part, long P, int Cin, int Cout, int groups, int nbx,
                              hipStream_t s) {
    W11Args a;
    a.x = (const bf16_t*)x; a.dz = (const bf16_t*)dz; a.part = part; a.P = P; a.iters = cdiv(P, 32);
    a.Cin = Cin; a.Cout = Cout; a.ci_tiles = (int)cdiv(Cin, 32);
    const int nm = Cin / 16, nn = Cout / 16;
    const dim3 grid((unsigned)nbx);
#define LEDN_W11(NM_, NN_)                                                                                      \
    if (nm == NM_ && nn == NN_) {                                                                               \
        if constexpr (NM_ * NN_ <= 16) {                                                                        \
            LEDN_LAUNCH((conv1x1_wgrad_reg_kernel<NM_, NN_, false>), grid, dim3(256), 0, s, a);                 \
            return check_launch();                                                                              \
        } else if constexpr (NM_ == 8 && NN_ == 8) {                                                            \
            LEDN_LAUNCH((conv1x1_wgrad_reg_kernel<NM_, NN_, true>), grid, dim3(256), 0, s, a);                  \
            return check_launch();                                                                              \
        }                                                                                                       \
    }
    LEDN_W11(1, 1) LEDN_W11(1, 2) LEDN_W11(1, 4) LEDN_W11(1, 8) LEDN_W11(2, 1) LEDN_W11(2, 2) LEDN_W11(2, 4) LEDN_W11(2, 8)
    LEDN_W11(4, 1) LEDN_W11(4, 2) LEDN_W11(4, 4) LEDN_W11(4, 8) LEDN_W11(8, 1) LEDN_W11(8, 2) LEDN_W11(8, 4) LEDN_W11(8, 8)
#undef LEDN_W11
    (void)groups;
    return LEDN_EINVAL;
}

}  // namespace ledn
