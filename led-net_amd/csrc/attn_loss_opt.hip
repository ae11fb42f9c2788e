// attn_loss_opt.hip -- window-attention backward, fused OHEM cross-entropy
// (radix-select threshold instead of the reference's full sort), multi-tensor SGD.
#include "ledn_rt.h"

namespace ledn {

// ===========================================================================
// Window attention backward.  One wavefront per (window, head); lane = token.
// Phase A (lane = query i): recompute P[i][:], dP = dO.V^T, dS = P*(dP - rowsum(P*dP)),
//   dQ[i] = scale * dS[i][:] . K ; dS kept in LDS (64x64 f32) and added to dbiasT.
// Phase B (lane = key j): dK[j] = scale * dS[:][j]^T . Q,  dV[j] = P[:][j]^T . dO.
// LDS: K, V, Q, dO (4 x 64 x D f32) + dS + P (2 x 16 KiB) = 48 KiB at D = 16.
// ===========================================================================
// The wavefront walks several windows of its head; the 64x64 score-gradient rows live in LDS with a
// 65-float row stride (lane = row in phase A, lane = column in phase B: both conflict-free; the
// 64-float stride of the first version put all 64 lanes on one bank), and the relative-position
// bias gradient is summed over the wavefront's windows in LDS and leaves as ONE partial table per
// wavefront (the first version issued one global atomic per score and window: 1024 same-address
// atomics per table entry at 16 x 64 x 64).
template <typename T, int D, bool PADDED>
__global__ void __launch_bounds__(64) window_attn_bwd_kernel(const T* qkv, const float* biasT, const T* dout,
                                                             float* dqkv, float* dbiasT, int N, int H,
                                                             int W, int C, int heads, int hh, int ww,
                                                             float* part, float* pad_out) {
    // LDS per wavefront: K, V, Q, dO (16 KB at D = 16) + ONE 64 x 65 score matrix that holds P and is then
    // overwritten by dS (dV = P^T dO is taken in between), the bias-gradient rows live in registers:
    // 33 KB -> four wavefronts per CU (the first version kept P, dS and the bias rows in LDS: 66 KB, two
    // wavefronts per CU, two of the four SIMDs idle).
    constexpr int WS = 8, T2 = 64, TP = 65;
    __shared__ float s_k[T2 * D], s_v[T2 * D], s_q[T2 * D], s_do[T2 * D];
    __shared__ float s_p[T2 * TP];
    const int head = blockIdx.y;
    const int t = threadIdx.x;
    const int nwin = N * hh * ww;
    const float scale = rsqrtf((float)D);
    const float* b = biasT + (long)head * T2 * T2 + t;
    float db[T2];                                  // bias-gradient row of query t, summed over this wave's windows
#pragma unroll
    for (int k = 0; k < T2; ++k) db[k] = 0.f;
    for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
        const int wx = win % ww, wy = (win / ww) % hh, n = win / (ww * hh);
        const int y = wy * WS + t / WS, x = wx * WS + t % WS;
        const bool inside = y < H && x < W;
        const int ys = y < H ? y : 2 * H - 2 - y;
        const int xs = x < W ? x : 2 * W - 2 - x;
        const long src = ((long)n * H + ys) * W + xs;
        const T* p = qkv + src * (3L * C) + head * D;
        float q[D], go[D];
        __syncthreads();     // the previous window's last phase is done with the LDS operands
#pragma unroll
        for (int j = 0; j < D; j += 4) {
            float kv[4], vv[4];
            ldv<4>(p + j, q + j);
            ldv<4>(p + C + j, kv);
            ldv<4>(p + 2 * C + j, vv);
            if (inside) ldv<4>(dout + src * C + head * D + j, go + j);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) go[j + i] = 0.f;  // cropped outputs carry no gradient
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s_k[t * D + j + i] = kv[i];
                s_v[t * D + j + i] = vv[i];
                s_q[t * D + j + i] = q[j + i];
                s_do[t * D + j + i] = go[j + i];
            }
        }
        __syncthreads();
        // ---- A1 (lane = query t): P[t][:] = softmax(q.K^T * scale + bias)
        float m = -3.0e38f;
        for (int k0 = 0; k0 < T2; k0 += 8) {
            float bk[8];                             // eight bias loads in flight (a dependent global load per
#pragma unroll                                       // score made this loop latency-bound)
            for (int u = 0; u < 8; ++u) bk[u] = b[(k0 + u) * T2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u;
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) dot = fmaf(q[j], s_k[k * D + j], dot);
                const float sc = dot * scale + bk[u];
                s_p[t * TP + k] = sc;
                m = fmaxf(m, sc);
            }
        }
        float l = 0.f;
        for (int k = 0; k < T2; ++k) {
            const float e = __expf(s_p[t * TP + k] - m);
            s_p[t * TP + k] = e;
            l += e;
        }
        const float inv = 1.f / l;
        float rs = 0.f;  // sum_k P*dP
        for (int k = 0; k < T2; ++k) {
            const float pk = s_p[t * TP + k] * inv;
            float dp = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) dp = fmaf(go[j], s_v[k * D + j], dp);
            s_p[t * TP + k] = pk;
            rs = fmaf(pk, dp, rs);
        }
        __syncthreads();
        // ---- B1 (lane = key t): dV[t] = P[:, t]^T . dO
        float dv[D];
#pragma unroll
        for (int j = 0; j < D; ++j) dv[j] = 0.f;
        for (int i = 0; i < T2; ++i) {
            const float pk = s_p[i * TP + t];
#pragma unroll
            for (int j = 0; j < D; ++j) dv[j] = fmaf(pk, s_do[i * D + j], dv[j]);
        }
        __syncthreads();
        // ---- A2 (lane = query t): dS = P * (dP - rs) overwrites P (dP recomputed: 16 FMAs), dQ, bias rows
        float dq[D];
#pragma unroll
        for (int j = 0; j < D; ++j) dq[j] = 0.f;
#pragma unroll
        for (int k = 0; k < T2; ++k) {
            float dp = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) dp = fmaf(go[j], s_v[k * D + j], dp);
            const float ds = s_p[t * TP + k] * (dp - rs);
            s_p[t * TP + k] = ds;
            db[k] += ds;
#pragma unroll
            for (int j = 0; j < D; ++j) dq[j] = fmaf(ds, s_k[k * D + j], dq[j]);
        }
        __syncthreads();
        // ---- B2 (lane = key t): dK[t] = scale * dS[:, t]^T . Q
        float dk[D];
#pragma unroll
        for (int j = 0; j < D; ++j) dk[j] = 0.f;
        for (int i = 0; i < T2; ++i) {
            const float ds = s_p[i * TP + t];
#pragma unroll
            for (int j = 0; j < D; ++j) dk[j] = fmaf(ds, s_q[i * D + j], dk[j]);
        }
        float* o = dqkv + src * (3L * C) + head * D;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (PADDED && pad_out) {    // deterministic mode: every padded position is stored once, window_attn_fold_kernel
                float* po = pad_out + (((long)n * (hh * WS) + y) * (ww * WS) + x) * (3L * C) + head * D;   // adds the copies
                po[j] = dq[j] * scale;
                po[C + j] = dk[j] * scale;
                po[2 * C + j] = dv[j];
            } else if (PADDED) {
                atomicAdd(o + j, dq[j] * scale);
                atomicAdd(o + C + j, dk[j] * scale);
                atomicAdd(o + 2 * C + j, dv[j]);
            } else {
                o[j] = dq[j] * scale;
                o[C + j] = dk[j] * scale;
                o[2 * C + j] = dv[j];
            }
        }
    }
    // bias-gradient table of this wavefront: dbiasT[head][k][t] (t = query row = lane)
#pragma unroll
    for (int k = 0; k < T2; ++k) {
        const long idx = (long)head * T2 * T2 + k * T2 + t;
        if (part) part[(long)blockIdx.x * heads * T2 * T2 + idx] = db[k];
        else atomicAdd(dbiasT + idx, db[k]);
    }
}

// ---------------------------------------------------------------------------
// Matrix-core variant (bf16, head dim 16, no reflect padding): the five 64 x 64 x 16 products of a
// (window, head) as v_mfma_f32_32x32x16_bf16.  Everything score-shaped is kept TRANSPOSED,
// S^T[key j][query i], so that an accumulator lane owns ONE query (column) and its registers run over
// the keys: the softmax / row-sum reductions are in-lane plus one exchange with lane ^ 32, and the
// bias-gradient table dbiasT[head][j][i] accumulates in the same register layout across the
// wavefront's windows.  S^T = K Q^T and dP^T = V dO^T take their fragments straight from global
// memory (16 B per lane); dQ^T = K^T dS^T uses the accumulator registers as the B operand (the
// contraction index is permuted consistently on the A side); dV^T = dO^T P and dK^T = Q^T dS need
// the score matrix transposed, which goes through one 64 x 64 bf16 LDS tile.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) window_attn_bwd_mfma_kernel(const bf16_t* qkv, const float* biasT,
                                                                  const bf16_t* dout, float* dqkv, float* dbiasT,
                                                                  int N, int H, int W, int C, int heads, int hh,
                                                                  int ww, float* part) {
    constexpr int D = 16, T2 = 64, LP = 72;      // LDS row stride (bf16 elements): 64 + 8 pad
    __shared__ __attribute__((aligned(16))) unsigned short s_kt[D * LP], s_qt[D * LP], s_dot[D * LP];
    __shared__ __attribute__((aligned(16))) unsigned short s_m[T2 * LP];
    const int head = blockIdx.y;
    const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
    const int nwin = N * hh * ww;
    const float scale = rsqrtf((float)D);
    // bias and its gradient in accumulator layout: [jt][it][reg] <-> key jt*32 + row(reg, lh), query it*32 + lr
    float bias[2][2][16], db[2][2][16];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, i = it * 32 + lr;
                bias[jt][it][r] = biasT[(long)head * T2 * T2 + j * T2 + i];
                db[jt][it][r] = 0.f;
            }
    const bf16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
        const int wx = win % ww, wy = (win / ww) % hh, n = win / (ww * hh);
        // fragments of tokens lr and 32 + lr (rows of the two 32-row tiles), dims lh*8 .. lh*8+7
        bf16x8_t kf[2], vf[2], qf[2], dof[2];
        long tokoff[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int tok = tl * 32 + lr;
            const long pix = ((long)n * H + wy * 8 + tok / 8) * W + wx * 8 + tok % 8;
            tokoff[tl] = pix;
            const bf16_t* p = qkv + pix * (3L * C) + head * D + lh * 8;
            qf[tl] = *reinterpret_cast<const bf16x8_t*>(p);
            kf[tl] = *reinterpret_cast<const bf16x8_t*>(p + C);
            vf[tl] = *reinterpret_cast<const bf16x8_t*>(p + 2 * C);
            dof[tl] = *reinterpret_cast<const bf16x8_t*>(dout + pix * C + head * D + lh * 8);
        }
        __syncthreads();     // previous window's readers of the LDS tiles are done
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int e = 0; e < 8; ++e) {          // transposed copies [dim][token]
                const int d = lh * 8 + e, tok = tl * 32 + lr;
                s_kt[d * LP + tok] = (unsigned short)kf[tl][e];
                s_qt[d * LP + tok] = (unsigned short)qf[tl][e];
                s_dot[d * LP + tok] = (unsigned short)dof[tl][e];
            }
        // ---- S^T = K Q^T, dP^T = V dO^T  (one MFMA per 32 x 32 tile: K = 16)
        f32x16_t st[2][2], dp[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                f32x16_t z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                st[jt][it] = mfma_32x32x16_bf16(kf[jt], qf[it], z);
                dp[jt][it] = mfma_32x32x16_bf16(vf[jt], dof[it], z);
            }
        // ---- softmax over the keys of each query column, dS^T = P^T * (dP^T - rowsum(P * dP))
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float m = -3.0e38f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[jt][it][r] = st[jt][it][r] * scale + bias[jt][it][r];
                    m = fmaxf(m, st[jt][it][r]);
                }
            m = fmaxf(m, __shfl_xor(m, 32));
            float l = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[jt][it][r] = __expf(st[jt][it][r] - m);
                    l += st[jt][it][r];
                }
            l += __shfl_xor(l, 32);
            const float inv = 1.f / l;
            float rs = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[jt][it][r] *= inv;
                    rs = fmaf(st[jt][it][r], dp[jt][it][r], rs);
                }
            rs += __shfl_xor(rs, 32);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dp[jt][it][r] = st[jt][it][r] * (dp[jt][it][r] - rs);     // dS^T
                    db[jt][it][r] += dp[jt][it][r];
                }
        }
        // ---- P^T -> LDS [j][i] (bf16), dV^T[d][j] = sum_i dO^T[d][i] P[i][j]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s_m[(jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + it * 32 + lr] = f32_to_bf16(st[jt][it][r]);
        __syncthreads();
        f32x16_t dvt[2], dkt[2], dqt[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) dvt[tl][r] = dkt[tl][r] = dqt[tl][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t a = lr < D ? *reinterpret_cast<const bf16x8_t*>(&s_dot[lr * LP + kk * 16 + lh * 8]) : zero8;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(&s_m[(jt * 32 + lr) * LP + kk * 16 + lh * 8]);
                dvt[jt] = mfma_32x32x16_bf16(a, b, dvt[jt]);
            }
        }
        __syncthreads();
        // ---- dS^T -> LDS, dK^T[d][j] = sum_i Q^T[d][i] dS[i][j]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s_m[(jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + it * 32 + lr] = f32_to_bf16(dp[jt][it][r]);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t a = lr < D ? *reinterpret_cast<const bf16x8_t*>(&s_qt[lr * LP + kk * 16 + lh * 8]) : zero8;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(&s_m[(jt * 32 + lr) * LP + kk * 16 + lh * 8]);
                dkt[jt] = mfma_32x32x16_bf16(a, b, dkt[jt]);
            }
        }
        // ---- dQ^T[d][i] = sum_j K^T[d][j] dS^T[j][i]: B straight from the dS^T registers.  k-step (jt, h)
        // covers keys jt*32 + 16h .. +15 in the accumulator's own order: logical k = lh*8 + e <->
        // key jt*32 + 16h + (e&3) + 8*(e>>2) + 4*lh, and the A operand (K^T) is gathered in that order.
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8_t a = zero8;
                if (lr < D) {
                    const int j0 = jt * 32 + 16 * h + 4 * lh;
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = (short)s_kt[lr * LP + j0 + (e & 3) + 8 * (e >> 2)];
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    bf16x8_t b;
#pragma unroll
                    for (int e = 0; e < 8; ++e) b[e] = (short)f32_to_bf16(dp[jt][it][8 * h + e]);
                    dqt[it] = mfma_32x32x16_bf16(a, b, dqt[it]);
                }
            }
        // ---- stores: lane (token tl*32 + lr, lh) holds dims 4*lh .. +3 (regs 0-3) and 8 + 4*lh .. +3 (regs 4-7)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            float* o = dqkv + tokoff[tl] * (3L * C) + head * D + 4 * lh;
            float t4[4];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[e] = dqt[tl][4 * g + e] * scale;
                st4(o + 8 * g, t4);
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[e] = dkt[tl][4 * g + e] * scale;
                st4(o + C + 8 * g, t4);
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[e] = dvt[tl][4 * g + e];
                st4(o + 2 * C + 8 * g, t4);
            }
        }
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, i = it * 32 + lr;
                const long idx = (long)head * T2 * T2 + j * T2 + i;
                if (part) part[(long)blockIdx.x * heads * T2 * T2 + idx] = db[jt][it][r];
                else atomicAdd(dbiasT + idx, db[jt][it][r]);
            }
}

// reflect-padded windows, deterministic mode: dqkv[n, ys, xs, :] += the gradients of the (up to four) padded positions whose
// source pixel is (ys, xs) -- itself, its mirror image in the bottom band, in the right band, in the corner -- in that order
__global__ void __launch_bounds__(256) window_attn_fold_kernel(const float* pad, float* dqkv, int N, int H, int W, int Hp,
                                                               int Wp, int C3) {
    const long total = (long)N * H * W * C3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C3);
    const long pix = idx / C3;
    const NhwcIdx ix_ = pix_split(pix, W, H);
    const int xs = ix_.x, ys = ix_.y, n = ix_.n;
    const int ym = 2 * H - 2 - ys, xm = 2 * W - 2 - xs;
    const bool my = ym >= H && ym < Hp, mx = xm >= W && xm < Wp;
    const float* base = pad + (long)n * Hp * Wp * C3 + c;
    float t = base[((long)ys * Wp + xs) * C3];
    if (my) t += base[((long)ym * Wp + xs) * C3];
    if (mx) t += base[((long)ys * Wp + xm) * C3];
    if (my && mx) t += base[((long)ym * Wp + xm) * C3];
    dqkv[idx] += t;
}

int window_attn_bwd_impl(const void* qkv, const float* biasT, const void* dout, float* dqkv, float* dbiasT,
                         int N, int H, int W, int C, int heads, int ws, int dtype, hipStream_t s) {
    LEDN_REQUIRE(qkv && biasT && dout && dqkv && dbiasT && N > 0 && H > 0 && W > 0 && C > 0 && heads > 0);
    LEDN_REQUIRE(ws == 8 && C % heads == 0);
    const int D = C / heads;
    const int hh = (H + ws - 1) / ws, ww = (W + ws - 1) / ws;
    LEDN_REQUIRE(hh * ws - H < H && ww * ws - W < W);
    const bool padded = (H % ws) || (W % ws);
    long nb = (long)N * hh * ww;
    if (nb > 256) nb = 256;                                   // wavefronts per head (each walks its windows)
    const long part_floats = nb * heads * 64 * 64;
    const long pad_floats = (padded && det()) ? (long)N * hh * ws * ww * ws * 3 * C : 0;
    float* part = (nb > 8 || det()) ? ws_take(part_floats + pad_floats) : nullptr;
    if (det() && !part) return LEDN_EINVAL;
    float* pad_out = pad_floats ? part + part_floats : nullptr;
    if (!part && nb > 32) nb = 32;                            // atomics fallback: <= 32 per table entry
    const dim3 grid((unsigned)nb, (unsigned)heads);
    if (dtype == LEDN_BF16 && D == 16 && !padded && C % 8 == 0) {   // matrix-core variant
        LEDN_LAUNCH(window_attn_bwd_mfma_kernel, grid, dim3(64), 0, s, (const bf16_t*)qkv, biasT, (const bf16_t*)dout,
                    dqkv, dbiasT, N, H, W, C, heads, hh, ww, part);
        if (part) return finish_partials(part, (int)nb, heads * 64 * 64, 1, dbiasT, nullptr, nullptr, s);
        return check_launch();
    }
#define LEDN_WB(T, DD)                                                                                  \
    do {                                                                                                \
        if (padded)                                                                                     \
            LEDN_LAUNCH((window_attn_bwd_kernel<T, DD, true>), grid, dim3(64), 0, s, (const T*)qkv, biasT, \
                        (const T*)dout, dqkv, dbiasT, N, H, W, C, heads, hh, ww, part, pad_out);        \
        else                                                                                            \
            LEDN_LAUNCH((window_attn_bwd_kernel<T, DD, false>), grid, dim3(64), 0, s, (const T*)qkv, biasT, \
                        (const T*)dout, dqkv, dbiasT, N, H, W, C, heads, hh, ww, part, (float*)nullptr); \
    } while (0)
#define LEDN_WBD(T)                        \
    do {                                   \
        if (D == 16) LEDN_WB(T, 16);       \
        else if (D == 32) LEDN_WB(T, 32);  \
        else return LEDN_EINVAL;           \
    } while (0)
    if (dtype == LEDN_F32) LEDN_WBD(float);
    else if (dtype == LEDN_BF16) LEDN_WBD(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_WBD
#undef LEDN_WB
    if (pad_out) {
        const long total = (long)N * H * W * 3 * C;
        LEDN_LAUNCH(window_attn_fold_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, pad_out, dqkv, N, H, W,
                    hh * ws, ww * ws, 3 * C);
    }
    if (part) return finish_partials(part, (int)nb, heads * 64 * 64, 1, dbiasT, nullptr, nullptr, s);
    return check_launch();
}

// ===========================================================================
// OHEM cross-entropy.
// work layout (floats): prob[P] | loss[P] | u32 hist[3][2048] | u32 state[16]
// state: 0 n_valid, 1 n_correct, 2 rank (remaining), 3 prefix bits, 4 thr bits,
//        5 n_selected, 6 (float) loss_sum
// ===========================================================================
constexpr int OH_BINS = 2048;
struct OhemWork {
    float* prob;
    float* loss;
    unsigned* hist;
    unsigned* state;
};
__host__ __device__ inline OhemWork ohem_work(float* work, long long P) {
    OhemWork w;
    w.prob = work;
    w.loss = work + P;
    w.hist = reinterpret_cast<unsigned*>(work + 2 * P);
    w.state = w.hist + 3 * OH_BINS;
    return w;
}
long long ohem_work_floats(long long P) { return 2 * P + 3 * OH_BINS + 16; }

__device__ __forceinline__ int oh_bin(unsigned u, int pass) {
    return pass == 0 ? (int)(u >> 21) : (pass == 1 ? (int)((u >> 10) & 2047u) : (int)(u & 1023u));
}

// pass over the pixels: prob of the target class, CE, validity, accuracy, level-0 histogram
template <int CT>
__global__ void __launch_bounds__(256) ohem_prob_kernel(const float* logits, const long long* target, long P,
                                                        int C, int ignore_label, float* work) {
    __shared__ unsigned s_hist[OH_BINS];
    __shared__ unsigned s_cnt[2];
    const OhemWork w = ohem_work(work, P);
    for (int i = threadIdx.x; i < OH_BINS; i += blockDim.x) s_hist[i] = 0u;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const long long tg = target[p];
        if (tg == ignore_label) {
            w.prob[p] = 2.0f;   // > any probability: never selected
            w.loss[p] = 0.f;
            continue;
        }
        const float* lg = logits + p * C;
        float mx = lg[0];
        int am = 0;
        const int cc = CT > 0 ? CT : C;
#pragma unroll
        for (int c = 1; c < cc; ++c)
            if (lg[c] > mx) { mx = lg[c]; am = c; }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < cc; ++c) se += __expf(lg[c] - mx);
        const float lt = lg[(int)tg] - mx;
        const float pr = __expf(lt) / se;
        w.prob[p] = pr;
        w.loss[p] = __logf(se) - lt;
        atomicAdd(&s_hist[oh_bin(__float_as_uint(pr), 0)], 1u);
        atomicAdd(&s_cnt[0], 1u);
        if (am == (int)tg) atomicAdd(&s_cnt[1], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < OH_BINS; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&w.hist[i], s_hist[i]);
    if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(&w.state[threadIdx.x], s_cnt[threadIdx.x]);
}

// single workgroup: locate the bucket holding the wanted rank at this level.  Thread i owns bins
// 8i..8i+7; a block-wide prefix sum of the per-thread counts finds the owner of the rank, which
// then walks its eight bins (the first version walked all 2048 bins with one thread: 75 us).
__global__ void __launch_bounds__(256) ohem_scan_kernel(float* work, long P, int pass, long long min_kept,
                                                        float thres) {
    __shared__ unsigned s_wave[4];
    const OhemWork w = ohem_work(work, P);
    const unsigned nv = w.state[0];
    if (nv == 0u) {   // workgroup-uniform
        if (pass == 0 && threadIdx.x == 0) {
            w.state[4] = __float_as_uint(thres);
            w.state[2] = 0u;
            w.state[3] = 0u;
        }
        return;
    }
    unsigned rank;
    if (pass == 0) {
        const long long k = min_kept < (long long)nv - 1 ? min_kept : (long long)nv - 1;
        rank = (unsigned)k;
    } else {
        rank = w.state[2];
    }
    const unsigned prefix_bits = pass == 0 ? 0u : w.state[3];
    const unsigned* h = w.hist + pass * OH_BINS;
    const int nb = pass == 2 ? 1024 : 2048;
    const int per = nb / 256;
    unsigned hb[8], mine = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hb[i] = i < per ? h[threadIdx.x * per + i] : 0u;
        mine += hb[i];
    }
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned nbr = __shfl(incl, lane >= o ? lane - o : lane);
        if (lane >= o) incl += nbr;
    }
    if (lane == 63) s_wave[wid] = incl;
    __syncthreads();          // also orders every thread's reads of state[2..3] before the writes below
    unsigned off = 0u;
    for (int i = 0; i < wid; ++i) off += s_wave[i];
    incl += off;
    const unsigned excl = incl - mine;
    const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    int b = -1;
    unsigned r = 0u;
    if (rank >= excl && rank < incl) {          // exactly one thread when rank < total
        r = rank - excl;
        b = threadIdx.x * per;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= per || r < hb[i]) break;
            r -= hb[i];
            ++b;
        }
    } else if (rank >= total && threadIdx.x == 255) {   // rank beyond the histogram: last bucket
        r = rank - total;
        b = nb - 1;
    }
    if (b >= 0) {
        const unsigned bits = pass == 0 ? ((unsigned)b << 21) : (pass == 1 ? ((unsigned)b << 10) : (unsigned)b);
        const unsigned full = prefix_bits | bits;
        w.state[3] = full;
        w.state[2] = r;
        if (pass == 2) {
            const float kth = __uint_as_float(full);
            w.state[4] = __float_as_uint(kth > thres ? kth : thres);   // threshold = max(min_value, thresh)
        }
    }
}

__global__ void __launch_bounds__(256) ohem_hist_kernel(float* work, long P, int pass) {
    __shared__ unsigned s_hist[OH_BINS];
    const OhemWork w = ohem_work(work, P);
    for (int i = threadIdx.x; i < OH_BINS; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
    const unsigned prefix = w.state[3];
    const unsigned mask = pass == 1 ? 0xffe00000u : 0xfffffc00u;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const unsigned u = __float_as_uint(w.prob[p]);
        if ((u & mask) == prefix) atomicAdd(&s_hist[oh_bin(u, pass)], 1u);
    }
    __syncthreads();
    unsigned* h = w.hist + pass * OH_BINS;
    for (int i = threadIdx.x; i < OH_BINS; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&h[i], s_hist[i]);
}

__global__ void __launch_bounds__(256) ohem_reduce_kernel(float* work, long P) {
    __shared__ float s_sum[4];
    __shared__ unsigned s_cnt[4];
    const OhemWork w = ohem_work(work, P);
    const float thr = __uint_as_float(w.state[4]);
    float sum = 0.f;
    unsigned cnt = 0u;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        if (w.prob[p] < thr) {
            sum += w.loss[p];
            ++cnt;
        }
    }
    sum = wave_sum(sum);
    float cf = wave_sum((float)cnt);   // exact below 2^24 per wave
    if ((threadIdx.x & 63) == 0) {
        s_sum[threadIdx.x >> 6] = sum;
        s_cnt[threadIdx.x >> 6] = (unsigned)cf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<float*>(&w.state[6]), s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
        atomicAdd(&w.state[5], s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
    }
}

__global__ void ohem_final_kernel(float* work, long P, float loss_weight, float* out) {
    const OhemWork w = ohem_work(work, P);
    const unsigned nv = w.state[0], nsel = w.state[5];
    const float sum = __uint_as_float(w.state[6]);
    out[0] = nv == 0u ? 0.f : loss_weight * (sum / (float)nsel);   // 0/0 -> NaN like the reference
    const float eps = 1.1920929e-07f;
    out[1] = ((float)w.state[1] + eps) * (100.0f / ((float)nv + eps));
    out[2] = __uint_as_float(w.state[4]);
    out[3] = (float)nsel;
}

int ohem_ce_fwd_impl(const float* logits, const long long* target, long long P, int C, float thres,
                     long long min_kept, float loss_weight, int ignore_label, float* work, float* out,
                     hipStream_t s) {
    LEDN_REQUIRE(logits && target && work && out && P > 0 && C > 1 && min_kept >= 1);
    LEDN_REQUIRE(P < (1LL << 31));
    const OhemWork w = ohem_work(work, P);
    if (hipMemsetAsync(w.hist, 0, sizeof(unsigned) * (3 * OH_BINS + 16), s) != hipSuccess) return LEDN_ELAUNCH;
    const dim3 grid((unsigned)(cdiv(P, 256) < 2048 ? cdiv(P, 256) : 2048));
    if (C == 2) LEDN_LAUNCH(ohem_prob_kernel<2>, grid, dim3(256), 0, s, logits, target, (long)P, C, ignore_label, work);
    else LEDN_LAUNCH(ohem_prob_kernel<0>, grid, dim3(256), 0, s, logits, target, (long)P, C, ignore_label, work);
    LEDN_LAUNCH(ohem_scan_kernel, dim3(1), dim3(256), 0, s, work, (long)P, 0, min_kept, thres);
    LEDN_LAUNCH(ohem_hist_kernel, grid, dim3(256), 0, s, work, (long)P, 1);
    LEDN_LAUNCH(ohem_scan_kernel, dim3(1), dim3(256), 0, s, work, (long)P, 1, min_kept, thres);
    LEDN_LAUNCH(ohem_hist_kernel, grid, dim3(256), 0, s, work, (long)P, 2);
    LEDN_LAUNCH(ohem_scan_kernel, dim3(1), dim3(256), 0, s, work, (long)P, 2, min_kept, thres);
    LEDN_LAUNCH(ohem_reduce_kernel, grid, dim3(256), 0, s, work, (long)P);
    LEDN_LAUNCH(ohem_final_kernel, dim3(1), dim3(1), 0, s, work, (long)P, loss_weight, out);
    return check_launch();
}

template <int CT>
__global__ void __launch_bounds__(256) ohem_bwd_kernel(const float* logits, const long long* target, long P,
                                                       int C, int ignore_label, const float* work,
                                                       const float* out, const float* dloss,
                                                       float loss_weight, float* dlogits) {
    const float* prob = work;
    const float thr = out[2];
    const float coef = dloss[0] * loss_weight / out[3];
    const long stride = (long)gridDim.x * blockDim.x;
    const int cc = CT > 0 ? CT : C;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const long long tg = target[p];
        float* dl = dlogits + p * C;
        if (tg == ignore_label || !(prob[p] < thr)) {
#pragma unroll
            for (int c = 0; c < cc; ++c) dl[c] = 0.f;
            continue;
        }
        const float* lg = logits + p * C;
        float mx = lg[0];
#pragma unroll
        for (int c = 1; c < cc; ++c) mx = fmaxf(mx, lg[c]);
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < cc; ++c) se += __expf(lg[c] - mx);
        const float inv = 1.f / se;
#pragma unroll
        for (int c = 0; c < cc; ++c)
            dl[c] = coef * (__expf(lg[c] - mx) * inv - (c == (int)tg ? 1.f : 0.f));
    }
}

int ohem_ce_bwd_impl(const float* logits, const long long* target, long long P, int C, int ignore_label,
                     const float* work, const float* out, const float* dloss, float loss_weight,
                     float* dlogits, hipStream_t s) {
    LEDN_REQUIRE(logits && target && work && out && dloss && dlogits && P > 0 && C > 1);
    const dim3 grid((unsigned)(cdiv(P, 256) < 4096 ? cdiv(P, 256) : 4096));
    if (C == 2)
        LEDN_LAUNCH(ohem_bwd_kernel<2>, grid, dim3(256), 0, s, logits, target, (long)P, C, ignore_label, work,
                    out, dloss, loss_weight, dlogits);
    else
        LEDN_LAUNCH(ohem_bwd_kernel<0>, grid, dim3(256), 0, s, logits, target, (long)P, C, ignore_label, work,
                    out, dloss, loss_weight, dlogits);
    return check_launch();
}

// ---------------------------------------------------------------------------
// OHEM-CE on logits that are the bilinear (align_corners=False) upsampling of src [N,Hs,Ws,2] to H x W, without
// materialising them: LEDHead.loss_by_feat resizes each fused output to the label size right in front of the loss
// (led_head.py:132-138); at 16 x 1024^2 that tensor and its gradient are 134 MB each, written once and read twice.
// Forward: ohem_prob with the four-tap interpolation in front (any ratio).  Backward (exact 2x): a workgroup owns a
// 16 x 16 tile of src; the softmax gradients of its 34 x 34 children are formed once in LDS, then every src pixel
// gathers its 4 x 4 children with the interpolation weights (the adjoint of the resize), so neither dlogits nor a
// scatter with atomics exists.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void up_logits2(const float* src, int Ws, const Lerp& ly, const Lerp& lx, float& l0, float& l1) {
    const float2 v00 = *reinterpret_cast<const float2*>(src + ((long)ly.i0 * Ws + lx.i0) * 2);
    const float2 v01 = *reinterpret_cast<const float2*>(src + ((long)ly.i0 * Ws + lx.i1) * 2);
    const float2 v10 = *reinterpret_cast<const float2*>(src + ((long)ly.i1 * Ws + lx.i0) * 2);
    const float2 v11 = *reinterpret_cast<const float2*>(src + ((long)ly.i1 * Ws + lx.i1) * 2);
    l0 = ly.w0 * (lx.w0 * v00.x + lx.w1 * v01.x) + ly.w1 * (lx.w0 * v10.x + lx.w1 * v11.x);
    l1 = ly.w0 * (lx.w0 * v00.y + lx.w1 * v01.y) + ly.w1 * (lx.w0 * v10.y + lx.w1 * v11.y);
}

__global__ void __launch_bounds__(256) ohem_prob_up_kernel(const float* src, int N, int Hs, int Ws, int H, int W,
                                                           const long long* target, int ignore_label, float* work) {
    __shared__ unsigned s_hist[OH_BINS];
    __shared__ unsigned s_cnt[2];
    const long P = (long)N * H * W;
    const OhemWork w = ohem_work(work, P);
    for (int i = threadIdx.x; i < OH_BINS; i += blockDim.x) s_hist[i] = 0u;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const long long tg = target[p];
        if (tg == ignore_label) {
            w.prob[p] = 2.0f;
            w.loss[p] = 0.f;
            continue;
        }
        const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((long)W * H));
        float lg[2];
        up_logits2(src + (long)n * Hs * Ws * 2, Ws, lerp_coord(y, Hs, H), lerp_coord(x, Ws, W), lg[0], lg[1]);
        const int am = lg[1] > lg[0] ? 1 : 0;
        const float mx = lg[am];
        const float se = __expf(lg[0] - mx) + __expf(lg[1] - mx);
        const float lt = lg[(int)tg] - mx;
        const float pr = __expf(lt) / se;
        w.prob[p] = pr;
        w.loss[p] = __logf(se) - lt;
        atomicAdd(&s_hist[oh_bin(__float_as_uint(pr), 0)], 1u);
        atomicAdd(&s_cnt[0], 1u);
        if (am == (int)tg) atomicAdd(&s_cnt[1], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < OH_BINS; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&w.hist[i], s_hist[i]);
    if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(&w.state[threadIdx.x], s_cnt[threadIdx.x]);
}

int ohem_ce_up_fwd_impl(const float* src, int N, int Hs, int Ws, int H, int W, const long long* target, float thres,
                        long long min_kept, float loss_weight, int ignore_label, float* work, float* out,
                        hipStream_t s) {
    LEDN_REQUIRE(src && target && work && out && N > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0);
    const long long P = (long long)N * H * W;
    LEDN_REQUIRE(P < (1LL << 31) && min_kept >= 1);
    const OhemWork w = ohem_work(work, P);
    if (hipMemsetAsync(w.hist, 0, sizeof(unsigned) * (3 * OH_BINS + 16), s) != hipSuccess) return LEDN_ELAUNCH;
    const dim3 grid((unsigned)(cdiv(P, 256) < 2048 ? cdiv(P, 256) : 2048));
    LEDN_LAUNCH(ohem_prob_up_kernel, grid, dim3(256), 0, s, src, N, Hs, Ws, H, W, target, ignore_label, work);
    LEDN_LAUNCH(ohem_scan_kernel, dim3(1), dim3(256), 0, s, work, (long)P, 0, min_kept, thres);
    LEDN_LAUNCH(ohem_hist_kernel, grid, dim3(256), 0, s, work, (long)P, 1);
    LEDN_LAUNCH(ohem_scan_kernel, dim3(1), dim3(256), 0, s, work, (long)P, 1, min_kept, thres);
    LEDN_LAUNCH(ohem_hist_kernel, grid, dim3(256), 0, s, work, (long)P, 2);
    LEDN_LAUNCH(ohem_scan_kernel, dim3(1), dim3(256), 0, s, work, (long)P, 2, min_kept, thres);
    LEDN_LAUNCH(ohem_reduce_kernel, grid, dim3(256), 0, s, work, (long)P);
    LEDN_LAUNCH(ohem_final_kernel, dim3(1), dim3(1), 0, s, work, (long)P, loss_weight, out);
    return check_launch();
}

__global__ void __launch_bounds__(256) ohem_bwd_up2_kernel(const float* src, int N, int Hs, int Ws,
                                                           const long long* target, int ignore_label,
                                                           const float* work, const float* out, const float* dloss,
                                                           float loss_weight, float* dsrc) {
    constexpr int T = 16, CH = 2 * T + 2;                   // src tile, children per side
    __shared__ float2 s_g[CH * CH];
    const int H = 2 * Hs, W = 2 * Ws;
    const int tw = (Ws + T - 1) / T, th = (Hs + T - 1) / T;
    const int bj = blockIdx.x % tw, bi = (blockIdx.x / tw) % th, n = blockIdx.x / (tw * th);
    const int i0 = bi * T, j0 = bj * T;
    const float* prob = work;
    const float thr = out[2];
    const float coef = dloss[0] * loss_weight / out[3];
    const float* sn = src + (long)n * Hs * Ws * 2;
    for (int k = threadIdx.x; k < CH * CH; k += 256) {
        const int y = 2 * i0 - 1 + k / CH, x = 2 * j0 - 1 + k % CH;
        float2 g = make_float2(0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const long p = ((long)n * H + y) * W + x;
            const long long tg = target[p];
            if (tg != ignore_label && prob[p] < thr) {
                float l0, l1;
                up_logits2(sn, Ws, lerp_coord(y, Hs, H), lerp_coord(x, Ws, W), l0, l1);
                const float mx = fmaxf(l0, l1);
                const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx), inv = 1.f / (e0 + e1);
                g.x = coef * (e0 * inv - (tg == 0 ? 1.f : 0.f));
                g.y = coef * (e1 * inv - (tg == 1 ? 1.f : 0.f));
            }
        }
        s_g[k] = g;
    }
    __syncthreads();
    const int a = threadIdx.x / T, b = threadIdx.x % T, i = i0 + a, j = j0 + b;
    if (i >= Hs || j >= Ws) return;
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
        const int y = 2 * i - 1 + dy;
        if (y < 0 || y >= H) continue;
        const Lerp ly = lerp_coord(y, Hs, H);
        const float wy = (ly.i0 == i ? ly.w0 : 0.f) + (ly.i1 == i ? ly.w1 : 0.f);
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int x = 2 * j - 1 + dx;
            if (x < 0 || x >= W) continue;
            const Lerp lx = lerp_coord(x, Ws, W);
            const float wgt = wy * ((lx.i0 == j ? lx.w0 : 0.f) + (lx.i1 == j ? lx.w1 : 0.f));
            const float2 g = s_g[(2 * a + dy) * CH + 2 * b + dx];
            acc.x += wgt * g.x;
            acc.y += wgt * g.y;
        }
    }
    *reinterpret_cast<float2*>(dsrc + (((long)n * Hs + i) * Ws + j) * 2) = acc;
}

int ohem_ce_up_bwd_impl(const float* src, int N, int Hs, int Ws, int H, int W, const long long* target,
                        int ignore_label, const float* work, const float* out, const float* dloss, float loss_weight,
                        float* dsrc, hipStream_t s) {
    LEDN_REQUIRE(src && target && work && out && dloss && dsrc && N > 0 && Hs > 0 && Ws > 0);
    LEDN_REQUIRE(H == 2 * Hs && W == 2 * Ws);               // the fused adjoint is written for the exact 2x resize
    const long nb = (long)N * cdiv(Hs, 16) * cdiv(Ws, 16);
    LEDN_REQUIRE(nb < (1L << 31));
    LEDN_LAUNCH(ohem_bwd_up2_kernel, dim3((unsigned)nb), dim3(256), 0, s, src, N, Hs, Ws, target, ignore_label, work, out,
                dloss, loss_weight, dsrc);
    return check_launch();
}

// ===========================================================================
// multi-tensor SGD: grid = (chunks, tensors)
// ===========================================================================
__global__ void __launch_bounds__(256) sgd_kernel(const ledn_sgd_entry* table, float lr_arg, const float* lr_dev,
                                                  float momentum, float wd, float gscale) {
    const float lr = lr_dev ? *lr_dev : lr_arg;   // device-resident lr: a captured graph replays with a new value
    const ledn_sgd_entry e = table[blockIdx.y];
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < e.n; i += stride) {
        const float p = e.p[i];
        const float g = e.g[i] * gscale + wd * p;
        const float m = momentum * e.m[i] + g;
        e.m[i] = m;
        e.p[i] = p - lr * m;
        e.g[i] = 0.f;
    }
}

int sgd_step_impl(const ledn_sgd_entry* table_dev, int n_tensors, long long max_n, float lr,
                  const float* lr_dev, float momentum, float weight_decay, float grad_scale, hipStream_t s) {
    LEDN_REQUIRE(table_dev && n_tensors > 0 && max_n > 0);
    long chunks = cdiv(max_n, 256 * 8);
    if (chunks > 64) chunks = 64;
    LEDN_LAUNCH(sgd_kernel, dim3((unsigned)chunks, (unsigned)n_tensors), dim3(256), 0, s, table_dev, lr,
                lr_dev, momentum, weight_decay, grad_scale);
    return check_launch();
}

}  // namespace ledn
