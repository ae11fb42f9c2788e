// dwconv.hip -- depthwise convolutions: generic KxK (SESP second stage, GETB 8x8)
// and the fused SESP pyramid (4 dilated branches + hierarchical adds).
//
// HBM-bound streaming stencils.  One thread = one output pixel x 4 channels
// (16 B f32 / 8 B bf16 per lane, lanes run along channels then pixels, so a
// wave reads whole NHWC rows).  Filter taps are read through L1 (wave-coherent).
#include "ledn_rt.h"
#include "stencil.h"

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
        // contiguous pixel range per workgroup, XCD-aware: the rows above / below come from the same L2
        const long npix = (long)d.N * d.Ho * d.Wo;
        const long ppb = cdiv(cdiv(npix, (long)gridDim.x), (long)rows) * rows;
        const long p0 = (long)xcd_block(blockIdx.x, gridDim.x) * ppb, p1 = min(npix, p0 + ppb);
        for (long pix = p0 + r; pix < p1; pix += rows) {
            const NhwcIdx ix_ = pix_split(pix, d.Wo, d.Ho);
            const int wo = ix_.x, ho = ix_.y, n = ix_.n;
            float acc[V];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = 0.f;
            if (KK > 0) {
                // 3x3: nine unconditional tap loads in flight, then the FMAs
                float xv[9][V];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    int hi = ho * d.stride - padh + (t / 3) * dl, wi = wo * d.stride - padw + (t % 3) * dl;
                    const bool valid = hi >= 0 && hi < Hx && wi >= 0 && wi < Wx;
                    if (hi == d.H) hi = d.H - 2;  // ext1 reflect row / column
                    if (wi == d.W) wi = d.W - 2;
                    ldv_if<V>(x, (((long)n * d.H + hi) * d.W + wi) * d.C + c, valid, xv[t]);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[v] = fmaf(xv[t][v], wreg[t][v], acc[v]);
            } else {
                for (int kh2 = 0; kh2 < d.KH; ++kh2) {
                    int hi = ho * d.stride - padh + kh2 * dl;
                    if (hi < 0 || hi >= Hx) continue;
                    if (hi == d.H) hi = d.H - 2;  // ext1 reflect row
                    // eight taps of the row at a time, unconditional loads in flight together (the first
                    // version was a chain of 64 dependent, branch-guarded loads for the GETB 8x8 filter)
                    for (int kw0 = 0; kw0 < d.KW; kw0 += 8) {
                        float xv[8][V], wv[8][V];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int kw2 = kw0 + j;
                            int wi = wo * d.stride - padw + kw2 * dl;
                            const bool ok = kw2 < d.KW && wi >= 0 && wi < Wx;
                            if (wi == d.W) wi = d.W - 2;
                            ldv_if<V>(x, (((long)n * d.H + hi) * d.W + wi) * d.C + c, ok, xv[j]);
                            ldv<V>(d.w + (long)(kh2 * d.KW + (kw2 < d.KW ? kw2 : 0)) * d.C + c, wv[j]);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int v = 0; v < V; ++v) acc[v] = fmaf(xv[j][v], wv[j][v], acc[v]);
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

// bf16 3x3 depthwise conv, stride 1, zero padding = dilation (SESP second stage): the vectorised
// form of stencil.h.  Thread = 8 channels x pixel row r; a workgroup walks one contiguous pixel
// range (XCD-aware numbering); the nine 16-byte tap loads of a pixel are in flight together.
// FLIP = 1 computes the data gradient (taps mirrored, optional `add` input, no epilogue).
template <int FLIP>
__global__ void __launch_bounds__(256) dw3x3_bf16_kernel(ledn_dw_desc d, const bf16_t* add, float* part) {
    constexpr int V = 8;
    __shared__ float s_part[2][256 * V];
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    f32x2_t st1[4], st2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) st1[i] = st2[i] = f32x2_t{0.f, 0.f};
    if (r < rows) {
        const int dl = d.dil[c / d.group_size];
        const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
        bf16_t* y = reinterpret_cast<bf16_t*>(d.y);
        f32x2_t w[9][4];
        int toff[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f32x8_load(d.w + (long)(FLIP ? 8 - t : t) * d.C + c, w[t]);
            toff[t] = ((t / 3 - 1) * dl * d.W + (t % 3 - 1) * dl) * d.C;
        }
        f32x2_t sc[4], sh[4], ng[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[i] = f32x2_t{1.f, 1.f};
            sh[i] = f32x2_t{0.f, 0.f};
            ng[i] = d.act_out == LEDN_ACT_NONE ? f32x2_t{1.f, 1.f} : f32x2_t{0.f, 0.f};
        }
        if (!FLIP) {
            if (d.out_scale) f32x8_load(d.out_scale + c, sc);
            if (d.out_shift) f32x8_load(d.out_shift + c, sh);
            if (d.act_out == LEDN_ACT_PRELU) f32x8_load(d.slope + c, ng);
        }
        const float hi = (!FLIP && d.act_out == LEDN_ACT_RELU6) ? 6.f : 3.0e38f;
        const bool has_act = !FLIP && d.act_out != LEDN_ACT_NONE;
        const long npix = (long)d.N * d.H * d.W;
        const long ppb = cdiv(cdiv(npix, (long)gridDim.x), (long)rows) * rows;
        const long p0 = (long)xcd_block(blockIdx.x, gridDim.x) * ppb, p1 = min(npix, p0 + ppb);
        PixCursor cur;
        cur.init(p0 + r, d.H, d.W);
        unsigned base = (unsigned)((p0 + r) * d.C + c);
        for (long p = p0 + r; p < p1; p += rows, base += (unsigned)(rows * d.C), cur.advance(rows, d.H, d.W)) {
            const unsigned mask = tap_mask(cur.y, cur.x, dl, d.H, d.W);
            uint4 raw[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) raw[t] = ld_tap(x, base + (unsigned)toff[t], base, (mask >> t) & 1u);
            f32x2_t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x2_t{0.f, 0.f};
            if (FLIP && add) bf16x8_unpack(*reinterpret_cast<const uint4*>(add + base), acc);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x2_t xv[4];
                bf16x8_unpack(raw[t], xv);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = pk_fma(xv[i], w[t][i], acc[i]);
            }
            if (!FLIP) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = pk_fma(acc[i], sc[i], sh[i]);
                    st1[i] += acc[i];
                    st2[i] = pk_fma(acc[i], acc[i], st2[i]);
                }
                if (has_act) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i].x = fminf(fmaxf(acc[i].x, 0.f) + ng[i].x * fminf(acc[i].x, 0.f), hi);
                        acc[i].y = fminf(fmaxf(acc[i].y, 0.f) + ng[i].y * fminf(acc[i].y, 0.f), hi);
                    }
                }
            }
            *reinterpret_cast<uint4*>(y + base) = bf16x8_pack(acc);
        }
    }
    if (FLIP || !d.stat_sum) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s_part[0][threadIdx.x * V + 2 * i] = st1[i].x;
        s_part[0][threadIdx.x * V + 2 * i + 1] = st1[i].y;
        s_part[1][threadIdx.x * V + 2 * i] = st2[i].x;
        s_part[1][threadIdx.x * V + 2 * i + 1] = st2[i].y;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < d.C; ch += 256) {
        float sa = 0.f, sb = 0.f;
        for (int rr = 0; rr < rows; ++rr) {
            sa += s_part[0][(rr * cvn + ch / V) * V + ch % V];
            sb += s_part[1][(rr * cvn + ch / V) * V + ch % V];
        }
        if (part) {
            part[(long)blockIdx.x * 2 * d.C + ch] = sa;
            part[(long)blockIdx.x * 2 * d.C + d.C + ch] = sb;
        } else {
            atomicAdd(d.stat_sum + ch, sa);
            atomicAdd(d.stat_sqsum + ch, sb);
        }
    }
}

// The same 3x3 depthwise conv with the input patch staged in LDS.  The kernel above re-reads every input pixel
// nine times through the vector L1, whose 32 KB per CU hold a fraction of the three image rows a workgroup's
// taps span (128 px x 128 B x (2*dil+1) rows): at 1/8 resolution the re-reads fall through to the L2 and the
// kernel runs at ~1.9 TB/s of algorithmic traffic.  Here a workgroup owns a TH x 32-pixel tile of 32 channels:
// the (TH + 2*HL) x (32 + 2*HL) patch is fetched ONCE (all 16-byte loads of a lane in flight before the first LDS
// write, zero fill outside the image), the nine taps are ds_read_b128 (a wave reads 16 consecutive pixels x 64 B:
// conflict-free), the output is written as 64-byte pixel halves.  Halo re-reads 1.41x (HL = 2, TH = 16).
// HL = the largest dilation among the workgroup's channel groups (SESP's second stage uses dil + 1: 2 on the
// spatial branch, 2..5 on the context branch).  Statistics: one partial row per tile.
// EPI: an output epilogue (folded BatchNorm / activation: the inference form) exists; the training forward writes the raw
// convolution + statistics and does without the 24 coefficient registers (186 -> 162: three workgroups per CU instead of two)
template <int FLIP, int HL, int TH, bool EPI = true>
__global__ void __launch_bounds__(256) dw3x3_tile_kernel(ledn_dw_desc d, const bf16_t* add, float* part) {
    constexpr int TW = 32, PW = TW + 2 * HL, PH = TH + 2 * HL, CW = 32, PXB = CW * 2;
    constexpr int NL = (PH * PW * 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char s_patch[PH * PW * PXB];
    static_assert(PH * PW * PXB >= 2 * 64 * CW * 4, "the statistics exchange reuses the patch");
    float(*s_red)[64 * CW] = reinterpret_cast<float(*)[64 * CW]>(s_patch);     // the patch is dead by then (barrier first)
    const int tid = threadIdx.x;
    const int tx = (d.W + TW - 1) / TW, ty = (d.H + TH - 1) / TH, nch = d.C / CW;
    const unsigned bid = xcd_block(blockIdx.x, gridDim.x);
    const int ch = (int)(bid % (unsigned)nch);
    const unsigned tile = bid / (unsigned)nch;
    const int txi = (int)(tile % (unsigned)tx), tyi = (int)((tile / (unsigned)tx) % (unsigned)ty);
    const int n = (int)(tile / (unsigned)(tx * ty));
    const int y0 = tyi * TH - HL, x0 = txi * TW - HL;
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(d.x) + (long)n * d.H * d.W * d.C + ch * CW;
    {
        uint4 stage[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256, px = e >> 2, q = e & 3;
            const int gy = y0 + px / PW, gx = x0 + px % PW;
            const bool ok = e < PH * PW * 4 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            stage[i] = ok ? *reinterpret_cast<const uint4*>(xin + ((long)gy * d.W + gx) * d.C + q * 8)
                          : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256;
            if (e < PH * PW * 4) *reinterpret_cast<uint4*>(s_patch + (long)e * 16) = stage[i];
        }
    }
    const int cg = tid & 3, pl = tid >> 2;
    const int c = ch * CW + cg * 8;
    const int dl = d.dil[c / d.group_size];
    f32x2_t w[9][4];
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        f32x8_load(d.w + (long)(FLIP ? 8 - t : t) * d.C + c, w[t]);
        toff[t] = (((t / 3 - 1) * dl) * PW + (t % 3 - 1) * dl) * PXB;
    }
    constexpr bool EP = EPI && !FLIP;
    f32x2_t sc[EP ? 4 : 1], sh[EP ? 4 : 1], ng[EP ? 4 : 1];
    if constexpr (EP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[i] = f32x2_t{1.f, 1.f};
            sh[i] = f32x2_t{0.f, 0.f};
            ng[i] = d.act_out == LEDN_ACT_NONE ? f32x2_t{1.f, 1.f} : f32x2_t{0.f, 0.f};
        }
        if (d.out_scale) f32x8_load(d.out_scale + c, sc);
        if (d.out_shift) f32x8_load(d.out_shift + c, sh);
        if (d.act_out == LEDN_ACT_PRELU) f32x8_load(d.slope + c, ng);
    }
    const float hi = (EP && d.act_out == LEDN_ACT_RELU6) ? 6.f : 3.0e38f;
    const bool has_act = EP && d.act_out != LEDN_ACT_NONE;
    f32x2_t st1[4], st2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) st1[i] = st2[i] = f32x2_t{0.f, 0.f};
    __syncthreads();
    bf16_t* y = reinterpret_cast<bf16_t*>(d.y) + (long)n * d.H * d.W * d.C + c;
    const bf16_t* addp = (FLIP && add) ? add + (long)n * d.H * d.W * d.C + c : nullptr;
    const int lx = pl & 31, gx = txi * TW + lx;
#pragma unroll 2
    for (int pass = 0; pass < TH / 2; ++pass) {
        const int ly = pass * 2 + (pl >> 5), gy = tyi * TH + ly;
        if (gy >= d.H || gx >= d.W) continue;
        const unsigned char* ctr = s_patch + ((ly + HL) * PW + lx + HL) * PXB + cg * 16;
        uint4 raw[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) raw[t] = *reinterpret_cast<const uint4*>(ctr + toff[t]);
        f32x2_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x2_t{0.f, 0.f};
        const long o = ((long)gy * d.W + gx) * d.C;
        if (FLIP && addp) bf16x8_unpack(*reinterpret_cast<const uint4*>(addp + o), acc);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f32x2_t xv[4];
            bf16x8_unpack(raw[t], xv);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = pk_fma(xv[i], w[t][i], acc[i]);
        }
        if (!FLIP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (EP) acc[i] = pk_fma(acc[i], sc[i], sh[i]);
                st1[i] += acc[i];
                st2[i] = pk_fma(acc[i], acc[i], st2[i]);
            }
            if constexpr (EP) if (has_act) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i].x = fminf(fmaxf(acc[i].x, 0.f) + ng[i].x * fminf(acc[i].x, 0.f), hi);
                    acc[i].y = fminf(fmaxf(acc[i].y, 0.f) + ng[i].y * fminf(acc[i].y, 0.f), hi);
                }
            }
        }
        *reinterpret_cast<uint4*>(y + o) = bf16x8_pack(acc);
    }
    if (FLIP || !part) return;
    // per-channel sums of the tile: [64 pixel lanes][32 channels] -> one partial row slice per workgroup
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s_red[0][pl * CW + cg * 8 + 2 * i] = st1[i].x;
        s_red[0][pl * CW + cg * 8 + 2 * i + 1] = st1[i].y;
        s_red[1][pl * CW + cg * 8 + 2 * i] = st2[i].x;
        s_red[1][pl * CW + cg * 8 + 2 * i + 1] = st2[i].y;
    }
    __syncthreads();
    if (tid < 2 * CW) {
        const int j = tid / CW, cc = tid % CW;
        float t0 = 0.f, t1 = 0.f;
        for (int r = 0; r < 64; r += 2) {
            t0 += s_red[j][r * CW + cc];
            t1 += s_red[j][(r + 1) * CW + cc];
        }
        part[(long)tile * 2 * d.C + (long)j * d.C + ch * CW + cc] = t0 + t1;
    }
}

// GETB's 8x8 depthwise convolution (UNetFormer_GETB.py:118,201-204: reflect-extended by one row / column, zero
// padding 3) with the input patch in LDS.  The generic kernel walks 8 rows x 8 taps of global loads per output as a
// serial chain (134 us for 16 x 64 x 64 x 128: the exposed tail of the context branch on the auxiliary stream); the
// arithmetic is ~10 us.  Workgroup = 8 x 32 outputs x 32 channels: (8+7) x (32+7) patch, 80-byte pixel rows (two-way
// bank conflicts at most), the 64 x 32 filter values in LDS; lane = (8-channel group, 4 adjacent outputs, row): per
// filter row the 11 pixels its four outputs share are read once and unpacked once (sliding window), the eight taps'
// filter values come as two broadcast 16-byte reads each.  FLIP = 1: the data gradient with respect to the
// reflect-EXTENDED map (flipped taps, padding 4); the fold of row H / column W into H-2 / W-2 is the caller's.
template <int FLIP>
__global__ void __launch_bounds__(256) dw8x8_tile_kernel(ledn_dw_desc d, float* part, int pad_lo) {
    constexpr int TH = 8, TW = 32, K = 8, PH = TH + K - 1, PW = TW + K - 1, CW = 32, PXB = 80;
    constexpr int NL = (PH * PW * 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char s_patch[PH * PW * PXB];
    __shared__ __attribute__((aligned(16))) float s_w[K * K * CW];
    float(*s_red)[64 * CW] = reinterpret_cast<float(*)[64 * CW]>(s_patch);     // statistics exchange: the patch is dead then
    const int tid = threadIdx.x;
    const int tx = (d.Wo + TW - 1) / TW, ty = (d.Ho + TH - 1) / TH, nch = d.C / CW;
    const unsigned bid = xcd_block(blockIdx.x, gridDim.x);
    const int ch = (int)(bid % (unsigned)nch);
    const unsigned tile = bid / (unsigned)nch;
    const int txi = (int)(tile % (unsigned)tx), tyi = (int)((tile / (unsigned)tx) % (unsigned)ty);
    const int n = (int)(tile / (unsigned)(tx * ty));
    const int y0 = tyi * TH - pad_lo, x0 = txi * TW - pad_lo;
    const int Hx = d.H + (d.ext1 ? 1 : 0), Wx = d.W + (d.ext1 ? 1 : 0);
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(d.x) + (long)n * d.H * d.W * d.C + ch * CW;
    {
        uint4 stage[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256, px = e >> 2, q = e & 3;
            int gy = y0 + px / PW, gx = x0 + px % PW;
            const bool ok = e < PH * PW * 4 && gy >= 0 && gy < Hx && gx >= 0 && gx < Wx;
            if (gy == d.H) gy = d.H - 2;                    // the reflected row / column (ext1)
            if (gx == d.W) gx = d.W - 2;
            stage[i] = ok ? *reinterpret_cast<const uint4*>(xin + ((long)gy * d.W + gx) * d.C + q * 8)
                          : uint4{0u, 0u, 0u, 0u};
        }
        for (int i = tid; i < K * K * CW; i += 256) {
            const int t = i / CW, cc = i % CW;
            s_w[i] = d.w[(long)(FLIP ? K * K - 1 - t : t) * d.C + ch * CW + cc];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256;
            if (e < PH * PW * 4) *reinterpret_cast<uint4*>(s_patch + (long)(e >> 2) * PXB + (e & 3) * 16) = stage[i];
        }
    }
    __syncthreads();
    const int cg = tid & 3, xg = (tid >> 2) & 7, r = tid >> 5;
    f32x2_t acc[4][4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[o][i] = f32x2_t{0.f, 0.f};
#pragma unroll 1
    for (int kh = 0; kh < K; ++kh) {
        const unsigned char* row = s_patch + ((r + kh) * PW + 4 * xg) * PXB + cg * 16;
        f32x2_t xr[11][4];
#pragma unroll
        for (int j = 0; j < 11; ++j) bf16x8_unpack(*reinterpret_cast<const uint4*>(row + j * PXB), xr[j]);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            f32x2_t w[4];
            f32x8_load(s_w + (kh * K + kw) * CW + cg * 8, w);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[o][i] = pk_fma(xr[o + kw][i], w[i], acc[o][i]);
        }
    }
    const int c = ch * CW + cg * 8;
    f32x2_t sc[4], sh[4], ng[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sc[i] = f32x2_t{1.f, 1.f};
        sh[i] = f32x2_t{0.f, 0.f};
        ng[i] = d.act_out == LEDN_ACT_NONE ? f32x2_t{1.f, 1.f} : f32x2_t{0.f, 0.f};
    }
    if (!FLIP) {
        if (d.out_scale) f32x8_load(d.out_scale + c, sc);
        if (d.out_shift) f32x8_load(d.out_shift + c, sh);
        if (d.act_out == LEDN_ACT_PRELU) f32x8_load(d.slope + c, ng);
    }
    const float hi = (!FLIP && d.act_out == LEDN_ACT_RELU6) ? 6.f : 3.0e38f;
    const bool has_act = !FLIP && d.act_out != LEDN_ACT_NONE;
    f32x2_t st1[4], st2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) st1[i] = st2[i] = f32x2_t{0.f, 0.f};
    const int gy = tyi * TH + r;
    bf16_t* y = reinterpret_cast<bf16_t*>(d.y) + (long)n * d.Ho * d.Wo * d.C + c;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int gx = txi * TW + 4 * xg + o;
        if (gy >= d.Ho || gx >= d.Wo) continue;
        if (!FLIP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[o][i] = pk_fma(acc[o][i], sc[i], sh[i]);
                st1[i] += acc[o][i];
                st2[i] = pk_fma(acc[o][i], acc[o][i], st2[i]);
            }
            if (has_act) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[o][i].x = fminf(fmaxf(acc[o][i].x, 0.f) + ng[i].x * fminf(acc[o][i].x, 0.f), hi);
                    acc[o][i].y = fminf(fmaxf(acc[o][i].y, 0.f) + ng[i].y * fminf(acc[o][i].y, 0.f), hi);
                }
            }
        }
        *reinterpret_cast<uint4*>(y + ((long)gy * d.Wo + gx) * d.C) = bf16x8_pack(acc[o]);
    }
    if (FLIP || !part) return;
    const int pl = tid >> 2;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s_red[0][pl * CW + cg * 8 + 2 * i] = st1[i].x;
        s_red[0][pl * CW + cg * 8 + 2 * i + 1] = st1[i].y;
        s_red[1][pl * CW + cg * 8 + 2 * i] = st2[i].x;
        s_red[1][pl * CW + cg * 8 + 2 * i + 1] = st2[i].y;
    }
    __syncthreads();
    if (tid < 2 * CW) {
        const int j = tid / CW, cc = tid % CW;
        float t0 = 0.f, t1 = 0.f;
        for (int q = 0; q < 64; q += 2) {
            t0 += s_red[j][q * CW + cc];
            t1 += s_red[j][(q + 1) * CW + cc];
        }
        part[(long)tile * 2 * d.C + (long)j * d.C + ch * CW + cc] = t0 + t1;
    }
}

static bool dw8x8_tile_ok(const ledn_dw_desc& d) {
    if (!(options().stream_fast & 2)) return false;
    if (d.dtype_x != LEDN_BF16 || d.dtype_y != LEDN_BF16 || d.KH != 8 || d.KW != 8 || d.stride != 1 || d.pad != 3) return false;
    if (d.C % 32 || d.Ho != d.H || d.Wo != d.W || !d.ext1 || d.act_out == LEDN_ACT_SIGMOID) return false;
    for (int g = 0; g * d.group_size < d.C; ++g)
        if (d.dil[g] != 1) return false;
    return (long)d.N * d.H * d.W >= 4096 && (long)d.N * d.H * d.W * d.C < (1L << 31);
}

// data gradient of the same convolution: the tiled kernel with flipped taps (padding 4) yields the gradient of the
// reflect-EXTENDED (H+1) x (W+1) map in workspace scratch; this pass folds the reflected row H / column W back
// into row H-2 / column W-2 (the adjoint of F.pad(.., (0,1,0,1), 'reflect')) and adds the fan-in addend
__global__ void __launch_bounds__(256) dw8x8_fold_kernel(const bf16_t* t, const bf16_t* add, bf16_t* dx, int N, int H,
                                                         int W, int C) {
    const int cv = C / 8;
    const long total = (long)N * H * W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, W, H);
    const int c = ix_.cv * 8;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    const int We = W + 1;
    const bf16_t* tn = t + (long)n * (H + 1) * We * C + c;
    float acc[8], v[8];
    ld8(tn + ((long)y * We + x) * C, acc);
    if (y == H - 2) {
        ld8(tn + ((long)H * We + x) * C, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
    if (x == W - 2) {
        ld8(tn + ((long)y * We + W) * C, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i];
        if (y == H - 2) {
            ld8(tn + ((long)H * We + W) * C, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
    }
    if (add) {
        ld8(add + pix * C + c, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
    st8(dx + pix * C + c, acc);
}

int dw8x8_bwd_data_tile(const ledn_dwbwd_desc& b, hipStream_t s) {   // used by backward.hip; -1 = shape not covered
    if (!(options().stream_fast & 2) || b.dtype != LEDN_BF16 || b.KH != 8 || b.KW != 8 || b.stride != 1 || b.pad != 3 ||
        !b.ext1 || b.C % 32 || b.Ho != b.H || b.Wo != b.W || b.H < 2 || b.W < 2)
        return -1;
    for (int g = 0; g * b.group_size < b.C; ++g)
        if (b.dil[g] != 1) return -1;
    const long npix = (long)b.N * b.H * b.W;
    const long ext = (long)b.N * (b.H + 1) * (b.W + 1) * b.C;
    if (npix < 4096 || ext >= (1L << 31)) return -1;
    float* scratch = ws_take(cdiv(ext, 2));
    if (!scratch) return -1;
    ledn_dw_desc d = {};
    d.x = b.dz; d.w = b.w; d.y = scratch;
    d.N = b.N; d.H = b.H; d.W = b.W; d.C = b.C; d.Ho = b.H + 1; d.Wo = b.W + 1;
    d.KH = d.KW = 8; d.stride = 1; d.pad = 3; d.group_size = b.group_size; d.ext1 = 0;
    for (int i = 0; i < 4; ++i) d.dil[i] = 1;
    d.dtype_x = d.dtype_y = LEDN_BF16;
    const long tiles = (long)d.N * cdiv(d.Ho, 8) * cdiv(d.Wo, 32);
    LEDN_LAUNCH((dw8x8_tile_kernel<1>), dim3((unsigned)(tiles * (d.C / 32))), dim3(256), 0, s, d, (float*)nullptr, 4);
    LEDN_LAUNCH(dw8x8_fold_kernel, dim3((unsigned)cdiv(npix * (b.C / 8), 256)), dim3(256), 0, s,
                reinterpret_cast<const bf16_t*>(scratch), reinterpret_cast<const bf16_t*>(b.add),
                reinterpret_cast<bf16_t*>(b.dx), b.N, b.H, b.W, b.C);
    return check_launch();
}

// weight gradient of the same convolution: dw[kh][kw][c] = sum_px xe[px + (kh, kw) - 3][c] * dz[px][c].  Workgroup =
// 8 x 32 outputs x 32 channels (the forward tile): the (8+7) x (32+7) window of the reflect-extended input and the
// dz tile sit in LDS; lane = (output row, 8-channel group, filter row kh): it walks the 32 pixels of its row and keeps
// the 8 x 8 sums of its filter row in registers; the eight output rows are adjacent lanes (three shuffle steps), one
// partial row [64 taps][C] per tile, finish_partials.  (The row-per-workgroup kernel: 132 us at 16 x 64 x 64 x 128.)
__global__ void __launch_bounds__(256) dw8x8_wgrad_tile_kernel(ledn_dwbwd_desc d, float* part) {
    constexpr int TH = 8, TW = 32, K = 8, PH = TH + K - 1, PW = TW + K - 1, CW = 32, PXB = 80;
    constexpr int NL = (PH * PW * 4 + 255) / 256, NZ = (TH * TW * 4) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char s_patch[PH * PW * PXB];
    __shared__ __attribute__((aligned(16))) unsigned char s_dz[TH * TW * 64];
    const int tid = threadIdx.x;
    const int tx = (d.Wo + TW - 1) / TW, ty = (d.Ho + TH - 1) / TH, nch = d.C / CW;
    const unsigned bid = xcd_block(blockIdx.x, gridDim.x);
    const int ch = (int)(bid % (unsigned)nch);
    const unsigned tile = bid / (unsigned)nch;
    const int txi = (int)(tile % (unsigned)tx), tyi = (int)((tile / (unsigned)tx) % (unsigned)ty);
    const int n = (int)(tile / (unsigned)(tx * ty));
    const int y0 = tyi * TH - 3, x0 = txi * TW - 3;
    const int Hx = d.H + 1, Wx = d.W + 1;
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(d.x) + (long)n * d.H * d.W * d.C + ch * CW;
    const bf16_t* zin = reinterpret_cast<const bf16_t*>(d.dz) + (long)n * d.Ho * d.Wo * d.C + ch * CW;
    {
        uint4 stage[NL], zs[NZ];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256, px = e >> 2, q = e & 3;
            int gy = y0 + px / PW, gx = x0 + px % PW;
            const bool ok = e < PH * PW * 4 && gy >= 0 && gy < Hx && gx >= 0 && gx < Wx;
            if (gy == d.H) gy = d.H - 2;
            if (gx == d.W) gx = d.W - 2;
            stage[i] = ok ? *reinterpret_cast<const uint4*>(xin + ((long)gy * d.W + gx) * d.C + q * 8)
                          : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int e = tid + i * 256, px = e >> 2, q = e & 3;
            const int gy = tyi * TH + px / TW, gx = txi * TW + px % TW;
            const bool ok = gy < d.Ho && gx < d.Wo;
            zs[i] = ok ? *reinterpret_cast<const uint4*>(zin + ((long)gy * d.Wo + gx) * d.C + q * 8) : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256;
            if (e < PH * PW * 4) *reinterpret_cast<uint4*>(s_patch + (long)(e >> 2) * PXB + (e & 3) * 16) = stage[i];
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) *reinterpret_cast<uint4*>(s_dz + (long)(tid + i * 256) * 16) = zs[i];
    }
    __syncthreads();
    const int r = tid & 7, cg = (tid >> 3) & 3, kh = tid >> 5;
    f32x2_t acc[K][4];
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[kw][i] = f32x2_t{0.f, 0.f};
    const unsigned char* xrow = s_patch + ((r + kh) * PW) * PXB + cg * 16;
    const unsigned char* zrow = s_dz + (r * TW) * 64 + cg * 16;
#pragma unroll 2
    for (int x = 0; x < TW; ++x) {
        f32x2_t g[4];
        bf16x8_unpack(*reinterpret_cast<const uint4*>(zrow + x * 64), g);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            f32x2_t xv[4];
            bf16x8_unpack(*reinterpret_cast<const uint4*>(xrow + (x + kw) * PXB), xv);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[kw][i] = pk_fma(xv[i], g[i], acc[kw][i]);
        }
    }
    // the eight output rows are lanes l, l^1, l^2, l^4
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = acc[kw][i].x, b = acc[kw][i].y;
            a += __shfl_xor(a, 1); b += __shfl_xor(b, 1);
            a += __shfl_xor(a, 2); b += __shfl_xor(b, 2);
            a += __shfl_xor(a, 4); b += __shfl_xor(b, 4);
            acc[kw][i].x = a; acc[kw][i].y = b;
        }
    if (r == 0) {
        float* row = part + (long)tile * (K * K) * d.C + ch * CW + cg * 8;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            float* dst = row + (long)(kh * K + kw) * d.C;
            *reinterpret_cast<float4*>(dst) = make_float4(acc[kw][0].x, acc[kw][0].y, acc[kw][1].x, acc[kw][1].y);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[kw][2].x, acc[kw][2].y, acc[kw][3].x, acc[kw][3].y);
        }
    }
}

int dw8x8_bwd_weight_tile(const ledn_dwbwd_desc& b, hipStream_t s) {   // used by backward.hip; -1 = shape not covered
    if (!(options().stream_fast & 2) || b.dtype != LEDN_BF16 || b.KH != 8 || b.KW != 8 || b.stride != 1 || b.pad != 3 ||
        !b.ext1 || b.C % 32 || b.Ho != b.H || b.Wo != b.W || b.H < 2 || b.W < 2)
        return -1;
    for (int g = 0; g * b.group_size < b.C; ++g)
        if (b.dil[g] != 1) return -1;
    const long npix = (long)b.N * b.H * b.W;
    if (npix < 4096 || npix * b.C >= (1L << 31)) return -1;
    const long tiles = (long)b.N * cdiv(b.Ho, 8) * cdiv(b.Wo, 32);
    float* part = ws_take(tiles * 64 * b.C);
    if (!part) return -1;
    LEDN_LAUNCH(dw8x8_wgrad_tile_kernel, dim3((unsigned)(tiles * (b.C / 32))), dim3(256), 0, s, b, part);
    return finish_partials(part, (int)tiles, 64 * b.C, 1, b.dw, nullptr, nullptr, s);
}

// launches the tiled kernel when it applies (returns the number of partial rows through *rows), else -1
static int dw3x3_tile_launch(const ledn_dw_desc& d, bool flip, const bf16_t* add, bool want_stats, float** part_out,
                             long* rows_out, hipStream_t s) {
    if (d.C % 32 || d.group_size % 8 || d.W < 32 || (long)d.N * d.H * d.W < 16384) return -1;
    int hl = 0;
    for (int g = 0; g * d.group_size < d.C; ++g) hl = d.dil[g] > hl ? d.dil[g] : hl;
    if (hl > 5) return -1;
    static const int th2 = (int)exp_knob("LEDN_DW_TH", 16);      // (A/B knob: 8 = more, shorter workgroups for dilation <= 2)
    const int TH = (hl <= 2 && th2 != 8) ? 16 : 8;
    const long tiles = (long)d.N * cdiv(d.H, TH) * cdiv(d.W, 32);
    const long nb = tiles * (d.C / 32);
    if (nb > (1L << 30) || tiles > 16384) return -1;
    float* part = nullptr;
    if (want_stats) {
        part = ws_take(tiles * 2 * d.C);
        if (!part) return -1;
    }
    const bool plain = !flip && !d.out_scale && !d.out_shift && d.act_out == LEDN_ACT_NONE;     // the training forward
    if (hl <= 2 && TH == 16) {
        if (flip) LEDN_LAUNCH((dw3x3_tile_kernel<1, 2, 16>), dim3((unsigned)nb), dim3(256), 0, s, d, add, part);
        else if (plain) LEDN_LAUNCH((dw3x3_tile_kernel<0, 2, 16, false>), dim3((unsigned)nb), dim3(256), 0, s, d, add, part);
        else LEDN_LAUNCH((dw3x3_tile_kernel<0, 2, 16>), dim3((unsigned)nb), dim3(256), 0, s, d, add, part);
    } else if (hl <= 2) {
        if (flip) LEDN_LAUNCH((dw3x3_tile_kernel<1, 2, 8>), dim3((unsigned)nb), dim3(256), 0, s, d, add, part);
        else LEDN_LAUNCH((dw3x3_tile_kernel<0, 2, 8>), dim3((unsigned)nb), dim3(256), 0, s, d, add, part);
    } else {
        if (flip) LEDN_LAUNCH((dw3x3_tile_kernel<1, 5, 8>), dim3((unsigned)nb), dim3(256), 0, s, d, add, part);
        else if (plain) LEDN_LAUNCH((dw3x3_tile_kernel<0, 5, 8, false>), dim3((unsigned)nb), dim3(256), 0, s, d, add, part);
        else LEDN_LAUNCH((dw3x3_tile_kernel<0, 5, 8>), dim3((unsigned)nb), dim3(256), 0, s, d, add, part);
    }
    *part_out = part;
    *rows_out = tiles;
    return 0;
}

// shape gate of the vectorised 3x3 kernels (forward and data gradient)
static bool dw3x3_bf16_ok(int C, int group_size, int KH, int KW, int stride, int pad, int ext1, const int* dil,
                          long numel) {
    if (KH != 3 || KW != 3 || stride != 1 || ext1 || C % 8 || group_size % 8 || 256 % (C / 8) || C / 8 > 256) return false;
    if (numel >= (1L << 31)) return false;
    for (int g = 0; g * group_size < C; ++g)
        if (pad >= 0 && pad != dil[g]) return false;
    return true;
}

int dw3x3_bwd_data_bf16(const ledn_dwbwd_desc& b, hipStream_t s) {   // used by backward.hip
    if (b.dtype != LEDN_BF16 || !dw3x3_bf16_ok(b.C, b.group_size, b.KH, b.KW, b.stride, b.pad, b.ext1, b.dil,
                                               (long)b.N * b.H * b.W * b.C))
        return -1;
    ledn_dw_desc d = {};
    d.x = b.dz; d.w = b.w; d.y = b.dx;
    d.N = b.N; d.H = b.H; d.W = b.W; d.C = b.C; d.Ho = b.H; d.Wo = b.W;
    d.KH = d.KW = 3; d.stride = 1; d.pad = b.pad; d.group_size = b.group_size;
    for (int i = 0; i < 4; ++i) d.dil[i] = b.dil[i];
    if (options().stream_fast & 2) {
        float* part = nullptr;
        long rows = 0;
        if (dw3x3_tile_launch(d, true, (const bf16_t*)b.add, false, &part, &rows, s) == 0) return check_launch();
    }
    const int rows = 256 / (d.C / 8);
    long nb = cdiv((long)d.N * d.H * d.W, rows * 4);
    if (nb > 2048) nb = 2048;
    LEDN_LAUNCH((dw3x3_bf16_kernel<1>), dim3((unsigned)nb), dim3(256), 0, s, d, (const bf16_t*)b.add, (float*)nullptr);
    return check_launch();
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
    if (dw8x8_tile_ok(d)) {
        const long tiles = (long)d.N * cdiv(d.Ho, 8) * cdiv(d.Wo, 32);
        float* part = nullptr;
        if (d.stat_sum) part = ws_take(tiles * 2 * d.C);
        if (!d.stat_sum || part) {
            LEDN_LAUNCH((dw8x8_tile_kernel<0>), dim3((unsigned)(tiles * (d.C / 32))), dim3(256), 0, s, d, part, 3);
            if (part) return finish_partials(part, (int)tiles, d.C, 2, d.stat_sum, d.stat_sqsum, nullptr, s);
            return check_launch();
        }
    }
    if (d.dtype_x == LEDN_BF16 && d.dtype_y == LEDN_BF16 && d.act_out != LEDN_ACT_SIGMOID && d.Ho == d.H &&
        d.Wo == d.W &&
        dw3x3_bf16_ok(d.C, d.group_size, d.KH, d.KW, d.stride, d.pad, d.ext1, d.dil, (long)d.N * d.H * d.W * d.C)) {
        if (options().stream_fast & 2) {
            float* partt = nullptr;
            long rowst = 0;
            if (dw3x3_tile_launch(d, false, nullptr, d.stat_sum != nullptr, &partt, &rowst, s) == 0) {
                if (partt) return finish_partials(partt, (int)rowst, d.C, 2, d.stat_sum, d.stat_sqsum, nullptr, s);
                return check_launch();
            }
        }
        const int rows8 = 256 / (d.C / 8);
        long nb8 = cdiv((long)d.N * d.H * d.W, rows8 * 4);
        if (nb8 > 2048) nb8 = 2048;
        float* part8 = nullptr;
        if (d.stat_sum) {
            if (nb8 > 64 || det()) part8 = ws_take(nb8 * 2 * d.C);
            if (!part8 && nb8 > 256) nb8 = 256;
        }
        LEDN_LAUNCH((dw3x3_bf16_kernel<0>), dim3((unsigned)nb8), dim3(256), 0, s, d, (const bf16_t*)nullptr, part8);
        if (part8) return finish_partials(part8, (int)nb8, d.C, 2, d.stat_sum, d.stat_sqsum, nullptr, s);
        return check_launch();
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
        if (nb > 64 || det()) part = ws_take(nb * 2 * d.C);
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
    const long idx = (long)xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, d.Wo, d.Ho);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int wo = ix_.x, ho = ix_.y, n = ix_.n;
    const TX* x = reinterpret_cast<const TX*>(d.x);
    TY* y = reinterpret_cast<TY*>(d.y) + pix * (4L * d.n) + c;
    float run[V];
#pragma unroll
    for (int v = 0; v < V; ++v) run[v] = 0.f;
    for (int b = 0; b < 4; ++b) {
        const int dl = d.dil[b];
        float xv[9][V], wv[9][V];
#pragma unroll
        for (int t = 0; t < 9; ++t) {   // nine unconditional tap loads (+ their weights) in flight
            const int hi = ho * d.stride + (t / 3 - 1) * dl, wi = wo * d.stride + (t % 3 - 1) * dl;
            const bool valid = hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
            ldv_if<V>(x, (((long)n * d.H + hi) * d.W + wi) * d.n + c, valid, xv[t]);
            ldv<V>(d.w + (long)(b * 9 + t) * d.n + c, wv[t]);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int v = 0; v < V; ++v) run[v] = fmaf(xv[t][v], wv[t][v], run[v]);
        stv<V>(y + (long)b * d.n, run);
    }
}

int pyr_fwd_bf16(const ledn_pyr_desc& d, hipStream_t s);   // stencil_bf16.hip; -1 = shape not covered

int sesp_pyramid_impl(const ledn_pyr_desc& d, hipStream_t s) {
    LEDN_REQUIRE(d.x && d.w && d.y);
    LEDN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.n > 0 && (d.stride == 1 || d.stride == 2));
    LEDN_REQUIRE(d.Ho == (d.H - 1) / d.stride + 1 && d.Wo == (d.W - 1) / d.stride + 1);
    for (int b = 0; b < 4; ++b) LEDN_REQUIRE(d.dil[b] > 0);
    {
        const int rc = pyr_fwd_bf16(d, s);
        if (rc >= 0) return rc;
    }
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

// ---------------------------------------------------------------------------
// depthwise filter layout bridge (see include/ledn.h: ledn_dwpack_desc)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dw_repack_kernel(ledn_dwpack_desc d, float* packed, const float* dpacked,
                                                        int ctot) {
    const int k = blockIdx.y;
    const int n = d.n[k];
    int c0 = 0;
    if (!d.stacked)
        for (int j = 0; j < k; ++j) c0 += d.n[j];
    const int total = n * d.taps;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / d.taps, t = i % d.taps;      // i = PyTorch's linear index [c][0][tap]
        const long pi = d.stacked ? ((long)k * d.taps + t) * n + c : (long)t * ctot + c0 + c;
        if (dpacked) d.dw[k][i] += dpacked[pi];
        else packed[pi] = d.w[k][i];
    }
}

static int dw_repack(const ledn_dwpack_desc& d, float* packed, const float* dpacked, hipStream_t s) {
    LEDN_REQUIRE(d.nsrc > 0 && d.nsrc <= 8 && d.taps > 0 && (packed != nullptr) != (dpacked != nullptr));
    int ctot = 0, nmax = 0;
    for (int k = 0; k < d.nsrc; ++k) {
        LEDN_REQUIRE(d.n[k] > 0 && (dpacked ? d.dw[k] != nullptr : d.w[k] != nullptr));
        LEDN_REQUIRE(!d.stacked || d.n[k] == d.n[0]);
        ctot += d.n[k];
        if (d.n[k] > nmax) nmax = d.n[k];
    }
    long bx = cdiv((long)nmax * d.taps, 256);
    if (bx > 64) bx = 64;
    LEDN_LAUNCH(dw_repack_kernel, dim3((unsigned)bx, (unsigned)d.nsrc), dim3(256), 0, s, d, packed, dpacked, ctot);
    return check_launch();
}

// table-driven variant: EVERY depthwise filter bank of the model in one launch (grid.z = bank).
//   dir 0: packed <- filters (start of a training step)
//   dir 1: filter gradients += packed gradient, and the packed gradient buffer is re-zeroed (end of the backward)
__global__ void __launch_bounds__(256) dw_repack_multi_kernel(const ledn_dwpack_entry* table, int dir) {
    // the entry goes through LDS (one dword per lane): indexing its arrays with the runtime source index from a per-lane
    // register copy put the whole entry into scratch (200 B per lane, 37 us per launch for a few KB of filters)
    __shared__ ledn_dwpack_entry s_e;
    static_assert(sizeof(ledn_dwpack_entry) % 4 == 0 && sizeof(ledn_dwpack_entry) <= 4 * 256, "entry staged by one dword per lane");
    if (threadIdx.x < sizeof(ledn_dwpack_entry) / 4)
        ((unsigned*)&s_e)[threadIdx.x] = ((const unsigned*)(table + blockIdx.z))[threadIdx.x];
    __syncthreads();
    const int k = blockIdx.y;
    const int nsrc = s_e.d.nsrc;
    if (k >= nsrc) return;
    const int n = s_e.d.n[k], taps = s_e.d.taps;
    int c0 = 0, ctot = 0;
    for (int j = 0; j < nsrc; ++j) {
        const int nj = s_e.d.n[j];
        if (j < k) c0 += nj;
        ctot += nj;
    }
    const int total = n * taps;
    const bool stacked = s_e.d.stacked;
    const float* w = s_e.d.w[k];
    float* dw = s_e.d.dw[k];
    float* packed = s_e.packed;
    float* dpacked = s_e.dpacked;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / taps, t = i % taps;
        const long pi = stacked ? ((long)k * taps + t) * n + c : (long)t * ctot + c0 + c;
        if (dir == 0) packed[pi] = w[i];
        else {
            dw[i] += dpacked[pi];
            dpacked[pi] = 0.f;
        }
    }
}

int dw_repack_multi_impl(const ledn_dwpack_entry* table_dev, int n, int max_elems, int dir, hipStream_t s) {
    LEDN_REQUIRE(table_dev && n > 0 && max_elems > 0 && (dir == 0 || dir == 1));
    long bx = cdiv((long)max_elems, 256);
    if (bx > 16) bx = 16;
    LEDN_LAUNCH(dw_repack_multi_kernel, dim3((unsigned)bx, 8u, (unsigned)n), dim3(256), 0, s, table_dev, dir);
    return check_launch();
}

int dw_pack_impl(const ledn_dwpack_desc& d, float* packed, hipStream_t s) { return dw_repack(d, packed, nullptr, s); }
int dw_unpack_grad_impl(const ledn_dwpack_desc& d, const float* dpacked, hipStream_t s) {
    return dw_repack(d, nullptr, dpacked, s);
}

}  // namespace ledn
