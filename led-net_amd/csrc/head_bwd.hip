// head_bwd.hip -- BatchNorm / activation backward of LEDHead's two-class heads (led_head.py:44-51: norm -> act -> 3x3 conv
// 32 -> num_classes on the 1/2- and 1/4-resolution stem maps) without ever writing the gradient of the activation:
//
//   forward   t = act(x * scale + shift)   (training-mode BatchNorm folded into scale / shift),   z = conv3x3(t, w) + b
//   backward  dy[p][c] = sum_{k,o} dz[p - off(k)][o] * w[o][c][k]                 (transposed 3x3, pad 1; off(k) = (kh-1, kw-1))
//             g = dy * act'(.),  sum_g, sum_g*xhat (, dslope)  ->  dx = scale * (g - mean_g - xhat * mean_gx) (+ addend)
//
// The layer-by-layer form writes dy (268 MB at 16 x 512 x 512 x 32 bf16) in a transposed-convolution launch and reads it
// twice (BatchNorm-backward reduce and apply): 88 + 117 + 130 us.  dy is a function of 18 numbers per pixel (the 3 x 3 x 2
// patch of dz), i.e. a [32 channels x 18] x [18 x pixels] product -- one and a bit v_mfma_f32_32x32x16_bf16 per 32 pixels:
//   head_bwd_reduce_kernel: x + the dz patch -> per-workgroup rows [sum_g | sum_gx | dslope]       (285 MB instead of 821)
//   head_bwd_apply_kernel:  x + the dz patch (+ fan-in addend) -> dx                               (no dy read)
// A wave owns tiles of 32 consecutive pixels.  B operand = the patch: lane (pixel n, half h) loads the taps 4h .. 4h+3 of its
// pixel as four dwords (bf16 pairs, both classes) + tap 8; A operand = the filter, [channel slot][tap, class], constant per
// lane.  The channel slots are permuted so that the accumulator of lane (n, h) holds channels 16h .. 16h+15 of pixel n:
// x, the addend and dx are then two 16-byte accesses per lane.  Next tile's loads are in flight under this tile's
// arithmetic (7 vector operations per element; a first form on the vector ALU alone -- 18 FMAs per element for dy and 18
// more for the head's weight gradient -- measured 375 us at 1/2 resolution against the 334 us of the layer-wise kernels).
// The head's weight gradient rides along in the reduce pass: dw[kidx][c] = sum_px P[px][kidx] * t[px][c], kidx = (tap, class)
// -- the SAME patch values the dy product takes as its B operand, and t = act(BatchNorm(x)), both through a wave-private LDS
// tile and the transposing read: two more matrix instructions and eight transposing reads per 32 pixels instead of
// conv3x3_wgrad_narrow_kernel's 18 + 36 in a launch of its own that reads x once more (148 us at 1/2 resolution).
// C = 32, num_classes = 2, bf16 only.
#include "stencil.h"

namespace ledn {

namespace {

constexpr int HB_C = 32, HB_CO = 2;

// channel of accumulator register `reg` in lane half h = 16 h + reg; MFMA row of that register = (reg & 3) + 8 (reg >> 2) + 4 h
// -> row m carries channel 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3)
__device__ __forceinline__ int hb_row_channel(int m) { return 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3); }

// A fragments: lane (m = lane & 31, hk = lane >> 5) holds A[m][16 s + 8 hk + j] = w[o][channel(m)][k9], 2 k9 + o = 16 s + 8 hk + j
__device__ __forceinline__ void hb_filter_frags(const float* w, bf16x8_t (&af)[2]) {
    const int lane = threadIdx.x & 63, ch = hb_row_channel(lane & 31), hk = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * s + 8 * hk + j, k9 = kk >> 1, o = kk & 1;
            af[s][j] = (short)(k9 < 9 ? f32_to_bf16(w[(o * HB_C + ch) * 9 + k9]) : (unsigned short)0);
        }
}

struct HbTaps {
    int dyo[5], dxo[5];     // position of the lane's taps relative to its pixel: taps 4h .. 4h+3, then tap 8
    int lin[5];             // the same as a pixel offset
};
__device__ __forceinline__ void hb_taps(int h, int W, HbTaps& t) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int k9 = j < 4 ? 4 * h + j : 8;
        t.dyo[j] = 1 - k9 / 3;
        t.dxo[j] = 1 - k9 % 3;
        t.lin[j] = t.dyo[j] * W + t.dxo[j];
    }
}

struct HbRaw {
    uint4 x0, x1;           // the lane's 16 channels of its pixel
    unsigned dz[5];
};

__device__ __forceinline__ void hb_fetch(const bf16_t* x, const unsigned* dz, const HbTaps& tp, long p, long pend, int y, int xx,
                                         int h, int H, int W, HbRaw& o) {
    const bool in = p < pend;
    const long q = in ? p : 0;
    const uint4* xp = reinterpret_cast<const uint4*>(x + q * HB_C + 16 * h);
    o.x0 = xp[0];
    o.x1 = xp[1];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int ty = y + tp.dyo[j], tx = xx + tp.dxo[j];
        const bool ok = in && ty >= 0 && ty < H && tx >= 0 && tx < W;
        const unsigned v = dz[ok ? q + tp.lin[j] : q];
        o.dz[j] = ok ? v : 0u;
    }
}

// dy of the lane's pixel, channels 16 h .. 16 h + 15
__device__ __forceinline__ f32x16_t hb_dy(const HbRaw& r, const bf16x8_t (&af)[2], int h) {
    const uint4 b0 = make_uint4(r.dz[0], r.dz[1], r.dz[2], r.dz[3]);
    const uint4 b1 = make_uint4(h == 0 ? r.dz[4] : 0u, 0u, 0u, 0u);
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = mfma_32x32x16_bf16(af[0], __builtin_bit_cast(bf16x8_t, b0), acc);
    acc = mfma_32x32x16_bf16(af[1], __builtin_bit_cast(bf16x8_t, b1), acc);
    return acc;
}

__device__ __forceinline__ void hb_unpack16(const uint4& x0, const uint4& x1, float* v) {
    const unsigned u[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[2 * i] = __uint_as_float(u[i] << 16);
        v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
}

// the wave's tile schedule: workgroup b owns tiles [t0, t1) of 32 pixels; wave w takes NT adjacent tiles per trip:
// t0 + NT w .. + NT - 1, then 4 NT further on (two tiles' loads in flight per wave: at two waves per SIMD one tile's 84 bytes
// per lane are ~11 MB on the whole chip, short of what 5 TB/s x the memory latency asks for)
constexpr int HB_NT = 2;
struct HbSched {
    long tile, t1;
    __device__ __forceinline__ void init(long ntiles) {
        const long per = cdiv(cdiv(ntiles, (long)gridDim.x), (long)(4 * HB_NT)) * (4 * HB_NT);
        const long t0 = (long)xcd_block(blockIdx.x, gridDim.x) * per;
        t1 = min(ntiles, t0 + per);
        tile = t0 + HB_NT * (threadIdx.x >> 6);
    }
};


// ---- weight gradient: dw[kidx][c] += P^T[kidx][px] t[px][c] per 32-pixel tile.  Both operands want 8 consecutive PIXELS per
// lane while the wave holds them pixel-per-lane (P = the two B operands of the dy product, t = act(BatchNorm(x))): each goes
// through a wave-private [32 px][32 columns] bf16 tile in LDS and comes back through the transposing read.
// fragment M[px = 16 s + 8 hk + j][col = lane & 31], j < 8, of a [32 px][32 col] bf16 tile: per 16-lane group (columns
// 16 (g & 1) ..), lane 4 q + p supplies the address of pixel row q, columns 4 p .. 4 p + 3
__device__ __forceinline__ bf16x8_t hb_tr_frag(const unsigned char* tile, int s) {
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, hk = lane >> 5;
    const unsigned char* ap = tile + (16 * s + 8 * hk + (i >> 2)) * 64 + (16 * (g & 1) + 4 * (i & 3)) * 2;
    const bf16x4_t lo = lds_read_tr16(ap), hi = lds_read_tr16(ap + 4 * 64);
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// row of partial sums per workgroup: [sum_g | sum_gx | dslope (3 x 32) | dw 2 x 32 x 9 | db 2]
constexpr int HB_ROW = 3 * HB_C, HB_ROW_W = 3 * HB_C + HB_CO * HB_C * 9 + HB_CO;

}  // namespace

template <int ACT, bool WG>
__global__ void __launch_bounds__(256) head_bwd_reduce_kernel(ledn_headbwd_desc d, float* part) {
    __shared__ float s_red[4][3][HB_C];
    __shared__ __attribute__((aligned(16))) unsigned char s_t[WG ? 4 : 1][WG ? 2 * 32 * 64 : 16];   // per wave: t tile [32 px][32 ch], P tile [32 px][32 (tap, class)], bf16
    __shared__ float s_w[WG ? 4 : 1][WG ? 18 * HB_C + 2 : 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    const int H = d.H, W = d.W;
    bf16x8_t af[2];
    hb_filter_frags(d.w, af);
    HbTaps tp;
    hb_taps(h, W, tp);
    f32x16_t dwacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) dwacc[i] = 0.f;
    float db0 = 0.f, db1 = 0.f;
    float sc[16], sh[16], mean[16], sl[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = 16 * h + i;
        sc[i] = d.bn.scale ? d.bn.scale[c] : 1.f;
        sh[i] = d.bn.shift ? d.bn.shift[c] : 0.f;
        mean[i] = d.bn.bn_mode ? d.bn.mean[c] : 0.f;
        sl[i] = ACT == LEDN_ACT_PRELU ? d.bn.slope[c] : 0.f;
    }
    float sg[16], sgx[16], sds[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sg[i] = sgx[i] = sds[i] = 0.f;
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.bn.z);
    const unsigned* dz = reinterpret_cast<const unsigned*>(d.head_dz);
    const long npix = (long)d.N * H * W;
    HbSched sch;
    sch.init(cdiv(npix, 32L));
    PixCursor cur[HB_NT];
    HbRaw raw[HB_NT];
#pragma unroll
    for (int u = 0; u < HB_NT; ++u) {
        cur[u].init(min((sch.tile + u) * 32 + n, npix - 1), H, W);
        hb_fetch(x, dz, tp, sch.tile < sch.t1 ? (sch.tile + u) * 32 + n : npix, npix, cur[u].y, cur[u].x, h, H, W, raw[u]);
    }
    for (; sch.tile < sch.t1; sch.tile += 4 * HB_NT) {
        HbRaw nxt[HB_NT];
        const bool more = sch.tile + 4 * HB_NT < sch.t1;
#pragma unroll
        for (int u = 0; u < HB_NT; ++u) {
            cur[u].advance(128 * HB_NT, H, W);
            hb_fetch(x, dz, tp, more ? (sch.tile + 4 * HB_NT + u) * 32 + n : npix, npix, cur[u].y, cur[u].x, h, H, W, nxt[u]);
        }
#pragma unroll
        for (int u = 0; u < HB_NT; ++u) {
            const long p = (sch.tile + u) * 32 + n;
            const f32x16_t dy = hb_dy(raw[u], af, h);
            float tv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) tv[i] = 0.f;
            if (p < npix) {
                float xv[16];
                hb_unpack16(raw[u].x0, raw[u].x1, xv);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = fmaf(xv[i], sc[i], sh[i]);
                    float g = dy[i];
                    tv[i] = v;
                    if (ACT == LEDN_ACT_RELU) {
                        g = v > 0.f ? g : 0.f;
                        tv[i] = fmaxf(v, 0.f);
                    } else if (ACT == LEDN_ACT_PRELU) {
                        sds[i] += v > 0.f ? 0.f : dy[i] * v;
                        g = v > 0.f ? g : g * sl[i];
                        tv[i] = v > 0.f ? v : v * sl[i];
                    }
                    sg[i] += g;
                    sgx[i] = fmaf(g, xv[i] - mean[i], sgx[i]);
                }
                if (WG && h == 1) {              // tap 4 = the pixel itself: the bias gradient
                    db0 += __uint_as_float(raw[u].dz[0] << 16);
                    db1 += __uint_as_float(raw[u].dz[0] & 0xffff0000u);
                }
            }
            if (WG) {
                uint4 t0, t1;
                t0.x = (unsigned)f32_to_bf16(tv[0]) | ((unsigned)f32_to_bf16(tv[1]) << 16);
                t0.y = (unsigned)f32_to_bf16(tv[2]) | ((unsigned)f32_to_bf16(tv[3]) << 16);
                t0.z = (unsigned)f32_to_bf16(tv[4]) | ((unsigned)f32_to_bf16(tv[5]) << 16);
                t0.w = (unsigned)f32_to_bf16(tv[6]) | ((unsigned)f32_to_bf16(tv[7]) << 16);
                t1.x = (unsigned)f32_to_bf16(tv[8]) | ((unsigned)f32_to_bf16(tv[9]) << 16);
                t1.y = (unsigned)f32_to_bf16(tv[10]) | ((unsigned)f32_to_bf16(tv[11]) << 16);
                t1.z = (unsigned)f32_to_bf16(tv[12]) | ((unsigned)f32_to_bf16(tv[13]) << 16);
                t1.w = (unsigned)f32_to_bf16(tv[14]) | ((unsigned)f32_to_bf16(tv[15]) << 16);
                wave_sync();                     // the previous tile's transposing reads are done
                uint4* tw = reinterpret_cast<uint4*>(s_t[wid] + n * 64 + h * 32);
                tw[0] = t0;
                tw[1] = t1;
                // P[px = n][kidx]: this lane's operands of the dy product ARE columns 8 h .. 8 h + 7 and 16 + 8 h .. of its pixel
                unsigned char* pw = s_t[wid] + 32 * 64 + n * 64 + h * 16;
                *reinterpret_cast<uint4*>(pw) = make_uint4(raw[u].dz[0], raw[u].dz[1], raw[u].dz[2], raw[u].dz[3]);
                *reinterpret_cast<uint4*>(pw + 32) = make_uint4(h == 0 ? raw[u].dz[4] : 0u, 0u, 0u, 0u);
                wave_sync();
                dwacc = mfma_32x32x16_bf16(hb_tr_frag(s_t[wid] + 32 * 64, 0), hb_tr_frag(s_t[wid], 0), dwacc);
                dwacc = mfma_32x32x16_bf16(hb_tr_frag(s_t[wid] + 32 * 64, 1), hb_tr_frag(s_t[wid], 1), dwacc);
            }
            raw[u] = nxt[u];
        }
    }
    // the 32 pixel lanes of each half: the two rows of a half exchange (sum g | sum g x) by a reduce-scatter, then a butterfly
    // inside the 16-lane row
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        unsigned a = __float_as_uint(sg[i]), b = __float_as_uint(sgx[i]);
        permlane16_swap(a, b);                  // even rows: both rows' sg[i]; odd rows: both rows' sgx[i]
        float v = __uint_as_float(a) + __uint_as_float(b);
        v = lane_step_sum(v, 8);
        v = lane_step_sum(v, 4);
        v = lane_step_sum(v, 2);
        v = lane_step_sum(v, 1);
        sg[i] = v;
        if (ACT == LEDN_ACT_PRELU) {
            float t = sds[i];
            t = lane_step_sum(t, 16);
            t = lane_step_sum(t, 8);
            t = lane_step_sum(t, 4);
            t = lane_step_sum(t, 2);
            t = lane_step_sum(t, 1);
            sds[i] = t;
        }
    }
    if ((lane & 15) == 0) {
        const int row = lane >> 4;               // rows 0, 1: half 0 (sum g, sum g x); rows 2, 3: half 1
#pragma unroll
        for (int i = 0; i < 16; ++i) s_red[wid][row & 1][16 * (row >> 1) + i] = sg[i];
        if (ACT == LEDN_ACT_PRELU && (row & 1) == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s_red[wid][2][16 * (row >> 1) + i] = sds[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * HB_C) {
        const int kind = threadIdx.x / HB_C, c = threadIdx.x % HB_C;
        float t = 0.f;
        if (kind < 2 || ACT == LEDN_ACT_PRELU)
            t = (s_red[0][kind][c] + s_red[1][kind][c]) + (s_red[2][kind][c] + s_red[3][kind][c]);
        if (kind == 1) t *= d.bn.bn_mode ? d.bn.invstd[c] : 0.f;
        part[(long)blockIdx.x * (WG ? HB_ROW_W : HB_ROW) + threadIdx.x] = t;
    }
    if (WG) {
        // accumulator register reg of lane (c = n, h) = row kidx = (reg & 3) + 8 (reg >> 2) + 4 h of dw[kidx][c], kidx = 2 k9 + o
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int kidx = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (kidx < 18) s_w[wid][kidx * HB_C + n] = dwacc[reg];
        }
        float b0 = db0, b1 = db1;               // (lanes of half 1 hold the pixels' sums, half 0 zeros)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            b0 = lane_step_sum(b0, m);
            b1 = lane_step_sum(b1, m);
        }
        if (lane == 0) {
            s_w[wid][18 * HB_C] = b0;
            s_w[wid][18 * HB_C + 1] = b1;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 18 * HB_C + 2; e += 256) {
            const float t = (s_w[0][e] + s_w[1][e]) + (s_w[2][e] + s_w[3][e]);
            int dst = HB_ROW + HB_CO * HB_C * 9 + (e - 18 * HB_C);          // db
            if (e < 18 * HB_C) {
                const int kidx = e / HB_C, c = e % HB_C;
                dst = HB_ROW + ((kidx & 1) * HB_C + c) * 9 + (kidx >> 1);    // dw[o][c][k9]
            }
            part[(long)blockIdx.x * HB_ROW_W + dst] = t;
        }
    }
}

// adds the workgroups' rows up and ACCUMULATES into the sinks: workgroup = 16 outputs x 16 row slots (slot q sums rows q,
// q + 16, ... with four loads in flight, the slots are then added in slot order: one fixed order in every mode).  (One lane
// per output walking all 512 rows was a chain of 128 dependent L2 round trips: ~60 us.)
__global__ void __launch_bounds__(256) head_bwd_finish_kernel(const float* part, int nrows, float* sum_g, float* sum_gx,
                                                              float* dslope, float* dw, float* db) {
    __shared__ float s_red[256];
    const int j = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + j;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < HB_ROW_W) {
        int b = slot;
        for (; b + 48 < nrows; b += 64) {
            a0 += part[(long)b * HB_ROW_W + e];
            a1 += part[(long)(b + 16) * HB_ROW_W + e];
            a2 += part[(long)(b + 32) * HB_ROW_W + e];
            a3 += part[(long)(b + 48) * HB_ROW_W + e];
        }
        for (; b < nrows; b += 16) a0 += part[(long)b * HB_ROW_W + e];
    }
    s_red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x >= 16 || e >= HB_ROW_W) return;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += s_red[q * 16 + threadIdx.x];
    float* dst;
    if (e < HB_C) dst = sum_g ? sum_g + e : nullptr;
    else if (e < 2 * HB_C) dst = sum_gx ? sum_gx + (e - HB_C) : nullptr;
    else if (e < 3 * HB_C) dst = dslope ? dslope + (e - 2 * HB_C) : nullptr;
    else if (e < HB_ROW + HB_CO * HB_C * 9) dst = dw + (e - HB_ROW);
    else dst = db ? db + (e - HB_ROW - HB_CO * HB_C * 9) : nullptr;
    if (dst) *dst += t;
}

template <int ACT>
__global__ void __launch_bounds__(256) head_bwd_apply_kernel(ledn_headbwd_desc d) {
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int H = d.H, W = d.W;
    bf16x8_t af[2];
    hb_filter_frags(d.w, af);
    HbTaps tp;
    hb_taps(h, W, tp);
    float sc[16], sh[16], ca[16], cb[16], sl[16];
    const float inv_count = (float)(1.0 / d.bn.count);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = 16 * h + i;
        const float s = d.bn.scale ? d.bn.scale[c] : 1.f;
        float a = 0.f, b = 0.f;
        if (d.bn.bn_mode) {           // dz = s * g + A * x + B,  A = -s * mean_gx * invstd,  B = -s * mean_g - A * mean
            a = -s * (d.bn.sum_gx[c] * inv_count) * d.bn.invstd[c];
            b = -s * (d.bn.sum_g[c] * inv_count) - a * d.bn.mean[c];
        }
        sc[i] = s;
        sh[i] = d.bn.shift ? d.bn.shift[c] : 0.f;
        sl[i] = ACT == LEDN_ACT_PRELU ? d.bn.slope[c] : 0.f;
        ca[i] = a;
        cb[i] = b;
    }
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.bn.z);
    const bf16_t* add = reinterpret_cast<const bf16_t*>(d.bn.dz_add);
    bf16_t* dx = reinterpret_cast<bf16_t*>(d.bn.dz);
    const unsigned* dz = reinterpret_cast<const unsigned*>(d.head_dz);
    const long npix = (long)d.N * H * W;
    HbSched sch;
    sch.init(cdiv(npix, 32L));
    PixCursor cur[HB_NT];
    HbRaw raw[HB_NT];
    uint4 a0[HB_NT], a1[HB_NT];
    auto fetch_add = [&](long p, uint4& o0, uint4& o1) {
        o0 = o1 = make_uint4(0u, 0u, 0u, 0u);
        if (!add) return;
        const uint4* ap = reinterpret_cast<const uint4*>(add + (p < npix ? p : 0) * HB_C + 16 * h);
        o0 = ap[0];
        o1 = ap[1];
    };
#pragma unroll
    for (int u = 0; u < HB_NT; ++u) {
        cur[u].init(min((sch.tile + u) * 32 + n, npix - 1), H, W);
        const long p = sch.tile < sch.t1 ? (sch.tile + u) * 32 + n : npix;
        hb_fetch(x, dz, tp, p, npix, cur[u].y, cur[u].x, h, H, W, raw[u]);
        fetch_add(p, a0[u], a1[u]);
    }
    for (; sch.tile < sch.t1; sch.tile += 4 * HB_NT) {
        HbRaw nxt[HB_NT];
        uint4 n0[HB_NT], n1[HB_NT];
        const bool more = sch.tile + 4 * HB_NT < sch.t1;
#pragma unroll
        for (int u = 0; u < HB_NT; ++u) {
            cur[u].advance(128 * HB_NT, H, W);
            const long pn = more ? (sch.tile + 4 * HB_NT + u) * 32 + n : npix;
            hb_fetch(x, dz, tp, pn, npix, cur[u].y, cur[u].x, h, H, W, nxt[u]);
            fetch_add(pn, n0[u], n1[u]);
        }
#pragma unroll
        for (int u = 0; u < HB_NT; ++u) {
            const long p = (sch.tile + u) * 32 + n;
            const f32x16_t dy = hb_dy(raw[u], af, h);
            if (p < npix) {
                float xv[16], av[16], out[16];
                hb_unpack16(raw[u].x0, raw[u].x1, xv);
                hb_unpack16(a0[u], a1[u], av);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = fmaf(xv[i], sc[i], sh[i]);
                    float g = dy[i];
                    if (ACT == LEDN_ACT_RELU) g = v > 0.f ? g : 0.f;
                    else if (ACT == LEDN_ACT_PRELU) g = v > 0.f ? g : g * sl[i];
                    out[i] = fmaf(g, sc[i], fmaf(xv[i], ca[i], cb[i]));
                    if (add) out[i] += av[i];
                }
                uint4 o0, o1;
                o0.x = (unsigned)f32_to_bf16(out[0]) | ((unsigned)f32_to_bf16(out[1]) << 16);
                o0.y = (unsigned)f32_to_bf16(out[2]) | ((unsigned)f32_to_bf16(out[3]) << 16);
                o0.z = (unsigned)f32_to_bf16(out[4]) | ((unsigned)f32_to_bf16(out[5]) << 16);
                o0.w = (unsigned)f32_to_bf16(out[6]) | ((unsigned)f32_to_bf16(out[7]) << 16);
                o1.x = (unsigned)f32_to_bf16(out[8]) | ((unsigned)f32_to_bf16(out[9]) << 16);
                o1.y = (unsigned)f32_to_bf16(out[10]) | ((unsigned)f32_to_bf16(out[11]) << 16);
                o1.z = (unsigned)f32_to_bf16(out[12]) | ((unsigned)f32_to_bf16(out[13]) << 16);
                o1.w = (unsigned)f32_to_bf16(out[14]) | ((unsigned)f32_to_bf16(out[15]) << 16);
                uint4* op = reinterpret_cast<uint4*>(dx + p * HB_C + 16 * h);
                op[0] = o0;
                op[1] = o1;
            }
            raw[u] = nxt[u];
            a0[u] = n0[u];
            a1[u] = n1[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward of the same heads: z[p][o] = b[o] + sum_{tap, c} t[p + off(tap)][c] w[o][c][tap], t = act(x * scale + shift).
// Instead of 9 taps x 2 K-steps of matrix instructions whose N dimension holds 2 useful columns of 32 (conv_mfma_kernel's
// narrow epilogue: 118 us at 16 x 512 x 512), the channel contraction runs ONCE per pixel for all 18 (tap, class)
// combinations -- Y[(tap, class)][px] = Wm[18 x 32] t[32 x px], two v_mfma_f32_32x32x16_bf16 per 32 pixels with t as the B
// operand in its natural layout (lane = pixel, 8 consecutive channels: the two 16-byte loads of the lane) -- and the nine
// taps become a shifted sum of 9 of those partial products per output: through a wave-private LDS row [18][34] for the
// column shifts, through a three-row register ring for the row shifts.  A wave owns a strip of 32 pixel columns (30 valid
// outputs: columns 30 j - 1 .. 30 j + 30) and walks down a chunk of rows; lane (n, h) emits class h of column n.
template <int ACT, bool OUT16>
__global__ void __launch_bounds__(256) head_fwd_kernel(ledn_conv_desc d, int strips, int chunks, int rows_per_chunk) {
    __shared__ float s_y[4][18 * 34];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    const int H = d.H, W = d.W;
    bf16x8_t af[2];                                            // A[m = 2 tap + class][c = 16 s + 8 hk + j]
    {
        const int m = lane & 31, hk = lane >> 5, k9 = m >> 1, o = m & 1;
        const float* w = reinterpret_cast<const float*>(d.w);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * s + 8 * hk + j;
                af[s][j] = (short)(m < 18 ? f32_to_bf16(w[(o * HB_C + c) * 9 + k9]) : (unsigned short)0);
            }
    }
    float sc[16], sh[16], sl[16];                              // channels 8 h + j (j < 8) and 16 + 8 h + j
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = (i < 8 ? 0 : 16) + 8 * h + (i & 7);
        sc[i] = d.in_scale ? d.in_scale[c] : 1.f;
        sh[i] = d.in_shift ? d.in_shift[c] : 0.f;
        sl[i] = ACT == LEDN_ACT_PRELU ? d.in_slope[c] : 0.f;
    }
    const float bias = d.out_shift ? d.out_shift[h] : 0.f;
    float* ys = s_y[wid];
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
    const long ntask = (long)d.N * strips * chunks;
    for (long task = (long)blockIdx.x * 4 + wid; task < ntask; task += (long)gridDim.x * 4) {
        const int ck = (int)(task % chunks), j = (int)((task / chunks) % strips), img = (int)(task / ((long)chunks * strips));
        const int r0 = ck * rows_per_chunk, r1 = min(H, r0 + rows_per_chunk);
        const int px = 30 * j - 1 + n;
        const bool col_ok = px >= 0 && px < W;
        const bf16_t* xi = x + (long)img * H * W * HB_C;
        auto fetch = [&](int r, uint4& v0, uint4& v1) {
            const bool ok = col_ok && r >= 0 && r < H;
            const uint4* p = reinterpret_cast<const uint4*>(xi + ((long)(ok ? r : 0) * W + (ok ? px : 0)) * HB_C + 8 * h);
            v0 = p[0];
            v1 = p[2];                                         // channels 16 + 8 h ..
            if (!ok) v0 = v1 = make_uint4(0u, 0u, 0u, 0u);
        };
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;                    // partial outputs of rows r - 1, r, r + 1
        auto process = [&](int r, const uint4& c0, const uint4& c1) {
            const bool row_ok = col_ok && r >= 0 && r < H;
            float xv[16];
            hb_unpack16(c0, c1, xv);
            unsigned tw[8];
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float a = fmaf(xv[i], sc[i], sh[i]), b = fmaf(xv[i + 1], sc[i + 1], sh[i + 1]);
                if (ACT == LEDN_ACT_RELU) {
                    a = fmaxf(a, 0.f);
                    b = fmaxf(b, 0.f);
                } else if (ACT == LEDN_ACT_PRELU) {
                    a = a > 0.f ? a : a * sl[i];
                    b = b > 0.f ? b : b * sl[i + 1];
                }
                tw[i >> 1] = row_ok ? ((unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16)) : 0u;   // zero padding of t
            }
            f32x16_t acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = mfma_32x32x16_bf16(af[0], __builtin_bit_cast(bf16x8_t, make_uint4(tw[0], tw[1], tw[2], tw[3])), acc);
            acc = mfma_32x32x16_bf16(af[1], __builtin_bit_cast(bf16x8_t, make_uint4(tw[4], tw[5], tw[6], tw[7])), acc);
            wave_sync();                                       // the previous row's reads of the LDS row are done
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int kidx = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (kidx < 18) ys[kidx * 34 + n + 1] = acc[reg];
            }
            wave_sync();
            // class h of column n: P[kh] = sum_kw Y[(3 kh + kw, h)][n + kw - 1]
            float P[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* row = ys + ((3 * kh) * 2 + h) * 34 + n;
                P[kh] = (row[0] + row[2 * 34 + 1]) + row[4 * 34 + 2];
            }
            // input row r feeds output rows r + 1 (kh = 0), r (kh = 1), r - 1 (kh = 2)
            const float done = t0 + P[2];
            t0 = t1 + P[1];
            t1 = t2 + P[0];
            t2 = 0.f;
            const int ro = r - 1;
            if (ro >= r0 && ro < r1 && n >= 1 && n <= 30 && px < W) {
                const long o = (((long)img * H + ro) * W + px) * HB_CO + h;
                if (OUT16) reinterpret_cast<bf16_t*>(d.y)[o].v = f32_to_bf16(done + bias);
                else reinterpret_cast<float*>(d.y)[o] = done + bias;
            }
        };
        // rows r0 - 1 .. r1 in groups of HF_D, the next group's loads in flight under this group's arithmetic (one row ahead
        // left every row waiting for its own load: 129 us at 16 x 512 x 512)
        constexpr int HF_D = 4;
        uint4 c0[HF_D], c1[HF_D];
#pragma unroll
        for (int u = 0; u < HF_D; ++u) fetch(min(r0 - 1 + u, r1), c0[u], c1[u]);
        for (int r = r0 - 1; r <= r1; r += HF_D) {
            uint4 n0[HF_D], n1[HF_D];
#pragma unroll
            for (int u = 0; u < HF_D; ++u) fetch(min(r + HF_D + u, r1), n0[u], n1[u]);
#pragma unroll
            for (int u = 0; u < HF_D; ++u) {
                if (r + u <= r1) process(r + u, c0[u], c1[u]);
                c0[u] = n0[u];
                c1[u] = n1[u];
            }
        }
    }
}

bool head_fwd_supported(const ledn_conv_desc& d) {
    static const bool on = exp_knob("LEDN_HEAD_FWD", 1) != 0;     // (A/B knob)
    if (!on || !(options().stream_fast & 1)) return false;
    if (d.dtype_x != LEDN_BF16 || (d.dtype_y != LEDN_BF16 && d.dtype_y != LEDN_F32) || !d.w) return false;
    if (d.Cin != HB_C || d.Cout != HB_CO || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1 || d.groups != 1)
        return false;
    if (d.transposed || d.xadd || d.res || d.res_mode != LEDN_RES_NONE || d.out_scale || d.stat_sum || d.act_out != LEDN_ACT_NONE)
        return false;
    if (d.Ho != d.H || d.Wo != d.W || d.W < 32 || (long)d.N * d.H * d.W < 16384 || (long)d.N * d.H * d.W * HB_C >= (1L << 31))
        return false;
    if ((d.in_scale == nullptr) != (d.in_shift == nullptr)) return false;
    if (d.in_act != LEDN_ACT_NONE && d.in_act != LEDN_ACT_RELU && !(d.in_act == LEDN_ACT_PRELU && d.in_slope)) return false;
    // the filter in its natural OIHW strides (ledn_conv_desc.ws_*)
    return d.ws_co == (long long)HB_C * 9 && d.ws_ci == 9 && d.ws_tap == 1;
}

int head_fwd(const ledn_conv_desc& d, hipStream_t s) {
    const int strips = (int)cdiv(d.W, 30);
    // rows per wave task: enough tasks to fill the chip twice over (a task is one serial chain of rows; 2 halo rows per chunk):
    // 16 x 512 x 512 -> 16 rows (101 us; 32 rows 110, 64 rows 126), 16 x 256 x 256 -> 8
    static const int rpc = (int)exp_knob("LEDN_HEAD_FWD_ROWS", 0);
    int rows_per_chunk = rpc;
    if (rows_per_chunk <= 0) {
        const long want = ((long)d.N * strips * d.H) / 8192;
        rows_per_chunk = want >= 32 ? 32 : (want >= 16 ? 16 : 8);
    }
    const int chunks = (int)cdiv(d.H, rows_per_chunk);
    const long ntask = (long)d.N * strips * chunks;
    long nb = cdiv(ntask, 4L);
    if (nb > 2048) nb = 2048;
#define LEDN_HF_GO(A_)                                                                                                       \
    do {                                                                                                                     \
        if (d.dtype_y == LEDN_BF16)                                                                                          \
            LEDN_LAUNCH((head_fwd_kernel<A_, true>), dim3((unsigned)nb), dim3(256), 0, s, d, strips, chunks, rows_per_chunk);  \
        else                                                                                                                 \
            LEDN_LAUNCH((head_fwd_kernel<A_, false>), dim3((unsigned)nb), dim3(256), 0, s, d, strips, chunks, rows_per_chunk); \
    } while (0)
    switch (d.in_act) {
        case LEDN_ACT_NONE: LEDN_HF_GO(LEDN_ACT_NONE); break;
        case LEDN_ACT_RELU: LEDN_HF_GO(LEDN_ACT_RELU); break;
        default: LEDN_HF_GO(LEDN_ACT_PRELU); break;
    }
#undef LEDN_HF_GO
    return check_launch();
}

int head_bwd_supported(const ledn_headbwd_desc& d) {
    if (!d.bn.z || !d.head_dz || !d.w || d.Co != HB_CO || d.bn.C != HB_C || d.dtype_dz != LEDN_BF16) return 0;
    if (d.bn.dtype_z != LEDN_BF16 || d.bn.res || d.bn.res_mode != LEDN_RES_NONE || d.bn.dres || d.bn.rows) return 0;
    if (d.bn.act != LEDN_ACT_NONE && d.bn.act != LEDN_ACT_RELU && d.bn.act != LEDN_ACT_PRELU) return 0;
    if (d.bn.act == LEDN_ACT_PRELU && !d.bn.slope) return 0;
    const long npix = (long)d.N * d.H * d.W;
    if (d.N < 1 || d.H < 2 || d.W < 2 || npix != d.bn.P || npix < 16384 || npix * HB_C >= (1L << 31)) return 0;
    if (d.bn.bn_mode && !(d.bn.mean && d.bn.invstd && d.bn.sum_g && d.bn.sum_gx)) return 0;
    return 1;
}

int head_bwd_reduce(const ledn_headbwd_desc& d, hipStream_t s) {
    LEDN_REQUIRE(head_bwd_supported(d));
    const bool wg = d.dw != nullptr;              // the head's weight (and bias) gradient in the same pass
    LEDN_REQUIRE(wg || !d.db);
    const long ntiles = cdiv((long)d.N * d.H * d.W, 32L);
    static const long cap = exp_knob("LEDN_HB_CAP", 512);
    long nb = cdiv(ntiles, 32L);                 // >= 4 trips of two tiles per wave
    if (nb > cap) nb = cap;
    float* part = ws_take(nb * (wg ? HB_ROW_W : HB_ROW));
    LEDN_REQUIRE(part);
#define LEDN_HB_GO(A_)                                                                                           \
    do {                                                                                                         \
        if (wg) LEDN_LAUNCH((head_bwd_reduce_kernel<A_, true>), dim3((unsigned)nb), dim3(256), 0, s, d, part);   \
        else LEDN_LAUNCH((head_bwd_reduce_kernel<A_, false>), dim3((unsigned)nb), dim3(256), 0, s, d, part);     \
    } while (0)
    switch (d.bn.act) {
        case LEDN_ACT_NONE: LEDN_HB_GO(LEDN_ACT_NONE); break;
        case LEDN_ACT_RELU: LEDN_HB_GO(LEDN_ACT_RELU); break;
        default: LEDN_HB_GO(LEDN_ACT_PRELU); break;
    }
#undef LEDN_HB_GO
    float* sum_g = d.bn.bn_mode ? d.bn.sum_g : nullptr;
    float* sum_gx = d.bn.bn_mode ? d.bn.sum_gx : nullptr;
    float* dslope = d.bn.act == LEDN_ACT_PRELU ? d.bn.dslope : nullptr;
    if (!wg) return finish_partials(part, (int)nb, HB_C, 3, sum_g, sum_gx, dslope, s);
    LEDN_LAUNCH(head_bwd_finish_kernel, dim3((unsigned)cdiv(HB_ROW_W, 16)), dim3(256), 0, s, part, (int)nb, sum_g, sum_gx, dslope,
                d.dw, d.db);
    return check_launch();
}

int head_bwd_apply(const ledn_headbwd_desc& d, hipStream_t s) {
    LEDN_REQUIRE(head_bwd_supported(d) && d.bn.dz);
    const long ntiles = cdiv((long)d.N * d.H * d.W, 32L);
    static const long cap = exp_knob("LEDN_HB_CAP_APPLY", 4096);
    static const long tpw = exp_knob("LEDN_HB_TRIPS_APPLY", 16);      // trips of two tiles per wave: 1 238 us, 2 138, 8 125, 16 128, 32 116 (1/2 resolution; a
                                                                      // workgroup starts with ~100 parameter loads per lane); step 12.46 / 12.43 / 12.41 / 12.44 ms
    long nb = cdiv(ntiles, 8L * (tpw > 0 ? tpw : 2));
    if (nb > cap) nb = cap;
    switch (d.bn.act) {
        case LEDN_ACT_NONE: LEDN_LAUNCH(head_bwd_apply_kernel<LEDN_ACT_NONE>, dim3((unsigned)nb), dim3(256), 0, s, d); break;
        case LEDN_ACT_RELU: LEDN_LAUNCH(head_bwd_apply_kernel<LEDN_ACT_RELU>, dim3((unsigned)nb), dim3(256), 0, s, d); break;
        default: LEDN_LAUNCH(head_bwd_apply_kernel<LEDN_ACT_PRELU>, dim3((unsigned)nb), dim3(256), 0, s, d); break;
    }
    return check_launch();
}

}  // namespace ledn
