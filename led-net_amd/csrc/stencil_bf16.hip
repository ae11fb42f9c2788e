// stencil_bf16.hip -- vectorised bf16 kernels (see stencil.h) for the depthwise weight gradient
// and the SESP pyramid forward / data gradient / weight gradient.  Each returns -1 when the shape
// is outside its gate so that the caller (dwconv.hip / backward.hip) falls back to the generic kernel.
#include <cstdlib>
#include "stencil.h"

namespace ledn {

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// sums `acc` (9 taps x 8 channels) over the pixel rows of one wave (lanes r*cvn + cv, cvn a power
// of two), then over the 4 waves through LDS; thread e < nine*C then owns element e of [9][C]
__device__ __forceinline__ void reduce_taps(f32x2_t (&acc)[9][4], int cvn, float* s_red /* [4][9*C] */, int C,
                                            int c) {
#define LEDN_RT_LEVEL(M)                                                        \
    if (cvn <= (M)) {                                                           \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                     \
                acc[t][i].x = lane_step_sum(acc[t][i].x, (M));                  \
                acc[t][i].y = lane_step_sum(acc[t][i].y, (M));                  \
            }                                                                   \
    }
    LEDN_RT_LEVEL(32)
    LEDN_RT_LEVEL(16)
    LEDN_RT_LEVEL(8)
    LEDN_RT_LEVEL(4)
    LEDN_RT_LEVEL(2)
    LEDN_RT_LEVEL(1)
#undef LEDN_RT_LEVEL
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane < cvn) {   // cvn <= 64: the first cvn lanes of the wave hold its sums (cvn == 64: one row per wave)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s_red[(wid * 9 + t) * C + c + 2 * i] = acc[t][i].x;
                s_red[(wid * 9 + t) * C + c + 2 * i + 1] = acc[t][i].y;
            }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// depthwise 3x3 weight gradient: dw[t][c] = sum_pix x[pix @ tap t] * dz[pix]   (stride 1, pad = dil)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dw3x3_bwd_weight_bf16_kernel(ledn_dwbwd_desc d, float* part) {
    constexpr int V = 8;
    LEDN_DYN_SHARED(float, s_red);   // [4][9*C]
    const int cvn = d.C / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    f32x2_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x2_t{0.f, 0.f};
    {
        const int dl = d.dil[c / d.group_size];
        const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
        const bf16_t* dz = reinterpret_cast<const bf16_t*>(d.dz);
        int toff[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * dl * d.W + (t % 3 - 1) * dl) * d.C;
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
            f32x2_t g[4];
            bf16x8_unpack(*reinterpret_cast<const uint4*>(dz + base), g);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x2_t xv[4];
                bf16x8_unpack(raw[t], xv);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = pk_fma(xv[i], g[i], acc[t][i]);
            }
        }
    }
    reduce_taps(acc, cvn, s_red, d.C, c);
    for (int e = threadIdx.x; e < 9 * d.C; e += 256) {
        const float sum = (s_red[e] + s_red[9 * d.C + e]) + (s_red[18 * d.C + e] + s_red[27 * d.C + e]);
        if (part) part[(long)blockIdx.x * 9 * d.C + e] = sum;
        else atomicAdd(d.dw + e, sum);
    }
}

int dw3x3_bwd_weight_bf16(const ledn_dwbwd_desc& d, hipStream_t s) {
    const int cvn = d.C / 8;
    if (d.dtype != LEDN_BF16 || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.ext1 || d.C % 8 || d.group_size % 8 ||
        !pow2(cvn) || cvn > 64 || d.Ho != d.H || d.Wo != d.W || (long)d.N * d.H * d.W * d.C >= (1L << 31))
        return -1;
    for (int g = 0; g * d.group_size < d.C; ++g)
        if (d.pad >= 0 && d.pad != d.dil[g]) return -1;
    const int rows = 256 / cvn;
    // 16 pixels per thread (512 workgroups at 16 x 128 x 128 x 64): 28.8 / 36.6 us against 34.0 / 39.7 with 8 (graph replay,
    // r03); 32 and 64 are slower again.  Cost experiments of the same visit: every tap reading the centre pixel, only one tap
    // accumulated, next pixel's loads requested ahead -- none moved the time
    static const int ppt = (int)exp_knob("LEDN_DW_PPT", 16);     // (A/B knob)
    long nb = cdiv((long)d.N * d.H * d.W, rows * (ppt > 0 ? ppt : 16));
    if (nb > 1024) nb = 1024;
    float* part = (nb > 32 || det()) ? ws_take(nb * 9 * d.C) : nullptr;
    if (!part && nb > 128) nb = 128;
    LEDN_LAUNCH(dw3x3_bwd_weight_bf16_kernel, dim3((unsigned)nb), dim3(256), (size_t)(36 * d.C) * sizeof(float), s, d,
                part);
    if (part) return finish_partials(part, (int)nb, 9 * d.C, 1, d.dw, nullptr, nullptr, s);
    return check_launch();
}

// ---------------------------------------------------------------------------
// SESP pyramid.  Weights [4][9][n] f32 sit in LDS (one copy per workgroup).
//   forward      : y[po][b*n + c] = sum_{b' <= b} dw3x3_{dil[b'], stride}(x)[po][c]
//   data gradient: dx[p][c] = sum_b sum_t g_b[p - tap_b(t)][c] * w[b][t][c]        (stride 1)
//   weight grad. : dw[b][t][c] = sum_po x[po*stride @ tap_b(t)][c] * g_b[po][c]
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pyr_stage_weights(const float* w, float* s_w, int n) {
    for (int i = threadIdx.x; i < 36 * n; i += 256) s_w[i] = w[i];
    __syncthreads();
}

// SAME = all four dilations equal (the spatial branch: [1,1,1,1]): the four branches read the same nine taps, which
// are then loaded once instead of four times (the re-reads miss the vector L1 at 1/8 resolution and queue on the L2).
template <bool SAME>
__global__ void __launch_bounds__(256) pyr_fwd_bf16_kernel(ledn_pyr_desc d) {
    constexpr int V = 8;
    LEDN_DYN_SHARED(float, s_w);   // [36][n]
    pyr_stage_weights(d.w, s_w, d.n);
    const int cvn = d.n / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
    bf16_t* y = reinterpret_cast<bf16_t*>(d.y);
    const long npo = (long)d.N * d.Ho * d.Wo;
    const long ppb = cdiv(cdiv(npo, (long)gridDim.x), (long)rows) * rows;
    const long p0 = (long)xcd_block(blockIdx.x, gridDim.x) * ppb, p1 = min(npo, p0 + ppb);
    PixCursor cur;
    cur.init(p0 + r, d.Ho, d.Wo);
    for (long p = p0 + r; p < p1; p += rows, cur.advance(rows, d.Ho, d.Wo)) {
        const int yi = cur.y * d.stride, xi = cur.x * d.stride;
        const unsigned base = (unsigned)((((long)cur.n * d.H + yi) * d.W + xi) * d.n + c);
        f32x2_t run[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) run[i] = f32x2_t{0.f, 0.f};
        uint4 raw[9];
        if (SAME) {
            const int dl = d.dil[0];
            const unsigned mask = tap_mask(yi, xi, dl, d.H, d.W);
            const int co = opaque(dl * d.n), ro = co * d.W;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                raw[t] = ld_tap(x, base + (unsigned)((t / 3 - 1) * ro + (t % 3 - 1) * co), base, (mask >> t) & 1u);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (!SAME) {
                const int dl = d.dil[b];
                const unsigned mask = tap_mask(yi, xi, dl, d.H, d.W);
                const int co = opaque(dl * d.n), ro = co * d.W;   // recomputed per pixel: 36 hoisted offsets cost a wave/SIMD
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    raw[t] = ld_tap(x, base + (unsigned)((t / 3 - 1) * ro + (t % 3 - 1) * co), base, (mask >> t) & 1u);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x2_t xv[4], wv[4];
                bf16x8_unpack(raw[t], xv);
                f32x8_load(s_w + (b * 9 + t) * d.n + c, wv);
#pragma unroll
                for (int i = 0; i < 4; ++i) run[i] = pk_fma(xv[i], wv[i], run[i]);
            }
            *reinterpret_cast<uint4*>(y + p * (4L * d.n) + (long)b * d.n + c) = bf16x8_pack(run);
            if (!SAME) sched_fence();   // one branch's nine taps in flight at a time (36 would cost the occupancy)
        }
    }
}

int pyr_fwd_bf16(const ledn_pyr_desc& d, hipStream_t s) {
    const int cvn = d.n / 8;
    if (d.dtype_x != LEDN_BF16 || d.dtype_y != LEDN_BF16 || d.n % 8 || 256 % cvn || cvn > 256 ||
        (long)d.N * d.H * d.W * d.n * 4 >= (1L << 31))
        return -1;
    const int rows = 256 / cvn;
    long nb = cdiv((long)d.N * d.Ho * d.Wo, rows * 2);
    if (nb > 2048) nb = 2048;
    if (d.dil[0] == d.dil[1] && d.dil[1] == d.dil[2] && d.dil[2] == d.dil[3])
        LEDN_LAUNCH(pyr_fwd_bf16_kernel<true>, dim3((unsigned)nb), dim3(256), (size_t)(36 * d.n) * sizeof(float), s, d);
    else
        LEDN_LAUNCH(pyr_fwd_bf16_kernel<false>, dim3((unsigned)nb), dim3(256), (size_t)(36 * d.n) * sizeof(float), s, d);
    return check_launch();
}

// Equal dilations of 1 (the spatial branch), stride 1, large maps: the data gradient straight from dy with the
// PREFIX-summed filters  dx = sum_b corr(dy_b, flip(W_b)),  W_b = sum_{b' <= b} w_b'  -- no suffix-sum pass over dy
// (a launch that read and wrote all 4n channels) -- and with the dy patch staged ONCE in LDS: the gather kernel below
// issues 36 global 16-byte loads per output that miss the 32 KB vector L1 at 1/8 resolution and queue on the L2
// (85 us for 16 x 128 x 128, n = 32).  Workgroup = 128 lanes = a 8 x 32-pixel tile x 16 channels: patch
// [10 rows][4 branches][34 px][16 ch] (43.5 KB, three workgroups per CU), lane = (column, 8-channel group, 4 output
// rows): per branch the 9 x 8 filter values sit in registers and each of the 6 x 3 staged input pieces feeds up to
// three output rows (18 instead of 36 LDS reads per 4 outputs).
__global__ void __launch_bounds__(128) pyr_bwd_data_tile_kernel(ledn_pyrbwd_desc d) {
    constexpr int TW = 32, TH = 8, PW = TW + 2, PH = TH + 2;
    constexpr int NPIECE = PH * 4 * PW * 2;                 // 16-byte pieces: [py][b][px][half]
    constexpr int NL = (NPIECE + 127) / 128;
    __shared__ __attribute__((aligned(16))) unsigned char s_g[NPIECE * 16];
    __shared__ __attribute__((aligned(16))) float s_w[4 * 9 * 16];
    const int tid = threadIdx.x;
    const int tx = (d.W + TW - 1) / TW, ty = (d.H + TH - 1) / TH, nc = d.n / 16;
    const unsigned bid = xcd_block(blockIdx.x, gridDim.x);
    const int cgp = (int)(bid % (unsigned)nc);
    const unsigned tile = bid / (unsigned)nc;
    const int txi = (int)(tile % (unsigned)tx), tyi = (int)((tile / (unsigned)tx) % (unsigned)ty);
    const int img = (int)(tile / (unsigned)(tx * ty));
    const int c0 = cgp * 16, n4 = 4 * d.n;
    const bf16_t* dy = reinterpret_cast<const bf16_t*>(d.dy) + (long)img * d.H * d.W * n4 + c0;
    const int y0 = tyi * TH - 1, x0 = txi * TW - 1;
    {
        uint4 stage[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 128, half = e & 1, q = e >> 1;
            const int px = q % PW, rb = q / PW, b = rb & 3, py = rb >> 2;
            const int gy = y0 + py, gx = x0 + px;
            const bool ok = e < NPIECE && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            stage[i] = ok ? *reinterpret_cast<const uint4*>(dy + ((long)gy * d.W + gx) * n4 + b * d.n + half * 8)
                          : uint4{0u, 0u, 0u, 0u};
        }
        for (int i = tid; i < 9 * 16; i += 128) {           // prefix sums of the 16 channels' filters
            const int t = i >> 4, cc = i & 15;
            float run = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                run += d.w[(long)(b * 9 + t) * d.n + c0 + cc];
                s_w[(b * 9 + t) * 16 + cc] = run;
            }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 128;
            if (e < NPIECE) *reinterpret_cast<uint4*>(s_g + (long)e * 16) = stage[i];
        }
    }
    __syncthreads();
    const int cg = tid & 1, lx = (tid >> 1) & 31, yg = tid >> 6;
    f32x2_t acc[4][4];
#pragma unroll
    for (int yo = 0; yo < 4; ++yo)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[yo][i] = f32x2_t{0.f, 0.f};
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
        f32x2_t w[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t) f32x8_load(s_w + (b * 9 + t) * 16 + cg * 8, w[t]);
#pragma unroll
        for (int r = 0; r < 6; ++r) {                        // staged row yg*4 + r = image row (first output row) - 1 + r
            const unsigned char* row = s_g + ((long)((yg * 4 + r) * 4 + b) * PW + lx) * 32 + cg * 16;
            f32x2_t in[3][4];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) bf16x8_unpack(*reinterpret_cast<const uint4*>(row + kx * 32), in[kx]);
#pragma unroll
            for (int yo = 0; yo < 4; ++yo) {
                const int ky = r - yo;                       // dx[y] += dy[y + ky - 1][x + kx - 1] * W[(2-ky)*3 + (2-kx)]
                if (ky < 0 || ky > 2) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[yo][i] = pk_fma(in[kx][i], w[8 - (ky * 3 + kx)][i], acc[yo][i]);
            }
        }
    }
    const int gx = txi * TW + lx;
    bf16_t* dx = reinterpret_cast<bf16_t*>(d.dx) + (long)img * d.H * d.W * d.n + c0 + cg * 8;
#pragma unroll
    for (int yo = 0; yo < 4; ++yo) {
        const int gy = tyi * TH + yg * 4 + yo;
        if (gy < d.H && gx < d.W) *reinterpret_cast<uint4*>(dx + ((long)gy * d.W + gx) * d.n) = bf16x8_pack(acc[yo]);
    }
}

// the tiled data gradient applies (and then nothing reads or writes gsum: the weight gradient forms its suffix sums
// from dy on the fly -- both entry points decide with this predicate)
bool pyr_tile_applies(const ledn_pyrbwd_desc& d) {
    return (options().stream_fast & 2) && d.dtype == LEDN_BF16 && d.stride == 1 && d.n % 16 == 0 && d.n <= 512 &&
           ((d.n / 8) & (d.n / 8 - 1)) == 0 && d.W >= 32 &&   // (n / 8 a power of two: the bf16 weight-gradient kernel's gate)
           d.dil[0] == 1 && d.dil[1] == 1 && d.dil[2] == 1 && d.dil[3] == 1 && (long)d.N * d.H * d.W >= 16384 &&
           (long)d.N * d.H * d.W * d.n * 4 < (1L << 31);
}

int pyr_bwd_data_tile(const ledn_pyrbwd_desc& d, hipStream_t s) {
    const long nb = (long)d.N * cdiv(d.H, 8) * cdiv(d.W, 32) * (d.n / 16);
    LEDN_LAUNCH(pyr_bwd_data_tile_kernel, dim3((unsigned)nb), dim3(128), 0, s, d);
    return check_launch();
}

// gsum holds the suffix sums g_b = sum_{b' >= b} dy_b' (pyr_suffix_kernel)
__global__ void __launch_bounds__(256) pyr_bwd_data_bf16_kernel(ledn_pyrbwd_desc d) {
    constexpr int V = 8;
    LEDN_DYN_SHARED(float, s_w);   // [36][n]
    pyr_stage_weights(d.w, s_w, d.n);
    const int cvn = d.n / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    const bf16_t* g = reinterpret_cast<const bf16_t*>(d.gsum);
    bf16_t* dx = reinterpret_cast<bf16_t*>(d.dx);
    const long npix = (long)d.N * d.H * d.W;
    const long ppb = cdiv(cdiv(npix, (long)gridDim.x), (long)rows) * rows;
    const long p0 = (long)xcd_block(blockIdx.x, gridDim.x) * ppb, p1 = min(npix, p0 + ppb);
    PixCursor cur;
    cur.init(p0 + r, d.H, d.W);
    const int n4 = 4 * d.n;
    for (long p = p0 + r; p < p1; p += rows, cur.advance(rows, d.H, d.W)) {
        const unsigned base = (unsigned)(p * n4 + c);
        f32x2_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x2_t{0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int dl = d.dil[b];
            const unsigned mask = tap_mask(cur.y, cur.x, dl, d.H, d.W);
            const int co = opaque(dl * n4), ro = co * d.W;
            uint4 raw[9];
#pragma unroll
            for (int t = 0; t < 9; ++t)   // source pixel of tap t is p - tap offset: mirrored index 8 - t
                raw[t] = ld_tap(g, base + (unsigned)(b * d.n) + (unsigned)((t / 3 - 1) * ro + (t % 3 - 1) * co), base,
                                (mask >> t) & 1u);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x2_t gv[4], wv[4];
                bf16x8_unpack(raw[t], gv);
                f32x8_load(s_w + (b * 9 + (8 - t)) * d.n + c, wv);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = pk_fma(gv[i], wv[i], acc[i]);
            }
            sched_fence();
        }
        *reinterpret_cast<uint4*>(dx + p * d.n + c) = bf16x8_pack(acc);
    }
}

int pyr_bwd_data_bf16(const ledn_pyrbwd_desc& d, hipStream_t s) {   // gsum already holds the suffix sums
    const int cvn = d.n / 8;
    if (d.dtype != LEDN_BF16 || d.stride != 1 || d.n % 8 || 256 % cvn || cvn > 256 ||
        (long)d.N * d.H * d.W * d.n * 4 >= (1L << 31))
        return -1;
    const int rows = 256 / cvn;
    long nb = cdiv((long)d.N * d.H * d.W, rows * 2);
    if (nb > 2048) nb = 2048;
    LEDN_LAUNCH(pyr_bwd_data_bf16_kernel, dim3((unsigned)nb), dim3(256), (size_t)(36 * d.n) * sizeof(float), s, d);
    return check_launch();
}

__global__ void __launch_bounds__(256) pyr_bwd_weight_bf16_kernel(ledn_pyrbwd_desc d, float* part, bool from_dy) {
    constexpr int V = 8;
    LEDN_DYN_SHARED(float, s_red);   // [4][9*n]
    const int b = blockIdx.y;
    const int dl = d.dil[b];
    const int cvn = d.n / V;
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    f32x2_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x2_t{0.f, 0.f};
    {
        const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
        const bf16_t* g = reinterpret_cast<const bf16_t*>(from_dy ? d.dy : d.gsum);
        const long npo = (long)d.N * d.Ho * d.Wo;
        const long ppb = cdiv(cdiv(npo, (long)gridDim.x), (long)rows) * rows;
        const long p0 = (long)xcd_block(blockIdx.x, gridDim.x) * ppb, p1 = min(npo, p0 + ppb);
        PixCursor cur;
        cur.init(p0 + r, d.Ho, d.Wo);
        for (long p = p0 + r; p < p1; p += rows, cur.advance(rows, d.Ho, d.Wo)) {
            const int yi = cur.y * d.stride, xi = cur.x * d.stride;
            const unsigned base = (unsigned)((((long)cur.n * d.H + yi) * d.W + xi) * d.n + c);
            const unsigned mask = tap_mask(yi, xi, dl, d.H, d.W);
            uint4 raw[9];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                raw[t] = ld_tap(x, base + (unsigned)(((t / 3 - 1) * dl * d.W + (t % 3 - 1) * dl) * d.n), base,
                                (mask >> t) & 1u);
            f32x2_t gv[4];
            if (from_dy) {      // g_b = sum_{b' >= b} dy_b' formed here (the tiled data gradient leaves no gsum behind)
#pragma unroll
                for (int i = 0; i < 4; ++i) gv[i] = f32x2_t{0.f, 0.f};
                for (int bb = 3; bb >= b; --bb) {
                    f32x2_t tv[4];
                    bf16x8_unpack(*reinterpret_cast<const uint4*>(g + p * (4L * d.n) + (long)bb * d.n + c), tv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[i] += tv[i];
                }
            } else {
                bf16x8_unpack(*reinterpret_cast<const uint4*>(g + p * (4L * d.n) + (long)b * d.n + c), gv);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x2_t xv[4];
                bf16x8_unpack(raw[t], xv);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = pk_fma(xv[i], gv[i], acc[t][i]);
            }
        }
    }
    reduce_taps(acc, cvn, s_red, d.n, c);
    for (int e = threadIdx.x; e < 9 * d.n; e += 256) {
        const float sum = (s_red[e] + s_red[9 * d.n + e]) + (s_red[18 * d.n + e] + s_red[27 * d.n + e]);
        // dw layout [4][3][3][n]; partial layout [blk][4*9*n]
        if (part) part[(long)blockIdx.x * 36 * d.n + (long)b * 9 * d.n + e] = sum;
        else atomicAdd(d.dw + (long)b * 9 * d.n + e, sum);
    }
}

// The same gradient when all four branches share one dilation (the spatial SESP blocks: eesp.py:40-62 with Spatial=True)
// and the tiled data gradient left no suffix sums behind: ONE pass over the pixels for the four branches instead of one
// workgroup column per branch -- the nine taps of x are the same for every branch, so a lane loads them once (and the four
// dy vectors once) and feeds 4 x 9 accumulators per channel.  Lane = 4 channels of a pixel (8-byte loads; 144 f32
// accumulators, two waves per SIMD), next pixel's 13 loads in flight under the 72 packed FMAs of this one.  The branch-per-
// column kernel read x through the texture path 36 times and dy 10 times (386 MB for a 42 MB problem at n = 16) with
// nothing in flight while it computed: 40 us at 16 x 128 x 128 x 16.
// Epilogue: reduce-scatter over the two wave halves and the two rows of each half (v_permlane32_swap / v_permlane16_swap:
// one swap + one add per register PAIR, 144 -> 36 registers), a butterfly inside the 16-lane row for the rest.
__global__ void __launch_bounds__(256, 2) pyr_bwd_weight_same_kernel(ledn_pyrbwd_desc d, float* part) {
    constexpr int V = 4;
    LEDN_DYN_SHARED(float, s_red);   // [4 waves][36][n]
    const int dl = d.dil[0], n = d.n;
    const int cvn = n / V;           // 4, 8 or 16 lanes per pixel
    const int rows = 256 / cvn;
    const int r = threadIdx.x / cvn, cv = threadIdx.x % cvn;
    const int c = cv * V;
    // acc[(b * 9 + t) * 2 + i]: branch b, tap t, channel pair i
    f32x2_t acc[72];
#pragma unroll
    for (int j = 0; j < 72; ++j) acc[j] = f32x2_t{0.f, 0.f};
    {
        const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
        const bf16_t* g = reinterpret_cast<const bf16_t*>(d.dy);
        int toff[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * dl * d.W + (t % 3 - 1) * dl) * n;
        const long npix = (long)d.N * d.H * d.W;
        const long ppb = cdiv(cdiv(npix, (long)gridDim.x), (long)rows) * rows;
        const long p0 = (long)xcd_block(blockIdx.x, gridDim.x) * ppb, p1 = min(npix, p0 + ppb);
        PixCursor cur;
        cur.init(p0 + r, d.H, d.W);
        uint2 xr[9], gr[4];
        auto fetch = [&](long p, uint2 (&xo)[9], uint2 (&go)[4]) {
            const bool in = p < p1;
            const unsigned base = in ? (unsigned)(p * n + c) : (unsigned)c;
            const unsigned mask = in ? tap_mask(cur.y, cur.x, dl, d.H, d.W) : 0u;
#pragma unroll
            for (int t = 0; t < 9; ++t) xo[t] = ld_tap8(x, base + (unsigned)toff[t], base, (mask >> t) & 1u);
            const unsigned gb = 4u * (base - (unsigned)c) + (unsigned)c;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) go[bb] = ld_tap8(g, gb + (unsigned)(bb * n), gb, in);
            cur.advance(rows, d.H, d.W);
        };
        long p = p0 + r;
        fetch(p, xr, gr);
        for (; p < p1; p += rows) {
            uint2 xn[9], gn[4];
            fetch(p + rows, xn, gn);
            f32x2_t gs[4][2];           // suffix sums g_b = sum_{b' >= b} dy_b' (the HFF adds of the forward, eesp.py:96-101)
            bf16x4_unpack(gr[3], gs[3]);
#pragma unroll
            for (int bb = 2; bb >= 0; --bb) {
                f32x2_t tv[2];
                bf16x4_unpack(gr[bb], tv);
                gs[bb][0] = gs[bb + 1][0] + tv[0];
                gs[bb][1] = gs[bb + 1][1] + tv[1];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x2_t xv[2];
                bf16x4_unpack(xr[t], xv);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    acc[(bb * 9 + t) * 2] = pk_fma(xv[0], gs[bb][0], acc[(bb * 9 + t) * 2]);
                    acc[(bb * 9 + t) * 2 + 1] = pk_fma(xv[1], gs[bb][1], acc[(bb * 9 + t) * 2 + 1]);
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) xr[t] = xn[t];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) gr[bb] = gn[bb];
        }
    }
    // lanes r * cvn + cv, cvn <= 16: the two wave halves and the two rows of a half always hold different pixels
#pragma unroll
    for (int j = 0; j < 36; ++j) swap_add32(acc[j], acc[j + 36]);
#pragma unroll
    for (int j = 0; j < 18; ++j) swap_add16(acc[j], acc[j + 18]);
    if (cvn <= 8) {
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            acc[j].x = lane_step_sum(acc[j].x, 8);
            acc[j].y = lane_step_sum(acc[j].y, 8);
        }
    }
    if (cvn <= 4) {
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            acc[j].x = lane_step_sum(acc[j].x, 4);
            acc[j].y = lane_step_sum(acc[j].y, 4);
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if ((lane & 15) < cvn) {
        // register j of this lane = float pair 2 j, 2 j + 1 of the 144 = [36 (branch, tap)][4 channels], offset by 36 floats
        // in the odd rows and by 72 in the upper half of the wave
        const int bt0 = 9 * ((lane >> 4) & 1) + 18 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            float* dst = s_red + (long)(wid * 36 + bt0 + j / 2) * n + c + 2 * (j % 2);
            dst[0] = acc[j].x;
            dst[1] = acc[j].y;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 36 * n; e += 256) {
        const float sum = (s_red[e] + s_red[36 * n + e]) + (s_red[72 * n + e] + s_red[108 * n + e]);
        if (part) part[(long)blockIdx.x * 36 * n + e] = sum;      // dw layout [4][3][3][n]
        else atomicAdd(d.dw + e, sum);
    }
}

int pyr_bwd_weight_bf16(const ledn_pyrbwd_desc& d, hipStream_t s) {
    const int cvn = d.n / 8;
    if (d.dtype != LEDN_BF16 || d.n % 8 || !pow2(cvn) || cvn > 64 || (long)d.N * d.H * d.W * d.n * 4 >= (1L << 31))
        return -1;
    const int rows = 256 / cvn;
    static const bool same_on = exp_knob("LEDN_PYR_WGRAD_SAME", 1) != 0;     // (A/B knob)
    if (same_on && pyr_tile_applies(d) && d.n <= 64 && d.dil[1] == d.dil[0] && d.dil[2] == d.dil[0] && d.dil[3] == d.dil[0]) {
        const int rows4 = 256 / (d.n / 4);
        long nb4 = cdiv((long)d.N * d.H * d.W, rows4 * 4);
        if (nb4 > 512) nb4 = 512;            // one resident round: two workgroups per CU by the registers
        float* part4 = (nb4 > 32 || det()) ? ws_take(nb4 * 36 * d.n) : nullptr;
        if (!part4 && nb4 > 128) nb4 = 128;
        LEDN_LAUNCH(pyr_bwd_weight_same_kernel, dim3((unsigned)nb4), dim3(256), (size_t)(144 * d.n) * sizeof(float), s, d,
                    part4);
        if (part4) return finish_partials(part4, (int)nb4, 36 * d.n, 1, d.dw, nullptr, nullptr, s);
        return check_launch();
    }
    long nb = cdiv((long)d.N * d.Ho * d.Wo, rows * 8);
    if (nb > 512) nb = 512;
    float* part = (nb > 32 || det()) ? ws_take(nb * 36 * d.n) : nullptr;
    if (!part && nb > 128) nb = 128;
    LEDN_LAUNCH(pyr_bwd_weight_bf16_kernel, dim3((unsigned)nb, 4u), dim3(256), (size_t)(36 * d.n) * sizeof(float), s, d,
                part, pyr_tile_applies(d));
    if (part) return finish_partials(part, (int)nb, 36 * d.n, 1, d.dw, nullptr, nullptr, s);
    return check_launch();
}

}  // namespace ledn
