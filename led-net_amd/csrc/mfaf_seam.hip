// mfaf_seam.hip -- MFAF (Muti_AFF) gate blend and the SEAM edge-map pipeline.
#include "ledn_rt.h"

namespace ledn {

// wei = sigmoid(aff0(xl) + sum_k aff_k(ctx_k nearest-upsampled) + aff4(xg))
// out = 2*x*wei + 2*r*(1-wei).   One thread = one pixel x 4 channels.
template <typename T, int V>
__global__ void __launch_bounds__(256) mfaf_gate_kernel(ledn_mfaf_desc d) {
    const int cv = d.C / V;
    const long total = (long)d.N * d.H * d.W * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const NhwcIdx ix_ = nhwc_split(idx, cv, d.W, d.H);
    const int c = ix_.cv * V;
    const long pix = ix_.pix;
    const int x = ix_.x, y = ix_.y, n = ix_.n;
    float s[V], t[V];
    ldv<V>(reinterpret_cast<const T*>(d.xl) + pix * d.C + c, t);
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = t[v] * d.scale[0][c + v] + d.shift[0][c + v];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int S = d.ctx_size[k];
        // F.interpolate(mode='nearest'): src = floor(dst * in / out)
        int sy = (int)((float)y * ((float)S / (float)d.H));
        int sx = (int)((float)x * ((float)S / (float)d.W));
        if (sy > S - 1) sy = S - 1;
        if (sx > S - 1) sx = S - 1;
        ldv<V>(d.ctx[k] + (((long)n * S + sy) * S + sx) * d.C + c, t);
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] += t[v] * d.scale[k + 1][c + v] + d.shift[k + 1][c + v];
    }
    float xv[V], rv[V], o[V];
    ldv<V>(reinterpret_cast<const T*>(d.x) + pix * d.C + c, xv);
    ldv<V>(reinterpret_cast<const T*>(d.r) + pix * d.C + c, rv);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const float w = 1.f / (1.f + __expf(-s[v]));
        o[v] = 2.f * xv[v] * w + 2.f * rv[v] * (1.f - w);
        if (d.act == LEDN_ACT_RELU) o[v] = fmaxf(o[v], 0.f);
    }
    stv<V>(reinterpret_cast<T*>(d.out) + pix * d.C + c, o);
}

// The same for bf16 maps with C a power of two in 8 .. 256, at the streaming rate: lane = 8 channels (16-byte accesses) of UNR
// pixels 256 / (C / 8) apart, all their loads issued before the first use; the five per-channel scales and the sum of the
// five shifts sit in registers (the form above issues ~40 parameter loads per lane for its three 8-byte tensor loads:
// 50 us for 134 MB at 16 x 128 x 128 x 64).
template <int UNR>
__global__ void __launch_bounds__(256) mfaf_gate_fast_kernel(ledn_mfaf_desc d) {
    const int cvn = d.C >> 3, rows = 256 / cvn;
    const int cv = (int)(threadIdx.x % (unsigned)cvn), r = (int)(threadIdx.x / (unsigned)cvn), c = cv * 8;
    float sc[5][8], shs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            sc[k][i] = d.scale[k][c + i];
            t += d.shift[k][c + i];
        }
        shs[i] = t;
    }
    const long npix = (long)d.N * d.H * d.W;
    const long p0 = (long)blockIdx.x * (rows * UNR) + r;
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.x);
    const bf16_t* rr = reinterpret_cast<const bf16_t*>(d.r);
    const bf16_t* xl = reinterpret_cast<const bf16_t*>(d.xl);
    uint4 xv[UNR], rv[UNR], lv[UNR];
    float4 cx[UNR][4][2];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long p = p0 + (long)u * rows, q = p < npix ? p : 0;
        xv[u] = *reinterpret_cast<const uint4*>(x + q * d.C + c);
        rv[u] = *reinterpret_cast<const uint4*>(rr + q * d.C + c);
        lv[u] = *reinterpret_cast<const uint4*>(xl + q * d.C + c);
        const int px = (int)(q % d.W), py = (int)((q / d.W) % d.H), n = (int)(q / ((long)d.W * d.H));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int S = d.ctx_size[k];
            // F.interpolate(mode='nearest'): src = floor(dst * in / out)
            int sy = (int)((float)py * ((float)S / (float)d.H));
            int sx = (int)((float)px * ((float)S / (float)d.W));
            if (sy > S - 1) sy = S - 1;
            if (sx > S - 1) sx = S - 1;
            const float4* cp = reinterpret_cast<const float4*>(d.ctx[k] + (((long)n * S + sy) * S + sx) * d.C + c);
            cx[u][k][0] = cp[0];
            cx[u][k][1] = cp[1];
        }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long p = p0 + (long)u * rows;
        if (p >= npix) break;
        const unsigned xw[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, rw[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
        const unsigned lw[4] = {lv[u].x, lv[u].y, lv[u].z, lv[u].w};
        unsigned ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o2[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int i = 2 * j + hh;
                const float xf = __uint_as_float(hh ? (xw[j] & 0xffff0000u) : (xw[j] << 16));
                const float rf = __uint_as_float(hh ? (rw[j] & 0xffff0000u) : (rw[j] << 16));
                const float lf = __uint_as_float(hh ? (lw[j] & 0xffff0000u) : (lw[j] << 16));
                float s = lf * sc[0][i] + shs[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 a = cx[u][k][i >> 2];
                    const float cvv = (i & 3) == 0 ? a.x : ((i & 3) == 1 ? a.y : ((i & 3) == 2 ? a.z : a.w));
                    s = fmaf(cvv, sc[k + 1][i], s);
                }
                const float w = 1.f / (1.f + __expf(-s));
                float o = 2.f * xf * w + 2.f * rf * (1.f - w);
                if (d.act == LEDN_ACT_RELU) o = fmaxf(o, 0.f);
                o2[hh] = o;
            }
            ow[j] = (unsigned)f32_to_bf16(o2[0]) | ((unsigned)f32_to_bf16(o2[1]) << 16);
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(d.out) + p * d.C + c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

int mfaf_gate_impl(const ledn_mfaf_desc& d, hipStream_t s) {
    LEDN_REQUIRE(d.x && d.r && d.xl && d.out && d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.C % 4 == 0);
    for (int k = 0; k < 4; ++k) LEDN_REQUIRE(d.ctx[k] && d.ctx_size[k] > 0);
    for (int k = 0; k < 5; ++k) LEDN_REQUIRE(d.scale[k] && d.shift[k]);
    static const bool fast_on = exp_knob("LEDN_MFAF_FAST", 1) != 0;     // (A/B knob)
    if (fast_on && (options().stream_fast & 1) && d.dtype == LEDN_BF16 && d.C >= 8 && d.C <= 256 && (d.C & (d.C - 1)) == 0 &&
        (long)d.N * d.H * d.W >= 4096) {
        constexpr int UNR = 2;
        const int rows = 256 / (d.C >> 3);
        LEDN_LAUNCH(mfaf_gate_fast_kernel<UNR>, dim3((unsigned)cdiv((long)d.N * d.H * d.W, (long)rows * UNR)), dim3(256), 0, s, d);
        return check_launch();
    }
    const long total = (long)d.N * d.H * d.W * (d.C / 4);
    const dim3 grid((unsigned)cdiv(total, 256));
    if (d.dtype == LEDN_F32) LEDN_LAUNCH((mfaf_gate_kernel<float, 4>), grid, dim3(256), 0, s, d);
    else if (d.dtype == LEDN_BF16) LEDN_LAUNCH((mfaf_gate_kernel<bf16_t, 4>), grid, dim3(256), 0, s, d);
    else return LEDN_EINVAL;
    return check_launch();
}

// ---------------------------------------------------------------------------
// SEAM edge map.  One 256-thread workgroup per image (h*w is 1/64 of the input
// pixel count: 16 K values at 1024^2).  Phases, separated by __syncthreads():
//   1 min / max of seg                      (wave + LDS reduction)
//   2 three clamped Laplacian responses, upsampled (nearest) -> scratch[3][h*w]
//   3 per scale: k-th order statistic by 4-pass 8-bit radix select on the f32
//     bit patterns (responses are >= 0, so the patterns are order-preserving)
//   4 binarise, 0.6/0.3/0.1 fuse, binarise -> edge
// ---------------------------------------------------------------------------
__device__ __forceinline__ float seam_lap(const float* seg, int h, int w, int cy, int cx, float mn,
                                          float den) {
    // 3x3 Laplacian [-1..8..-1] on the min-max normalised map, zero padding
    float acc = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = cy + dy, xx = cx + dx;
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            const float v = (seg[yy * w + xx] - mn) / den;
            acc += (dy == 0 && dx == 0) ? 8.f * v : -v;
        }
    return acc > 0.f ? acc : 0.f;
}

// One workgroup of 1024 threads per image (the per-image order statistic needs a workgroup-wide view;
// at batch 1 this kernel is on the latency path of the whole network: 256 threads and a one-thread
// 256-bin scan per radix pass took 0.5-0.7 ms).
constexpr int SEAM_T = 1024;
__global__ void __launch_bounds__(SEAM_T) seam_edge_kernel(const float* seg_all, float* edge_all,
                                                           float* scratch, int h, int w, int kth,
                                                           float thr, float final_thr) {
    __shared__ float s_red[2][SEAM_T / 64];
    __shared__ unsigned s_hist[3][256];
    __shared__ unsigned s_wave[3][4];
    __shared__ unsigned s_sel[3][2];  // per level: prefix, remaining rank
    __shared__ float s_thr[3];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int hw = h * w;
    const float* seg = seg_all + (long)n * hw;
    float* edge = edge_all + (long)n * hw;
    float* resp = scratch + (long)n * 3 * hw;

    // phase 1: min / max of the image
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int i = tid; i < hw; i += SEAM_T) {
        const float v = seg[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((tid & 63) == 0) {
        s_red[0][tid >> 6] = mn;
        s_red[1][tid >> 6] = mx;
    }
    __syncthreads();
    mn = s_red[0][0];
    mx = s_red[1][0];
    for (int i = 1; i < SEAM_T / 64; ++i) {
        mn = fminf(mn, s_red[0][i]);
        mx = fmaxf(mx, s_red[1][i]);
    }
    const float den = mx - mn;

    // phase 2: the three Laplacian responses (strides 1 / 2 / 4, nearest-upsampled)
    for (int i = tid; i < hw; i += SEAM_T) {
        const int y = i / w, x = i % w;
        resp[i] = seam_lap(seg, h, w, y, x, mn, den);
#pragma unroll
        for (int sidx = 1; sidx < 3; ++sidx) {
            const int st = sidx == 1 ? 2 : 4;
            const int hs = (h - 1) / st + 1, wss = (w - 1) / st + 1;  // conv2d(stride, pad 1) size
            int sy = (int)((float)y * ((float)hs / (float)h));
            int sx = (int)((float)x * ((float)wss / (float)w));
            if (sy > hs - 1) sy = hs - 1;
            if (sx > wss - 1) sx = wss - 1;
            resp[sidx * hw + i] = seam_lap(seg, h, w, sy * st, sx * st, mn, den);
        }
    }
    __syncthreads();

    // phase 3: k-th smallest response of the THREE levels together (4 x 8-bit radix select, one histogram per
    // level).  The clamped Laplacian is exactly 0 on about half of the pixels: those would all hit bin 0 of every
    // pass (same-address LDS atomics serialise), so a wavefront counts its zeros with one ballot per level.
    if (kth > 0) {
        const int kk = kth > hw ? hw : kth;
        if (tid < 3) {
            s_sel[tid][0] = 0u;
            s_sel[tid][1] = (unsigned)(kk - 1);     // 0-based rank still to locate
        }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 768) s_hist[tid >> 8][tid & 255] = 0u;
            __syncthreads();
            const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            unsigned prefix[3], rank[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                prefix[l] = s_sel[l][0];
                rank[l] = s_sel[l][1];
            }
            for (int i0 = 0; i0 < hw; i0 += SEAM_T) {
                const int i = i0 + tid;
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    const unsigned u = i < hw ? __float_as_uint(resp[l * hw + i]) : 0xffffffffu;
                    const bool zero = u == 0u && prefix[l] == 0u;
                    const unsigned long long zm = __ballot(zero);
                    if (zm != 0ull && (tid & 63) == 0) atomicAdd(&s_hist[l][0], (unsigned)__popcll(zm));
                    if (i < hw && !zero && (u & himask) == prefix[l]) atomicAdd(&s_hist[l][(u >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            // bucket holding the rank: prefix sum over the 256 bins of level l by wavefronts 4l .. 4l+3
            unsigned mine = 0u, incl = 0u;
            const int l = tid >> 8, b = tid & 255;
            if (tid < 768) {
                mine = s_hist[l][b];
                incl = mine;
                const int lane = tid & 63;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned nbr = __shfl(incl, lane >= o ? lane - o : lane);
                    if (lane >= o) incl += nbr;
                }
                if (lane == 63) s_wave[l][b >> 6] = incl;
            }
            __syncthreads();
            if (tid < 768) {
                const unsigned rank0 = l == 0 ? rank[0] : l == 1 ? rank[1] : rank[2];
                const unsigned pre = l == 0 ? prefix[0] : l == 1 ? prefix[1] : prefix[2];
                for (int i = 0; i < (b >> 6); ++i) incl += s_wave[l][i];
                const unsigned excl = incl - mine;
                const unsigned total = s_wave[l][0] + s_wave[l][1] + s_wave[l][2] + s_wave[l][3];
                if (rank0 >= excl && rank0 < incl) {                // exactly one thread when rank0 < total
                    s_sel[l][0] = pre | ((unsigned)b << shift);
                    s_sel[l][1] = rank0 - excl;
                } else if (rank0 >= total && b == 255) {            // (cannot happen: rank < count; kept as the
                    s_sel[l][0] = pre | (256u << shift);            //  serial scan's fall-through: b = 256)
                    s_sel[l][1] = rank0 - total;
                }
            }
            __syncthreads();
        }
        if (tid < 3) s_thr[tid] = __uint_as_float(s_sel[tid][0]);
        __syncthreads();
    } else {
        if (tid < 3) s_thr[tid] = thr;
        __syncthreads();
    }

    // phase 4: binarise, fuse, binarise
    for (int i = tid; i < hw; i += SEAM_T) {
        const float f = 0.6f * (resp[i] > s_thr[0] ? 1.f : 0.f) +
                        0.3f * (resp[hw + i] > s_thr[1] ? 1.f : 0.f) +
                        0.1f * (resp[2 * hw + i] > s_thr[2] ? 1.f : 0.f);
        edge[i] = f > final_thr ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------------------
// The same map for h * w <= 16 K pixels (every size the network produces up to 1024 x 1024 inputs) with the image
// RESIDENT: the min-max normalised map lives in LDS (64 KB), each lane keeps its 16 pixels x 3 responses in
// registers.  The generic kernel above reads the map from global memory inside the Laplacian's border branches (27
// dependent L2 round trips per pixel and lane: 88 of its 133 us at 16 x 128 x 128) and the responses back from a
// global scratch in every radix pass.  Arithmetic, summation order and the radix select are the generic kernel's
// (bit-identical responses and thresholds).  Histogram: 4 interleaved copies per level (lane & 3) -- most responses
// share their top byte, and same-address LDS atomics of a wavefront serialise.
// ---------------------------------------------------------------------------
constexpr int SEAM_RPT = 16, SEAM_HC = 4;
__device__ __forceinline__ float seam_lap_lds(const float* s, int h, int w, int cy, int cx) {
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = cy + dy, xx = cx + dx;
            const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
            const float v = s[min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
            const float term = (dy == 0 && dx == 0) ? 8.f * v : -v;
            if (in) acc += term;
        }
    return acc > 0.f ? acc : 0.f;
}

__global__ void __launch_bounds__(SEAM_T) seam_edge_lds_kernel(const float* seg_all, float* edge_all, int h, int w,
                                                               int kth, float thr, float final_thr) {
    __shared__ float s_seg[SEAM_T * SEAM_RPT];
    __shared__ float s_red[2][SEAM_T / 64];
    __shared__ unsigned s_hist[3][SEAM_HC][256];
    __shared__ unsigned s_wave[3][4];
    __shared__ unsigned s_sel[3][2];  // per level: prefix, remaining rank
    __shared__ float s_thr[3];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int hw = h * w;
    const float* seg = seg_all + (long)n * hw;
    float* edge = edge_all + (long)n * hw;

    // phase 1: min / max, then the normalised map -> LDS
    float raw[SEAM_RPT];
    float mn = 3.0e38f, mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < SEAM_RPT; ++r) {
        const int i = tid + r * SEAM_T;
        raw[r] = i < hw ? seg[i] : 0.f;
        if (i < hw) {
            mn = fminf(mn, raw[r]);
            mx = fmaxf(mx, raw[r]);
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((tid & 63) == 0) {
        s_red[0][tid >> 6] = mn;
        s_red[1][tid >> 6] = mx;
    }
    __syncthreads();
    mn = s_red[0][0];
    mx = s_red[1][0];
    for (int i = 1; i < SEAM_T / 64; ++i) {
        mn = fminf(mn, s_red[0][i]);
        mx = fmaxf(mx, s_red[1][i]);
    }
    const float den = mx - mn;
#pragma unroll
    for (int r = 0; r < SEAM_RPT; ++r) {
        const int i = tid + r * SEAM_T;
        if (i < hw) s_seg[i] = (raw[r] - mn) / den;
    }
    __syncthreads();

    // phase 2: the three Laplacian responses (strides 1 / 2 / 4, nearest-upsampled) of this lane's pixels
    float resp[3][SEAM_RPT];
#pragma unroll
    for (int r = 0; r < SEAM_RPT; ++r) {
        const int i = tid + r * SEAM_T;
        const int ic = min(i, hw - 1);
        const int y = ic / w, x = ic % w;
        resp[0][r] = seam_lap_lds(s_seg, h, w, y, x);
#pragma unroll
        for (int sidx = 1; sidx < 3; ++sidx) {
            const int st = sidx == 1 ? 2 : 4;
            const int hs = (h - 1) / st + 1, wss = (w - 1) / st + 1;  // conv2d(stride, pad 1) size
            int sy = (int)((float)y * ((float)hs / (float)h));
            int sx = (int)((float)x * ((float)wss / (float)w));
            if (sy > hs - 1) sy = hs - 1;
            if (sx > wss - 1) sx = wss - 1;
            resp[sidx][r] = seam_lap_lds(s_seg, h, w, sy * st, sx * st);
        }
    }

    // phase 3: k-th smallest response of the three levels together (4 x 8-bit radix select)
    if (kth > 0) {
        const int kk = kth > hw ? hw : kth;
        if (tid < 3) {
            s_sel[tid][0] = 0u;
            s_sel[tid][1] = (unsigned)(kk - 1);     // 0-based rank still to locate
        }
        const int hc = tid & (SEAM_HC - 1);
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 3 * SEAM_HC * 256; i += SEAM_T) (&s_hist[0][0][0])[i] = 0u;
            __syncthreads();
            const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            unsigned prefix[3], rank[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                prefix[l] = s_sel[l][0];
                rank[l] = s_sel[l][1];
            }
#pragma unroll
            for (int r = 0; r < SEAM_RPT; ++r) {
                const bool live = tid + r * SEAM_T < hw;
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    const unsigned u = __float_as_uint(resp[l][r]);
                    const bool zero = live && u == 0u && prefix[l] == 0u;     // (exact zeros: about half of the map)
                    const unsigned long long zm = __ballot(zero);
                    if (zm != 0ull && (tid & 63) == 0) atomicAdd(&s_hist[l][0][0], (unsigned)__popcll(zm));
                    if (live && !zero && (u & himask) == prefix[l]) atomicAdd(&s_hist[l][hc][(u >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            // bucket holding the rank: prefix sum over the 256 bins of level l by wavefronts 4l .. 4l+3
            unsigned mine = 0u, incl = 0u;
            const int l = tid >> 8, b = tid & 255;
            if (tid < 768) {
#pragma unroll
                for (int c = 0; c < SEAM_HC; ++c) mine += s_hist[l][c][b];
                incl = mine;
                const int lane = tid & 63;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned nbr = __shfl(incl, lane >= o ? lane - o : lane);
                    if (lane >= o) incl += nbr;
                }
                if (lane == 63) s_wave[l][b >> 6] = incl;
            }
            __syncthreads();
            if (tid < 768) {
                const unsigned rank0 = l == 0 ? rank[0] : l == 1 ? rank[1] : rank[2];
                const unsigned pre = l == 0 ? prefix[0] : l == 1 ? prefix[1] : prefix[2];
                for (int i = 0; i < (b >> 6); ++i) incl += s_wave[l][i];
                const unsigned excl = incl - mine;
                const unsigned total = s_wave[l][0] + s_wave[l][1] + s_wave[l][2] + s_wave[l][3];
                if (rank0 >= excl && rank0 < incl) {                // exactly one thread when rank0 < total
                    s_sel[l][0] = pre | ((unsigned)b << shift);
                    s_sel[l][1] = rank0 - excl;
                } else if (rank0 >= total && b == 255) {            // (cannot happen: rank < count)
                    s_sel[l][0] = pre | (256u << shift);
                    s_sel[l][1] = rank0 - total;
                }
            }
            __syncthreads();
        }
        if (tid < 3) s_thr[tid] = __uint_as_float(s_sel[tid][0]);
        __syncthreads();
    } else {
        if (tid < 3) s_thr[tid] = thr;
        __syncthreads();
    }

    // phase 4: binarise, fuse, binarise
    const float t0 = s_thr[0], t1 = s_thr[1], t2 = s_thr[2];
#pragma unroll
    for (int r = 0; r < SEAM_RPT; ++r) {
        const int i = tid + r * SEAM_T;
        const float f = 0.6f * (resp[0][r] > t0 ? 1.f : 0.f) + 0.3f * (resp[1][r] > t1 ? 1.f : 0.f) +
                        0.1f * (resp[2][r] > t2 ? 1.f : 0.f);
        if (i < hw) edge[i] = f > final_thr ? 1.f : 0.f;
    }
}

int seam_edge_impl(const float* seg, float* edge, float* scratch, int N, int h, int w, int kth, float thr,
                   float final_thr, hipStream_t s) {
    LEDN_REQUIRE(seg && edge && scratch && N > 0 && h > 0 && w > 0);
    LEDN_REQUIRE(kth >= 0);
    if ((long)h * w <= SEAM_T * SEAM_RPT)
        LEDN_LAUNCH(seam_edge_lds_kernel, dim3((unsigned)N), dim3(SEAM_T), 0, s, seg, edge, h, w, kth, thr, final_thr);
    else
        LEDN_LAUNCH(seam_edge_kernel, dim3((unsigned)N), dim3(SEAM_T), 0, s, seg, edge, scratch, h, w, kth, thr,
                    final_thr);
    return check_launch();
}

}  // namespace ledn
