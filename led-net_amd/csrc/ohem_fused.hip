// ohem_fused.hip -- BOTH OhemCrossEntropy losses of LEDHead.loss_by_feat (led_head.py:132-146: loss_context on the
// fused context logits, loss_spatial on the fused spatial logits, same labels) in ONE launch set, with the last
// (exact 2x) resize of each fused output folded in (as ledn_ohem_ce_up_*).
//
// What it changes against two ledn_ohem_ce_up_fwd / _bwd calls (attn_loss_opt.hip), per step at 16 x 1024 x 1024:
//   * the int64 label plane (134 MB) is read ONCE; the kernels after the first read a uint8 copy (16.7 MB) that the
//     first one writes on the way (4 x 134 MB of label reads -> 134 + 5 x 16.7 MB);
//   * no per-pixel loss array: the masked mean re-forms -log p from the stored probability (2 x 67 MB written and
//     read less);
//   * four consecutive pixels per thread (16-byte stores of the probabilities, 4-byte label stores, one row
//     interpolation per thread);
//   * no same-address global atomics on the critical path: at random initialisation every probability is ~0.5, i.e. a
//     dozen of the 2048 level-0 bins, and thousands of workgroups flushing their LDS histogram into the same few
//     words serialise at ~25 ns per atomic (measured r3e: 344 us for the first version of the probability kernel,
//     109 us for a masked mean that ended in 4096 x 4 atomics on four words).  Histograms and counters are kept in
//     O2_REP replicas (workgroup b adds into replica b % O2_REP, the single-workgroup scan sums them), the masked
//     mean writes one partial per workgroup and the final kernel sums them in a fixed order;
//   * rows are walked by persistent workgroups (image / row from the row index, no 64-bit division per pixel);
//   * the three radix-select levels, the masked mean and the final scalars handle both losses per launch.
// Selection semantics are exactly ohem_cross_entropy_loss.py:62-90 (k-th order statistic over the valid pixels,
// threshold = max(k-th, thres), strict <), bit-exact as the single-loss kernels.
//
// work layout (floats): prob0[P] | prob1[P] | lab8[P bytes, padded to 16 B] | u32 hist[2][3][O2_REP][2048] |
//                       u32 cnt[O2_REP][2] (n_valid, n_correct) | float part[O2_GRID][4] | u32 state[2][16]
// state (per loss): 0 n_valid, 1 n_correct, 2 rank (remaining), 3 prefix bits, 4 thr bits, 5 n_selected, 6 (float) loss sum
#include "ledn_rt.h"

namespace ledn {

constexpr int O2_BINS = 2048;
constexpr int O2_REP = 16;          // replicas of the histograms / counters (workgroup b uses replica b % O2_REP)
constexpr int O2_GRID = 1024;       // persistent workgroups of the per-pixel passes (4 per CU; 2048 measured slower)

struct Ohem2Work {
    float* prob[2];
    unsigned char* lab8;
    unsigned* hist;     // [2][3][O2_REP][O2_BINS]
    unsigned* cnt;      // [O2_REP][2]
    float* part;        // [O2_GRID][4]: per-workgroup (sum0, sum1, count0, count1) of the masked mean
    unsigned* state;    // [2][16]
};
__host__ __device__ inline long long o2_lab_floats(long long P) { return (P + 15) / 16 * 4; }
__host__ __device__ inline Ohem2Work o2_work(float* work, long long P) {
    Ohem2Work w;
    w.prob[0] = work;
    w.prob[1] = work + P;
    w.lab8 = reinterpret_cast<unsigned char*>(work + 2 * P);
    w.hist = reinterpret_cast<unsigned*>(work + 2 * P + o2_lab_floats(P));
    w.cnt = w.hist + 2 * 3 * O2_REP * O2_BINS;
    w.part = reinterpret_cast<float*>(w.cnt + O2_REP * 2);
    w.state = reinterpret_cast<unsigned*>(w.part + O2_GRID * 4);
    return w;
}
constexpr long long O2_CLEAR_WORDS = 2LL * 3 * O2_REP * O2_BINS + O2_REP * 2;     // histograms + counters (zeroed per call)
long long ohem2_work_floats(long long P) { return 2 * P + o2_lab_floats(P) + O2_CLEAR_WORDS + O2_GRID * 4 + 32; }

// radix digits of the probability's f32 bit pattern, most significant first: 11 + 11 + 8 bits starting at bit 29.  A
// probability is <= 1.0 = 0x3F800000 < 2^30, so bits 31..30 carry nothing; starting the first digit there (as the
// single-loss kernels do: 11 + 11 + 10 from bit 31) leaves it two mantissa bits -- four bins per octave, and at random
// initialisation (p ~ 0.5) a wave's 64 LDS atomics land on ~5 addresses.  From bit 29 it has four mantissa bits.
constexpr int O2_SH0 = 19, O2_SH1 = 8;
constexpr unsigned O2_MASK1 = 0xfff80000u, O2_MASK2 = 0xffffff00u;     // prefix bits known after level 0 / level 1
__device__ __forceinline__ int o2_bin(unsigned u, int pass) {
    return pass == 0 ? (int)(u >> O2_SH0) : (pass == 1 ? (int)((u >> O2_SH1) & 2047u) : (int)(u & 255u));
}

// lerp_coord (ledn_rt.h) with the scale in / out formed once by the caller: the same expressions, bit-identical
__device__ __forceinline__ Lerp o2_lerp(int dst, int in, float scale) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Lerp l;
    l.i0 = (int)src;
    if (l.i0 > in - 1) l.i0 = in - 1;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

__device__ __forceinline__ float2 o2_mix(float wy0, float wy1, float wx0, float wx1, float2 v00, float2 v01, float2 v10,
                                          float2 v11) {
    float2 r;
    r.x = wy0 * (wx0 * v00.x + wx1 * v01.x) + wy1 * (wx0 * v10.x + wx1 * v11.x);
    r.y = wy0 * (wx0 * v00.y + wx1 * v01.y) + wy1 * (wx0 * v10.y + wx1 * v11.y);
    return r;
}
__device__ __forceinline__ float2 o2_up(const float* src, int Ws, const Lerp& ly, const Lerp& lx) {
    const float2 v00 = *reinterpret_cast<const float2*>(src + ((long)ly.i0 * Ws + lx.i0) * 2);
    const float2 v01 = *reinterpret_cast<const float2*>(src + ((long)ly.i0 * Ws + lx.i1) * 2);
    const float2 v10 = *reinterpret_cast<const float2*>(src + ((long)ly.i1 * Ws + lx.i0) * 2);
    const float2 v11 = *reinterpret_cast<const float2*>(src + ((long)ly.i1 * Ws + lx.i1) * 2);
    return o2_mix(ly.w0, ly.w1, lx.w0, lx.w1, v00, v01, v10, v11);
}

// pass over the pixels, four per thread (W % 4 == 0: a quad never crosses a row): probabilities of the target class for
// both fused outputs, validity / accuracy counts, level-0 histograms, the uint8 label plane.  Persistent workgroups walk
// the N x H rows.
__global__ void __launch_bounds__(256) ohem2_prob_kernel(const float* src0, const float* src1, int N, int Hs, int Ws,
                                                         int H, int W, const long long* target, int ignore_label,
                                                         float* work) {
    __shared__ unsigned s_hist[2][O2_BINS];
    __shared__ unsigned s_cnt[2];
    const long P = (long)N * H * W;
    const Ohem2Work w = o2_work(work, P);
    for (int i = threadIdx.x; i < 2 * O2_BINS; i += blockDim.x) (&s_hist[0][0])[i] = 0u;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    unsigned nvalid = 0u, ncorr = 0u;
    const int rows = N * H, qpr = W / 4;
    const float sy = (float)Hs / (float)H, sx = (float)Ws / (float)W;
    const bool exact2x = H == 2 * Hs && W == 2 * Ws;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const int n = r / H, y = r - n * H;
        const Lerp ly = o2_lerp(y, Hs, sy);
        const float* s0 = src0 + (long)n * Hs * Ws * 2;
        const float* s1 = src1 + (long)n * Hs * Ws * 2;
        const long rowp = (long)r * W;
        for (int qx = threadIdx.x; qx < qpr; qx += blockDim.x) {
            const long p = rowp + 4 * qx;
            long long tg[4];
            {
                const uint4 a = *reinterpret_cast<const uint4*>(target + p);          // 2 x int64 per 16-byte load
                const uint4 b = *reinterpret_cast<const uint4*>(target + p + 2);
                tg[0] = (long long)(((unsigned long long)a.y << 32) | a.x);
                tg[1] = (long long)(((unsigned long long)a.w << 32) | a.z);
                tg[2] = (long long)(((unsigned long long)b.y << 32) | b.x);
                tg[3] = (long long)(((unsigned long long)b.w << 32) | b.z);
            }
            float pr[2][4];
            unsigned lab = 0u;
            // exact 2x resize, quad away from the left / right border: pixels 4q .. 4q+3 interpolate source columns
            // 2q-1 .. 2q+2 with the weights (.25 .75) (.75 .25) (.25 .75) (.75 .25) -- what o2_lerp returns there
            // (scale 0.5: every coordinate is an exact multiple of 0.25), so the eight source values of a row pair are
            // loaded once per quad instead of four gathers per pixel and no coordinate arithmetic is left
            const bool fast = exact2x && qx >= 1 && qx < qpr - 1;
            float2 c[2][2][4];           // [source][row][column 2q-1+i]
            if (fast) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const float* rowp2 = (j ? s1 : s0) + ((long)(rr ? ly.i1 : ly.i0) * Ws + 2 * qx - 1) * 2;
#pragma unroll
                        for (int i = 0; i < 4; ++i) c[j][rr][i] = *reinterpret_cast<const float2*>(rowp2 + 2 * i);
                    }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = tg[k] != ignore_label;
                lab |= ((unsigned)tg[k] & 0xffu) << (8 * k);
                Lerp lx;
                if (!fast) lx = o2_lerp(4 * qx + k, Ws, sx);
                const int t = ok ? (int)tg[k] : 0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float2 lg;
                    if (fast) {
                        constexpr int ci[4] = {0, 1, 1, 2};
                        const float wx1 = (k & 1) ? 0.25f : 0.75f, wx0 = 1.f - wx1;
                        lg = o2_mix(ly.w0, ly.w1, wx0, wx1, c[j][0][ci[k]], c[j][0][ci[k] + 1], c[j][1][ci[k]], c[j][1][ci[k] + 1]);
                    } else {
                        lg = o2_up(j ? s1 : s0, Ws, ly, lx);
                    }
                    const int am = lg.y > lg.x ? 1 : 0;
                    const float mx = am ? lg.y : lg.x;
                    const float se = __expf(lg.x - mx) + __expf(lg.y - mx);
                    const float lt = (t ? lg.y : lg.x) - mx;
                    pr[j][k] = ok ? __expf(lt) / se : 2.0f;      // 2 > any probability: never selected
                    if (ok) atomicAdd(&s_hist[j][o2_bin(__float_as_uint(pr[j][k]), 0)], 1u);
                    if (j == 0 && ok && am == t) ++ncorr;
                }
                if (ok) ++nvalid;
            }
            *reinterpret_cast<float4*>(w.prob[0] + p) = make_float4(pr[0][0], pr[0][1], pr[0][2], pr[0][3]);
            *reinterpret_cast<float4*>(w.prob[1] + p) = make_float4(pr[1][0], pr[1][1], pr[1][2], pr[1][3]);
            *reinterpret_cast<unsigned*>(w.lab8 + p) = lab;
        }
    }
    const float fv = wave_sum((float)nvalid), fc = wave_sum((float)ncorr);      // exact: < 2^24 per wave
    if (lane_id() == 0) {
        atomicAdd(&s_cnt[0], (unsigned)fv);
        atomicAdd(&s_cnt[1], (unsigned)fc);
    }
    __syncthreads();
    const int rep = blockIdx.x % O2_REP;
    for (int i = threadIdx.x; i < 2 * O2_BINS; i += blockDim.x) {
        const unsigned v = (&s_hist[0][0])[i];
        if (v) atomicAdd(&w.hist[(((i / O2_BINS) * 3 + 0) * O2_REP + rep) * O2_BINS + i % O2_BINS], v);
    }
    if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(&w.cnt[rep * 2 + threadIdx.x], s_cnt[threadIdx.x]);
}

// one workgroup per loss: locate the bucket holding the wanted rank at this level (as ohem_scan_kernel)
__global__ void __launch_bounds__(256) ohem2_scan_kernel(float* work, long P, int pass, long long min_kept0,
                                                         long long min_kept1, float thres0, float thres1) {
    __shared__ unsigned s_wave[4];
    const Ohem2Work wk = o2_work(work, P);
    const int L = blockIdx.x;
    unsigned* st = wk.state + 16 * L;
    const long long min_kept = L ? min_kept1 : min_kept0;
    const float thres = L ? thres1 : thres0;
    unsigned nv = st[0];
    if (pass == 0) {          // the counters of the probability pass, summed over their replicas (every thread: uniform)
        unsigned c0 = 0u, c1 = 0u;
        for (int r = 0; r < O2_REP; ++r) {
            c0 += wk.cnt[r * 2];
            c1 += wk.cnt[r * 2 + 1];
        }
        nv = c0;
        __syncthreads();
        if (threadIdx.x == 0) {
            st[0] = c0;
            st[1] = c1;
        }
    }
    if (nv == 0u) {   // workgroup-uniform
        if (pass == 0 && threadIdx.x == 0) {
            st[4] = __float_as_uint(thres);
            st[2] = 0u;
            st[3] = 0u;
        }
        return;
    }
    unsigned rank;
    if (pass == 0) {
        const long long k = min_kept < (long long)nv - 1 ? min_kept : (long long)nv - 1;
        rank = (unsigned)k;
    } else {
        rank = st[2];
    }
    const unsigned prefix_bits = pass == 0 ? 0u : st[3];
    const unsigned* h = wk.hist + (long)(L * 3 + pass) * O2_REP * O2_BINS;
    const int nb = pass == 2 ? 256 : 2048;
    const int per = nb / 256;
    unsigned hb[8], mine = 0u;
    if (pass == 2) {                     // one bin per thread
        unsigned v[O2_REP];
#pragma unroll
        for (int r = 0; r < O2_REP; ++r) v[r] = h[r * O2_BINS + threadIdx.x];
        hb[0] = 0u;
#pragma unroll
        for (int r = 0; r < O2_REP; ++r) hb[0] += v[r];
#pragma unroll
        for (int i = 1; i < 8; ++i) hb[i] = 0u;
        mine = hb[0];
    } else {                             // eight consecutive bins per thread: two 16-byte loads per replica, all in flight
        uint4 v[O2_REP][2];
#pragma unroll
        for (int r = 0; r < O2_REP; ++r) {
            v[r][0] = *reinterpret_cast<const uint4*>(h + r * O2_BINS + threadIdx.x * 8);
            v[r][1] = *reinterpret_cast<const uint4*>(h + r * O2_BINS + threadIdx.x * 8 + 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) hb[i] = 0u;
#pragma unroll
        for (int r = 0; r < O2_REP; ++r) {
            hb[0] += v[r][0].x; hb[1] += v[r][0].y; hb[2] += v[r][0].z; hb[3] += v[r][0].w;
            hb[4] += v[r][1].x; hb[5] += v[r][1].y; hb[6] += v[r][1].z; hb[7] += v[r][1].w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) mine += hb[i];
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
        const unsigned bits = pass == 0 ? ((unsigned)b << O2_SH0) : (pass == 1 ? ((unsigned)b << O2_SH1) : (unsigned)b);
        const unsigned full = prefix_bits | bits;
        st[3] = full;
        st[2] = r;
        if (pass == 2) {
            const float kth = __uint_as_float(full);
            st[4] = __float_as_uint(kth > thres ? kth : thres);   // threshold = max(min_value, thresh)
        }
    }
}

// histogram of the next radix level among the probabilities that share the prefix found so far; both losses
__global__ void __launch_bounds__(256) ohem2_hist_kernel(float* work, long P, int pass) {
    __shared__ unsigned s_hist[2][O2_BINS];
    const Ohem2Work w = o2_work(work, P);
    for (int i = threadIdx.x; i < 2 * O2_BINS; i += blockDim.x) (&s_hist[0][0])[i] = 0u;
    __syncthreads();
    const unsigned prefix[2] = {w.state[3], w.state[16 + 3]};
    const unsigned mask = pass == 1 ? O2_MASK1 : O2_MASK2;
    const long nq = P / 4, stride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(w.prob[j] + q * 4);
            const unsigned u[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((u[k] & mask) == prefix[j]) atomicAdd(&s_hist[j][o2_bin(u[k], pass)], 1u);
        }
    }
    __syncthreads();
    const int rep = blockIdx.x % O2_REP;
    for (int i = threadIdx.x; i < 2 * O2_BINS; i += blockDim.x) {
        const unsigned v = (&s_hist[0][0])[i];
        if (v) atomicAdd(&w.hist[(((i / O2_BINS) * 3 + pass) * O2_REP + rep) * O2_BINS + i % O2_BINS], v);
    }
}

// masked mean: sum of -log p over p < threshold, and the count, for both losses
__global__ void __launch_bounds__(256) ohem2_reduce_kernel(float* work, long P) {
    __shared__ float s_sum[2][4];
    __shared__ unsigned s_cnt[2][4];
    const Ohem2Work w = o2_work(work, P);
    const float thr[2] = {__uint_as_float(w.state[4]), __uint_as_float(w.state[16 + 4])};
    float sum[2] = {0.f, 0.f};
    unsigned cnt[2] = {0u, 0u};
    const long nq = P / 4, stride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(w.prob[j] + q * 4);
            const float pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (pv[k] < thr[j]) {
                    sum[j] -= __logf(pv[k]);
                    ++cnt[j];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float s = wave_sum(sum[j]), c = wave_sum((float)cnt[j]);
        if ((threadIdx.x & 63) == 0) {
            s_sum[j][threadIdx.x >> 6] = s;
            s_cnt[j][threadIdx.x >> 6] = (unsigned)c;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) {          // one partial per workgroup (summed in a fixed order by the final kernel)
        const int j = threadIdx.x;
        w.part[blockIdx.x * 4 + j] = s_sum[j][0] + s_sum[j][1] + s_sum[j][2] + s_sum[j][3];
        w.part[blockIdx.x * 4 + 2 + j] = (float)(s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3]);   // exact < 2^24
    }
}

// out[2][4]: loss, accuracy (of output 0), threshold, n_selected.  One workgroup: the per-workgroup partials of the
// masked mean are summed in a fixed order (deterministic), counts exactly (each < 2^24, the total in integers).
__global__ void __launch_bounds__(256) ohem2_final_kernel(float* work, long P, int nparts, float lw0, float lw1, float* out) {
    __shared__ float s_sum[2][256];
    __shared__ unsigned s_cnt[2][256];
    const Ohem2Work w = o2_work(work, P);
    float sum[2] = {0.f, 0.f};
    unsigned cnt[2] = {0u, 0u};
    for (int b = threadIdx.x; b < nparts; b += 256) {
        sum[0] += w.part[b * 4];
        sum[1] += w.part[b * 4 + 1];
        cnt[0] += (unsigned)w.part[b * 4 + 2];
        cnt[1] += (unsigned)w.part[b * 4 + 3];
    }
    for (int j = 0; j < 2; ++j) {
        s_sum[j][threadIdx.x] = sum[j];
        s_cnt[j][threadIdx.x] = cnt[j];
    }
    __syncthreads();
    const int j = threadIdx.x;
    if (j >= 2) return;
    float tot = 0.f;
    unsigned nsel = 0u;
    for (int t = 0; t < 256; ++t) {
        tot += s_sum[j][t];
        nsel += s_cnt[j][t];
    }
    unsigned* st = w.state + 16 * j;
    st[5] = nsel;
    st[6] = __float_as_uint(tot);
    const unsigned nv = st[0];
    out[4 * j + 0] = nv == 0u ? 0.f : (j ? lw1 : lw0) * (tot / (float)nsel);   // 0/0 -> NaN like the reference
    const float eps = 1.1920929e-07f;
    out[4 * j + 1] = ((float)w.state[1] + eps) * (100.0f / ((float)nv + eps));
    out[4 * j + 2] = __uint_as_float(st[4]);
    out[4 * j + 3] = (float)nsel;
}

int ohem2_up_fwd_impl(const float* src0, const float* src1, int N, int Hs, int Ws, int H, int W,
                      const long long* target, float thres0, long long min_kept0, float lw0, float thres1,
                      long long min_kept1, float lw1, int ignore_label, float* work, float* out, hipStream_t s) {
    LEDN_REQUIRE(src0 && src1 && target && work && out && N > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0);
    LEDN_REQUIRE(W % 4 == 0 && ignore_label >= 0 && ignore_label <= 255);
    const long long P = (long long)N * H * W;
    LEDN_REQUIRE(P < (1LL << 31) && min_kept0 >= 1 && min_kept1 >= 1);
    const Ohem2Work w = o2_work(work, P);
    if (hipMemsetAsync(w.hist, 0, sizeof(unsigned) * O2_CLEAR_WORDS, s) != hipSuccess) return LEDN_ELAUNCH;
    if (hipMemsetAsync(w.state, 0, sizeof(unsigned) * 32, s) != hipSuccess) return LEDN_ELAUNCH;
    const long nq = P / 4;
    const dim3 grid((unsigned)(cdiv(nq, 256) < O2_GRID ? cdiv(nq, 256) : O2_GRID));
    const long rows = (long)N * H;
    const dim3 grid_rows((unsigned)(rows < O2_GRID ? rows : O2_GRID));
    LEDN_LAUNCH(ohem2_prob_kernel, grid_rows, dim3(256), 0, s, src0, src1, N, Hs, Ws, H, W, target, ignore_label, work);
    LEDN_LAUNCH(ohem2_scan_kernel, dim3(2), dim3(256), 0, s, work, (long)P, 0, min_kept0, min_kept1, thres0, thres1);
    LEDN_LAUNCH(ohem2_hist_kernel, grid, dim3(256), 0, s, work, (long)P, 1);
    LEDN_LAUNCH(ohem2_scan_kernel, dim3(2), dim3(256), 0, s, work, (long)P, 1, min_kept0, min_kept1, thres0, thres1);
    LEDN_LAUNCH(ohem2_hist_kernel, grid, dim3(256), 0, s, work, (long)P, 2);
    LEDN_LAUNCH(ohem2_scan_kernel, dim3(2), dim3(256), 0, s, work, (long)P, 2, min_kept0, min_kept1, thres0, thres1);
    LEDN_LAUNCH(ohem2_reduce_kernel, grid, dim3(256), 0, s, work, (long)P);
    LEDN_LAUNCH(ohem2_final_kernel, dim3(1), dim3(256), 0, s, work, (long)P, (int)grid.x, lw0, lw1, out);
    return check_launch();
}

// Backward (exact 2x): a workgroup owns a 16 x 16 tile of the half-resolution maps; the softmax gradients of its
// 34 x 34 children are formed once in LDS for BOTH losses (labels: the uint8 plane; selection: stored probability
// < threshold; the gradient itself from the stored probability), then every source pixel gathers its 4 x 4 children
// with the interpolation weights.  src0 / src1 are not read.
__global__ void __launch_bounds__(512) ohem2_bwd_up2_kernel(const float* src0, const float* src1, int N, int Hs, int Ws,
                                                            int ignore_label, const float* work, const float* out,
                                                            const float* dloss0, const float* dloss1, float lw0,
                                                            float lw1, float* dsrc0, float* dsrc1) {
    // 8 x 64 source pixels per workgroup: the 18 x 130 children are read as 520-byte row segments (the first version's
    // 16 x 16 tile read 136-byte segments at a 4 KB stride: two cache lines for 34 floats, 99 us at 2.2 TB/s)
    constexpr int TH = 8, TW = 64, CHH = 2 * TH + 2, CHW = 2 * TW + 2;
    __shared__ float4 s_g[CHH * CHW];                         // (g0.x, g0.y, g1.x, g1.y) of one child
    const int H = 2 * Hs, W = 2 * Ws;
    const long P = (long)N * H * W;
    const Ohem2Work w = o2_work(const_cast<float*>(work), P);
    const int tw = (Ws + TW - 1) / TW, th = (Hs + TH - 1) / TH;
    const int bj = blockIdx.x % tw, bi = (blockIdx.x / tw) % th, n = blockIdx.x / (tw * th);
    const int i0 = bi * TH, j0 = bj * TW;
    const float thr0 = out[2], thr1 = out[4 + 2];
    const float coef0 = dloss0[0] * lw0 / out[3], coef1 = dloss1[0] * lw1 / out[4 + 3];
    (void)src0; (void)src1;
    const float sy = (float)Hs / (float)H, sx = (float)Ws / (float)W;
    for (int k = threadIdx.x; k < CHH * CHW; k += 512) {
        const int y = 2 * i0 - 1 + k / CHW, x = 2 * j0 - 1 + k % CHW;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const long p = ((long)n * H + y) * W + x;
            const int tg = w.lab8[p];
            if (tg != ignore_label) {
                // two classes: softmax = (p_t, 1 - p_t) with p_t the STORED probability of the target class, so
                // softmax - onehot = -(1 - p_t) at the target class and +(1 - p_t) at the other: no logits, no
                // interpolation, no exponentials in the backward (the first version re-formed them: 143 us, VALU-bound)
                const float p0 = w.prob[0][p], p1 = w.prob[1][p];
                if (p0 < thr0) {
                    const float d = coef0 * (1.f - p0);
                    g.x = tg == 0 ? -d : d;
                    g.y = tg == 1 ? -d : d;
                }
                if (p1 < thr1) {
                    const float d = coef1 * (1.f - p1);
                    g.z = tg == 0 ? -d : d;
                    g.w = tg == 1 ? -d : d;
                }
            }
        }
        s_g[k] = g;
    }
    __syncthreads();
    const int a = threadIdx.x / TW, b = threadIdx.x % TW, i = i0 + a, j = j0 + b;
    if (i >= Hs || j >= Ws) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float wys[4], wxs[4];         // the interpolation weight of child (dy, dx) onto this source pixel = wys[dy] * wxs[dx]
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int y = 2 * i - 1 + d, x = 2 * j - 1 + d;
        wys[d] = wxs[d] = 0.f;
        if (y >= 0 && y < H) {
            const Lerp ly = o2_lerp(y, Hs, sy);
            wys[d] = (ly.i0 == i ? ly.w0 : 0.f) + (ly.i1 == i ? ly.w1 : 0.f);
        }
        if (x >= 0 && x < W) {
            const Lerp lx = o2_lerp(x, Ws, sx);
            wxs[d] = (lx.i0 == j ? lx.w0 : 0.f) + (lx.i1 == j ? lx.w1 : 0.f);
        }
    }
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const float wgt = wys[dy] * wxs[dx];
            const float4 g = s_g[(2 * a + dy) * CHW + 2 * b + dx];
            acc.x += wgt * g.x;
            acc.y += wgt * g.y;
            acc.z += wgt * g.z;
            acc.w += wgt * g.w;
        }
    }
    const long o = (((long)n * Hs + i) * Ws + j) * 2;
    *reinterpret_cast<float2*>(dsrc0 + o) = make_float2(acc.x, acc.y);
    *reinterpret_cast<float2*>(dsrc1 + o) = make_float2(acc.z, acc.w);
}

int ohem2_up_bwd_impl(const float* src0, const float* src1, int N, int Hs, int Ws, int H, int W, int ignore_label,
                      const float* work, const float* out, const float* dloss0, const float* dloss1, float lw0,
                      float lw1, float* dsrc0, float* dsrc1, hipStream_t s) {
    LEDN_REQUIRE(src0 && src1 && work && out && dloss0 && dloss1 && dsrc0 && dsrc1 && N > 0 && Hs > 0 && Ws > 0);
    LEDN_REQUIRE(H == 2 * Hs && W == 2 * Ws);               // the fused adjoint is written for the exact 2x resize
    const long nb = (long)N * cdiv(Hs, 8) * cdiv(Ws, 64);
    LEDN_REQUIRE(nb < (1L << 31));
    LEDN_LAUNCH(ohem2_bwd_up2_kernel, dim3((unsigned)nb), dim3(512), 0, s, src0, src1, N, Hs, Ws, ignore_label, work,
                out, dloss0, dloss1, lw0, lw1, dsrc0, dsrc1);
    return check_launch();
}

}  // namespace ledn
