// conv_mfma.hip -- implicit-GEMM convolution on the CDNA4 matrix cores (gfx950).
//
// bf16 activations / bf16 packed weights / f32 accumulation, v_mfma_f32_32x32x16_bf16.
//
// Forward + data-gradient kernel (conv_mfma_kernel)
//   GEMM view: M = output pixels, N = Cout, K = taps x Cin.
//   Workgroup = 4 wavefronts = an output patch of TR rows x 32 columns (one MFMA M-tile
//   = 32 consecutive pixels of a row) x WN*32 output channels.  Wave (wm, wn) owns MT
//   rows and one 32-channel N-tile: MT accumulator tiles (16 f32 VGPRs each).
//   The NHWC input patch (with halo) of one 32-channel K-chunk is staged ONCE per chunk
//   in LDS as [patch pixel][32 ch + 8 pad] bf16: the 80-byte pixel stride makes the
//   A-fragment read (16 B of 8 consecutive channels per lane, lanes = 32 pixels of a
//   row) hit 16 distinct 16-B slots per 16-lane group -> conflict-free ds_read_b128.
//   Every tap re-reads the same LDS patch (im2col never materialised); the B fragment
//   (8 consecutive ci of one cout) is a 16-B global load from the [tap][co][ci] bf16
//   weight pack (L1/L2 resident) reused for the wave's MT tiles.
//   BatchNorm/activation of the producer can be applied while staging (prologue),
//   BN-affine / residual / activation / per-channel statistics in the epilogue.
//   Data gradient = the same kernel: stride-1 convs use the flipped/transposed weight
//   pack; stride-2 convs read dz as a zero-upsampled patch (UP = 2).
//
// Weight-gradient kernel (conv_wgrad_mfma_kernel)
//   GEMM view: M = Cout tile (32), N = Cin tile (32), K = output pixels.
//   dz and the x patch are staged pixel-major exactly as above; the K-contiguous
//   fragments (8 consecutive pixels of one channel) come out of the SAME layout through
//   ds_read_b64_tr_b16 (hardware transpose read), so tap shifts are whole-pixel address
//   offsets (always aligned).  Each wave accumulates 9 taps x [32 x 32] in registers over
//   its share of the pixels; partial tiles are reduced across the 4 waves through LDS and
//   added to dW with one f32 atomic per element per workgroup.
#include "ledn_rt.h"

namespace ledn {

constexpr int CK = 32;            // channels per K-chunk
constexpr int PIXB = CK * 2 + 16; // bytes per staged pixel (80)

struct MfmaConvArgs {
    const bf16_t* x;
    const bf16_t* wp;
    bf16_t* y;
    const bf16_t* res;
    const float* in_scale;
    const float* in_shift;
    const float* out_scale;
    const float* out_shift;
    const float* slope;
    float* stat_sum;
    float* stat_sqsum;
    int N, H, W, Cin, Ho, Wo, Cout;
    int pad, in_act, act_out, res_mode;
    int tiles_h, tiles_w;
    float* part;      // optional workspace for the statistics: [gridDim.x * WM][2][Cout]
};

// stage one 16-byte (8-channel) piece of a patch pixel into LDS, prologue applied
__device__ __forceinline__ void stage_piece(unsigned char* dst, const bf16_t* src, bool valid,
                                            const float* in_scale, const float* in_shift, int in_act,
                                            int c) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (valid) {
        v = *reinterpret_cast<const uint4*>(src);
        if (in_scale || in_act) {
            unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
                if (in_scale) {
                    lo = lo * in_scale[c + 2 * i] + in_shift[c + 2 * i];
                    hi = hi * in_scale[c + 2 * i + 1] + in_shift[c + 2 * i + 1];
                }
                if (in_act == LEDN_ACT_RELU) {
                    lo = fmaxf(lo, 0.f);
                    hi = fmaxf(hi, 0.f);
                }
                w[i] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    *reinterpret_cast<uint4*>(dst) = v;
}

template <int WM, int WN, int MT, int K, int S, int UP>
__global__ void __launch_bounds__(256) conv_mfma_kernel(MfmaConvArgs a) {
    constexpr int TR = WM * MT;
    constexpr int PR = (TR - 1) * S + K, PC = 31 * S + K;
    __shared__ __attribute__((aligned(16))) unsigned char s_patch[PR * PC * PIXB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int lr = lane & 31, lh = lane >> 5;
    int b = blockIdx.x;
    const int tw = b % a.tiles_w; b /= a.tiles_w;
    const int th = b % a.tiles_h;
    const int n = b / a.tiles_h;
    const int ho0 = th * TR, wo0 = tw * 32;
    const int co = blockIdx.y * (WN * 32) + wn * 32 + lr;   // this lane's output channel (B column)
    const bool co_ok = co < a.Cout;                         // Cout % 32 == 16 tail (grouped 1x1 reduce convs)

    f32x16_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;

    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        // ---- stage the input patch of this K-chunk (4 threads x 16 B per pixel)
        for (int p = tid >> 2; p < PR * PC; p += 64) {
            const int pr = p / PC, pc = p % PC;
            const int part = tid & 3;
            const int uh = ho0 * S - a.pad + pr, uw = wo0 * S - a.pad + pc;
            bool valid;
            int hi, wi;
            if (UP == 1) {
                hi = uh; wi = uw;
                valid = uh >= 0 && uh < a.H && uw >= 0 && uw < a.W;
            } else {
                hi = uh / UP; wi = uw / UP;
                valid = uh >= 0 && uw >= 0 && (uh % UP) == 0 && (uw % UP) == 0 && hi < a.H && wi < a.W;
            }
            const int c = c0 + part * 8;
            stage_piece(s_patch + (long)p * PIXB + part * 16,
                        a.x + (((long)n * a.H + hi) * a.W + wi) * a.Cin + c, valid && c < a.Cin, a.in_scale,
                        a.in_shift, a.in_act, c);
        }
        __syncthreads();
        // ---- taps x k-steps: one B fragment (global, L1/L2) feeds MT MFMAs
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const bf16_t* wrow = a.wp + ((long)(kh * K + kw) * a.Cout + co) * a.Cin + c0 + lh * 8;
#pragma unroll
                for (int kk = 0; kk < CK / 16; ++kk) {
                    bf16x8_t bf = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (co_ok && c0 + kk * 16 + lh * 8 < a.Cin) bf = *reinterpret_cast<const bf16x8_t*>(wrow + kk * 16);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int row = wm * MT + m;
                        const unsigned char* ap = s_patch + (long)((row * S + kh) * PC + lr * S + kw) * PIXB +
                                                  (kk * 16 + lh * 8) * 2;
                        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(ap);
                        acc[m] = mfma_32x32x16_bf16(af, bf, acc[m]);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds channel `co` for 16 pixels of each of its MT rows
    if (!co_ok) return;    // (no wave collective below needs the masked lanes: shfl partner lane^32 has the same co)
    const float sc = a.out_scale ? a.out_scale[co] : 1.f;
    const float sh = a.out_shift ? a.out_shift[co] : 0.f;
    const float sl = a.slope ? a.slope[co] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ho = ho0 + wm * MT + m;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int wo = wo0 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (ho >= a.Ho || wo >= a.Wo) continue;
            float v = acc[m][i] * sc + sh;
            s1 += v;
            s2 = fmaf(v, v, s2);
            const long off = (((long)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co;
            if (a.res_mode != LEDN_RES_NONE) {
                const float r = ld(a.res + off);
                v = a.res_mode == LEDN_RES_ADD ? v + r : v * r + r;
            }
            v = act_apply(a.act_out, v, sl);
            st(a.y + off, v);
        }
    }
    if (a.stat_sum) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lh == 0) {
            if (a.part) {
                float* row = a.part + ((long)blockIdx.x * WM + wm) * 2 * a.Cout;
                row[co] = s1;
                row[a.Cout + co] = s2;
            } else {
                atomicAdd(a.stat_sum + co, s1);
                atomicAdd(a.stat_sqsum + co, s2);
            }
        }
    }
}

template <int WM, int WN, int MT, int K, int S, int UP>
static int launch_cfg(const MfmaConvArgs& a0, hipStream_t s) {
    MfmaConvArgs a = a0;
    constexpr int TR = WM * MT;
    a.tiles_h = (int)cdiv(a.Ho, TR);
    a.tiles_w = (int)cdiv(a.Wo, 32);
    const long nbx = (long)a.N * a.tiles_h * a.tiles_w;
    const dim3 grid((unsigned)nbx, (unsigned)cdiv(a.Cout, WN * 32));
    a.part = (a.stat_sum && nbx * WM > 16) ? ws_take(nbx * WM * 2 * a.Cout) : nullptr;
    LEDN_LAUNCH((conv_mfma_kernel<WM, WN, MT, K, S, UP>), grid, dim3(256), 0, s, a);
    if (a.part) return finish_partials(a.part, (int)(nbx * WM), a.Cout, 2, a.stat_sum, a.stat_sqsum, nullptr, s);
    return check_launch();
}

template <int K, int S, int UP>
static int launch_shape(const MfmaConvArgs& a, hipStream_t s) {
    constexpr int MTS = S == 2 ? 1 : 4;   // stride 2 needs a (2*TR+1) x 65 pixel patch: keep TR = 4
    if (a.Cout % 128 == 0) return launch_cfg<1, 4, 4, K, S, UP>(a, s);
    if (a.Cout % 64 == 0) return launch_cfg<2, 2, S == 2 ? 2 : 4, K, S, UP>(a, s);
    return launch_cfg<4, 1, MTS, K, S, UP>(a, s);
}

bool conv_mfma_supported(const ledn_conv_desc& d) {
    if (!d.w_bf16 || d.dtype_x != LEDN_BF16 || d.dtype_y != LEDN_BF16) return false;
    if (d.dil != 1 || d.xadd) return false;
    if (d.groups != 1 && !(d.KH == 1 && d.KW == 1)) return false;   // grouped 1x1: densified weight pack
    if (d.Cin % 16 || (d.Cout % 16 && d.Cout > 8)) return false;   // 16-channel tails are masked
    if (d.KH != d.KW || !((d.KH == 3 && d.pad == 1) || (d.KH == 1 && d.pad == 0))) return false;
    if (d.stride != 1 && d.stride != 2) return false;
    if (d.transposed && d.stride == 2 && d.KH == 1) return false;
    return true;
}

int conv_mfma(const ledn_conv_desc& d, hipStream_t s) {
    MfmaConvArgs a;
    a.x = (const bf16_t*)d.x; a.wp = (const bf16_t*)d.w_bf16; a.y = (bf16_t*)d.y; a.res = (const bf16_t*)d.res;
    a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.out_scale = d.out_scale; a.out_shift = d.out_shift;
    a.slope = d.slope; a.stat_sum = d.stat_sum; a.stat_sqsum = d.stat_sqsum;
    a.N = d.N; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout;
    a.in_act = d.in_act; a.act_out = d.act_out; a.res_mode = d.res_mode;
    a.tiles_h = a.tiles_w = 0;
    a.part = nullptr;
    if (!d.transposed) {
        a.pad = d.pad;
        if (d.KH == 3) return d.stride == 1 ? launch_shape<3, 1, 1>(a, s) : launch_shape<3, 2, 1>(a, s);
        return d.stride == 1 ? launch_shape<1, 1, 1>(a, s) : launch_shape<1, 2, 1>(a, s);
    }
    a.pad = d.KH - 1 - d.pad;   // stride-1 correlation over the (zero-upsampled) dz with flipped taps
    if (d.KH == 3) return d.stride == 1 ? launch_shape<3, 1, 1>(a, s) : launch_shape<3, 1, 2>(a, s);
    return launch_shape<1, 1, 1>(a, s);
}

// ---------------------------------------------------------------------------
// weight pack (f32 OIHW -> bf16 [tap][co][ci], or the dgrad variant)
// ---------------------------------------------------------------------------
// w: [Cout][Cin/groups][KK] f32; the pack is DENSE over the full Cin (zeros outside the group)
__global__ void pack_weights_kernel(const float* w, bf16_t* out, int Cout, int Cin, int KK, int mode,
                                    int groups) {
    const long total = (long)Cout * Cin * KK;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int tap = (int)(i % KK);
    const int ci = (int)((i / KK) % Cin);
    const int co = (int)(i / ((long)KK * Cin));
    const int cig = Cin / groups, cog = Cout / groups;
    const int g = co / cog;
    float v = 0.f;
    if (ci / cig == g) v = w[((long)co * cig + (ci - g * cig)) * KK + tap];
    long o;
    if (mode == 0) o = ((long)tap * Cout + co) * Cin + ci;
    else o = ((long)(KK - 1 - tap) * Cin + ci) * Cout + co;
    st(out + o, v);
}

// table-driven variant: every packed weight of the model in ONE launch (grid.y = tensor)
__global__ void __launch_bounds__(256) pack_weights_multi_kernel(const ledn_pack_entry* table) {
    const ledn_pack_entry e = table[blockIdx.y];
    const long total = (long)e.Cout * e.Cin * e.KK;
    const int cig = e.Cin / e.groups, cog = e.Cout / e.groups;
    bf16_t* out = reinterpret_cast<bf16_t*>(e.out);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % e.KK);
        const int ci = (int)((i / e.KK) % e.Cin);
        const int co = (int)(i / ((long)e.KK * e.Cin));
        const int g = co / cog;
        float v = 0.f;
        if (ci / cig == g) v = e.w[((long)co * cig + (ci - g * cig)) * e.KK + tap];
        const long o = e.mode == 0 ? ((long)tap * e.Cout + co) * e.Cin + ci
                                   : ((long)(e.KK - 1 - tap) * e.Cin + ci) * e.Cout + co;
        st(out + o, v);
    }
}

int pack_conv_weights_multi_impl(const ledn_pack_entry* table_dev, int n, long long max_elems, hipStream_t s) {
    LEDN_REQUIRE(table_dev && n > 0 && max_elems > 0);
    long chunks = cdiv(max_elems, 256 * 4);
    if (chunks > 32) chunks = 32;
    LEDN_LAUNCH(pack_weights_multi_kernel, dim3((unsigned)chunks, (unsigned)n), dim3(256), 0, s, table_dev);
    return check_launch();
}

int pack_conv_weights_impl(const float* w, void* out, int Cout, int Cin, int KH, int KW, int mode,
                           int groups, hipStream_t s) {
    LEDN_REQUIRE(w && out && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && (mode == 0 || mode == 1));
    LEDN_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0);
    const long total = (long)Cout * Cin * KH * KW;
    LEDN_LAUNCH(pack_weights_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, (bf16_t*)out, Cout,
                Cin, KH * KW, mode, groups);
    return check_launch();
}

// ---------------------------------------------------------------------------
// im2col of the 3-channel stem (3x3, stride 2, pad 1): p[n,ho,wo, (kh*3+kw)*C + c] (27 of 32
// columns used, the rest zero) so that the stem runs as a K=32 1x1 GEMM on the matrix cores
// (forward and weight gradient); the input needs no gradient.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) im2col_stem_kernel(const bf16_t* x, bf16_t* p, int N, int H, int W, int C,
                                                          int Ho, int Wo) {
    const long total = (long)N * Ho * Wo;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total) return;
    const int wo = (int)(pix % Wo);
    const int ho = (int)((pix / Wo) % Ho);
    const int n = (int)(pix / ((long)Wo * Ho));
    unsigned short e[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) e[i] = 0;
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - 1 + kh;
        if (hi < 0 || hi >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - 1 + kw;
            if (wi < 0 || wi >= W) continue;
            const bf16_t* src = x + (((long)n * H + hi) * W + wi) * C;
            for (int c = 0; c < C; ++c) e[(kh * 3 + kw) * C + c] = src[c].v;
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(p + pix * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        dst[q] = make_uint4(e[8 * q] | ((unsigned)e[8 * q + 1] << 16), e[8 * q + 2] | ((unsigned)e[8 * q + 3] << 16),
                            e[8 * q + 4] | ((unsigned)e[8 * q + 5] << 16), e[8 * q + 6] | ((unsigned)e[8 * q + 7] << 16));
}

int im2col_stem_impl(const void* x, void* p, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    LEDN_REQUIRE(x && p && N > 0 && H > 0 && W > 0 && C > 0 && 9 * C <= 32);
    LEDN_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1);
    LEDN_LAUNCH(im2col_stem_kernel, dim3((unsigned)cdiv((long)N * Ho * Wo, 256)), dim3(256), 0, s,
                (const bf16_t*)x, (bf16_t*)p, N, H, W, C, Ho, Wo);
    return check_launch();
}

// ---------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------
struct MfmaWgradArgs {
    const bf16_t* x;
    const bf16_t* dz;
    float* dw;
    const float* in_scale;
    const float* in_shift;
    long long ws_co, ws_ci, ws_tap;
    int N, H, W, Cin, Ho, Wo, Cout;
    int pad, in_act, groups;
    int tiles_h, tiles_w, tiles_per_block, ci_tiles;
    float* part;      // optional workspace: [gridDim.x][gridDim.y][KK*1024] per-workgroup partial tiles
};

template <int K, int S>
__global__ void __launch_bounds__(256) conv_wgrad_mfma_kernel(MfmaWgradArgs a) {
    constexpr int TR = S == 2 ? 4 : 8;             // output rows per staged tile (TR x 32 pixels = 2*TR k-steps)
    constexpr int PR = (TR - 1) * S + K, PC = 31 * S + K;
    constexpr int XB = PR * PC * PIXB, ZB = TR * 32 * PIXB;
    constexpr int KK = K * K;
    constexpr int RED = KK * 32 * 32 * 4;          // bytes of one wave's partial tiles
    constexpr int LDSB = (XB + ZB) > RED ? (XB + ZB) : RED;
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[LDSB];
    unsigned char* s_x = s_mem;
    unsigned char* s_z = s_mem + XB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g16 = lane >> 4, l16 = lane & 15;
    const int colblk = g16 & 1, lh = g16 >> 1;     // channel half (16) and k half (8 pixels)
    const int q = l16 >> 2, pp = l16 & 3;          // tr-read: this lane supplies row q, columns 4pp..4pp+3
    const int ci_tile = blockIdx.y % a.ci_tiles, co_tile = blockIdx.y / a.ci_tiles;
    const int ci0 = ci_tile * 32, co0 = co_tile * 32;
    const int cig = a.Cin / a.groups, cog = a.Cout / a.groups;
    if (a.groups > 1) {   // grouped conv: only tile pairs that touch the block diagonal (workgroup-uniform exit)
        const int g_lo = co0 / cog, g_hi = min(co0 + 31, a.Cout - 1) / cog;
        if (ci0 + 31 < g_lo * cig || ci0 >= (g_hi + 1) * cig) return;
    }

    f32x16_t acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const long ntiles = (long)a.N * a.tiles_h * a.tiles_w;
    const long t0 = (long)blockIdx.x * a.tiles_per_block;
    const long t1 = min(ntiles, t0 + a.tiles_per_block);
    for (long tile = t0; tile < t1; ++tile) {
        long b = tile;
        const int tw = (int)(b % a.tiles_w); b /= a.tiles_w;
        const int th = (int)(b % a.tiles_h);
        const int n = (int)(b / a.tiles_h);
        const int ho0 = th * TR, wo0 = tw * 32;
        // ---- stage x patch (32 input channels) and dz tile (32 output channels); zeros outside
        for (int p = tid >> 2; p < PR * PC; p += 64) {
            const int pr = p / PC, pc = p % PC, part = tid & 3;
            const int hi = ho0 * S - a.pad + pr, wi = wo0 * S - a.pad + pc;
            const bool valid = hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            const int c = ci0 + part * 8;
            stage_piece(s_x + (long)p * PIXB + part * 16, a.x + (((long)n * a.H + hi) * a.W + wi) * a.Cin + c,
                        valid && c < a.Cin, a.in_scale, a.in_shift, a.in_act, c);
        }
        for (int p = tid >> 2; p < TR * 32; p += 64) {
            const int pr = p / 32, pc = p % 32, part = tid & 3;
            const int ho = ho0 + pr, wo = wo0 + pc;
            const bf16_t* zsrc = a.dz + (((long)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co0 + part * 8;
            if (a.Cout % 8 == 0) {
                const bool valid = ho < a.Ho && wo < a.Wo && co0 + part * 8 < a.Cout;
                stage_piece(s_z + (long)p * PIXB + part * 16, zsrc, valid, nullptr, nullptr, 0, 0);
            } else {   // narrow heads (Cout = 1, 2, 4): element-wise staging, zero-filled to 8 channels
                unsigned short e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    e[j] = (ho < a.Ho && wo < a.Wo && co0 + part * 8 + j < a.Cout) ? zsrc[j].v : (unsigned short)0;
                *reinterpret_cast<uint4*>(s_z + (long)p * PIXB + part * 16) =
                    make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16),
                               e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16));
            }
        }
        __syncthreads();
        // ---- 2*TR k-steps of 16 pixels (half a row each); wave w takes k-steps w, w+4, ...
#pragma unroll 1
        for (int ks = wid; ks < TR * 2; ks += 4) {
            const int row = ks >> 1, cb = (ks & 1) * 16;       // pixels (row, cb + 0..15)
            // A = dz^T: lane needs 8 pixels cb+8*lh+0..7 of channel colblk*16+l16
            const unsigned char* zp = s_z + (long)(row * 32 + cb + lh * 8 + q) * PIXB + (colblk * 16 + pp * 4) * 2;
            const bf16x4_t a_lo = lds_read_tr16(zp);
            const bf16x4_t a_hi = lds_read_tr16(zp + 4 * PIXB);
            bf16x8_t af;
            af[0] = a_lo[0]; af[1] = a_lo[1]; af[2] = a_lo[2]; af[3] = a_lo[3];
            af[4] = a_hi[0]; af[5] = a_hi[1]; af[6] = a_hi[2]; af[7] = a_hi[3];
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const unsigned char* xp = s_x + (long)((row * S + kh) * PC + (cb + lh * 8 + q) * S + kw) * PIXB +
                                              (colblk * 16 + pp * 4) * 2;
                    const bf16x4_t b_lo = lds_read_tr16(xp);
                    const bf16x4_t b_hi = lds_read_tr16(xp + 4 * S * PIXB);
                    bf16x8_t bfv;
                    bfv[0] = b_lo[0]; bfv[1] = b_lo[1]; bfv[2] = b_lo[2]; bfv[3] = b_lo[3];
                    bfv[4] = b_hi[0]; bfv[5] = b_hi[1]; bfv[6] = b_hi[2]; bfv[7] = b_hi[3];
                    acc[kh * K + kw] = mfma_32x32x16_bf16(af, bfv, acc[kh * K + kw]);
                }
            }
        }
        __syncthreads();
    }

    // ---- reduce the 4 waves' partial tiles through LDS, then one atomic per element
    float* red = reinterpret_cast<float*>(s_mem);
    for (int w = 0; w < 4; ++w) {
        if (wid == w) {
#pragma unroll
            for (int t = 0; t < KK; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int co_l = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), ci_l = lane & 31;
                    float* r = red + (t * 32 + co_l) * 32 + ci_l;
                    *r = (w == 0 ? 0.f : *r) + acc[t][i];
                }
        }
        __syncthreads();
    }
    if (a.part) {
        float* dst = a.part + ((long)blockIdx.x * gridDim.y + blockIdx.y) * (KK * 1024);
        for (int e = tid; e < KK * 1024; e += 256) dst[e] = red[e];
        return;
    }
    for (int e = tid; e < KK * 1024; e += 256) {
        const int t = e / 1024, co = co0 + (e / 32) % 32, ci = ci0 + e % 32;
        if (co >= a.Cout || ci >= a.Cin) continue;
        const int g = co / cog;
        if (ci / cig != g) continue;   // off-diagonal element of a grouped conv
        atomicAdd(a.dw + (long)co * a.ws_co + (long)(ci - g * cig) * a.ws_ci + (long)t * a.ws_tap, red[e]);
    }
}

// second stage: dW element = sum over the pixel-range workgroups of its partial tile
template <int KK>
__global__ void __launch_bounds__(256) conv_wgrad_finish_kernel(MfmaWgradArgs a, int nbx, int pairs) {
    const int pair = blockIdx.y;
    const int ci_tile = pair % a.ci_tiles, co_tile = pair / a.ci_tiles;
    const int ci0 = ci_tile * 32, co0 = co_tile * 32;
    const int cig = a.Cin / a.groups, cog = a.Cout / a.groups;
    if (a.groups > 1) {
        const int g_lo = co0 / cog, g_hi = min(co0 + 31, a.Cout - 1) / cog;
        if (ci0 + 31 < g_lo * cig || ci0 >= (g_hi + 1) * cig) return;
    }
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= KK * 1024) return;
    const int t = e / 1024, co = co0 + (e / 32) % 32, ci = ci0 + e % 32;
    if (co >= a.Cout || ci >= a.Cin) return;
    const int g = co / cog;
    if (ci / cig != g) return;
    const int per = (nbx + gridDim.z - 1) / gridDim.z;       // blockIdx.z: split of the partial rows
    const int b0 = blockIdx.z * per, b1 = min(nbx, b0 + per);
    const long stride = (long)pairs * (KK * 1024);
    const float* src = a.part + (long)pair * (KK * 1024) + e;
    float acc = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4)
        acc += (src[(long)b * stride] + src[(long)(b + 1) * stride]) +
               (src[(long)(b + 2) * stride] + src[(long)(b + 3) * stride]);
    for (; b < b1; ++b) acc += src[(long)b * stride];
    atomicAdd(a.dw + (long)co * a.ws_co + (long)(ci - g * cig) * a.ws_ci + (long)t * a.ws_tap, acc);
}

bool wgrad_mfma_supported(const ledn_wgrad_desc& d) {
    if (d.dtype_x != LEDN_BF16 || d.dtype_dz != LEDN_BF16 || d.dil != 1 || d.xadd) return false;
    if (d.groups != 1 && !(d.KH == 1 && d.KW == 1)) return false;
    if (d.Cin % 16 || (d.Cout % 16 && d.Cout > 4)) return false;   // Cout <= 4: narrow heads, scalar dz staging
    if (d.KH != d.KW || !((d.KH == 3 && d.pad == 1) || (d.KH == 1 && d.pad == 0))) return false;
    return d.stride == 1 || d.stride == 2;
}

template <int K, int S>
static int launch_wgrad(MfmaWgradArgs a, hipStream_t s) {
    a.tiles_h = (int)cdiv(a.Ho, S == 2 ? 4 : 8);
    a.tiles_w = (int)cdiv(a.Wo, 32);
    a.ci_tiles = (int)cdiv(a.Cin, 32);
    const long ntiles = (long)a.N * a.tiles_h * a.tiles_w;
    const int pairs = a.ci_tiles * (int)cdiv(a.Cout, 32);
    long blocks_x = cdiv(512, pairs);                // ~512 workgroups; partial tiles go to the workspace
    if (blocks_x < 32) blocks_x = 32;
    if (blocks_x > ntiles) blocks_x = ntiles;
    if (blocks_x > ntiles) blocks_x = ntiles;
    a.tiles_per_block = (int)cdiv(ntiles, blocks_x);
    const int nbx = (int)cdiv(ntiles, a.tiles_per_block);
    const dim3 grid((unsigned)nbx, (unsigned)pairs);
    a.part = nbx > 4 ? ws_take((long)nbx * pairs * K * K * 1024) : nullptr;
    LEDN_LAUNCH((conv_wgrad_mfma_kernel<K, S>), grid, dim3(256), 0, s, a);
    if (a.part) {
        int split = (int)cdiv(nbx, 16);
        if (split > 32) split = 32;
        LEDN_LAUNCH((conv_wgrad_finish_kernel<K * K>),
                    dim3((unsigned)cdiv(K * K * 1024, 256), (unsigned)pairs, (unsigned)split), dim3(256), 0, s, a,
                    nbx, pairs);
    }
    return check_launch();
}

int conv_wgrad_mfma(const ledn_wgrad_desc& d, hipStream_t s) {
    MfmaWgradArgs a;
    a.x = (const bf16_t*)d.x; a.dz = (const bf16_t*)d.dz; a.dw = d.dw;
    a.in_scale = d.in_scale; a.in_shift = d.in_shift;
    a.ws_co = d.ws_co; a.ws_ci = d.ws_ci; a.ws_tap = d.ws_tap;
    a.N = d.N; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout;
    a.pad = d.pad; a.in_act = d.in_act; a.groups = d.groups;
    a.tiles_h = a.tiles_w = a.tiles_per_block = a.ci_tiles = 0;
    a.part = nullptr;
    if (d.KH == 3) return d.stride == 1 ? launch_wgrad<3, 1>(a, s) : launch_wgrad<3, 2>(a, s);
    return d.stride == 1 ? launch_wgrad<1, 1>(a, s) : launch_wgrad<1, 2>(a, s);
}

}  // namespace ledn
