// conv_mfma.hip -- implicit-GEMM convolution on the CDNA4 matrix cores (gfx950).
//
// bf16 activations / bf16 packed weights / f32 accumulation, v_mfma_f32_32x32x16_bf16.
//
// Forward + data-gradient kernel (conv_mfma_kernel)
//   GEMM view: M = Cout, N = output pixels, K = taps x Cin  (A = weights, B = pixels, so that an
//   accumulator lane owns one pixel and 4 x 4 consecutive channels of the NHWC output).
//   Workgroup = 4 wavefronts = an output patch of TR rows x 32 columns x WN*32 output channels;
//   wave (wm, wn) owns MT rows and one 32-channel tile: MT accumulator tiles (16 f32 VGPRs each).
//   Workgroups are PERSISTENT (about two per CU): each walks a contiguous range of tiles and the
//   flattened (tile, 32-channel K-chunk) sequence is software-pipelined -- the 16-byte global loads
//   of step i+1 are issued right after the LDS commit of step i and fly during its MFMA phase and
//   epilogue.  vmcnt retires in order, so nothing inside a step may wait on vector memory: the
//   weight fragments of the K-chunk live in LDS too, and the epilogue flavour is a template
//   parameter (no runtime-switched parameter loads between the stores).
//   The NHWC input patch (with halo) of one K-chunk is staged in LDS as [patch pixel][32 ch + 8 pad]
//   bf16: the 80-byte pixel stride makes the fragment read (16 B of 8 consecutive channels per
//   lane, lanes = 32 pixels of a row) conflict-free.  Every tap re-reads the same patch (im2col is
//   never materialised).  The producer's BatchNorm/ReLU can be applied while staging (prologue);
//   BN-affine / residual / activation / per-channel statistics in the epilogue; the bf16 output
//   tile goes through a per-wave LDS transposition so that global stores are 16 B per lane and
//   cover whole 64-byte channel rows.
//   Data gradient = the same kernel: stride-1 convs use the flipped/transposed weight pack;
//   stride-2 convs read dz as a zero-upsampled patch (UP = 2).
//   The hot loop is kept small on purpose (~30 KB of code): the first unrolled version was 90 KB and
//   ran instruction-cache bound.
//
// Weight-gradient kernel (conv_wgrad_mfma_kernel)
//   GEMM view: M = Cout tile (32), N = Cin tile (32), K = output pixels.
//   dz and the x patch are staged pixel-major exactly as above (one tile ahead, through
//   registers); the K-contiguous fragments (8 consecutive pixels of one channel) come out of the
//   SAME layout through ds_read_b64_tr_b16 (hardware transpose read), so tap shifts are whole-pixel
//   address offsets.  3x3: the nine taps are split over the four waves (3 accumulator tiles per wave,
//   no cross-wave reduction); 1x1: the waves split the pixels and reduce through LDS.  Per-workgroup
//   partial tiles go to the workspace and a second kernel sums them (no same-address atomics).
#include <cstdlib>
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
    const float* in_slope;
    const float* out_scale;
    const float* out_shift;
    const float* slope;
    float* stat_sum;
    float* stat_sqsum;
    int N, H, W, Cin, Ho, Wo, Cout;
    int pad, in_act, act_out, res_mode;
    int tiles_h, tiles_w, tiles_per_block;
    int tile_stride;        // 0: a workgroup owns a contiguous tile range; > 0: tiles blockIdx.x, + stride, ...
    int y_f32;        // narrow heads (Cout < 8) only: y is float (the logits of LEDHead stay f32)
    float* part;      // optional workspace for the statistics: [gridDim.x * WM][2][Cout]
};

// producer BatchNorm / ReLU applied to one 16-byte (8-channel) piece while it is staged
__device__ __forceinline__ uint4 prologue_piece(uint4 v, const float* in_scale, const float* in_shift,
                                                const float* in_slope, int in_act, int c) {
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
            } else if (in_act == LEDN_ACT_PRELU) {
                lo = lo > 0.f ? lo : lo * in_slope[c + 2 * i];
                hi = hi > 0.f ? hi : hi * in_slope[c + 2 * i + 1];
            }
            w[i] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

// The patch of one 32-channel K-chunk, fetched in two phases so that every global load of a
// thread is in flight before the first one is consumed (and, across K-chunks / tiles, while the
// matrix cores work on the previous chunk): fetch() issues NL unconditional 16-byte loads
// (addresses of out-of-image pieces are clamped to the tensor base), commit() zeroes the invalid
// pieces, applies the prologue and writes the [pixel][32 ch + 8 pad] LDS layout.
template <int PR, int PC, int S, int UP>
struct PatchStage {
    static constexpr int NPIX = PR * PC;
    static constexpr int NL = (NPIX * 4 + 255) / 256;
    uint4 v[NL];
    unsigned ok;   // bit j: piece j holds image data

    __device__ __forceinline__ void fetch(const bf16_t* x, int n, int H, int W, int C, int h0, int w0, int c0,
                                          int tid) {
        tid = opaque(tid);   // recompute the piece coordinates per call instead of keeping 3 x NL registers
        ok = 0u;
        const int part = tid & 3;
        const int c = c0 + part * 8;
        const bf16_t* img = x + (long)n * H * W * C;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int p = (tid >> 2) + j * 64;
            const int pr = p / PC, pc = p % PC;
            const int uh = h0 + pr, uw = w0 + pc;
            bool valid;
            int hi, wi;
            if (UP == 1) {
                hi = uh; wi = uw;
                valid = uh >= 0 && uh < H && uw >= 0 && uw < W;
            } else {
                hi = uh / UP; wi = uw / UP;
                valid = uh >= 0 && uw >= 0 && (uh % UP) == 0 && (uw % UP) == 0 && hi < H && wi < W;
            }
            valid = valid && p < NPIX && c < C;
            // 32-bit element offset inside image n (the host checks H*W*C < 2^31): scalar base + vector offset
            const unsigned off = valid ? (unsigned)((hi * W + wi) * C + c) : 0u;
            v[j] = *reinterpret_cast<const uint4*>(img + off);
            ok |= valid ? (1u << j) : 0u;
        }
    }

    // Interior tile (the whole patch, halo included, lies inside the image; UP == 1): no bounds tests, and the
    // piece coordinates are advanced incrementally (piece j+1 is 64 patch pixels after piece j: Q rows and R
    // columns further, one row more when the column wraps) instead of a division per piece -- the general
    // fetch() spends ~25 vector instructions per piece on index arithmetic, this one ~5.
    __device__ __forceinline__ void fetch_interior(const bf16_t* x, int n, int H, int W, int C, int h0, int w0, int c0,
                                                   int tid) {
        constexpr int Q = 64 / PC, R = 64 % PC;
        tid = opaque(tid);
        const int part = tid & 3;
        const int c = c0 + part * 8;
        const bool cok = c < C;
        const bf16_t* img = x + (long)n * H * W * C;
        const int p0 = tid >> 2;
        const int pr0 = p0 / PC;
        int pc = p0 - pr0 * PC;
        unsigned off = (unsigned)(((h0 + pr0) * W + (w0 + pc)) * C + c);
        const unsigned a1 = (unsigned)((Q * W + R) * C), a2 = (unsigned)(((Q + 1) * W + R - PC) * C);
        ok = 0u;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const bool valid = cok && (j * 64 + 63 < NPIX || p0 + j * 64 < NPIX);
            v[j] = *reinterpret_cast<const uint4*>(img + (valid ? off : 0u));
            ok |= valid ? (1u << j) : 0u;
            if (R != 0) {
                const bool wrap = pc + R >= PC;
                pc += wrap ? R - PC : R;
                off += wrap ? a2 : a1;
            } else {
                off += a1;
            }
        }
    }
    // workgroup-uniform: the patch [h0, h0+PR) x [w0, w0+PC) lies inside the H x W image
    static __device__ __forceinline__ bool interior(int H, int W, int h0, int w0) {
        return UP == 1 && h0 >= 0 && w0 >= 0 && h0 + PR <= H && w0 + PC <= W;
    }

    // raw pieces -> LDS; when the producer's BatchNorm / ReLU is folded into this conv (prologue) the
    // thread then transforms its own valid pieces in place (rolled loop: keeps the hot loop's code
    // small; zero padding stays zero, i.e. padding is applied AFTER the prologue)
    __device__ __forceinline__ void commit(unsigned char* s_patch, const float* in_scale, const float* in_shift,
                                           const float* in_slope, int in_act, int c0, int tid) const {
        const int part = tid & 3;
        // input prologue (BatchNorm affine + activation of the producing layer, folded into this convolution): applied
        // to the staged registers on their way into LDS -- the thread's eight channels are the same for all its pieces,
        // so their parameters are loaded once per commit.  (The first version wrote the raw patch and re-read, transformed
        // and re-wrote every slot: the LDS read-modify-write cost what the saved elementwise pass had cost.)
        if (!(in_scale || in_act)) {            // no prologue (most layers): the plain copy, no per-piece branching
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int p = (tid >> 2) + j * 64;
                if (p >= NPIX) continue;
                const bool valid = ok & (1u << j);
                uint4 q = v[j];
                q.x = valid ? q.x : 0u; q.y = valid ? q.y : 0u; q.z = valid ? q.z : 0u; q.w = valid ? q.w : 0u;
                *reinterpret_cast<uint4*>(s_patch + (long)p * PIXB + part * 16) = q;
            }
            return;
        }
        float sc[8], sh[8], sl[8];
        {
            const int c = c0 + part * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sc[i] = 1.f; sh[i] = 0.f; sl[i] = 0.f;
            }
            if (in_scale) {
                ld8(in_scale + c, sc);
                ld8(in_shift + c, sh);
            }
            if (in_act == LEDN_ACT_PRELU) ld8(in_slope + c, sl);
            else if (in_act == LEDN_ACT_NONE) {
#pragma unroll
                for (int i = 0; i < 8; ++i) sl[i] = 1.f;     // act(v) = max(v,0) + sl * min(v,0)
            }
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int p = (tid >> 2) + j * 64;
            if (p >= NPIX) continue;
            const bool valid = ok & (1u << j);
            unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
                lo = lo * sc[2 * i] + sh[2 * i];
                hi = hi * sc[2 * i + 1] + sh[2 * i + 1];
                lo = fmaxf(lo, 0.f) + sl[2 * i] * fminf(lo, 0.f);
                hi = fmaxf(hi, 0.f) + sl[2 * i + 1] * fminf(hi, 0.f);
                w[i] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
            }
            // (zero padding and tails pad the ACTIVATION, not its pre-image: zero after the prologue)
            uint4 q;
            q.x = valid ? w[0] : 0u; q.y = valid ? w[1] : 0u; q.z = valid ? w[2] : 0u; q.w = valid ? w[3] : 0u;
            *reinterpret_cast<uint4*>(s_patch + (long)p * PIXB + part * 16) = q;
        }
    }
};

// dz tile of a narrow head (Cout = 1..4, pixel stride not 16-byte aligned): the first 4 channels
// of each pixel are fetched element-wise by the pixel's first thread, the rest of the 32-channel
// LDS row is zero.
template <int NPIX>
struct NarrowStage {
    static constexpr int NL = (NPIX * 4 + 255) / 256;
    unsigned short e[NL][4];

    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < NL; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) e[j][k] = 0;
    }
    __device__ __forceinline__ void fetch(const bf16_t* dz, int n, int Ho, int Wo, int C, int h0, int w0, int tid) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int p = (tid >> 2) + j * 64;
            const int ho = h0 + p / 32, wo = w0 + p % 32;
            const bool valid = (tid & 3) == 0 && p < NPIX && ho < Ho && wo < Wo;
            const bf16_t* src = valid ? dz + (((long)n * Ho + ho) * Wo + wo) * C : dz;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned short u = src[k < C ? k : 0].v;
                e[j][k] = (valid && k < C) ? u : (unsigned short)0;
            }
        }
    }
    __device__ __forceinline__ void commit(unsigned char* s_z, int tid) const {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int p = (tid >> 2) + j * 64;
            if (p >= NPIX) continue;
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if ((tid & 3) == 0) {
                q.x = e[j][0] | ((unsigned)e[j][1] << 16);
                q.y = e[j][2] | ((unsigned)e[j][3] << 16);
            }
            *reinterpret_cast<uint4*>(s_z + (long)p * PIXB + (tid & 3) * 16) = q;
        }
    }
};

// sum over the 32 lanes that share lane>>5 of 16 per-lane values: every lane ends up with the total
// of value index ((l>>4)&1)*8 + ((l>>3)&1)*4 + ((l>>2)&1)*2 + ((l>>1)&1).  16 shuffles instead of 80
// (each of the first four steps halves the number of live values).
__device__ __forceinline__ float reduce16_over32(const float* v, int lane) {
    float r8[8], r4[4], r2[2];
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float keep = b4 ? v[8 + j] : v[j], send = b4 ? v[j] : v[8 + j];
        r8[j] = keep + __shfl_xor(send, 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = b3 ? r8[4 + j] : r8[j], send = b3 ? r8[j] : r8[4 + j];
        r4[j] = keep + __shfl_xor(send, 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b2 ? r4[2 + j] : r4[j], send = b2 ? r4[j] : r4[2 + j];
        r2[j] = keep + __shfl_xor(send, 4);
    }
    const float keep = b1 ? r2[1] : r2[0], send = b1 ? r2[0] : r2[1];
    float r = keep + __shfl_xor(send, 2);
    r += __shfl_xor(r, 1);
    return r;
}

// compile-time loop: f(int_c<0>) ... f(int_c<N-1>)
template <int V> struct int_c { static constexpr int value = V; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(int_c<I>{});
        static_for<N, I + 1>(f);
    }
}

// epilogue flavours (compile-time: the runtime-switched version cost a vmcnt(0) drain per store group)
constexpr int EPI_RAW = 0;        // y = bf16(z)                                  (data gradients)
constexpr int EPI_RAW_STATS = 1;  // y = bf16(z) + per-channel sum / sum of squares (training forward)
constexpr int EPI_FULL = 2;       // scale/shift, residual, activation + statistics
constexpr int EPI_FULL_NS = 3;    // the same without statistics (inference: 32 VGPRs less, no spills)
constexpr int EPI_RAW_ACC = 4;    // y = bf16(z) + res, added at the 16-byte store stage: the gradient fan-in of the
                                  // training backward (the partial gradient of another consumer of the same tensor);
                                  // EPI_FULL_NS read the addend as 8-byte pieces per accumulator quad and cost as much
                                  // as the elementwise add it replaced
constexpr bool epi_full(int e) { return e == EPI_FULL || e == EPI_FULL_NS; }
constexpr bool epi_stats(int e) { return e == EPI_RAW_STATS || e == EPI_FULL; }

// Persistent workgroups: each owns a contiguous range of output tiles and walks the flattened
// (tile, K-chunk) sequence.  The global loads of step i+1 are issued right after the LDS commit of
// step i and fly during its MFMA phase and epilogue.  Because vmcnt retires in order, nothing in
// the MFMA phase or the RAW epilogues waits on vector memory: the weight fragments of the current
// K-chunk sit in LDS ([tap][cout][32 ci + 8 pad], zero-filled tails; loaded once when Cin <= 32).
// RW: K-chunks whose weight slices stay RESIDENT in LDS for the lifetime of the workgroup (1: only the Cin <= 32 case;
// 2: Cin <= 64 too -- the 64-output-channel configuration already owns its CU, and re-staging 46 KB of weights per
// K-chunk and tile moved more L2 -> LDS bytes than the activations it multiplies them with)
#ifndef LEDN_EXP
#define LEDN_EXP 0      // phase-cost experiments (tools/gpu_exp.sh): 1 no stores, 2 no matrix phase, 3 no fetch, 4 no fetch / commit
#endif
template <int WM, int WN, int MT, int K, int S, int UP, int EPI, bool VEC, int RW = 1>
__global__ void __launch_bounds__(256, 2) conv_mfma_kernel(MfmaConvArgs a) {
    constexpr int TR = WM * MT;
    constexpr int PR = (TR - 1) * S + K, PC = 31 * S + K;
    constexpr int KK = K * K;
    constexpr int NCO = WN * 32;                            // output channels per workgroup
    constexpr int WROWS = KK * NCO;                         // weight rows of one K-chunk
    constexpr int NLW = (WROWS * 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char s_patch[PR * PC * PIXB];
    __shared__ __attribute__((aligned(16))) unsigned char s_w[RW * WROWS * PIXB];
    __shared__ float s_par[3 * NCO];                        // EPI_FULL: out_scale, out_shift, slope
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int cbw = blockIdx.y * NCO;                       // first output channel of the workgroup
    const int cb0 = cbw + wn * 32;                          // ... of this wave's N-tile
    // VEC: Cout % 8 == 0, 16-byte NHWC stores through LDS; else the narrow heads (Cout < 8), scalar stores.
    // Compile-time: as a run-time flag it put a branch around every accumulator quad of the epilogue.
    constexpr bool vec = VEC;
    constexpr int EG = 1;

    const long ntiles = (long)a.N * a.tiles_h * a.tiles_w;
    const long tstep = a.tile_stride > 0 ? a.tile_stride : 1;
    const long tb = a.tile_stride > 0 ? (long)blockIdx.x : (long)blockIdx.x * a.tiles_per_block;
    const long te = a.tile_stride > 0 ? ntiles : min(ntiles, tb + a.tiles_per_block);
    if (tb >= te) return;                                   // workgroup-uniform
    const bool one_chunk = a.Cin <= RW * CK;            // every K-chunk's weights resident: loaded once

    // weights of K-chunk c0 -> LDS (16-byte pieces; rows beyond Cout / channels beyond Cin are zero)
#define LEDN_CONV_WEIGHTS(c0_, slot_)                                                                 \
    do {                                                                                              \
        unsigned char* s_wd_ = s_w + (slot_) * (WROWS * PIXB);                                        \
        uint4 wv_[NLW];                                                                               \
        _Pragma("unroll") for (int j = 0; j < NLW; ++j) {                                             \
            const int e_ = tid + j * 256, row_ = e_ >> 2, part_ = e_ & 3;                             \
            const int co_ = cbw + row_ % NCO, ci_ = (c0_) + part_ * 8;                                \
            const bool ok_ = row_ < WROWS && co_ < a.Cout && ci_ < a.Cin;                             \
            const long off_ = ok_ ? ((long)(row_ / NCO) * a.Cout + co_) * a.Cin + ci_ : 0;            \
            wv_[j] = *reinterpret_cast<const uint4*>(a.wp + off_);                                    \
            if (!ok_) wv_[j] = make_uint4(0u, 0u, 0u, 0u);                                            \
        }                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NLW; ++j) {                                             \
            const int e_ = tid + j * 256;                                                             \
            if (e_ < WROWS * 4) *reinterpret_cast<uint4*>(s_wd_ + (e_ >> 2) * PIXB + (e_ & 3) * 16) = wv_[j]; \
        }                                                                                             \
    } while (0)
    // DEEP: two register staging sets, the loads of step i+2 issued while step i computes and step i+1's are still
    // outstanding (twice the bytes in flight).  Built and measured (r03r): no change (3x3 32->32 data gradient 31.9
    // vs 32.0 us) -- the per-phase cycle counters (r03s) put the time in the epilogue and the index arithmetic of
    // fetch / commit, not in waiting for memory.  Kept off: it costs ~28 VGPRs.
    constexpr bool DEEP = false;
    PatchStage<PR, PC, S, UP> stage, stage2;
#define LEDN_CONV_FETCH(st_, tile_, c0_)                                                              \
    do {                                                                                              \
        long b_ = (tile_);                                                                            \
        const int tw_ = (int)(b_ % a.tiles_w); b_ /= a.tiles_w;                                       \
        const int th_ = (int)(b_ % a.tiles_h);                                                        \
        st_.fetch(a.x, (int)(b_ / a.tiles_h), a.H, a.W, a.Cin, th_ * TR * S - a.pad,                  \
                  tw_ * 32 * S - a.pad, (c0_), tid);                                                  \
    } while (0)
    if (one_chunk) {
#pragma unroll
        for (int ch = 0; ch < RW; ++ch)
            if (ch * CK < a.Cin) LEDN_CONV_WEIGHTS(ch * CK, ch);
    }
    if (epi_full(EPI)) {
        for (int i = tid; i < NCO; i += 256) {
            const int c = cbw + i;
            s_par[i] = (a.out_scale && c < a.Cout) ? a.out_scale[c] : 1.f;
            s_par[NCO + i] = (a.out_shift && c < a.Cout) ? a.out_shift[c] : 0.f;
            s_par[2 * NCO + i] = (a.act_out == LEDN_ACT_PRELU && c < a.Cout) ? a.slope[c]
                                 : (a.act_out == LEDN_ACT_NONE ? 1.f : 0.f);
        }
    }

    f32x16_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    constexpr int NST = epi_stats(EPI) ? 16 : 1;            // per-lane running channel statistics
    float st1[NST], st2[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) st1[i] = st2[i] = 0.f;

    // one (tile, K-chunk) step: `stg` holds its patch; afterwards stg is refilled with the step `look` ahead
    auto do_step = [&](auto& stg, const long tile, const int c0, const int look) {
        if (LEDN_EXP != 4 || a.N < 0) stg.commit(s_patch, a.in_scale, a.in_shift, a.in_slope, a.in_act, c0, tid);
        if (!one_chunk) LEDN_CONV_WEIGHTS(c0, 0);
        __syncthreads();
        long ntile = tile;
        int nc0 = c0 + CK;
        if (nc0 >= a.Cin) { nc0 = 0; ntile = tile + tstep; }
        {
            long ft = ntile;
            int fc = nc0;
            if (look == 2) {
                fc += CK;
                if (fc >= a.Cin) { fc = 0; ft += tstep; }
            }
            if (ft < te && ((LEDN_EXP != 3 && LEDN_EXP != 4) || a.N < 0)) LEDN_CONV_FETCH(stg, ft, fc);
        }
        // ---- taps x k-steps, all operands from LDS.  A = weights (M = cout), B = pixels (N = 32
        // pixels of a row): the accumulator lane owns ONE pixel and 4 x 4 consecutive channels.
        // The fragments of product step i+1 (one weight fragment + MT pixel fragments, ds_read_b128 each) are read
        // BEFORE the MT matrix instructions of step i are issued, so the LDS latency hides behind 4 x 32 cycles of
        // MFMA.  The compiler's own schedule kept pixel fragments alive across the kh shifts instead (fewer LDS
        // reads, but ~60 more VGPRs: the statistics flavours spilled 100-180 B per lane) and read most fragments
        // right in front of their first use.  Measured r03h: 3x3 32->32 forward + statistics 43.6 -> 37.4 us,
        // 64->64 37.8 -> 33.3 us, no scratch in any flavour.
        constexpr int NSTEP = KK * (CK / 16);
        const unsigned char* wbase = s_w + (one_chunk && RW > 1 ? (c0 / CK) * (WROWS * PIXB) : 0) + (wn * 32 + lr) * PIXB + lh * 16;
        const unsigned char* xbase = s_patch + (long)((wm * MT * S) * PC + lr * S) * PIXB + lh * 16;
        bf16x8_t wf_n, xf_n[MT];
#define LEDN_FRAGS(i_, wf_, xf_)                                                                          \
        do {                                                                                              \
            constexpr int t_ = (i_) / (CK / 16), kk_ = (i_) % (CK / 16), kh_ = t_ / K, kw_ = t_ % K;      \
            wf_ = *reinterpret_cast<const bf16x8_t*>(wbase + t_ * NCO * PIXB + kk_ * 32);                 \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                \
                xf_[m] = *reinterpret_cast<const bf16x8_t*>(xbase + (long)((m * S + kh_) * PC + kw_) * PIXB + kk_ * 32); \
        } while (0)
        LEDN_FRAGS(0, wf_n, xf_n);
        if (LEDN_EXP != 2 || a.N < 0)
        static_for<NSTEP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const bf16x8_t wf = wf_n;
            bf16x8_t xf[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) xf[m] = xf_n[m];
            if constexpr (i + 1 < NSTEP) LEDN_FRAGS(i + 1, wf_n, xf_n);
            sched_fence();
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = mfma_32x32x16_bf16(wf, xf[m], acc[m]);
            sched_fence();
        });
#undef LEDN_FRAGS
        __syncthreads();     // patch and chunk weights may be overwritten from here on

        if (nc0 == 0) {
            // ---- epilogue of `tile`: lane = pixel (ho, wo0 + lr); register i = channel
            // cb0 + (i&3) + 8*(i>>2) + 4*lh
            long b = tile;
            const int tw = (int)(b % a.tiles_w); b /= a.tiles_w;
            const int th = (int)(b % a.tiles_h);
            const int n = (int)(b / a.tiles_h);
            const int ho0 = th * TR + wm * MT, wo0 = tw * 32;
            const int wo = wo0 + lr;
            const float hi = a.act_out == LEDN_ACT_RELU6 ? 6.f : 3.0e38f;
            // the two 16-byte pieces this lane stores per output row, straight from registers: after the
            // v_permlane32_swap of the channel quads of lanes l / l + 32 (below) the lane holds channels
            // cb0 + 8 lh + 0..7 and cb0 + 16 + 8 lh + 0..7 of ITS pixel -- no LDS transposition, no wave
            // synchronisation (each was a full LDS drain: the epilogue was the longest phase of a RAW step, cycle
            // counters r03s), and no workgroup barrier after the epilogue.  Element offset of row ho0 (rows advance
            // by Wo * Cout).
            bool px_ok[2];
            long yoff[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                px_ok[h] = cb0 + 16 * h + 8 * lh < a.Cout && wo < a.Wo;
                yoff[h] = (((long)n * a.Ho + ho0) * a.Wo + wo) * a.Cout + cb0 + 16 * h + 8 * lh;
            }
            const long row_elems = (long)a.Wo * a.Cout;
#pragma unroll
            for (int mg = 0; mg < MT; mg += EG) {
#pragma unroll
            for (int m = mg; m < mg + EG; ++m) {
                const int ho = ho0 + m;
                const bool pix_ok = ho < a.Ho && wo < a.Wo;
                const long pix = (((long)n * a.Ho + ho) * a.Wo + wo) * a.Cout;
                unsigned pk[4][2];
                uint4 radd[2];
                if (vec && EPI == EPI_RAW_ACC) {    // the addend's pieces, in flight during the conversion
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        radd[h] = *reinterpret_cast<const uint4*>(a.res + ((px_ok[h] && ho < a.Ho) ? yoff[h] + m * row_elems : 0L));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = wn * 32 + 8 * q + 4 * lh;         // channel inside the workgroup's slice
                    const int c4 = cbw + cl;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[m][4 * q + j];
                        acc[m][4 * q + j] = 0.f;
                    }
                    if (epi_full(EPI)) {
                        float sc[4], sh[4];
                        ld4(s_par + cl, sc);
                        ld4(s_par + NCO + cl, sh);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] * sc[j] + sh[j];
                    }
                    if (epi_stats(EPI)) {   // statistics of the BatchNorm input (pre-residual, pre-activation)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float vm = pix_ok ? v[j] : 0.f;
                            st1[4 * q + j] += vm;
                            st2[4 * q + j] = fmaf(vm, vm, st2[4 * q + j]);
                        }
                    }
                    if (epi_full(EPI)) {
                        float ng[4];
                        ld4(s_par + 2 * NCO + cl, ng);
                        if (a.res_mode != LEDN_RES_NONE) {
                            float r[4] = {0.f, 0.f, 0.f, 0.f};
                            if constexpr (vec) {
                                if (pix_ok && c4 < a.Cout) ld4(a.res + pix + c4, r);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (pix_ok && c4 + j < a.Cout) r[j] = ld(a.res + pix + c4 + j);
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = a.res_mode == LEDN_RES_ADD ? v[j] + r[j] : v[j] * r[j] + r[j];
                        }
                        // activation as min(max(v,0) + neg * min(v,0), hi): none (1, inf), relu (0, inf),
                        // relu6 (0, 6), prelu (slope[c], inf)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j], 0.f) + ng[j] * fminf(v[j], 0.f), hi);
                    }
                    if constexpr (vec) {
                        pk[q][0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        pk[q][1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    } else if (q == 0) {   // narrow heads (Cout < 8): channels 4*lh + j
                        if (a.Cout == 2) {   // the two-class heads: ONE store per pixel (2-byte stores are partial writes)
                            if (pix_ok && lh == 0) {
                                if (a.y_f32) *reinterpret_cast<float2*>(reinterpret_cast<float*>(a.y) + pix) = make_float2(v[0], v[1]);
                                else *reinterpret_cast<unsigned*>(a.y + pix) = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (pix_ok && c4 + j < a.Cout) {
                                    if (a.y_f32) reinterpret_cast<float*>(a.y)[pix + c4 + j] = v[j];
                                    else st(a.y + pix + c4 + j, v[j]);
                                }
                        }
                    }
                }
                if constexpr (vec) {
                    // lanes l / l + 32 hold channels {0-3, 8-11, 16-19, 24-27} (+4): after the swaps l has 0-7 and
                    // 16-23, l + 32 has 8-15 and 24-31
                    permlane32_swap(pk[0][0], pk[1][0]);
                    permlane32_swap(pk[0][1], pk[1][1]);
                    permlane32_swap(pk[2][0], pk[3][0]);
                    permlane32_swap(pk[2][1], pk[3][1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        uint4 o = make_uint4(pk[2 * h][0], pk[2 * h][1], pk[2 * h + 1][0], pk[2 * h + 1][1]);
                        if (EPI == EPI_RAW_ACC) {
                            float fo[8], fr[8];
                            ld8(reinterpret_cast<const bf16_t*>(&o), fo);
                            ld8(reinterpret_cast<const bf16_t*>(&radd[h]), fr);
#pragma unroll
                            for (int k2 = 0; k2 < 8; ++k2) fo[k2] += fr[k2];
                            st8(reinterpret_cast<bf16_t*>(&o), fo);
                        }
                        if (px_ok[h] && ho < a.Ho && (LEDN_EXP != 1 || a.N < 0)) *reinterpret_cast<uint4*>(a.y + yoff[h] + m * row_elems) = o;
                    }
                }
            }
            }
        }
    };
    {
        long tile = tb;
        int c0 = 0;
        auto advance = [&]() {
            c0 += CK;
            if (c0 >= a.Cin) { c0 = 0; tile += tstep; }
        };
        LEDN_CONV_FETCH(stage, tile, 0);   // (requesting it before the weight load measured 1 % slower in inference, r04h)
        if (DEEP) {
            long t1 = tile;
            int c1 = CK;
            if (c1 >= a.Cin) { c1 = 0; t1 += tstep; }
            if (t1 < te) LEDN_CONV_FETCH(stage2, t1, c1);
            while (tile < te) {
                do_step(stage, tile, c0, 2);
                advance();
                if (!(tile < te)) break;
                do_step(stage2, tile, c0, 2);
                advance();
            }
        } else {
            while (tile < te) {
                do_step(stage, tile, c0, 1);
                advance();
            }
        }
    }
#undef LEDN_CONV_FETCH
#undef LEDN_CONV_WEIGHTS

    if (epi_stats(EPI) && a.stat_sum) {   // workgroup-uniform; every lane takes part in the exchange
        const float t1 = reduce16_over32(st1, lane), t2 = reduce16_over32(st2, lane);
        const int idx = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        const int cl = wn * 32 + (idx & 3) + 8 * (idx >> 2) + 4 * lh;      // channel inside the workgroup's slice
        // the WM waves of a channel slice add up in LDS first: ONE partial row per workgroup (a quarter of the rows
        // the BatchNorm finalize has to walk: it is a latency chain of row batches)
        float* s_st = reinterpret_cast<float*>(s_patch);                     // [WM][2][NCO]; the patch is dead here
        __syncthreads();
        if ((lane & 1) == 0) {
            s_st[(wm * 2 + 0) * NCO + cl] = t1;
            s_st[(wm * 2 + 1) * NCO + cl] = t2;
        }
        __syncthreads();
        for (int i = tid; i < 2 * NCO; i += 256) {
            const int j = i / NCO, cc = i % NCO, c = cbw + cc;
            if (c >= a.Cout) continue;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += s_st[(w * 2 + j) * NCO + cc];
            if (a.part) a.part[(long)blockIdx.x * 2 * a.Cout + (long)j * a.Cout + c] = t;
            else atomicAdd((j ? a.stat_sqsum : a.stat_sum) + c, t);
        }
    }
}

template <int WM, int WN, int MT, int K, int S, int UP, int EPI>
static int launch_epi(MfmaConvArgs a, hipStream_t s) {
    constexpr int TR = WM * MT;
    a.tiles_h = (int)cdiv(a.Ho, TR);
    a.tiles_w = (int)cdiv(a.Wo, 32);
    const long ntiles = (long)a.N * a.tiles_h * a.tiles_w;
    const long gy = cdiv(a.Cout, WN * 32);
    long want = options().conv_workgroups;           // default ~2 resident workgroups per CU
    if (K == 1 && S == 1 && want == 512) want = 768;  // 1x1: three per CU measured faster (r03i: 139 -> 113 us, 32->32 @512^2)
    long nbx = cdiv(want, gy);
    if (nbx > ntiles) nbx = ntiles;
    a.tiles_per_block = (int)cdiv(ntiles, nbx);
    nbx = cdiv(ntiles, a.tiles_per_block);
    a.tile_stride = (options().stream_fast & 4) ? (int)nbx : 0;
    const dim3 grid((unsigned)nbx, (unsigned)gy);
    a.part = (epi_stats(EPI) && a.stat_sum && (nbx > 16 || det())) ? ws_take(nbx * 2 * a.Cout) : nullptr;
    if constexpr (WN == 2 && K == 3 && S == 1) {     // 64 output channels per workgroup, 3x3: one workgroup per CU anyway
        if (a.Cin > CK && a.Cin <= 2 * CK && (a.Cout & 7) == 0 && (options().stream_fast & 32)) {
            LEDN_LAUNCH((conv_mfma_kernel<WM, WN, MT, K, S, UP, EPI, true, 2>), grid, dim3(256), 0, s, a);
            if (a.part) return finish_partials(a.part, (int)nbx, a.Cout, 2, a.stat_sum, a.stat_sqsum, nullptr, s);
            return check_launch();
        }
    }
    if ((a.Cout & 7) == 0) LEDN_LAUNCH((conv_mfma_kernel<WM, WN, MT, K, S, UP, EPI, true>), grid, dim3(256), 0, s, a);
    else LEDN_LAUNCH((conv_mfma_kernel<WM, WN, MT, K, S, UP, EPI, false>), grid, dim3(256), 0, s, a);
    if (a.part) return finish_partials(a.part, (int)nbx, a.Cout, 2, a.stat_sum, a.stat_sqsum, nullptr, s);
    return check_launch();
}

template <int WM, int WN, int MT, int K, int S, int UP>
static int launch_cfg(const MfmaConvArgs& a, hipStream_t s) {
    const bool raw = !a.out_scale && !a.out_shift && a.act_out == LEDN_ACT_NONE && a.res_mode == LEDN_RES_NONE;
    if (!a.out_scale && !a.out_shift && a.act_out == LEDN_ACT_NONE && a.res_mode == LEDN_RES_ADD && !a.stat_sum &&
        (a.Cout & 7) == 0 && !a.y_f32)
        return launch_epi<WM, WN, MT, K, S, UP, EPI_RAW_ACC>(a, s);
    if (!raw && a.stat_sum) return launch_epi<WM, WN, MT, K, S, UP, EPI_FULL>(a, s);
    if (!raw) return launch_epi<WM, WN, MT, K, S, UP, EPI_FULL_NS>(a, s);
    if (a.stat_sum) return launch_epi<WM, WN, MT, K, S, UP, EPI_RAW_STATS>(a, s);
    return launch_epi<WM, WN, MT, K, S, UP, EPI_RAW>(a, s);
}

template <int K, int S, int UP>
static int launch_shape(const MfmaConvArgs& a, hipStream_t s) {
    // stride 2 stages a (2*TR+1) x 65 pixel patch: TR = 4.  The LDS weight slice (taps x 32/64 cout)
    // and the patch leave room for two workgroups per CU.
    if constexpr (S == 2) {
        return launch_cfg<4, 1, 1, K, S, UP>(a, s);
    } else {
        // 64 output channels per workgroup (one read of the patch for both 32-channel halves) pays for 1x1 only: the
        // 3x3 slices of 32 channels at 2 workgroups per CU measured 6-8 % faster (r03 exp7: 64->64 39.7 -> 37.3 us,
        // data gradient 33.5 -> 31.0, 128->64 65.1 -> 59.9) -- the second read of the patch comes from L2
        if (a.Cout % 64 == 0 && K == 1) return launch_cfg<2, 2, 4, K, S, UP>(a, s);
        // fewer 16-row tiles than workgroups wanted (1/8-resolution maps, small batches): 8-row tiles double the
        // parallelism of these latency-bound launches
        if ((options().stream_fast & 8) &&
            (long)a.N * cdiv(a.Ho, 16) * cdiv(a.Wo, 32) * cdiv(a.Cout, 32) < (K == 1 ? 768 : 512))
            return launch_cfg<4, 1, 2, K, S, UP>(a, s);
        return launch_cfg<4, 1, 4, K, S, UP>(a, s);
    }
}

bool conv_mfma_supported(const ledn_conv_desc& d) {
    if (!d.w_bf16 || d.dtype_x != LEDN_BF16) return false;
    // f32 output: narrow heads only (scalar-store epilogue), no residual (it would be f32 too)
    if (d.dtype_y != LEDN_BF16 && !(d.dtype_y == LEDN_F32 && d.Cout < 8 && !d.res)) return false;
    if (d.dil != 1 || d.xadd) return false;
    if (d.act_out == LEDN_ACT_SIGMOID) return false;
    if (d.in_act != LEDN_ACT_NONE && d.in_act != LEDN_ACT_RELU && !(d.in_act == LEDN_ACT_PRELU && d.in_slope)) return false;
    if (d.groups != 1 && !(d.KH == 1 && d.KW == 1)) return false;   // grouped 1x1: densified weight pack
    if (d.Cin % 16 || (d.Cout % 16 && d.Cout > 8)) return false;   // 16-channel tails are masked
    if (d.KH != d.KW || !((d.KH == 3 && d.pad == 1) || (d.KH == 1 && d.pad == 0))) return false;
    if (d.stride != 1 && d.stride != 2) return false;
    return true;
}

int conv_mfma(const ledn_conv_desc& d, hipStream_t s) {
    MfmaConvArgs a;
    a.x = (const bf16_t*)d.x; a.wp = (const bf16_t*)d.w_bf16; a.y = (bf16_t*)d.y; a.res = (const bf16_t*)d.res;
    a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.in_slope = d.in_slope; a.out_scale = d.out_scale; a.out_shift = d.out_shift;
    a.slope = d.slope; a.stat_sum = d.stat_sum; a.stat_sqsum = d.stat_sqsum;
    a.N = d.N; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout;
    a.in_act = d.in_act; a.act_out = d.act_out; a.res_mode = d.res_mode;
    a.tiles_h = a.tiles_w = a.tiles_per_block = a.tile_stride = 0;
    a.y_f32 = d.dtype_y == LEDN_F32;
    a.part = nullptr;
    if (!d.transposed) {
        a.pad = d.pad;
        if (d.KH == 3) return d.stride == 1 ? launch_shape<3, 1, 1>(a, s) : launch_shape<3, 2, 1>(a, s);
        return d.stride == 1 ? launch_shape<1, 1, 1>(a, s) : launch_shape<1, 2, 1>(a, s);
    }
    a.pad = d.KH - 1 - d.pad;   // stride-1 correlation over the (zero-upsampled) dz with flipped taps
    if (d.KH == 3) return d.stride == 1 ? launch_shape<3, 1, 1>(a, s) : launch_shape<3, 1, 2>(a, s);
    // 1x1 stride 2: a 1x1 correlation over the zero-upsampled dz (three of four outputs are zero)
    return d.stride == 1 ? launch_shape<1, 1, 1>(a, s) : launch_shape<1, 1, 2>(a, s);
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
    const NhwcIdx ix_ = pix_split(pix, Wo, Ho);
    const int wo = ix_.x, ho = ix_.y, n = ix_.n;
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

// planar source: thread = (output pixel, quarter of the 32 columns): the four lanes of a pixel store its
// whole 64-byte row (16 B each); along a wave the planar reads run along W (stride 2 pixels)
__device__ __forceinline__ float ldp(const unsigned char* p) { return (float)*p; }
__device__ __forceinline__ float ldp(const float* p) { return *p; }
__device__ __forceinline__ float ldp(const bf16_t* p) { return bf16_to_f32(p->v); }

// Workgroup = 64 consecutive output pixels of one output row: the 3 input rows x (2*64+1) columns x C
// planes it needs are read once with coalesced loads, normalised and kept as bf16 in LDS; thread
// (pixel, quarter) then assembles 8 of the pixel's 32 columns and stores 16 bytes (the four lanes of
// a pixel write its whole 64-byte row).  (The first version gathered straight from global memory:
// 32 scattered scalar loads per thread, 1 TB/s.)
// (C is a template parameter: the column -> (tap, channel) split divides by it 16 times per thread, and a
// run-time divisor made the kernel ALU-bound at 1.1 TB/s)
template <typename TX, int C>
__global__ void __launch_bounds__(256) im2col_stem_planar_kernel(const TX* x, bf16_t* p, int N, int H, int W,
                                                                 int Ho, int Wo, const float* scale,
                                                                 const float* shift, const int* map,
                                                                 const int* valid_hw, float pad_val) {
    constexpr int PXB = 64, COLS = 2 * PXB + 1;
    __shared__ unsigned short s_in[3 * 3 * COLS];             // [c][row][col], zero outside the image
    const int wtiles = (Wo + PXB - 1) / PXB;
    const int wt = blockIdx.x % wtiles;
    const int ho = (blockIdx.x / wtiles) % Ho;
    const int n = blockIdx.x / (wtiles * Ho);
    const int wo0 = wt * PXB;
    const long plane = (long)H * W;
    // unconditional loads (out-of-image elements read the plane's first byte and are zeroed): all of a
    // thread's loads are in flight together instead of a chain of five dependent round trips
    constexpr int NE = (C * 3 * COLS + 255) / 256;
    float f[NE];
    bool ok[NE], data[NE];
    // batch padding (stack_batch): rows >= vh / columns >= vw of image n read as pad_val (normalised domain)
    const int vh = valid_hw ? valid_hw[2 * n] : H, vw = valid_hw ? valid_hw[2 * n + 1] : W;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = threadIdx.x + j * 256;
        const int col = e % COLS, r = (e / COLS) % 3, c = min(e / (3 * COLS), C - 1);
        const int hi = ho * 2 - 1 + r, wi = wo0 * 2 - 1 + col;
        ok[j] = e < C * 3 * COLS && hi >= 0 && hi < H && wi >= 0 && wi < W;
        data[j] = hi < vh && wi < vw;
        const int cs = map ? map[c] : c;
        f[j] = ldp(x + ((long)n * C + cs) * plane + (ok[j] ? (long)hi * W + wi : 0L));
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = threadIdx.x + j * 256;
        if (e >= C * 3 * COLS) continue;
        const int c = e / (3 * COLS);
        float v = f[j];
        if (scale) v = v * scale[c] + shift[c];
        v = data[j] ? v : pad_val;
        s_in[e] = ok[j] ? f32_to_bf16(v) : (unsigned short)0;
    }
    __syncthreads();
    const int q = threadIdx.x & 3, px = threadIdx.x >> 2;
    const int wo = wo0 + px;
    if (wo >= Wo) return;
    unsigned short e8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int col = q * 8 + i;                 // column (kh*3+kw)*C + c of the patch row
        const int tap = col / C, c = col - tap * C;
        const int kh = tap / 3, kw = tap - kh * 3;
        e8[i] = col < 9 * C ? s_in[(c * 3 + kh) * COLS + px * 2 + kw] : (unsigned short)0;
    }
    const long pix = ((long)n * Ho + ho) * Wo + wo;
    *reinterpret_cast<uint4*>(p + pix * 32 + q * 8) =
        make_uint4(e8[0] | ((unsigned)e8[1] << 16), e8[2] | ((unsigned)e8[3] << 16), e8[4] | ((unsigned)e8[5] << 16),
                   e8[6] | ((unsigned)e8[7] << 16));
}

// ---------------------------------------------------------------------------
// The first stem convolution (3x3, stride 2, 3 -> 32) straight from the planar input batch: normalisation, batch
// padding, im2col and the K = 32 GEMM in ONE kernel -- the patch matrix [pixels][32] (268 MB at 16 x 1024^2, written
// by ledn_im2col_stem_planar and read back by the 1x1 MFMA conv: 307 us together) is never materialised.
// Workgroup = 4 output rows x 64 pixels per tile (tiles dealt round-robin): the 9 x 129 x 3 input window is
// normalised into LDS as bf16; wave w owns output row w: its two 32-pixel blocks assemble the K-fragment of a pixel
// from 8 LDS halfwords (k = (kh*3+kw)*3 + c; k >= 27 is zero), weights stay in registers (A operand: cout rows),
// the accumulator lane owns one pixel and 16 channels, v_permlane32_swap merges the channel quads of lanes l and
// l + 32 into 16-byte stores (no LDS transposition).  FULL: y = act(acc * out_scale + out_shift) (inference: folded
// BatchNorm + ReLU); else raw z + per-workgroup statistics rows (training).
// ---------------------------------------------------------------------------
struct StemArgs {
    const void* x;
    const bf16_t* wp;            // [32 cout][32 k] bf16 (ledn_pack_conv_weights of the [32][32][1][1] view)
    bf16_t* y;
    const float *in_scale, *in_shift, *out_scale, *out_shift;
    const int* map;
    const int* valid_hw;
    float* stat_sum;
    float* stat_sqsum;
    float* part;
    int N, H, W, Ho, Wo, act_out;
    float pad_val;
};

template <typename TX, bool FULL>
__global__ void __launch_bounds__(256) stem_conv_kernel(StemArgs a) {
    constexpr int C = 3, TRO = 4, PXB = 64, COLS = 2 * PXB + 1, ROWS = 2 * TRO + 1;
    constexpr int NE = (C * ROWS * COLS + 255) / 256;
    __shared__ unsigned short s_in[C * ROWS * COLS];          // [c][row][col], zero outside the image
    __shared__ float s_st[4][2][32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const TX* x = reinterpret_cast<const TX*>(a.x);
    bf16x8_t wf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(a.wp + lr * 32 + ks * 16 + lh * 8);
    int koff[2][8];                                           // LDS offset of column k of the patch row, -1: zero
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + lh * 8 + j, tap = k / C, c = k - tap * C, kh = tap / 3, kw = tap - kh * 3;
            koff[ks][j] = k < 9 * C ? (c * ROWS + kh) * COLS + kw : -1;
        }
    float sc[16], sh[16];
    if (FULL) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = (i & 3) + 8 * (i >> 2) + 4 * lh;
            sc[i] = a.out_scale ? a.out_scale[c] : 1.f;
            sh[i] = a.out_shift ? a.out_shift[c] : 0.f;
        }
    }
    float st1[16], st2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st1[i] = st2[i] = 0.f;
    const float hi_clip = a.act_out == LEDN_ACT_RELU6 ? 6.f : 3.0e38f;
    const int tw = (a.Wo + PXB - 1) / PXB, th = (a.Ho + TRO - 1) / TRO;
    const long ntiles = (long)a.N * th * tw;
    const long plane = (long)a.H * a.W;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int txi = (int)(t % tw), tyi = (int)((t / tw) % th), n = (int)(t / ((long)tw * th));
        const int ho0 = tyi * TRO, wo0 = txi * PXB;
        const int vh = a.valid_hw ? a.valid_hw[2 * n] : a.H, vw = a.valid_hw ? a.valid_hw[2 * n + 1] : a.W;
        float f[NE];
        bool ok[NE], data[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int e = tid + j * 256;
            const int col = e % COLS, r = (e / COLS) % ROWS, c = min(e / (ROWS * COLS), C - 1);
            const int hi = ho0 * 2 - 1 + r, wi = wo0 * 2 - 1 + col;
            ok[j] = e < C * ROWS * COLS && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            data[j] = hi < vh && wi < vw;
            const int cs = a.map ? a.map[c] : c;
            f[j] = ldp(x + ((long)n * C + cs) * plane + (ok[j] ? (long)hi * a.W + wi : 0L));
        }
        __syncthreads();                                      // the previous tile's fragments have been read
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int e = tid + j * 256;
            if (e >= C * ROWS * COLS) continue;
            const int c = e / (ROWS * COLS);
            float v = f[j];
            if (a.in_scale) v = v * a.in_scale[c] + a.in_shift[c];
            v = data[j] ? v : a.pad_val;
            s_in[e] = ok[j] ? f32_to_bf16(v) : (unsigned short)0;
        }
        __syncthreads();
        const int ho = ho0 + wid;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int px = mb * 32 + lr, wo = wo0 + px;
            const int base = (2 * wid) * COLS + 2 * px;
            f32x16_t acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t xf;
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[j] = koff[ks][j] >= 0 ? (short)s_in[base + koff[ks][j]] : (short)0;
                acc = mfma_32x32x16_bf16(wf[ks], xf, acc);
            }
            const bool pix_ok = ho < a.Ho && wo < a.Wo;
            unsigned pk[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[4 * q + j];
                    if (FULL) {
                        v[j] = v[j] * sc[4 * q + j] + sh[4 * q + j];
                        if (a.act_out != LEDN_ACT_NONE) v[j] = fminf(fmaxf(v[j], 0.f), hi_clip);
                    } else {
                        const float vm = pix_ok ? v[j] : 0.f;
                        st1[4 * q + j] += vm;
                        st2[4 * q + j] = fmaf(vm, vm, st2[4 * q + j]);
                    }
                }
                pk[q][0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                pk[q][1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            }
            // lanes l / l + 32 hold channels {0-3, 8-11, 16-19, 24-27} (+4): after the swaps l has 0-7 and 16-23,
            // l + 32 has 8-15 and 24-31
            permlane32_swap(pk[0][0], pk[1][0]);
            permlane32_swap(pk[0][1], pk[1][1]);
            permlane32_swap(pk[2][0], pk[3][0]);
            permlane32_swap(pk[2][1], pk[3][1]);
            if (pix_ok) {
                bf16_t* dst = a.y + (((long)n * a.Ho + ho) * a.Wo + wo) * 32 + 8 * lh;
                *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
                *reinterpret_cast<uint4*>(dst + 16) = make_uint4(pk[2][0], pk[2][1], pk[3][0], pk[3][1]);
            }
        }
    }
    if (FULL || !a.stat_sum) return;
    const float t1 = reduce16_over32(st1, lane), t2 = reduce16_over32(st2, lane);
    const int idx = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    const int cl = (idx & 3) + 8 * (idx >> 2) + 4 * lh;
    __syncthreads();
    if ((lane & 1) == 0) {
        s_st[wid][0][cl] = t1;
        s_st[wid][1][cl] = t2;
    }
    __syncthreads();
    if (tid < 64) {
        const int j = tid >> 5, c = tid & 31;
        const float t = (s_st[0][j][c] + s_st[1][j][c]) + (s_st[2][j][c] + s_st[3][j][c]);
        if (a.part) a.part[(long)blockIdx.x * 64 + j * 32 + c] = t;
        else atomicAdd((j ? a.stat_sqsum : a.stat_sum) + c, t);
    }
}

bool stem_conv_reg_enabled();
int stem_conv_reg_impl(const void* x, int dtype_x, const void* wp, void* y, int N, int H, int W, int Ho, int Wo,
                       const float* in_scale, const float* in_shift, const int* map, const int* valid_hw, float pad_val,
                       const float* out_scale, const float* out_shift, int act_out, float* stat_sum, float* stat_sqsum,
                       hipStream_t s);

int stem_conv_impl(const void* x, int dtype_x, const void* wp, void* y, int N, int H, int W, int C, int Ho, int Wo,
                   int Cout, const float* in_scale, const float* in_shift, const int* map, const int* valid_hw,
                   float pad_val, const float* out_scale, const float* out_shift, int act_out, float* stat_sum,
                   float* stat_sqsum, hipStream_t s) {
    LEDN_REQUIRE(x && wp && y && N > 0 && H > 0 && W > 0 && C == 3 && Cout == 32);
    LEDN_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1);
    LEDN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (stat_sum == nullptr) == (stat_sqsum == nullptr));
    LEDN_REQUIRE(act_out == LEDN_ACT_NONE || act_out == LEDN_ACT_RELU || act_out == LEDN_ACT_RELU6);
    const bool full = out_scale || out_shift || act_out != LEDN_ACT_NONE;
    LEDN_REQUIRE(!(full && stat_sum));                        // inference epilogue XOR training statistics
    if (stem_conv_reg_enabled())                              // the register-direct form (conv3x3.hip)
        return stem_conv_reg_impl(x, dtype_x, wp, y, N, H, W, Ho, Wo, in_scale, in_shift, map, valid_hw, pad_val, out_scale,
                                  out_shift, act_out, stat_sum, stat_sqsum, s);
    StemArgs a;
    a.x = x; a.wp = (const bf16_t*)wp; a.y = (bf16_t*)y;
    a.in_scale = in_scale; a.in_shift = in_shift; a.out_scale = out_scale; a.out_shift = out_shift;
    a.map = map; a.valid_hw = valid_hw; a.stat_sum = stat_sum; a.stat_sqsum = stat_sqsum;
    a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.act_out = act_out; a.pad_val = pad_val;
    const long ntiles = (long)N * cdiv(Ho, 4) * cdiv(Wo, 64);
    long nb = ntiles < 2048 ? ntiles : 2048;      // (one-shot workgroups measured slower: 198 vs 182 us)
    a.part = (stat_sum && (nb > 16 || det())) ? ws_take(nb * 64) : nullptr;
#define LEDN_STEM(TX)                                                                                      \
    do {                                                                                                   \
        if (full) LEDN_LAUNCH((stem_conv_kernel<TX, true>), dim3((unsigned)nb), dim3(256), 0, s, a);       \
        else LEDN_LAUNCH((stem_conv_kernel<TX, false>), dim3((unsigned)nb), dim3(256), 0, s, a);           \
    } while (0)
    if (dtype_x == LEDN_U8) LEDN_STEM(unsigned char);
    else if (dtype_x == LEDN_F32) LEDN_STEM(float);
    else if (dtype_x == LEDN_BF16) LEDN_STEM(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_STEM
    if (a.part) return finish_partials(a.part, (int)nb, 32, 2, stat_sum, stat_sqsum, nullptr, s);
    return check_launch();
}

int im2col_stem_planar_impl(const void* x, int dtype_x, void* p, int N, int H, int W, int C, int Ho, int Wo,
                            const float* scale, const float* shift, const int* map, const int* valid_hw,
                            float pad_val, hipStream_t s) {
    LEDN_REQUIRE(x && p && N > 0 && H > 0 && W > 0 && C > 0 && 9 * C <= 32);
    LEDN_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1);
    LEDN_REQUIRE((scale == nullptr) == (shift == nullptr));
    const dim3 grid((unsigned)((long)N * Ho * cdiv(Wo, 64)));
#define LEDN_IPC(TX, CC)                                                                                    \
    LEDN_LAUNCH((im2col_stem_planar_kernel<TX, CC>), grid, dim3(256), 0, s, (const TX*)x, (bf16_t*)p, N, H, W, Ho, \
                Wo, scale, shift, map, valid_hw, pad_val)
#define LEDN_IP(TX)                  \
    do {                             \
        if (C == 3) LEDN_IPC(TX, 3); \
        else if (C == 2) LEDN_IPC(TX, 2); \
        else LEDN_IPC(TX, 1);        \
    } while (0)
    if (dtype_x == LEDN_U8) LEDN_IP(unsigned char);
    else if (dtype_x == LEDN_F32) LEDN_IP(float);
    else if (dtype_x == LEDN_BF16) LEDN_IP(bf16_t);
    else return LEDN_EINVAL;
#undef LEDN_IP
#undef LEDN_IPC
    return check_launch();
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
    const float* in_slope;
    long long ws_co, ws_ci, ws_tap;
    int N, H, W, Cin, Ho, Wo, Cout;
    int pad, in_act, groups;
    int tiles_h, tiles_w, tiles_per_block, ci_tiles;
    float* part;      // optional workspace: [gridDim.x][gridDim.y][KK*1024] per-workgroup partial tiles
};

// the 2*TR k-steps (16 pixels each) of one staged tile for NTW taps: the transposing LDS reads of k-step ks+1 are
// issued before the NTW matrix instructions of k-step ks
template <int NTW, int S, int PC, int TR, bool TAPSPLIT, int NT>
__device__ __forceinline__ void wgrad_ksteps(f32x16_t (&acc)[NT], const unsigned char* s_x, const unsigned char* s_z,
                                             const int (&tapoff)[NT], int ks0, int kstep, int lh, int q, int colblk,
                                             int pp) {
    bf16x4_t fa[2], fb[NTW][2];
    auto frags = [&](int ks) {
        const int row = ks >> 1, cb = (ks & 1) * 16;       // pixels (row, cb + 0..15)
        // A = dz^T: the lane needs 8 pixels cb+8*lh+0..7 of channel colblk*16+l16; B = the x patch at the tap shift
        const unsigned char* zp = s_z + (long)(row * 32 + cb + lh * 8 + q) * PIXB + (colblk * 16 + pp * 4) * 2;
        fa[0] = lds_read_tr16(zp);
        fa[1] = lds_read_tr16(zp + 4 * PIXB);
        const unsigned char* xp0 = s_x + (long)(row * S * PC + (cb + lh * 8 + q) * S) * PIXB + (colblk * 16 + pp * 4) * 2;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            fb[t][0] = lds_read_tr16(xp0 + tapoff[t]);
            fb[t][1] = lds_read_tr16(xp0 + tapoff[t] + 4 * S * PIXB);
        }
    };
    if (ks0 < TR * 2) frags(ks0);
#pragma unroll 2
    for (int ks = ks0; ks < TR * 2; ks += kstep) {
        bf16x8_t af, bfv[NTW];
        af[0] = fa[0][0]; af[1] = fa[0][1]; af[2] = fa[0][2]; af[3] = fa[0][3];
        af[4] = fa[1][0]; af[5] = fa[1][1]; af[6] = fa[1][2]; af[7] = fa[1][3];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            bfv[t][0] = fb[t][0][0]; bfv[t][1] = fb[t][0][1]; bfv[t][2] = fb[t][0][2]; bfv[t][3] = fb[t][0][3];
            bfv[t][4] = fb[t][1][0]; bfv[t][5] = fb[t][1][1]; bfv[t][6] = fb[t][1][2]; bfv[t][7] = fb[t][1][3];
        }
        if (ks + kstep < TR * 2) frags(ks + kstep);
        sched_fence();
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = mfma_32x32x16_bf16(af, bfv[t], acc[t]);
        sched_fence();
    }
}

template <int K, int S>
__global__ void __launch_bounds__(256) conv_wgrad_mfma_kernel(MfmaWgradArgs a) {
    constexpr int TR = S == 2 ? 4 : 8;             // output rows per staged tile (TR x 32 pixels = 2*TR k-steps)
    constexpr int PR = (TR - 1) * S + K, PC = 31 * S + K;
    constexpr int XB = PR * PC * PIXB, ZB = TR * 32 * PIXB;
    constexpr int KK = K * K;
    constexpr int RED = 32 * 32 * 4;               // K = 1: cross-wave reduction of the single tile
    constexpr int LDSB = (XB + ZB) > RED ? (XB + ZB) : RED;
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[LDSB];
    unsigned char* s_x = s_mem;
    unsigned char* s_z = s_mem + XB;
    const int tid = threadIdx.x, lane = tid & 63, wid = wave_uniform(tid >> 6);
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

    // K = 3: the nine taps are split over the four waves (wave w owns taps w, w+4, w+8): three
    // accumulator tiles per wave instead of nine (3 workgroups per CU instead of 1) and no cross-wave
    // reduction.  K = 1: the waves split the pixels and reduce through LDS.
    constexpr bool TAPSPLIT = K == 3;
    constexpr int NT = TAPSPLIT ? 3 : 1;
    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    int tapoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = TAPSPLIT ? wid + 4 * t : 0;
        tapoff[t] = ((tap / K) * PC + tap % K) * PIXB;
    }

    const long ntiles = (long)a.N * a.tiles_h * a.tiles_w;
    const long t0 = (long)blockIdx.x * a.tiles_per_block;
    const long t1 = min(ntiles, t0 + a.tiles_per_block);
    const bool narrow = (a.Cout & 7) != 0;   // heads with Cout = 1, 2, 4: element-wise dz staging
    // (Two register staging sets -- loads two tiles ahead -- measured no faster, r03k: the loop was not waiting on
    // global memory but on a branch + full LDS wait in front of every matrix instruction; see wgrad_ksteps.)
    PatchStage<PR, PC, S, 1> sx;
    PatchStage<TR, 32, 1, 1> sz;
    NarrowStage<TR * 32> szn;
    szn.clear();
    // every tile's loads are issued one tile ahead: they fly while the matrix cores work
#define LEDN_WGRAD_FETCH(tile_)                                                                          \
    do {                                                                                                 \
        long b_ = (tile_);                                                                               \
        const int tw_ = (int)(b_ % a.tiles_w); b_ /= a.tiles_w;                                          \
        const int th_ = (int)(b_ % a.tiles_h);                                                           \
        const int n_ = (int)(b_ / a.tiles_h);                                                            \
        const int hx_ = th_ * TR * S - a.pad, wx_ = tw_ * 32 * S - a.pad;                                \
        if (sx.interior(a.H, a.W, hx_, wx_)) sx.fetch_interior(a.x, n_, a.H, a.W, a.Cin, hx_, wx_, ci0, tid); \
        else sx.fetch(a.x, n_, a.H, a.W, a.Cin, hx_, wx_, ci0, tid);                                     \
        if (!narrow) {                                                                                   \
            if (sz.interior(a.Ho, a.Wo, th_ * TR, tw_ * 32)) sz.fetch_interior(a.dz, n_, a.Ho, a.Wo, a.Cout, th_ * TR, tw_ * 32, co0, tid); \
            else sz.fetch(a.dz, n_, a.Ho, a.Wo, a.Cout, th_ * TR, tw_ * 32, co0, tid);                   \
        } else szn.fetch(a.dz, n_, a.Ho, a.Wo, a.Cout, th_ * TR, tw_ * 32, tid);                         \
    } while (0)
    if (t0 < t1) LEDN_WGRAD_FETCH(t0);
    for (long tile = t0; tile < t1; ++tile) {
        sx.commit(s_x, a.in_scale, a.in_shift, a.in_slope, a.in_act, ci0, tid);
        if (!narrow) sz.commit(s_z, nullptr, nullptr, nullptr, 0, co0, tid);
        else szn.commit(s_z, tid);
        __syncthreads();
        if (tile + 1 < t1) LEDN_WGRAD_FETCH(tile + 1);
        // wave w owns taps w, w+4, w+8: wave 0 three of them, the others two -- a compile-time count per code path
        // (a per-tap validity test inside the loop put a branch and a full LDS wait in front of every matrix
        // instruction, and with the wave index in a vector register the accumulators travelled AGPR <-> VGPR)
        if constexpr (TAPSPLIT) {
            if (wid != 0) wgrad_ksteps<NT - 1, S, PC, TR, true>(acc, s_x, s_z, tapoff, 0, 1, lh, q, colblk, pp);
            else wgrad_ksteps<NT, S, PC, TR, true>(acc, s_x, s_z, tapoff, 0, 1, lh, q, colblk, pp);
        } else {
            wgrad_ksteps<NT, S, PC, TR, false>(acc, s_x, s_z, tapoff, wid, 4, lh, q, colblk, pp);
        }
        __syncthreads();
    }
#undef LEDN_WGRAD_FETCH

    // accumulator register i of lane l = dW[co0 + (i&3) + 8*(i>>2) + 4*(l>>5)][ci0 + (l&31)]
    if (TAPSPLIT) {   // every wave owns whole taps: straight to the workspace / dW
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tap = wid + 4 * t;
            if (tap >= KK) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co_l = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), ci_l = lane & 31;
                if (a.part) {
                    a.part[((long)blockIdx.x * gridDim.y + blockIdx.y) * (KK * 1024) + (tap * 32 + co_l) * 32 + ci_l] =
                        acc[t][i];
                } else {
                    const int co = co0 + co_l, ci = ci0 + ci_l;
                    if (co >= a.Cout || ci >= a.Cin) continue;
                    const int g = co / cog;
                    if (ci / cig != g) continue;
                    atomicAdd(a.dw + (long)co * a.ws_co + (long)(ci - g * cig) * a.ws_ci + (long)tap * a.ws_tap,
                              acc[t][i]);
                }
            }
        }
        return;
    }
    // ---- K = 1: reduce the 4 waves' partial tiles through LDS, then one atomic per element
    float* red = reinterpret_cast<float*>(s_mem);
    for (int w = 0; w < 4; ++w) {
        if (wid == w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co_l = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), ci_l = lane & 31;
                float* r = red + co_l * 32 + ci_l;
                *r = (w == 0 ? 0.f : *r) + acc[0][i];
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

// second stage: dW element = sum over the pixel-range workgroups of its partial tile.
// Workgroup = 64 consecutive elements of one (ci, co) tile pair x 16 row groups (1024 lanes): a wave reads 256
// contiguous bytes of one partial tile per load, 8 loads in flight, rows rg, rg+16, ...; the 16 row-group sums meet
// in LDS and ONE lane per element adds the total to dW (plain read-modify-write: the element has one owner).
// (First version: 32 workgroups per element ending in 32 same-address atomics = 18 us after a 44 us main kernel.)
template <int KK>
__global__ void __launch_bounds__(1024) conv_wgrad_finish_kernel(MfmaWgradArgs a, int nbx, int pairs) {
    __shared__ float s_red[16][64];
    const int pair = blockIdx.y;
    const int ci_tile = pair % a.ci_tiles, co_tile = pair / a.ci_tiles;
    const int ci0 = ci_tile * 32, co0 = co_tile * 32;
    const int cig = a.Cin / a.groups, cog = a.Cout / a.groups;
    if (a.groups > 1) {
        const int g_lo = co0 / cog, g_hi = min(co0 + 31, a.Cout - 1) / cog;
        if (ci0 + 31 < g_lo * cig || ci0 >= (g_hi + 1) * cig) return;
    }
    const int el = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;                       // KK*1024 is a multiple of 64
    const long stride = (long)pairs * (KK * 1024);
    const float* src = a.part + (long)pair * (KK * 1024) + e;
    float acc = 0.f;
    int b = rg;
    for (; b + 112 < nbx; b += 128) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long)(b + 16 * u) * stride];
        acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; b < nbx; b += 16) acc += src[(long)b * stride];
    s_red[rg][el] = acc;
    __syncthreads();
    if (rg != 0) return;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += s_red[r][el];
    const int tap = e / 1024, co = co0 + (e / 32) % 32, ci = ci0 + e % 32;
    if (co >= a.Cout || ci >= a.Cin) return;
    const int g = co / cog;
    if (ci / cig != g) return;
    float* dst = a.dw + (long)co * a.ws_co + (long)(ci - g * cig) * a.ws_ci + (long)tap * a.ws_tap;
    *dst += t;
}

bool wgrad_mfma_supported(const ledn_wgrad_desc& d) {
    if (d.dtype_x != LEDN_BF16 || d.dtype_dz != LEDN_BF16 || d.dil != 1 || d.xadd) return false;
    if (d.groups != 1 && !(d.KH == 1 && d.KW == 1)) return false;
    if (d.Cin % 16 || (d.Cout % 16 && d.Cout > 4)) return false;   // Cout <= 4: narrow heads, scalar dz staging
    if (d.KH != d.KW || !((d.KH == 3 && d.pad == 1) || (d.KH == 1 && d.pad == 0))) return false;
    return d.stride == 1 || d.stride == 2;
}

// deferred reduction (ledn_conv2d_wgrad_partial): the partial tiles go to a caller-owned buffer and the summing launch is
// left to ledn_conv2d_wgrad_finish_multi (one launch for every convolution of the backward pass)
struct WgradDefer {
    float* part;                     // caller's buffer (nullptr: classic path)
    long long part_floats;
    ledn_wgrad_finish_entry* entry;  // filled on success
    bool query;                      // only report the buffer size in entry->nbx/pairs/KK
};

bool conv1x1_wgrad_reg_ok(int Cin, int Cout, int groups);
int conv1x1_wgrad_reg_partial(const void* x, const void* dz, float* part, long P, int Cin, int Cout, int groups, int nbx,
                              hipStream_t s);

// does ledn_conv2d_wgrad run this descriptor on conv1x1_wgrad_reg_kernel?  (the same test as in launch_wgrad<1, 1>)
bool conv1x1_wgrad_reg_applies(const ledn_wgrad_desc& d) {
    return wgrad_mfma_supported(d) && d.KH == 1 && d.stride == 1 && !d.in_scale && d.in_act == LEDN_ACT_NONE &&
           (long)d.N * d.Ho * d.Wo >= 32768 && conv1x1_wgrad_reg_ok(d.Cin, d.Cout, d.groups);
}

template <int K, int S>
static int launch_wgrad(MfmaWgradArgs a, hipStream_t s, WgradDefer* df = nullptr) {
    a.tiles_h = (int)cdiv(a.Ho, S == 2 ? 4 : 8);
    a.tiles_w = (int)cdiv(a.Wo, 32);
    a.ci_tiles = (int)cdiv(a.Cin, 32);
    const long ntiles = (long)a.N * a.tiles_h * a.tiles_w;
    const int pairs = a.ci_tiles * (int)cdiv(a.Cout, 32);
    if constexpr (K == 1 && S == 1) {
        // the wave-autonomous form (conv3x3.hip: conv1x1_wgrad_reg_kernel): every workgroup forms ALL pairs from one read
        // of x and dz and writes them in this kernel's partial-tile layout -- the summing kernels below serve both
        const long P = (long)a.N * a.Ho * a.Wo;
        if (!a.in_scale && a.in_act == LEDN_ACT_NONE && P >= 32768 && conv1x1_wgrad_reg_ok(a.Cin, a.Cout, a.groups)) {
            static const int ipw = (int)exp_knob("LEDN_W11_ITERS", 8);   // (A/B knob: 4 and 8 measured equal, 8 = half the partial tiles)
            long nb = cdiv(P, 32L * 4 * (ipw > 0 ? ipw : 8));    // >= ipw iterations of 32 pixels per wave
            if (nb > options().wgrad_workgroups) nb = options().wgrad_workgroups;
            const int nbx = (int)nb;
            if (df) {
                ledn_wgrad_finish_entry& e = *df->entry;
                e.nbx = nbx; e.pairs = pairs; e.KK = 1; e.ci_tiles = a.ci_tiles; e.Cin = a.Cin; e.Cout = a.Cout;
                e.groups = a.groups; e.chunk0 = 0;
                e.ws_co = a.ws_co; e.ws_ci = a.ws_ci; e.ws_tap = a.ws_tap;
                e.dw = a.dw; e.part = df->part;
                if (df->query) return LEDN_OK;
                if (!df->part || df->part_floats < (long long)nbx * pairs * 1024) return LEDN_EINVAL;
                return conv1x1_wgrad_reg_partial(a.x, a.dz, df->part, P, a.Cin, a.Cout, a.groups, nbx, s);
            }
            a.part = ws_take((long)nbx * pairs * 1024);
            if (a.part) {
                const int rc = conv1x1_wgrad_reg_partial(a.x, a.dz, a.part, P, a.Cin, a.Cout, a.groups, nbx, s);
                if (rc != LEDN_OK) return rc;
                LEDN_LAUNCH((conv_wgrad_finish_kernel<1>), dim3((unsigned)(1024 / 64), (unsigned)pairs), dim3(1024), 0, s, a, nbx,
                            pairs);
                return check_launch();
            }
        }
    }
    long blocks_x = cdiv(options().wgrad_workgroups, pairs);   // partial tiles go to the workspace
    if (blocks_x < 32) blocks_x = 32;
    if (blocks_x > ntiles) blocks_x = ntiles;
    if (blocks_x > ntiles) blocks_x = ntiles;
    a.tiles_per_block = (int)cdiv(ntiles, blocks_x);
    const int nbx = (int)cdiv(ntiles, a.tiles_per_block);
    const dim3 grid((unsigned)nbx, (unsigned)pairs);
    if (df) {
        ledn_wgrad_finish_entry& e = *df->entry;
        e.nbx = nbx; e.pairs = pairs; e.KK = K * K; e.ci_tiles = a.ci_tiles; e.Cin = a.Cin; e.Cout = a.Cout;
        e.groups = a.groups; e.chunk0 = 0;
        e.ws_co = a.ws_co; e.ws_ci = a.ws_ci; e.ws_tap = a.ws_tap;
        e.dw = a.dw; e.part = df->part;
        if (df->query) return LEDN_OK;
        if (!df->part || df->part_floats < (long long)nbx * pairs * K * K * 1024) return LEDN_EINVAL;
        a.part = df->part;
        LEDN_LAUNCH((conv_wgrad_mfma_kernel<K, S>), grid, dim3(256), 0, s, a);
        return check_launch();
    }
    a.part = (nbx > 4 || det()) ? ws_take((long)nbx * pairs * K * K * 1024) : nullptr;
    LEDN_LAUNCH((conv_wgrad_mfma_kernel<K, S>), grid, dim3(256), 0, s, a);
    if (a.part)
        LEDN_LAUNCH((conv_wgrad_finish_kernel<K * K>), dim3((unsigned)(K * K * 1024 / 64), (unsigned)pairs), dim3(1024), 0,
                    s, a, nbx, pairs);
    return check_launch();
}

static int conv_wgrad_mfma_x(const ledn_wgrad_desc& d, hipStream_t s, WgradDefer* df);
int conv_wgrad_mfma(const ledn_wgrad_desc& d, hipStream_t s) { return conv_wgrad_mfma_x(d, s, nullptr); }
// -> entry filled; query: no launch
int conv_wgrad_mfma_partial(const ledn_wgrad_desc& d, float* part, long long part_floats, ledn_wgrad_finish_entry* entry,
                            bool query, hipStream_t s) {
    WgradDefer df = {part, part_floats, entry, query};
    return conv_wgrad_mfma_x(d, s, &df);
}

static int conv_wgrad_mfma_x(const ledn_wgrad_desc& d, hipStream_t s, WgradDefer* df) {
    MfmaWgradArgs a;
    a.x = (const bf16_t*)d.x; a.dz = (const bf16_t*)d.dz; a.dw = d.dw;
    a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.in_slope = d.in_slope;
    a.ws_co = d.ws_co; a.ws_ci = d.ws_ci; a.ws_tap = d.ws_tap;
    a.N = d.N; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout;
    a.pad = d.pad; a.in_act = d.in_act; a.groups = d.groups;
    a.tiles_h = a.tiles_w = a.tiles_per_block = a.ci_tiles = 0;
    a.part = nullptr;
    if (d.KH == 3) return d.stride == 1 ? launch_wgrad<3, 1>(a, s, df) : launch_wgrad<3, 2>(a, s, df);
    return d.stride == 1 ? launch_wgrad<1, 1>(a, s, df) : launch_wgrad<1, 2>(a, s, df);
}

// One launch for the deferred reductions of a whole backward pass.  Workgroup (256 lanes) = ONE 32 x 32 tile (4 KB) of one
// (co, ci) tile pair and tap of one table entry; a lane owns four consecutive elements and walks the partial rows with
// eight 16-byte loads in flight, so every wave-load is 1 KB contiguous and a workgroup reads 4 KB per row.  (Round 3's
// form -- 1024 lanes = 64 elements x 16 row groups -- read 256 B per row and workgroup, rows >= 36 KB apart: 429 MB in
// 239 us = 1.8 TB/s, on the critical path between the backward and the optimizer.)  The entry is found from the chunk
// prefix (chunk0, in 64-element chunks: 16 per tile) of the table.  Summation order: rows in order, 8 interleaved
// partial sums -- fixed.
__global__ void __launch_bounds__(256) conv_wgrad_finish_multi_kernel(const ledn_wgrad_finish_entry* tab, int n) {
    __shared__ int s_e;
    const int chunk = (int)blockIdx.x * 16;
    if (threadIdx.x < 64) {
        int cnt = 0;
        for (int i = threadIdx.x; i < n; i += 64) cnt += tab[i].chunk0 <= chunk ? 1 : 0;
        cnt = (int)wave_sum((float)cnt);
        if (threadIdx.x == 0) s_e = cnt - 1;
    }
    __syncthreads();
    const ledn_wgrad_finish_entry e = tab[s_e];
    const int KK = e.KK;
    const int local = ((int)blockIdx.x * 16 - e.chunk0) / 16;          // tile index inside the entry: pair * KK + tap
    const int pair = local / KK, tap = local % KK;
    const int ci_tile = pair % e.ci_tiles, co_tile = pair / e.ci_tiles;
    const int ci0 = ci_tile * 32, co0 = co_tile * 32;
    const int cig = e.Cin / e.groups, cog = e.Cout / e.groups;
    if (e.groups > 1) {
        const int g_lo = co0 / cog, g_hi = min(co0 + 31, e.Cout - 1) / cog;
        if (ci0 + 31 < g_lo * cig || ci0 >= (g_hi + 1) * cig) return;
    }
    const long stride = (long)e.pairs * (KK * 1024);
    const float4* src = reinterpret_cast<const float4*>(e.part + (long)pair * (KK * 1024) + (long)tap * 1024) + threadIdx.x;
    const long s4 = stride / 4;
    float4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int b = 0;
    for (; b + 8 <= e.nbx; b += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long)(b + u) * s4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w;
        }
    }
    for (; b < e.nbx; ++b) {
        const float4 v = src[(long)b * s4];
        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
    }
    float t[4];
    t[0] = ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
    t[1] = ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
    t[2] = ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
    t[3] = ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e1 = (int)threadIdx.x * 4 + j;
        const int co = co0 + e1 / 32, ci = ci0 + e1 % 32;
        if (co >= e.Cout || ci >= e.Cin) continue;
        const int g = co / cog;
        if (ci / cig != g) continue;
        float* dst = e.dw + (long)co * e.ws_co + (long)(ci - g * cig) * e.ws_ci + (long)tap * e.ws_tap;
        *dst += t[j];
    }
}

int conv_wgrad_finish_multi_impl(const ledn_wgrad_finish_entry* table_dev, int n, int total_chunks, hipStream_t s) {
    LEDN_REQUIRE(table_dev && n > 0 && total_chunks > 0);
    LEDN_REQUIRE(total_chunks % 16 == 0);
    LEDN_LAUNCH(conv_wgrad_finish_multi_kernel, dim3((unsigned)(total_chunks / 16)), dim3(256), 0, s, table_dev, n);
    return check_launch();
}

}  // namespace ledn
