// ledn_rt.h -- common device-side helpers for the LED-Net HIP kernels (gfx950).
//
// The only place that knows about the test-only CPU emulation build
// (LEDN_CPU_EMU, see emu.h); kernel sources are written once against this
// header.  Wave = 64 lanes everywhere.
#pragma once
#include <stdint.h>

#ifdef LEDN_CPU_EMU
#include "emu.h"
#else
#include <hip/hip_runtime.h>
#define LEDN_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__)
#endif

#include "../../include/ledn.h"

namespace ledn {

// ---- storage types ----------------------------------------------------------
struct bf16_t {
    unsigned short v;
};

__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
#ifdef LEDN_CPU_EMU
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {  // round-to-nearest-even, NaN stays NaN
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
#else
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even, quiet NaN)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
#endif

__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->v); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(bf16_t* p, float v) { p->v = f32_to_bf16(v); }

// 4-element vector access (pointer must be 16 B / 8 B aligned)
__device__ __forceinline__ void ld4(const float* p, float* o) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float* o) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(bf16_t* p, const float* v) {
    uint2 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = o;
}
// 8-element access: 16 B per lane for bf16 (the width the HBM-bound elementwise kernels want)
__device__ __forceinline__ void ld8(const float* p, float* o) {
    ld4(p, o);
    ld4(p + 4, o + 4);
}
__device__ __forceinline__ void ld8(const bf16_t* p, float* o) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ void st8(float* p, const float* v) {
    st4(p, v);
    st4(p + 4, v + 4);
}
__device__ __forceinline__ void st8(bf16_t* p, const float* v) {
    uint4 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
    o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = o;
}
// 2-element access (2-channel logits: one 8-byte / 4-byte access per pixel)
__device__ __forceinline__ void ld2(const float* p, float* o) {
    const float2 v = *reinterpret_cast<const float2*>(p);
    o[0] = v.x; o[1] = v.y;
}
__device__ __forceinline__ void ld2(const bf16_t* p, float* o) {
    const unsigned v = *reinterpret_cast<const unsigned*>(p);
    o[0] = __uint_as_float(v << 16); o[1] = __uint_as_float(v & 0xffff0000u);
}
__device__ __forceinline__ void st2(float* p, const float* v) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
__device__ __forceinline__ void st2(bf16_t* p, const float* v) {
    *reinterpret_cast<unsigned*>(p) = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
}
template <int V, typename T> __device__ __forceinline__ void ldv(const T* p, float* o) {
    if constexpr (V == 8) ld8(p, o);
    else if constexpr (V == 2) ld2(p, o);
    else if constexpr (V == 4) ld4(p, o);
    else {
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = ld(p + i);
    }
}
template <int V, typename T> __device__ __forceinline__ void stv(T* p, const float* v) {
    if constexpr (V == 8) st8(p, v);
    else if constexpr (V == 2) st2(p, v);
    else if constexpr (V == 4) st4(p, v);
    else {
#pragma unroll
        for (int i = 0; i < V; ++i) st(p + i, v[i]);
    }
}

// stencil tap: an UNCONDITIONAL load (out-of-image taps read the tensor base and are zeroed) --
// no branch around the load, so all taps of a stencil are in flight together
template <int V, typename T>
__device__ __forceinline__ void ldv_if(const T* base, long off, bool valid, float* o) {
    ldv<V>(base + (valid ? off : 0L), o);
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = valid ? o[i] : 0.f;
}

// ---- activations --------------------------------------------------------------
__device__ __forceinline__ float act_apply(int act, float v, float slope) {
    switch (act) {
        case LEDN_ACT_RELU: return v > 0.f ? v : 0.f;
        case LEDN_ACT_RELU6: return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
        case LEDN_ACT_PRELU: return v > 0.f ? v : v * slope;
        case LEDN_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}
// d act(v) / dv, as a function of the pre-activation v
__device__ __forceinline__ float act_grad(int act, float v, float slope) {
    switch (act) {
        case LEDN_ACT_RELU: return v > 0.f ? 1.f : 0.f;
        case LEDN_ACT_RELU6: return (v > 0.f && v < 6.f) ? 1.f : 0.f;
        case LEDN_ACT_PRELU: return v > 0.f ? 1.f : slope;
        case LEDN_ACT_SIGMOID: { float s = 1.f / (1.f + __expf(-v)); return s * (1.f - s); }
        default: return 1.f;
    }
}

// ---- wave reductions (64 lanes; every lane of the wave must call) --------------
__device__ __forceinline__ float wave_sum(float v);      // (defined below, after the lane-crossing helpers)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// dynamically sized LDS (the launch's shared-memory argument), 16-byte aligned
#ifdef LEDN_CPU_EMU
#define LEDN_DYN_SHARED(T, name) T* name = reinterpret_cast<T*>(LEDN_DYN_SMEM)
#else
#define LEDN_DYN_SHARED(T, name)                                                  \
    extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_[];   \
    T* name = reinterpret_cast<T*>(name##_raw_)
#endif

// ---- exchange between the two halves of a wave (gfx950: v_permlane32_swap_b32) -------------------------------
// after the call: a = {a.low, b.low}, b = {a.high, b.high}  (x.low / x.high = the values lanes 0-31 / 32-63 held in x)
#ifdef LEDN_CPU_EMU
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {
    const bool hi = lane_id() >= 32;
    const unsigned recv = __shfl_xor(hi ? a : b, 32);
    if (hi) a = recv;
    else b = recv;
}
// the same between neighbouring 16-lane rows (v_permlane16_swap_b32): a = {a.r0, b.r0, a.r2, b.r2}, b = {a.r1, b.r1, a.r3, b.r3}
__device__ __forceinline__ void permlane16_swap(unsigned& a, unsigned& b) {
    const bool odd = (lane_id() >> 4) & 1;
    const unsigned recv = __shfl_xor(odd ? a : b, 16);
    if (odd) a = recv;
    else b = recv;
}
#else
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r.x;
    b = r.y;
}
__device__ __forceinline__ void permlane16_swap(unsigned& a, unsigned& b) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r.x;
    b = r.y;
}
#endif

// v + the value of the lane `m` away in the coset pattern of an all-reduce (m a power of two): on the vector ALU where the
// hardware has a lane-crossing form for it -- DPP row_ror inside a 16-lane row (m = 1, 2, 4, 8), v_permlane32_swap for the
// two halves of the wave (m = 32) -- and through the LDS crossbar (ds_bpermute) only for m = 16.  The butterfly of
// reduce_taps (72 accumulators x up to 6 levels per lane) was 360 ds_bpermute per lane for 16-channel maps: the epilogue
// of the pyramid / depthwise weight-gradient kernels cost as much as their pixel loops.
__device__ __forceinline__ float lane_step_sum(float v, int m) {
#ifdef LEDN_CPU_EMU
    return v + __shfl_xor(v, m);
#else
    if (m == 32) {
        unsigned a = __float_as_uint(v), b = a;
        permlane32_swap(a, b);                               // a = {low, low}, b = {high, high}
        return __uint_as_float(a) + __uint_as_float(b);
    }
    if (m == 16) return v + __shfl_xor(v, 16);
    const int iv = __float_as_int(v);
    int r;
    if (m == 8) r = __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xf, 0xf, false);        // row_ror:8
    else if (m == 4) r = __builtin_amdgcn_update_dpp(0, iv, 0x124, 0xf, 0xf, false);   // row_ror:4
    else if (m == 2) r = __builtin_amdgcn_update_dpp(0, iv, 0x122, 0xf, 0xf, false);   // row_ror:2
    else r = __builtin_amdgcn_update_dpp(0, iv, 0x121, 0xf, 0xf, false);               // row_ror:1
    return v + __int_as_float(r);
#endif
}

// sum over the 64 lanes of the wave, returned in every lane (the xor butterfly 32, 16, .. 1: bit-identical to six __shfl_xor
// steps, one ds_bpermute instead of six)
__device__ __forceinline__ float wave_sum(float v) {
    v = lane_step_sum(v, 32);
    v = lane_step_sum(v, 16);
    v = lane_step_sum(v, 8);
    v = lane_step_sum(v, 4);
    v = lane_step_sum(v, 2);
    return lane_step_sum(v, 1);
}

// F.interpolate(mode='bilinear', align_corners=False) source coordinates
// (ATen area_pixel_compute_source_index: scale*(dst+0.5)-0.5 clamped at 0).
struct Lerp {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ Lerp lerp_coord(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
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

// ---- flat index -> (channel vector, x, y, image) --------------------------------------------------------
// 32-bit divisions whenever the index fits (every tensor of this network: < 2^31 vectors); a 64-bit integer division is
// ~100 vector instructions on this ISA, and the five of them a lane spent on `idx % cv, idx / cv, pix % W, (pix / W) % H,
// pix / (W H)` made the gather kernels (bilinear and its adjoints, the pool adjoints, the MFAF combine) VALU-bound.
struct NhwcIdx {
    long pix;
    int cv, x, y, n;
};
__device__ __forceinline__ NhwcIdx pix_split(long pix, int W, int H) {
    NhwcIdx r;
    r.pix = pix;
    r.cv = 0;
    if (pix < 0x7fffffffL) {
        const unsigned p = (unsigned)pix, row = p / (unsigned)W, n = row / (unsigned)H;
        r.x = (int)(p - row * (unsigned)W);
        r.y = (int)(row - n * (unsigned)H);
        r.n = (int)n;
    } else {
        r.x = (int)(pix % W);
        r.y = (int)((pix / W) % H);
        r.n = (int)(pix / ((long)W * H));
    }
    return r;
}
__device__ __forceinline__ NhwcIdx nhwc_split(long idx, int cvn, int W, int H) {
    if (idx < 0x7fffffffL) {
        const unsigned i = (unsigned)idx, pix = i / (unsigned)cvn;
        NhwcIdx r = pix_split((long)pix, W, H);
        r.cv = (int)(i - pix * (unsigned)cvn);
        return r;
    }
    NhwcIdx r = pix_split(idx / cvn, W, H);
    r.cv = (int)(idx % cvn);
    return r;
}

// ---- XCD-aware block numbering ----------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs, each with a private L2.  xcd_block() renumbers
// them so that the workgroups resident on one XCD own ONE contiguous eighth of the logical block
// range: neighbouring tiles / pixel rows then find their halo lines in the same L2 instead of
// every XCD pulling them over the fabric again.
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nblocks) {
    if (nblocks < 16u) return bid;
    const unsigned x = bid & 7u, q = nblocks >> 3, rem = nblocks & 7u;
    return x * q + (x < rem ? x : rem) + (bid >> 3);
}

// ---- matrix-core (MFMA) fragments and the LDS transposing read ------------------
typedef short bf16x8_t __attribute__((ext_vector_type(8)));   // 8 bf16 = one 32x32x16 A/B fragment
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));  // one 32x32 f32 accumulator tile

#ifdef LEDN_CPU_EMU
__device__ __forceinline__ f32x16_t mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return emu::mfma_32x32x16_bf16(a, b, c);
}
__device__ __forceinline__ bf16x4_t lds_read_tr16(const void* p) { return emu::lds_read_tr16(p); }
__device__ __forceinline__ void sched_fence() {}
__device__ __forceinline__ int opaque(int x) { return x; }
__device__ __forceinline__ int wave_uniform(int x) { return x; }
__device__ __forceinline__ void wave_sync() { emu::barrier_wait(LEDN_EMU_CUR->wave_bar); }
#else
// value the optimiser must treat as freshly computed: keeps loop-invariant address arithmetic from
// being hoisted out of a persistent loop into (spilled) registers
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// a value that is the same in every lane of the wave (wave index, ...) moved to a scalar register: branches on it
// become scalar branches instead of exec-mask regions (which, around an MFMA, also made the compiler shuttle the
// accumulator between AGPRs and VGPRs: 32 moves per matrix instruction in the weight-gradient loop)
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// orders one wave's LDS accesses across its lanes (the DS unit executes a wave's instructions in
// order; this keeps the compiler from reordering them).  Every lane of the wave must call.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// keeps the instruction scheduler from hoisting later loads above this point (bounds live registers)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
typedef __bf16 hw_bf16x8_t __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_bf16: lane l (r=l&31,h=l>>5) holds A[r][8h+j], B[8h+j][r];
// D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5)   (cdna_hip_programming.md section 3)
__device__ __forceinline__ f32x16_t mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hw_bf16x8_t, a),
                                                   __builtin_bit_cast(hw_bf16x8_t, b), c, 0, 0, 0);
}
// ds_read_b64_tr_b16; needs all 64 lanes active and an 8-byte aligned LDS address per lane
__device__ __forceinline__ bf16x4_t lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) bf16x4_t*)(const_cast<void*>(p)));
}
#endif

// ---- host-side helpers ----------------------------------------------------------
inline int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LEDN_OK : LEDN_ELAUNCH;
}
__host__ __device__ inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// Caller-provided scratch (ledn_set_workspace): cross-workgroup reductions write per-workgroup
// partials here and a tiny second kernel sums them -- no same-address atomics (which serialise
// at ~0.3 us each on gfx950), deterministic.  Single-stream use.
struct Workspace {
    float* ptr;
    long nfloats;
};
Workspace& workspace();
inline float* ws_take(long nfloats) {
    Workspace& w = workspace();
    return (w.ptr && nfloats <= w.nfloats) ? w.ptr : nullptr;
}
// launch-shape knobs (ledn_set_option)
struct Options {
    int conv_workgroups;
    int wgrad_workgroups;
    int stream_fast;
    int deterministic;      // LEDN_OPT_DETERMINISTIC: every cross-workgroup reduction in a fixed order (no f32 atomics)
    int bn_fused;           // LEDN_OPT_BN_FUSED: the persistent one-pass BatchNorm backward may be launched
};
Options& options();
// A/B measurement knobs (getenv) are honoured only in experimental mode (LEDN_EXPERIMENTAL=1): an old shell export must
// not change the launch shape of a normal run
inline long exp_knob(const char* name, long def) {
    const char* e = getenv("LEDN_EXPERIMENTAL");
    const char* v = (e && atoi(e)) ? getenv(name) : nullptr;
    return v ? atol(v) : def;
}
// deterministic mode: reductions that would end in float atomics for small grids take the partial-row path too
inline bool det() { return options().deterministic != 0; }
// ledn_conv2d_deferred_stats: the MFMA conv leaves its per-workgroup statistic rows [rows][2][C] in the
// workspace (no finish launch) and reports them here; ledn_bn_finalize_rows sums them itself
struct DeferredStats {
    bool want;
    float* part;
    int rows;
};
DeferredStats& deferred_stats();
// out_j[c] += sum_b part[b*K + j*C + c], j < nout (K = nout*C)
int finish_partials(const float* part, int nblk, int C, int nout, float* o0, float* o1, float* o2,
                    hipStream_t s);
// stream_fast.hip: -1 = not handled (the generic kernel takes the call)
int affine_act_fast(const ledn_affine_desc& d, hipStream_t s);
int bn_act_bwd_reduce_fast(const ledn_bnbwd_desc& d, hipStream_t s);
int bn_act_bwd_apply_fast(const ledn_bnbwd_desc& d, hipStream_t s);
int bn_act_bwd_fused(const ledn_bnbwd_desc& d, hipStream_t s);
int bn_act_bwd_fused_check(int C, hipStream_t s);
// head_bwd.hip
int head_bwd_supported(const ledn_headbwd_desc& d);
int head_bwd_reduce(const ledn_headbwd_desc& d, hipStream_t s);
int head_bwd_apply(const ledn_headbwd_desc& d, hipStream_t s);
bool head_fwd_supported(const ledn_conv_desc& d);
int head_fwd(const ledn_conv_desc& d, hipStream_t s);
int channel_stats_fast(const void* x, const void* xadd, long long P, int C, int dtype, float* sum, float* sqsum,
                       hipStream_t s);

}  // namespace ledn

#define LEDN_REQUIRE(cond) \
    do {                   \
        if (!(cond)) return LEDN_EINVAL; \
    } while (0)
