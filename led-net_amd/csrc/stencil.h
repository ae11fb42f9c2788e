// stencil.h -- building blocks of the bf16 depthwise / SESP-pyramid stencil kernels.
//
// These kernels are VALU-issue bound, not HBM bound, when written per 4 channels with 64-bit
// address arithmetic per tap (a wave instruction costs 4 cycles: ~250 instructions per output pixel
// and 4 channels ran at ~1 TB/s).  The vectorised form below spends ~20 instructions per tap and
// EIGHT channels: one 16-byte load at a 32-bit offset from a scalar base, shift/mask bf16->f32,
// packed f32 FMAs (v_pk_fma_f32), pixel coordinates advanced incrementally instead of divided.
#pragma once
#include "ledn_rt.h"

namespace ledn {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// 8 consecutive bf16 channels <-> four packed f32 pairs
__device__ __forceinline__ void bf16x8_unpack(const uint4& r, f32x2_t* o) {
    o[0] = f32x2_t{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
    o[1] = f32x2_t{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    o[2] = f32x2_t{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u)};
    o[3] = f32x2_t{__uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}
__device__ __forceinline__ uint4 bf16x8_pack(const f32x2_t* v) {
    uint4 o;
    o.x = (unsigned)f32_to_bf16(v[0].x) | ((unsigned)f32_to_bf16(v[0].y) << 16);
    o.y = (unsigned)f32_to_bf16(v[1].x) | ((unsigned)f32_to_bf16(v[1].y) << 16);
    o.z = (unsigned)f32_to_bf16(v[2].x) | ((unsigned)f32_to_bf16(v[2].y) << 16);
    o.w = (unsigned)f32_to_bf16(v[3].x) | ((unsigned)f32_to_bf16(v[3].y) << 16);
    return o;
}
// 8 f32 (two 16-byte loads) as four pairs
__device__ __forceinline__ void f32x8_load(const float* p, f32x2_t* o) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = f32x2_t{a.x, a.y}; o[1] = f32x2_t{a.z, a.w}; o[2] = f32x2_t{b.x, b.y}; o[3] = f32x2_t{b.z, b.w};
}
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }

// 16-byte load of 8 bf16 channels at a 32-bit element offset; `valid` = false reads `safe` and yields zeros
__device__ __forceinline__ uint4 ld_tap(const bf16_t* base, unsigned off, unsigned safe, bool valid) {
    uint4 r = *reinterpret_cast<const uint4*>(base + (valid ? off : safe));
    r.x = valid ? r.x : 0u; r.y = valid ? r.y : 0u; r.z = valid ? r.z : 0u; r.w = valid ? r.w : 0u;
    return r;
}

// pixel cursor over an [N][H][W] raster advanced by a fixed step (< W*H) without divisions
struct PixCursor {
    int n, y, x;
    __device__ __forceinline__ void init(long p, int H, int W) {
        x = (int)(p % W);
        y = (int)((p / W) % H);
        n = (int)(p / ((long)W * H));
    }
    __device__ __forceinline__ void advance(int step, int H, int W) {
        x += step;
        while (x >= W) {
            x -= W;
            if (++y >= H) { y = 0; ++n; }
        }
    }
};

// 3x3 tap geometry with dilation dl (stride 1, zero padding dl): bit t = kh*3+kw of the mask is set
// when tap (y + (kh-1)*dl, x + (kw-1)*dl) lies inside the H x W image
__device__ __forceinline__ unsigned tap_mask(int y, int x, int dl, int H, int W) {
    const unsigned r0 = y - dl >= 0, r2 = y + dl < H, c0 = x - dl >= 0, c2 = x + dl < W;
    const unsigned cols = c0 | 2u | (c2 << 2);
    return (r0 ? cols : 0u) | (cols << 3) | (r2 ? cols << 6 : 0u);
}

// ---- 8-byte (4-channel) forms and the reduce-scatter steps between wave halves / neighbouring rows
__device__ __forceinline__ uint2 ld_tap8(const bf16_t* base, unsigned off, unsigned safe, bool valid) {
    uint2 r = *reinterpret_cast<const uint2*>(base + (valid ? off : safe));
    r.x = valid ? r.x : 0u;
    r.y = valid ? r.y : 0u;
    return r;
}
__device__ __forceinline__ void bf16x4_unpack(const uint2& r, f32x2_t* o) {
    o[0] = f32x2_t{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
    o[1] = f32x2_t{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
__device__ __forceinline__ void swap_add32(f32x2_t& a, f32x2_t& b) {
    unsigned ax = __float_as_uint(a.x), bx = __float_as_uint(b.x), ay = __float_as_uint(a.y), by = __float_as_uint(b.y);
    permlane32_swap(ax, bx);
    permlane32_swap(ay, by);
    a = f32x2_t{__uint_as_float(ax) + __uint_as_float(bx), __uint_as_float(ay) + __uint_as_float(by)};
}
__device__ __forceinline__ void swap_add16(f32x2_t& a, f32x2_t& b) {
    unsigned ax = __float_as_uint(a.x), bx = __float_as_uint(b.x), ay = __float_as_uint(a.y), by = __float_as_uint(b.y);
    permlane16_swap(ax, bx);
    permlane16_swap(ay, by);
    a = f32x2_t{__uint_as_float(ax) + __uint_as_float(bx), __uint_as_float(ay) + __uint_as_float(by)};
}

}  // namespace ledn
