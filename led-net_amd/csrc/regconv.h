// regconv.h -- shared pieces of the register-direct convolution kernels (conv1x1.hip, conv3x3.hip): the
// v_mfma_f32_16x16x32_bf16 wrapper, the output-channel permutation of the weight fragments and the 16-lane reductions.
#pragma once
#include "ledn_rt.h"

namespace ledn {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

#ifdef LEDN_CPU_EMU
__device__ __forceinline__ f32x4_t mfma_16x16x32_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return emu::mfma_16x16x32_bf16(a, b, c);
}
#else
// v_mfma_f32_16x16x32_bf16: lane l holds A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15]; D: col = l&15, row = 4(l>>4)+reg
__device__ __forceinline__ f32x4_t mfma_16x16x32_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hw_bf16x8_t, a),
                                                   __builtin_bit_cast(hw_bf16x8_t, b), c, 0, 0, 0);
}
#endif

template <int V> struct c11_int { static constexpr int value = V; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void c11_for(F&& f) {
    if constexpr (I < N) {
        f(c11_int<I>{});
        c11_for<N, I + 1>(f);
    }
}

// channel of row 4q + i of M-tile mt (see the header): pairs of tiles interleave in blocks of four
template <int NMT>
__device__ __forceinline__ int c11_channel(int mt, int q, int i) {
    if ((mt | 1) < NMT) return 32 * (mt >> 1) + 8 * q + 4 * (mt & 1) + i;
    return 16 * mt + 4 * q + i;                    // unpaired last tile (Cout % 32 == 16): natural order
}

// sum over the 16 lanes that share (lane >> 4) of V per-lane values (V = 4, 8 or 16): every lane ends up with the
// total of ONE value, index c11_red_index<V>(lane); V - 1 + (4 - log2 V) * ... shuffles instead of 4 V
template <int V>
__device__ __forceinline__ float c11_reduce16(const float* v, int lane) {
    static_assert(V == 4 || V == 8 || V == 16, "");
    float r8[8], r4[4], r2[2], r;
    if constexpr (V == 16) {
        const bool b = lane & 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) r8[j] = (b ? v[8 + j] : v[j]) + __shfl_xor(b ? v[j] : v[8 + j], 8);
    } else if constexpr (V == 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r8[j] = v[j] + __shfl_xor(v[j], 8);
    }
    if constexpr (V >= 8) {
        const bool b = lane & 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) r4[j] = (b ? r8[4 + j] : r8[j]) + __shfl_xor(b ? r8[j] : r8[4 + j], 4);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = v[j] + __shfl_xor(v[j], 8);
            r4[j] = t + __shfl_xor(t, 4);
        }
    }
    {
        const bool b = lane & 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) r2[j] = (b ? r4[2 + j] : r4[j]) + __shfl_xor(b ? r4[j] : r4[2 + j], 2);
    }
    {
        const bool b = lane & 1;
        r = (b ? r2[1] : r2[0]) + __shfl_xor(b ? r2[0] : r2[1], 1);
    }
    return r;
}
template <int V>
__device__ __forceinline__ int c11_red_index(int lane) {
    int idx = ((lane >> 1) & 1) * 2 + (lane & 1);
    if (V >= 8) idx += ((lane >> 2) & 1) * 4;
    if (V == 16) idx += ((lane >> 3) & 1) * 8;
    return idx;
}

// pre(x) on one fragment: 8 consecutive input channels of one pixel, per-channel coefficients in registers
__device__ __forceinline__ bf16x8_t c11_prologue(bf16x8_t f, const float* sc, const float* sh, const float* ng) {
    const uint4 v = __builtin_bit_cast(uint4, f);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
        lo = lo * sc[2 * i] + sh[2 * i];
        hi = hi * sc[2 * i + 1] + sh[2 * i + 1];
        lo = fmaxf(lo, 0.f) + ng[2 * i] * fminf(lo, 0.f);          // none: ng = 1, relu: 0, prelu: slope
        hi = fmaxf(hi, 0.f) + ng[2 * i + 1] * fminf(hi, 0.f);
        o[i] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
    }
    return __builtin_bit_cast(bf16x8_t, make_uint4(o[0], o[1], o[2], o[3]));
}

}  // namespace ledn
