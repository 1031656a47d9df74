// Device-side helpers shared by the gfx950 kernels: 16-bit storage formats and the MFMA wrapper.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dffw_internal.h"

namespace dffw {

typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- 16-bit number formats ---------------------------------------------------------------------
__device__ __forceinline__ uint16_t f2bf(float f) {  // round-to-nearest-even (finite inputs)
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float h2f(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }

template <int PREC>
struct Fmt {
    static constexpr int PARTS = (PREC == P_BF16X3) ? 2 : 1;
    // value of channel c of a pixel whose storage starts at p (layout [part][C])
    static __device__ __forceinline__ float load(const uint16_t *p, int C, int c) {
        if constexpr (PREC == P_BF16X3) return bf2f(p[c]) + bf2f(p[C + c]);
        else if constexpr (PREC == P_FP16) return h2f(p[c]);
        else return bf2f(p[c]);
    }
    static __device__ __forceinline__ void split(float v, uint16_t &hi, uint16_t &lo) {
        if constexpr (PREC == P_BF16X3) {
            hi = f2bf(v);
            lo = f2bf(v - bf2f(hi));
        } else if constexpr (PREC == P_FP16) {
            hi = f2h(v);
            lo = 0;
        } else {
            hi = f2bf(v);
            lo = 0;
        }
    }
    static __device__ __forceinline__ float join(uint16_t hi, uint16_t lo) {
        if constexpr (PREC == P_BF16X3) return bf2f(hi) + bf2f(lo);
        else if constexpr (PREC == P_FP16) return h2f(hi);
        else return bf2f(hi);
    }
    static __device__ __forceinline__ void store(uint16_t *p, int C, int c, float v) {
        uint16_t hi, lo;
        split(v, hi, lo);
        p[c] = hi;
        if constexpr (PARTS == 2) p[C + c] = lo;
    }
};

template <bool F16>
__device__ __forceinline__ f32x4 mma(short8 a, short8 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace dffw
