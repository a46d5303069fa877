// bf16x3_split.h — the exact three-plane bf16 split of f32 operands and the MFMA it feeds (shared by dense_bf16x3.h: the
// first-layer products, and class_bf16x3.h: the class-layer products).  See dense_bf16x3.h for the arithmetic and its error.
#pragma once
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

#define MFMA_BF16(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a_), (b_), (c_), 0, 0, 0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));


typedef float f32x16 __attribute__((ext_vector_type(16)));

// two f32 -> their bf16 roundings (one v_cvt_pk_bf16_f32) as a packed word
__device__ __forceinline__ uint32_t bx_cvt2(float a, float b) {
    const bf16x2_t h = __builtin_convertvector((float2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, h);
}
// (a, b) -> word i of the three planes: a = hi + mid + lo exactly (see the header)
__device__ __forceinline__ void bx_split2(float a, float b, uint32_t &h, uint32_t &m, uint32_t &l) {
    // (v_pk_add_f32 beside MFMAs is slower than two scalar subtractions: cdna guide, "anti-lever")
    h = bx_cvt2(a, b);
    float ra = a - __uint_as_float(h << 16);
    asm("" : "+v"(ra));                                      // opaque: keeps hipcc from packing the two subtractions into v_pk_add_f32
    const float rb = b - __uint_as_float(h & 0xFFFF0000u);
    m = bx_cvt2(ra, rb);
    float sa = ra - __uint_as_float(m << 16);
    asm("" : "+v"(sa));
    const float sb = rb - __uint_as_float(m & 0xFFFF0000u);
    l = bx_cvt2(sa, sb);
}
struct BxPlanes { uint32_t w[3][4]; };                       // [plane][word]: 8 bf16 per plane
__device__ __forceinline__ bf16x8 bx_plane(const BxPlanes &P, int p) {
    const uint4 q = make_uint4(P.w[p][0], P.w[p][1], P.w[p][2], P.w[p][3]);
    return __builtin_bit_cast(bf16x8, q);
}

