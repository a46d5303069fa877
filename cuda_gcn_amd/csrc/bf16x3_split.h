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


// ---- vector-memory loads as inline asm (one hand-kept count covers them; hipcc neither sees nor moves nor waits for them) -----
// Rules measured in round 5 (dense_bf16x3.h, "COUNTED WAITS"): `s_waitcnt vmcnt(N)` before the use of a load is safe when N is
// the number of YOUNGER LOADS of this wave (stores and LDS-DMA pieces, older or younger, may be ignored: they are in the count
// on both sides); the destination registers of a load must be tied ("+v") to the wait that precedes their first use — and, for
// loads whose data is never used, to a final vmcnt(0) — or hipcc reuses them while the load is in flight.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 bx_make_rsrc(const void *p, uint32_t bytes) {
    const uint64_t q = (uint64_t)(uintptr_t)p;
    return (u32x4){(uint32_t)q, (uint32_t)(q >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
// dst = 4 bytes at rsrc.base + voff + soff; past the descriptor's byte count: 0
__device__ __forceinline__ void bx_bload4(float &dst, uint32_t voff, u32x4 rsrc, uint32_t soff) {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// dst = 16 bytes at rsrc.base + voff + OFF (OFF < 4096)
template <int OFF>
__device__ __forceinline__ void bx_bload16(f32x4 &dst, uint32_t voff, u32x4 rsrc) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(dst) : "v"(voff), "s"(rsrc), "n"(OFF) : "memory");
}
