// common.h — internals shared by the gfx950 kernels of libgcnhip.so.
// Wave = 64 lanes everywhere in this directory (CDNA4); nothing here is
// written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gcnhip.h"

#define GCNHIP_TRY(expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return (int)_e;              \
    } while (0)
#define GCNHIP_LAUNCH_CHECK() GCNHIP_TRY(hipGetLastError())

constexpr int WAVE = 64;

struct gcnhip_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int n_cu;
    // scratch for block-level partial reductions (xent, sumsq, adam sumsq)
    float *red_f;       // [RED_SLOTS * 4]
    int32_t *red_i;     // [RED_SLOTS * 4]
    uint32_t *ticket;   // arrival counters for last-block reductions
    // split-K slabs for the dense weight-gradient GEMMs
    float *slab;
    size_t slab_bytes;
};
constexpr int RED_SLOTS = 4096;

struct gcnhip_graph {
    int n_rows, n_cols, nnz;
    int *indptr;        // [n_rows+1]
    int *indices;       // [nnz]
    float *coef;        // [nnz]
    // long-row splitting (rows above SPLIT_EDGES are cut into segments)
    int n_tasks;        // number of (row, e0, e1) tasks; 0 => one task per row
    int4 *tasks;        // {row, e_begin, e_end, partial_slot or -1}
    int n_split_rows;   // rows that own partial slots
    int4 *split_rows;   // {row, first_slot, n_slots, 0}
    float *partials;    // [n_slots * part_ld]
    int part_ld;
    int n_slots;
};

struct gcnhip_feat {
    int n_rows, n_cols;
    int64_t nnz;
    bool dense;
    int *indptr;        // [n_rows+1] (kept for dense too: row r starts at r*n_cols)
    int *indices;       // [nnz]; NULL when dense
    float *values;      // [nnz] pristine X
    // CSC view for the weight gradient (sparse X only)
    int *csc_ptr;       // [n_cols+1]
    int *csc_row;       // [nnz] source row of each entry
    int *csc_pos;       // [nnz] position jj in CSR order (selects value + dropout decision)
};

// ---- Philox4x32-10 (Salmon et al., SC'11): counter-based, so the dropout
// decision of an element depends only on (seed, epoch, global element index).
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// threshold of the reference: int(p * MY_RAND_MAX) with float arithmetic
// (src/seq/module.cpp:211); keep <=> (int)r31 >= threshold.
__host__ __device__ inline int dropout_threshold(float p) { return (int)(p * (float)0x7fffffff); }

// keep decisions of the 4 elements j0..j0+3 (j0 % 4 == 0) -> bit i set = keep
__device__ inline uint32_t keep4(uint64_t quad, uint32_t epoch, uint64_t seed, int thr) {
    uint32_t r[4];
    philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), epoch, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) bits |= ((int)(r[i] & 0x7fffffffu) >= thr ? 1u : 0u) << i;
    return bits;
}
__device__ inline bool keep1(uint64_t j, uint32_t epoch, uint64_t seed, int thr) {
    uint32_t r[4];
    const uint64_t quad = j >> 2;
    philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), epoch, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return (int)(r[j & 3] & 0x7fffffffu) >= thr;
}

// keep decisions of V (1, 2 or 4) consecutive elements starting at j, j % V == 0:
// they share one Philox block.  Bit s set = keep element j + s.
template <int V>
__device__ inline uint32_t keepv(uint64_t j, uint32_t epoch, uint64_t seed, int thr) {
    uint32_t r[4];
    const uint64_t quad = j >> 2;
    philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), epoch, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int w0 = (int)(j & 3);
    uint32_t bits = 0;
#pragma unroll
    for (int s = 0; s < V; s++) bits |= ((int)(r[(w0 + s) & 3] & 0x7fffffffu) >= thr ? 1u : 0u) << s;
    return bits;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }
static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
