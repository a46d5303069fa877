// graphsum.hip — CSR row-gather  out[r,:] = sum_e coef(e) * in[col(e),:]
// (the reference's GraphSum forward AND backward: src/seq/module.cpp:83-119,
//  src/cuda/cuda_kernel.cu:126-162).
//
// Bound: gather bandwidth (0.5 FLOP/B).  Design for gfx950:
//  * one wave64 per row task; the wave reads 64 (index, coef) pairs with one
//    coalesced load each and hands them to its lane groups by cross-lane
//    shuffle, so the only per-edge memory instruction is the feature-row load;
//  * a feature row is read with 16-byte lane loads: L lanes per row slice,
//    64/L rows per wave instruction (d=128: 16 lanes per 64-float slice, 4 rows /
//    1 KiB per instruction); gather_chunk issues GS_U (4; 2 past the Infinity
//    Cache) row loads per lane group before the first is used, from a
//    wave-uniform loop so that no load sits inside a per-lane branch;
//  * partial sums of the lane groups are combined with wave shuffles (no LDS,
//    no atomics) — results are bitwise reproducible run to run;
//  * rows longer than SPLIT_EDGES are cut into segments processed by separate
//    waves (heavy segments dispatched first) and summed in segment order by a
//    small second kernel, so a hub row does not serialise the tail;
//  * XCD-aware launch for rows of whole 256-byte pieces: the columns are cut into
//    64-float (256-byte) slices and each slice is bound to the workgroups of one XCD
//    group (blockIdx % 8), so every XCD's private 4 MiB L2 caches 1/slices of the
//    table (d=128: half of it) and the (index, coef) stream is re-read once per slice;
//  * the ReLU + dropout of the first layer is an optional store epilogue.
// The reference launches one block per row with `dim` threads and does a
// global read-modify-write per edge (cuda_kernel.cu:126-143).
#include "dense_kernels.h"
#include <algorithm>
#include <string.h>
#include <stdlib.h>

struct GsArgs {
    const int *indptr, *indices;
    const float *coef;
    const int4 *tasks;
    int n_tasks, n_rows, nnz;
    size_t table_bytes;      // n_cols * ld_in * 4: decides the batch depth (cache-resident or HBM regime)
    const float *in;
    const uint16_t *in_bf;   // bf16 copy of the gathered table (opt-in storage format); row stride ld_in values
    float *out;
    float *partials;
    int ld_in, ld_out, part_ld, dim;
    int n_slices;     // >= 1: the columns are cut into n_slices slices of L*4 floats, one slice per XCD group
    int bounds[9];          // task range [bounds[g], bounds[g+1]) of XCD group g (equal edge counts)
    // epilogue
    int fuse, training, thr;
    float scale;
    uint64_t seed, elem_offset;
    const uint32_t *d_epoch;
    const uint8_t *keep_mask;
    const uint32_t *row_bits;   // optional: bit j == 0 -> row j of `in` is all zero and is not read
    const uint32_t *out_bits;   // optional: bit r == 0 -> nobody reads row r of `out`: it is not computed (left untouched)
    uint32_t *pos_bits;         // optional (fuse, dim % 32 == 0, vector kernel): bit (c & 31) of pos_bits[r * wpr + (c >> 5)] =
    int wpr;                    // (out[r, c] > 0) AFTER the ReLU/dropout epilogue — the mask their backward needs, so that it
                                // does not have to read the activations again (gcnhip_matmul_bwd_fused_bits)
    // in-kernel segment sum (vector kernel): the wave that finishes a split row's LAST outstanding segment adds the row's
    // partials in segment order and runs the store epilogue itself — no second launch.  NULL: a finalize launch follows.
    const int2 *slot_info;      // [slot] = {first slot of the row, segments of the row}
    uint32_t *seg_count;        // [first_slot * 8 + column slice] arrivals; zero before and after every launch
    int n_slots_bytes;          // size of `partials` in bytes (buffer descriptor of the write-through stores)
    int accumulate;             // 1: out[r,:] = out[r,:] + sum (the second of two operators that share the rows of `out`:
                                // the remote-column part of a row-partitioned aggregation, gcnhip_graphsum_part)
    // factored coefficients (gcnhip_graphsum_ex): coef == NULL -> every edge counts 1 and the row's total is multiplied by
    // post[row] (NULL: by nothing) before the epilogue.  The input rows then hold dinv(col) * x.
    const float *post;
    // loss epilogue (gcnhip_gs_loss, gcnhip.h): xe_truth == NULL -> off.  dim <= 64: one lane group holds the row.
    const int32_t *xe_truth;
    float *xe_grad; int xe_ld_grad; int xe_training;
    float xe_count;
    const float *xe_grad_scale;
    float *xe_terms;
};

__device__ inline bool row_wanted(const GsArgs &a, int row) {
    return !a.out_bits || ((a.out_bits[row >> 5] >> (row & 31)) & 1u);
}

__device__ inline float4 f4_fma(float c, float4 v, float4 a) {
    a.x += c * v.x; a.y += c * v.y; a.z += c * v.z; a.w += c * v.w;
    return a;
}
__device__ inline float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ inline float4 f4_shfl_xor(float4 v, int m) {
    return make_float4(__shfl_xor(v.x, m, WAVE), __shfl_xor(v.y, m, WAVE), __shfl_xor(v.z, m, WAVE), __shfl_xor(v.w, m, WAVE));
}

// ReLU (module.cpp:179-181: keep = x > 0) then dropout on one float4 of row r
__device__ inline float4 relu_dropout4(float4 v, const GsArgs &a, int64_t r, int col0) {
    float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = x[i] > 0.f ? x[i] : 0.f;
    if (a.training) {
        const uint32_t epoch = a.d_epoch ? *a.d_epoch : 0u;
        const int64_t e0 = r * a.dim + col0;          // element index inside this matrix
        uint32_t bits;
        if (a.keep_mask) {
            bits = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) bits |= (col0 + i < a.dim && a.keep_mask[e0 + i] ? 1u : 0u) << i;
        } else {
            const uint64_t j0 = a.elem_offset + (uint64_t)e0;
            if ((j0 & 3) == 0) {
                bits = keep4(j0 >> 2, epoch, a.seed, a.thr);
            } else {
                bits = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) bits |= (keep1(j0 + i, epoch, a.seed, a.thr) ? 1u : 0u) << i;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] *= (bits >> i & 1u) ? a.scale : 0.f;   // module.cpp:216
    }
    return make_float4(x[0], x[1], x[2], x[3]);
}

// ---- loss epilogue (gcnhip_gs_loss) ------------------------------------------------------------------------------------
// The arithmetic of xent_lane_kernel (xent.hip; CrossEntropyLoss::forward, module.cpp:124-161), statement for statement,
// on a row that sits in a wave's registers instead of in memory: max (order-free), exp of the shifted logits, their sum
// LEFT TO RIGHT (v_readlane of column after column into a wave-uniform chain of adds: the reference's order and
// xent_lane_kernel's), the loss term, the accuracy test, the gradient quad of each lane.  Same bits as the loss kernel run
// on the stored row.
// Layout A (vector kernel): lane l of group 0 holds columns 4l..4l+3 in z; every lane of the wave calls.
template <int L>
__device__ __forceinline__ void xent_row_epilogue4(const GsArgs &a, int row, float4 z, int lane, int l, int g, int col0) {
#pragma clang fp contract(off)
    const int t = a.xe_truth[row];                          // wave-uniform
    const int nv4 = (a.dim + 3) >> 2;
    float *gr = a.xe_grad ? a.xe_grad + (size_t)row * a.xe_ld_grad : nullptr;
    if (t < 0) {                                            // not scored: zero gradient row, zero terms (module.cpp:129,132)
        if (a.xe_training && gr && g == 0 && l < nv4) *reinterpret_cast<float4 *>(gr + col0) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane == 0) *reinterpret_cast<float2 *>(a.xe_terms + 2 * (size_t)row) = make_float2(0.f, 0.f);
        return;
    }
    float x[4] = {z.x, z.y, z.z, z.w};
    float mx = -1e30f;                                      // module.cpp:135
#pragma unroll
    for (int i = 0; i < 4; i++) if (col0 + i < a.dim) mx = fmaxf(mx, x[i]);
#pragma unroll
    for (int m = 1; m < L; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, WAVE));
    mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx)));       // lane 0 is in group 0
    const int ti = t & 3;
    const float sel = ti == 0 ? x[0] : (ti == 1 ? x[1] : (ti == 2 ? x[2] : x[3]));
    const float tv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sel), t >> 2));
    const bool correct = !(mx > tv);                        // gcn.cpp:88-93
    float ex[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        x[i] -= mx;                                         // module.cpp:140
        ex[i] = col0 + i < a.dim ? expf(x[i]) : 0.f;
    }
    float se = 0.f;
    for (int q = 0; q < nv4; q++) {                         // columns 4q .. 4q+3 live in lane q (group 0): left to right
#pragma unroll
        for (int i = 0; i < 4; i++) se += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex[i]), q));
    }
    const float term = logf(se) - (tv - mx);                // module.cpp:143
    if (lane == 0) *reinterpret_cast<float2 *>(a.xe_terms + 2 * (size_t)row) = make_float2(term, correct ? 1.f : 0.f);
    if (a.xe_training && gr && g == 0 && l < nv4) {
        const float gs = a.xe_grad_scale ? a.xe_grad_scale[row] : 1.f;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float p = ex[i] / se;                           // module.cpp:147
            if (col0 + i == t) p = (float)((double)p - 1.0);
            o[i] = col0 + i < a.dim ? (a.xe_grad_scale ? (p / a.xe_count) * gs : p / a.xe_count) : 0.f;   // module.cpp:157
        }
        *reinterpret_cast<float4 *>(gr + col0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// Layout B (finalize kernel, split rows): lane c of the block's first wave holds column c in v; that whole wave calls.
__device__ __forceinline__ void xent_row_epilogue1(const GsArgs &a, int row, float v, int lane) {
#pragma clang fp contract(off)
    const int t = a.xe_truth[row];
    const int nv4 = (a.dim + 3) >> 2;
    float *gr = a.xe_grad ? a.xe_grad + (size_t)row * a.xe_ld_grad : nullptr;
    if (t < 0) {
        if (a.xe_training && gr && lane < 4 * nv4) gr[lane] = 0.f;
        if (lane == 0) *reinterpret_cast<float2 *>(a.xe_terms + 2 * (size_t)row) = make_float2(0.f, 0.f);
        return;
    }
    float mx = lane < a.dim ? fmaxf(-1e30f, v) : -1e30f;
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, WAVE));
    const float tv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), t));
    const bool correct = !(mx > tv);
    const float x = v - mx;
    const float ex = lane < a.dim ? expf(x) : 0.f;
    float se = 0.f;
    for (int j = 0; j < 4 * nv4; j++) se += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex), j));
    const float term = logf(se) - (tv - mx);
    if (lane == 0) *reinterpret_cast<float2 *>(a.xe_terms + 2 * (size_t)row) = make_float2(term, correct ? 1.f : 0.f);
    if (a.xe_training && gr && lane < 4 * nv4) {
        float p = ex / se;
        if (lane == t) p = (float)((double)p - 1.0);
        const float gs = a.xe_grad_scale ? a.xe_grad_scale[row] : 1.f;
        gr[lane] = lane < a.dim ? (a.xe_grad_scale ? (p / a.xe_count) * gs : p / a.xe_count) : 0.f;
    }
}

constexpr int GS_U = 4;        // row loads in flight per lane group when the table is cache resident (bf16 kernel: always)
// One chunk of <= 64 edges whose (index, coef) pairs sit in the wave's lanes: acc += sum over the chunk, lane group g
// taking edges g, g + G, ... in order (the order every form of the kernel keeps, so all of them agree bit for bit).
// NT: the row loads carry the non-temporal hint (the line is not to be kept in L2 / Infinity Cache at the expense of others)
template <int L, int GS_U, bool NT = false>
__device__ __forceinline__ float4 gather_chunk(const GsArgs &a, const float *in, int my_idx, float my_c, int cnt, int g, float4 acc) {
    auto ldrow = [&](const float *p) __attribute__((always_inline)) -> float4 {
        if (NT) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
            return make_float4(v[0], v[1], v[2], v[3]);
        }
        return *reinterpret_cast<const float4 *>(p);
    };
    constexpr int G = WAVE / L;
    const int iters = (cnt + G - 1) / G;
    // GS_U row loads per lane group are in flight together whenever the next GS_U iterations are all real edges
    // (a wave-uniform test, so the loads need no predicate and nothing has to wait inside a branch).  Until round
    // 2 this was a `#pragma unroll 8` over a loop with the load inside a per-lane branch: the compiler kept ONE
    // load in flight per wave — a 100-edge row was 13 dependent round trips — and the kernel was bound by
    // latency x resident waves (4 instead of 8 waves per SIMD: 1.08 -> 1.87 ms).  Same order of the sum.
    int k = 0;
    if (!a.row_bits) {
        for (; (k + GS_U) * G <= cnt; k += GS_U) {
            float4 v[GS_U];
            float cc[GS_U];
#pragma unroll
            for (int u = 0; u < GS_U; u++) {
                const int src = (k + u) * G + g;
                const int j = __shfl(my_idx, src, WAVE);
                cc[u] = __shfl(my_c, src, WAVE);
                v[u] = ldrow(in + (size_t)j * a.ld_in);
            }
#pragma unroll
            for (int u = 0; u < GS_U; u++) acc = f4_fma(cc[u], v[u], acc);
        }
    }
    // the tail of a row, and rows read through an input mask: the same batches, with the lanes that have no edge
    // (or an edge whose row is known to be zero) reading the chunk's first neighbour instead and keeping their sum
    const int j_safe = __shfl(my_idx, 0, WAVE);
    for (; k < iters; k += GS_U) {
        float4 v[GS_U];
        float cc[GS_U];
        bool on[GS_U];
#pragma unroll
        for (int u = 0; u < GS_U; u++) {
            const int src = (k + u) * G + g;
            const int j = __shfl(my_idx, src & 63, WAVE);
            cc[u] = __shfl(my_c, src & 63, WAVE);
            on[u] = src < cnt && cc[u] != 0.f;         // padded lanes and known-zero rows contribute nothing
            v[u] = ldrow(in + (size_t)(on[u] ? j : j_safe) * a.ld_in);
        }
#pragma unroll
        for (int u = 0; u < GS_U; u++) {
            const float4 n = f4_fma(cc[u], v[u], acc);
            acc.x = on[u] ? n.x : acc.x; acc.y = on[u] ? n.y : acc.y; acc.z = on[u] ? n.z : acc.z; acc.w = on[u] ? n.w : acc.w;
        }
    }
    return acc;
}

// L lanes per feature row (float4 each), G = 64/L rows per wave instruction.
// SLICED: the launch binds one column slice to each XCD group (it only changes the block -> (tasks, columns) mapping; the
// flag is a template argument so that profiles name the hidden-width launches apart from the class-width ones).
template <int L, int U = GS_U, bool SLICED = false, bool NT = false>
__global__ __launch_bounds__(256) void graphsum_vec_kernel(GsArgs a) {
    constexpr int G = WAVE / L;
    const int lane = threadIdx.x & 63;
    // XCD-aware mapping (speed only, never correctness): workgroups are dealt round-robin over the
    // 8 XCDs, so blockIdx % 8 names the group of blocks that share an L2.  Each group gets
    //  (a) one column slice (wide rows): its 4 MiB L2 then sees 1/n_slices of the gathered table;
    //  (b) a CONTIGUOUS range of the task list, so rows that are neighbours in the node order —
    //      the same community after reordering — are gathered through the same L2.
    int t, cslice;
    {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        cslice = SLICED ? xcd % a.n_slices : blockIdx.y;
        const int g_id = SLICED ? xcd / a.n_slices : xcd;  // which share of the tasks
        t = a.bounds[g_id] + q * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (t >= a.bounds[g_id + 1]) return;               // wave-uniform
    }
    int row, e0, e1, slot;
    if (a.n_tasks) {
        const int4 tk = a.tasks[t];
        row = tk.x; e0 = tk.y; e1 = tk.z; slot = tk.w;
    } else {
        row = t; e0 = a.indptr[t]; e1 = a.indptr[t + 1]; slot = -1;
    }
    if (!row_wanted(a, row)) return;                       // wave-uniform
    const int g = lane / L, l = lane % L;
    const int col0 = (cslice * L + l) * 4;                  // first column of this lane's float4
    const bool active = col0 < a.dim;
    const float *in = a.in + (active ? col0 : 0);           // lanes past the last column read (and discard) columns 0..3
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = e0; base < e1; base += WAVE) {
        const int cnt = min(WAVE, e1 - base);
        int my_idx = 0;
        float my_c = 0.f;
        if (lane < cnt) {
            my_idx = a.indices[base + lane];
            // > 0 for every real edge.  Factored operator (a.coef == NULL, wave-uniform): every edge counts 1 and no coefficient
            // stream is read — the saving is the stream (measured 0.756 -> 0.720 ms at Reddit scale); instantiations that also
            // drop the hand-out shuffle and the multiply measured the same 0.720 ms and were not kept
            my_c = a.coef ? a.coef[base + lane] : 1.f;
            if (a.row_bits && !((a.row_bits[my_idx >> 5] >> (my_idx & 31)) & 1u)) my_c = 0.f;
        }
        acc = gather_chunk<L, U, NT>(a, in, my_idx, my_c, cnt, g, acc);
    }
#pragma unroll
    for (int m = L; m < WAVE; m <<= 1) acc = f4_add(acc, f4_shfl_xor(acc, m));
#ifndef GCNHIP_EXPERIMENTS
    if (slot >= 0) {                                        // a finalize launch will add the segments
#else
    if (slot >= 0 && !a.seg_count) {                        // a finalize launch will add the segments (seg_count: the in-launch sum, an experiment)
#endif
        if (g == 0 && active) *reinterpret_cast<float4 *>(a.partials + (size_t)slot * a.part_ld + col0) = acc;
        return;
    }
#ifdef GCNHIP_EXPERIMENTS
    if (slot >= 0) {
        // Hand-off between workgroups on any XCDs (cdna guide, Guideline 16): the partial leaves WRITE-THROUGH (sc1: no
        // release fence, which would write back this XCD's whole dirty L2), the wave drains its stores, one lane takes a
        // ticket; the wave that draws the last ticket invalidates its CU's L1 (agent-scope acquire) and reads every
        // segment's partial — in segment order, so the row's sum has the bits the finalize kernel gives it.
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(a.partials, 0, a.n_slots_bytes, 0x00020000);
        if (g == 0 && active) {
            const v4u raw = {__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)};
            __builtin_amdgcn_raw_buffer_store_b128(raw, prsrc, (int)(((uint32_t)slot * (uint32_t)a.part_ld + (uint32_t)col0) * 4u), 0, 16 /* sc1 */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int2 info = a.slot_info[slot];
        uint32_t *cnt = a.seg_count + (size_t)info.x * 8 + cslice;
        unsigned prev = 0;
        if (lane == 0) prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // (the write-through store + drain above
                                                                                                              //  make the release cheap: nothing of this wave is left dirty)
        prev = __builtin_amdgcn_readfirstlane(prev);
        if ((int)prev != info.y - 1) return;                // other segments of this row are still on their way
        if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // as the next launch expects it
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g == 0 && active) {
            if (a.accumulate) {                            // (what the first operator left) + p0 + p1 + ..., as the finalize kernel adds
                const float *o = a.out + (size_t)row * a.ld_out + col0;
                float x[4] = {0.f, 0.f, 0.f, 0.f};
                for (int i = 0; i < 4 && col0 + i < a.dim; i++) x[i] = o[i];
                acc = make_float4(x[0], x[1], x[2], x[3]);
            }
            const float *pp = a.partials + (size_t)info.x * a.part_ld + col0;
            int k = 0;
            for (; k + 4 <= info.y; k += 4) {               // four partials in flight; left-to-right sum
                const float4 p0 = *reinterpret_cast<const float4 *>(pp + (size_t)k * a.part_ld);
                const float4 p1 = *reinterpret_cast<const float4 *>(pp + (size_t)(k + 1) * a.part_ld);
                const float4 p2 = *reinterpret_cast<const float4 *>(pp + (size_t)(k + 2) * a.part_ld);
                const float4 p3 = *reinterpret_cast<const float4 *>(pp + (size_t)(k + 3) * a.part_ld);
                acc = f4_add(acc, p0); acc = f4_add(acc, p1); acc = f4_add(acc, p2); acc = f4_add(acc, p3);
            }
            for (; k < info.y; k++) acc = f4_add(acc, *reinterpret_cast<const float4 *>(pp + (size_t)k * a.part_ld));
        }
    }
#endif
    uint32_t nib = 0;                                       // this lane's four (out > 0) bits, at their place in the row's word
    if (g == 0 && active) {
        float *o = a.out + (size_t)row * a.ld_out + col0;
        if (a.accumulate && slot < 0) {                    // what the first operator left in this row, then this one's terms
            if (col0 + 4 <= a.dim) {
                acc = f4_add(*reinterpret_cast<const float4 *>(o), acc);
            } else {
                float x[4] = {acc.x, acc.y, acc.z, acc.w};
                for (int i = 0; col0 + i < a.dim; i++) x[i] = o[i] + x[i];
                acc = make_float4(x[0], x[1], x[2], x[3]);
            }
        }
        if (a.post) { const float ps = a.post[row]; acc.x *= ps; acc.y *= ps; acc.z *= ps; acc.w *= ps; }
        if (a.fuse) acc = relu_dropout4(acc, a, row, col0);
        if (col0 + 4 <= a.dim) {
            *reinterpret_cast<float4 *>(o) = acc;
        } else {                                           // ragged tail: dim % 4 != 0
            const float x[4] = {acc.x, acc.y, acc.z, acc.w};
            for (int i = 0; col0 + i < a.dim; i++) o[i] = x[i];
        }
        nib = ((acc.x > 0.f ? 1u : 0u) | (acc.y > 0.f ? 2u : 0u) | (acc.z > 0.f ? 4u : 0u) | (acc.w > 0.f ? 8u : 0u)) << (4 * (l & 7));
    }
    if constexpr (!SLICED && L <= 16)
        if (a.xe_truth) xent_row_epilogue4<L>(a, row, acc, lane, l, g, col0);     // wave-uniform; one column chunk (launch site)
    if (L >= 8 && a.pos_bits) {                             // wave-uniform; dim % 32 == 0 (launch site): 8 lanes hold one word
        nib |= __shfl_xor(nib, 1, WAVE);
        nib |= __shfl_xor(nib, 2, WAVE);
        nib |= __shfl_xor(nib, 4, WAVE);
        if (g == 0 && active && (l & 7) == 0) a.pos_bits[(size_t)row * a.wpr + (col0 >> 5)] = nib;
    }
}

#ifdef GCNHIP_EXPERIMENTS   // measured-slower variants: compiled only by `make EXPERIMENTS=1`
// ---- EXPERIMENT: persistent form that also keeps the next chunk's (index, coef) pairs in flight -------------------
// A wave walks tasks t, t + stride, ... of its XCD group's range; while the rows of chunk k are gathered, the pairs of
// chunk k+1 (of the same task, or the first chunk of the wave's next task, whose record was fetched a task earlier) are
// on their way, so one round trip per batch of rows remains.  Same lane groups, same edge order, same reduction tree:
// bit-identical results.  Measured SLOWER than a wave per task (1.15 vs 0.85 ms, hidden width, Reddit scale) and
// therefore not the default (GCNHIP_GS_PIPE selects it): see the launch site.
template <int L>
__global__ __launch_bounds__(256, 8) void graphsum_pipe_kernel(GsArgs a) {
    constexpr int G = WAVE / L;
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int cslice = a.n_slices > 1 ? xcd % a.n_slices : blockIdx.y;
    const int g_id = xcd / a.n_slices;
    const int t_end = a.bounds[g_id + 1];
    const int stride = (gridDim.x >> 3) * 4;                // waves of this XCD
    // wave-uniform by construction; saying so lets the task records live in scalar registers
    int t = __builtin_amdgcn_readfirstlane(a.bounds[g_id] + q * 4 + (int)(threadIdx.x >> 6));
    if (t >= t_end) return;
    const int g = lane / L, l = lane % L;
    const int col0 = (cslice * L + l) * 4;
    const bool active = col0 < a.dim;
    const float *in = a.in + (active ? col0 : 0);
    const int last_edge = a.nnz - 1;

    // every prefetch below is an UNCONDITIONAL load from a clamped address: a load inside a branch makes the compiler
    // wait for it at the end of that branch, which would put the round trip back on the critical path
    int4 tk = a.tasks[t];
    int t_nx = t + stride;
    int4 tk_nx = a.tasks[min(t_nx, t_end - 1)];
    int base = tk.y;
    int cnt = min(WAVE, tk.z - base);
    int my_idx = a.indices[min(base + lane, last_edge)];
    float my_c = a.coef[min(base + lane, last_edge)];
    if (a.row_bits && !((a.row_bits[my_idx >> 5] >> (my_idx & 31)) & 1u)) my_c = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (;;) {
        // ---- the item after this one: next chunk of the task, or first chunk of the wave's next task
        const bool last = base + WAVE >= tk.z;
        const bool more = !last || t_nx < t_end;
        const int nbase = last ? tk_nx.y : base + WAVE;
        const int nend = last ? tk_nx.z : tk.z;
        const int ncnt = more ? min(WAVE, nend - nbase) : 0;
        const int n_idx = a.indices[min(max(nbase + lane, 0), last_edge)];      // in flight while this chunk's rows are gathered
        float n_c = a.coef[min(max(nbase + lane, 0), last_edge)];
        // ---- this chunk
        acc = gather_chunk<L, GS_U>(a, in, my_idx, my_c, cnt, g, acc);
        if (a.row_bits && !((a.row_bits[n_idx >> 5] >> (n_idx & 31)) & 1u)) n_c = 0.f;
        if (last) {
#pragma unroll
            for (int m = L; m < WAVE; m <<= 1) acc = f4_add(acc, f4_shfl_xor(acc, m));
            if (g == 0 && active) {
                if (tk.w >= 0) {
                    *reinterpret_cast<float4 *>(a.partials + (size_t)tk.w * a.part_ld + col0) = acc;
                } else {
                    if (a.fuse) acc = relu_dropout4(acc, a, tk.x, col0);
                    float *o = a.out + (size_t)tk.x * a.ld_out + col0;
                    if (col0 + 4 <= a.dim) {
                        *reinterpret_cast<float4 *>(o) = acc;
                    } else {
                        const float x[4] = {acc.x, acc.y, acc.z, acc.w};
                        for (int i = 0; col0 + i < a.dim; i++) o[i] = x[i];
                    }
                }
            }
            if (!more) break;
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            tk = tk_nx;
            t_nx += stride;
            tk_nx = a.tasks[min(t_nx, t_end - 1)];             // needed at the new task's last chunk, one round trip away at least
        }
        base = nbase; cnt = ncnt; my_idx = n_idx; my_c = n_c;
    }
}

// ---- packed rows, exact f32 (the backward of the hidden layer) ---------------------------------------
// The gathered matrix is dH1 = mask . (dZ0 . W2^T): three quarters of it are zeros whose positions are known
// (ReLU and dropout of H1).  Its producer (gemm_rowstream, PACK) writes every 64-column half of a row as one
// 128-byte slot — 64-bit mask + the masked-in values in column order — so an edge costs ONE line per half
// instead of two, with the f32 values untouched.  Edges are split as in graphsum_vec_kernel<16> (the dense gather
// at widths that are multiples of 64): edge k*4+g of a 64-edge chunk goes to group g of FOUR, the four sums are
// combined by the same xor tree — so every output element sees the same non-zero terms in the same order and the
// result is bit-identical to the dense gather (adding c * 0 never changes a sum that started at +0).  A wave covers a
// PAIR of halves: lane groups 0-3 (8 lanes each) take the even half, groups 4-7 the odd one.
// A group loads its slot as 8 x 16 bytes (one request), parks it in the wave's LDS scratch, reads the mask back
// (broadcast) and then each lane picks the values of its 8 columns: position = popcount of the mask below them.
__global__ __launch_bounds__(256) void graphsum_packed_kernel(GsArgs a, const uint32_t *__restrict__ slots, int halves) {
    __shared__ uint4 stage[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int t, pair;
    {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        pair = a.n_slices > 1 ? xcd % a.n_slices : blockIdx.y;
        const int g_id = xcd / a.n_slices;
        t = a.bounds[g_id] + q * 4 + wave;
        if (t >= a.bounds[g_id + 1]) return;               // wave-uniform
    }
    int row, e0, e1, slot;
    if (a.n_tasks) {
        const int4 tk = a.tasks[t];
        row = tk.x; e0 = tk.y; e1 = tk.z; slot = tk.w;
    } else {
        row = t; e0 = a.indptr[t]; e1 = a.indptr[t + 1]; slot = -1;
    }
    const int g8 = lane >> 3, g = g8 & 3, l = lane & 7;
    const int half = pair * 2 + (g8 >> 2);
    const bool live = half < halves;                       // an odd number of halves: the last wave's upper groups idle
    const uint32_t *sbase = slots + (size_t)(live ? half : 0) * 32 + l * 4;
    const size_t row_stride = (size_t)halves * 32;
    const uint32_t *mine = reinterpret_cast<const uint32_t *>(&stage[wave][g8 * 8]);
    const uint32_t sh8 = 8u * (l & 3);                                            // this lane's byte inside its mask word
    const uint32_t below_lo = l < 4 ? (1u << (8 * l)) - 1u : 0xFFFFFFFFu;         // mask bits of the columns before this lane's
    const uint32_t below_hi = l < 4 ? 0u : (1u << (8 * (l - 4))) - 1u;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0.f;
    for (int base = e0; base < e1; base += WAVE) {
        const int cnt = min(WAVE, e1 - base);
        int my_idx = 0;
        float my_c = 0.f;
        if (lane < cnt) {
            my_idx = a.indices[base + lane];
            my_c = a.coef[base + lane];
        }
        const int iters = (cnt + 3) >> 2;
        // PACKED_U edges per lane group in flight: their slot loads are issued together, then each is parked in LDS and
        // decoded without branches (a branch per column serialises the loop: one load in flight per wave)
        constexpr int PACKED_U = 4;
        for (int k0 = 0; k0 < iters; k0 += PACKED_U) {
            uint4 piece[PACKED_U];
            int jj[PACKED_U];
            float cc[PACKED_U];
#pragma unroll
            for (int u = 0; u < PACKED_U; u++) {
                const int src = (k0 + u) * 4 + g;              // >= 64 wraps in the shuffle; masked by `on`
                jj[u] = __shfl(my_idx, src & 63, WAVE);
                cc[u] = __shfl(my_c, src & 63, WAVE);
                const bool on = src < cnt && live;
                piece[u] = make_uint4(0u, 0u, 0u, 0u);         // empty mask: contributes nothing
                if (on) piece[u] = *reinterpret_cast<const uint4 *>(sbase + (size_t)jj[u] * row_stride);
            }
#pragma unroll
            for (int u = 0; u < PACKED_U; u++) {
                stage[wave][lane] = piece[u];
                __builtin_amdgcn_wave_barrier();
                // 32-bit arithmetic only; per column: rank below it (and + bcnt), LDS address, read, bit -> all-ones,
                // and, fmac.  A masked-out column adds c * (+0), exactly what the dense gather adds there.
                const uint32_t mlo = mine[0], mhi = mine[1];
                const int n_half = __popc(mlo) + __popc(mhi);
                const bool fits = n_half <= PACK_CAP;
                const uint32_t bits8 = fits ? ((l < 4 ? mlo : mhi) >> sh8) & 0xFFu : 0u;
                const int off0 = fits ? 2 + __popc(mlo & below_lo) + __popc(mhi & below_hi) : 0;
                const float c = cc[u];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int off = off0 + __popc(bits8 & ((1u << i) - 1u));          // <= 31 when the half fits
                    const uint32_t keep = (uint32_t)(((int32_t)(bits8 << (31 - i))) >> 31);   // bit i -> 0 or ~0
                    acc[i] = fmaf(c, __uint_as_float(mine[off] & keep), acc[i]);
                }
                if (!fits) {                                   // this half did not fit its slot: dense image (rare)
                    const float *d = a.in + (size_t)jj[u] * a.ld_in + half * 64 + 8 * l;
                    const float4 v0 = *reinterpret_cast<const float4 *>(d), v1 = *reinterpret_cast<const float4 *>(d + 4);
                    acc[0] = fmaf(c, v0.x, acc[0]); acc[1] = fmaf(c, v0.y, acc[1]); acc[2] = fmaf(c, v0.z, acc[2]); acc[3] = fmaf(c, v0.w, acc[3]);
                    acc[4] = fmaf(c, v1.x, acc[4]); acc[5] = fmaf(c, v1.y, acc[5]); acc[6] = fmaf(c, v1.z, acc[6]); acc[7] = fmaf(c, v1.w, acc[7]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int m = 8; m < 32; m <<= 1) acc[i] += __shfl_xor(acc[i], m, WAVE);
    if (g == 0 && live) {
        const int col0 = half * 64 + 8 * l;
        float *o = slot >= 0 ? a.partials + (size_t)slot * a.part_ld + col0 : a.out + (size_t)row * a.ld_out + col0;
        *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// slots -> dense image (tests, introspection): halves that fit are written from their slot, the others are already dense
__global__ __launch_bounds__(256) void rowpack_expand_kernel(const uint32_t *__restrict__ slots, int halves, int64_t n_half_rows,
                                                             float *__restrict__ dense, int ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_half_rows) return;
    const uint32_t *s = slots + i * 32;
    const uint64_t m = (uint64_t)s[0] | ((uint64_t)s[1] << 32);
    if (__popcll(m) > PACK_CAP) return;
    float *d = dense + (i / halves) * ld + (i % halves) * 64;
    int off = 2;
    for (int c = 0; c < 64; c++) d[c] = ((m >> c) & 1ull) ? __uint_as_float(s[off++]) : 0.f;
}

#endif  // GCNHIP_EXPERIMENTS

// ---- bf16 table, f32 accumulate (opt-in storage format) ------------------------
// The gathered rows are stored as bfloat16 (round-to-nearest-even of the f32 values, made by
// gcnhip_f32_to_bf16), so a row of d values is d*2 bytes: half the 128-byte lines per edge.  coef, the
// sum and the output stay f32.  L lanes per row slice, 8 values (16 bytes) per lane; slices of 64 columns
// = one line, bound to XCD groups like the f32 kernel.
__device__ inline void bf8_fma(float c, const uint4 v, float4 &lo, float4 &hi) {
    lo.x += c * __uint_as_float(v.x << 16); lo.y += c * __uint_as_float(v.x & 0xFFFF0000u);
    lo.z += c * __uint_as_float(v.y << 16); lo.w += c * __uint_as_float(v.y & 0xFFFF0000u);
    hi.x += c * __uint_as_float(v.z << 16); hi.y += c * __uint_as_float(v.z & 0xFFFF0000u);
    hi.z += c * __uint_as_float(v.w << 16); hi.w += c * __uint_as_float(v.w & 0xFFFF0000u);
}

template <int L>
__global__ __launch_bounds__(256) void graphsum_bf16_kernel(GsArgs a) {
    constexpr int G = WAVE / L;
    const int lane = threadIdx.x & 63;
    int t, cslice;
    {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        cslice = a.n_slices > 1 ? xcd % a.n_slices : blockIdx.y;
        const int g_id = xcd / a.n_slices;
        t = a.bounds[g_id] + q * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (t >= a.bounds[g_id + 1]) return;
    }
    int row, e0, e1, slot;
    if (a.n_tasks) {
        const int4 tk = a.tasks[t];
        row = tk.x; e0 = tk.y; e1 = tk.z; slot = tk.w;
    } else {
        row = t; e0 = a.indptr[t]; e1 = a.indptr[t + 1]; slot = -1;
    }
    if (!row_wanted(a, row)) return;                       // wave-uniform
    const int g = lane / L, l = lane % L;
    const int col0 = (cslice * L + l) * 8;
    const bool active = col0 < a.dim;
    const uint16_t *in = a.in_bf + (active ? col0 : 0);     // lanes past the last column read (and discard) columns 0..7
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    for (int base = e0; base < e1; base += WAVE) {
        const int cnt = min(WAVE, e1 - base);
        int my_idx = 0;
        float my_c = 0.f;
        if (lane < cnt) {
            my_idx = a.indices[base + lane];
            my_c = a.coef[base + lane];
            if (a.row_bits && !((a.row_bits[my_idx >> 5] >> (my_idx & 31)) & 1u)) my_c = 0.f;
        }
        const int iters = (cnt + G - 1) / G;
        // batches of GS_U row loads in flight, as gather_chunk does for the f32 table
        int k = 0;
        if (!a.row_bits) {
            for (; (k + GS_U) * G <= cnt; k += GS_U) {
                uint4 v[GS_U];
                float cc[GS_U];
#pragma unroll
                for (int u = 0; u < GS_U; u++) {
                    const int src = (k + u) * G + g;
                    const int j = __shfl(my_idx, src, WAVE);
                    cc[u] = __shfl(my_c, src, WAVE);
                    v[u] = *reinterpret_cast<const uint4 *>(in + (size_t)j * a.ld_in);
                }
#pragma unroll
                for (int u = 0; u < GS_U; u++) bf8_fma(cc[u], v[u], lo, hi);
            }
        }
        const int j_safe = __shfl(my_idx, 0, WAVE);
        for (; k < iters; k += GS_U) {
            uint4 v[GS_U];
            float cc[GS_U];
            bool on[GS_U];
#pragma unroll
            for (int u = 0; u < GS_U; u++) {
                const int src = (k + u) * G + g;
                const int j = __shfl(my_idx, src & 63, WAVE);
                cc[u] = __shfl(my_c, src & 63, WAVE);
                on[u] = src < cnt && cc[u] != 0.f;
                v[u] = *reinterpret_cast<const uint4 *>(in + (size_t)(on[u] ? j : j_safe) * a.ld_in);
            }
#pragma unroll
            for (int u = 0; u < GS_U; u++) {
                float4 nlo = lo, nhi = hi;
                bf8_fma(cc[u], v[u], nlo, nhi);
                lo.x = on[u] ? nlo.x : lo.x; lo.y = on[u] ? nlo.y : lo.y; lo.z = on[u] ? nlo.z : lo.z; lo.w = on[u] ? nlo.w : lo.w;
                hi.x = on[u] ? nhi.x : hi.x; hi.y = on[u] ? nhi.y : hi.y; hi.z = on[u] ? nhi.z : hi.z; hi.w = on[u] ? nhi.w : hi.w;
            }
        }
    }
#pragma unroll
    for (int m = L; m < WAVE; m <<= 1) { lo = f4_add(lo, f4_shfl_xor(lo, m)); hi = f4_add(hi, f4_shfl_xor(hi, m)); }
    if (g == 0 && active) {
        if (slot >= 0) {
            float *pp = a.partials + (size_t)slot * a.part_ld + col0;
            *reinterpret_cast<float4 *>(pp) = lo;
            *reinterpret_cast<float4 *>(pp + 4) = hi;
        } else {
            if (a.fuse) { lo = relu_dropout4(lo, a, row, col0); hi = relu_dropout4(hi, a, row, col0 + 4); }
            float *o = a.out + (size_t)row * a.ld_out + col0;
            const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (col0 + 8 <= a.dim && (a.ld_out & 3) == 0) {
                *reinterpret_cast<float4 *>(o) = lo;
                *reinterpret_cast<float4 *>(o + 4) = hi;
            } else {
                for (int i = 0; i < 8 && col0 + i < a.dim; i++) o[i] = x[i];
            }
        }
    }
}

// f32 -> bf16 (round to nearest even; NaN stays NaN), columns dim..ld_dst-1 of every row written as 0
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *__restrict__ src, int ld_src, uint16_t *__restrict__ dst, int ld_dst,
                                                          int64_t rows, int dim) {
    const int per_row = ld_dst / 8;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int c0 = (int)(i % per_row) * 8;
        uint32_t h[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float f = c0 + k < dim ? src[r * ld_src + c0 + k] : 0.f;
            const uint32_t u = __float_as_uint(f);
            h[k] = (f != f) ? ((u >> 16) | 0x40u) : ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
        }
        uint4 v;
        v.x = (h[0] & 0xFFFFu) | (h[1] << 16); v.y = (h[2] & 0xFFFFu) | (h[3] << 16);
        v.z = (h[4] & 0xFFFFu) | (h[5] << 16); v.w = (h[6] & 0xFFFFu) | (h[7] << 16);
        *reinterpret_cast<uint4 *>(dst + r * ld_dst + c0) = v;
    }
}

template <int L>
static void launch_bf16(GsArgs a, const int (*xb)[9], hipStream_t s) {
    const int ychunks = ceil_div(a.dim, L * 8);
    const bool sliced = ychunks > 1 && 8 % ychunks == 0;
    a.n_slices = sliced ? ychunks : 1;
    const int G = 8 / a.n_slices;
    const int lg = G == 8 ? 3 : (G == 4 ? 2 : (G == 2 ? 1 : 0));
    int max_blocks = 1;
    for (int k = 0; k <= 8; k++) a.bounds[k] = xb[lg][k];
    for (int k = 0; k < G; k++) max_blocks = std::max(max_blocks, ceil_div(a.bounds[k + 1] - a.bounds[k], 4));
    graphsum_bf16_kernel<L><<<dim3(max_blocks * 8, sliced ? 1 : ychunks), 256, 0, s>>>(a);
}

// Any ld / alignment: scalar loads, lane l owns columns l, l+L, ...
template <int L>
__global__ __launch_bounds__(256) void graphsum_scalar_kernel(GsArgs a) {
    constexpr int G = WAVE / L;
    constexpr int MAXC = 4;                                // columns per lane per pass
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nt = a.n_tasks ? a.n_tasks : a.n_rows;
    if (t >= nt) return;
    int row, e0, e1, slot;
    if (a.n_tasks) {
        const int4 tk = a.tasks[t];
        row = tk.x; e0 = tk.y; e1 = tk.z; slot = tk.w;
    } else {
        row = t; e0 = a.indptr[t]; e1 = a.indptr[t + 1]; slot = -1;
    }
    if (!row_wanted(a, row)) return;                       // wave-uniform
    const int g = lane / L, l = lane % L;
    for (int cbase = 0; cbase < a.dim; cbase += L * MAXC) {
        float acc[MAXC] = {0.f, 0.f, 0.f, 0.f};
        for (int base = e0; base < e1; base += WAVE) {
            const int cnt = min(WAVE, e1 - base);
            int my_idx = 0;
            float my_c = 0.f;
            if (lane < cnt) {
                my_idx = a.indices[base + lane];
                my_c = a.coef[base + lane];
                if (a.row_bits && !((a.row_bits[my_idx >> 5] >> (my_idx & 31)) & 1u)) my_c = 0.f;
            }
            const int iters = (cnt + G - 1) / G;
#pragma unroll 4
            for (int k = 0; k < iters; k++) {
                const int src = k * G + g;
                const int j = __shfl(my_idx, src, WAVE);
                const float c = __shfl(my_c, src, WAVE);
                const float *p = a.in + (size_t)j * a.ld_in + cbase + l;
                if (src < cnt && c != 0.f) {
#pragma unroll
                    for (int q = 0; q < MAXC; q++)
                        if (cbase + l + q * L < a.dim) acc[q] += c * p[q * L];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < MAXC; q++) {
#pragma unroll
            for (int m = L; m < WAVE; m <<= 1) acc[q] += __shfl_xor(acc[q], m, WAVE);
        }
        if (g == 0) {
#pragma unroll
            for (int q = 0; q < MAXC; q++) {
                const int col = cbase + l + q * L;
                if (col >= a.dim) continue;
                float v = acc[q];
                if (slot >= 0) {
                    a.partials[(size_t)slot * a.part_ld + col] = v;
                } else {
                    if (a.accumulate) v = a.out[(size_t)row * a.ld_out + col] + v;
                    if (a.fuse) {
                        v = v > 0.f ? v : 0.f;
                        if (a.training) {
                            const int64_t e = (int64_t)row * a.dim + col;
                            const bool keep = a.keep_mask ? a.keep_mask[e] != 0
                                                          : keep1(a.elem_offset + (uint64_t)e, a.d_epoch ? *a.d_epoch : 0u, a.seed, a.thr);
                            v *= keep ? a.scale : 0.f;
                        }
                    }
                    a.out[(size_t)row * a.ld_out + col] = v;
                }
            }
        }
    }
}

// sum the segment partials of each split row in segment order, apply the epilogue
__global__ __launch_bounds__(256) void graphsum_finalize_kernel(GsArgs a, const int4 *split_rows, int n_split_rows) {
    const int s = blockIdx.x;
    if (s >= n_split_rows) return;
    const int4 sr = split_rows[s];
    const int row = sr.x, first = sr.y, ns = sr.z;
    if (!row_wanted(a, row)) return;
    for (int c0 = 0; c0 < a.dim; c0 += blockDim.x) {        // whole waves stay in the loop: the ballot below needs them
        const int col = c0 + threadIdx.x;
        float v = 0.f;
        if (col < a.dim) {
            v = a.accumulate ? a.out[(size_t)row * a.ld_out + col] : 0.f;
            // four partials in flight per batch (unconditional loads from clamped slots; the same left-to-right sum)
            const float *pp = a.partials + (size_t)first * a.part_ld + col;
            int k = 0;
            for (; k + 4 <= ns; k += 4) {
                const float p0 = pp[(size_t)k * a.part_ld], p1 = pp[(size_t)(k + 1) * a.part_ld];
                const float p2 = pp[(size_t)(k + 2) * a.part_ld], p3 = pp[(size_t)(k + 3) * a.part_ld];
                v += p0; v += p1; v += p2; v += p3;
            }
            {
                const float p0 = pp[(size_t)min(k, ns - 1) * a.part_ld], p1 = pp[(size_t)min(k + 1, ns - 1) * a.part_ld];
                const float p2 = pp[(size_t)min(k + 2, ns - 1) * a.part_ld];
                if (k < ns) v += p0;
                if (k + 1 < ns) v += p1;
                if (k + 2 < ns) v += p2;
            }
            if (a.post) v *= a.post[row];
            if (a.fuse) {
                v = v > 0.f ? v : 0.f;
                if (a.training) {
                    const int64_t e = (int64_t)row * a.dim + col;
                    const bool keep = a.keep_mask ? a.keep_mask[e] != 0
                                                  : keep1(a.elem_offset + (uint64_t)e, a.d_epoch ? *a.d_epoch : 0u, a.seed, a.thr);
                    v *= keep ? a.scale : 0.f;
                }
            }
            a.out[(size_t)row * a.ld_out + col] = v;
        }
        if (a.xe_truth && threadIdx.x < WAVE) xent_row_epilogue1(a, row, v, threadIdx.x);   // dim <= 64 (launch site): the first wave holds the row
        if (a.pos_bits) {                                   // 64 consecutive columns per wave: two words of the row
            const uint64_t b = __ballot(col < a.dim && v > 0.f);
            const int w0 = (c0 + (int)(threadIdx.x & ~63u)) >> 5;
            if ((threadIdx.x & 63) == 0 && w0 < a.wpr) a.pos_bits[(size_t)row * a.wpr + w0] = (uint32_t)b;
            if ((threadIdx.x & 63) == 32 && w0 + 1 < a.wpr) a.pos_bits[(size_t)row * a.wpr + w0 + 1] = (uint32_t)(b >> 32);
        }
    }
}

template <int L>
static void launch_vec(GsArgs &a, const int (*xb)[9], const gcnhip_ctx *c) {
    hipStream_t s = c->stream;
    const int ychunks = ceil_div(a.dim, L * 4);
    const bool sliced = ychunks > 1 && 8 % ychunks == 0;   // XCD-sliced columns (1-D grid)
    a.n_slices = sliced ? ychunks : 1;
    if (ychunks > 8) { a.seg_count = nullptr; a.slot_info = nullptr; }   // one arrival counter per (split row, column chunk), 8 per row
    const int G = 8 / a.n_slices;                          // XCD groups that share the task list
    const int lg = G == 8 ? 3 : (G == 4 ? 2 : (G == 2 ? 1 : 0));
    int max_blocks = 1;
    for (int k = 0; k <= 8; k++) a.bounds[k] = xb[lg][k];
    for (int k = 0; k < G; k++) max_blocks = std::max(max_blocks, ceil_div(a.bounds[k + 1] - a.bounds[k], 4));
    // EXPERIMENT (GCNHIP_GS_PIPE): the persistent form that also prefetches the next chunk's indices.  Measured slower
    // than a wave per task once the row loads are batched (1.15 vs 0.85 ms at Reddit scale): fresh waves arriving in
    // task order keep the XCD's window of active rows tight, statically strided persistent waves drift apart.
#ifdef GCNHIP_EXPERIMENTS
    const bool pipe = c->opt.gs_pipe != 0;
    if (pipe && a.n_tasks && !a.out_bits && !a.accumulate && !a.pos_bits && a.coef) {
        a.seg_count = nullptr; a.slot_info = nullptr;      // the experiment keeps the finalize launch
        const int per_xcd = std::min(max_blocks, 32 * 8);
        graphsum_pipe_kernel<L><<<dim3(per_xcd * 8, sliced ? 1 : ychunks), 256, 0, s>>>(a);
        return;
    }
#endif
    // Batch depth by regime.  A table that fits the 256 MiB Infinity Cache is gathered with 4 row loads in flight per lane
    // group (latency-bound otherwise: 1.08 -> 0.85 ms at Reddit scale).  Past it the kernel is HBM-bound and 2 is the
    // optimum (R-MAT scale 21, 1 GiB table, d = 128: 5.33 / 5.16 / 5.51 ms with 1 / 2 / 4 in flight).
    const int force_u = c->opt.gs_u;
    const int u = force_u ? force_u : (a.table_bytes > ((size_t)256 << 20) ? 2 : 4);
    const dim3 grid(max_blocks * 8, sliced ? 1 : ychunks);
#ifdef GCNHIP_EXPERIMENTS
    const bool nt_all = c->opt.gs_nt != 0;                            // EXPERIMENT: every row load non-temporal
    if (sliced && nt_all && L == 16 && a.coef) { graphsum_vec_kernel<16, 4, true, true><<<grid, 256, 0, s>>>(a); return; }
#endif
    if (sliced) {
        if (u >= 4) graphsum_vec_kernel<L, 4, true><<<grid, 256, 0, s>>>(a);
        else if (u >= 2) graphsum_vec_kernel<L, 2, true><<<grid, 256, 0, s>>>(a);
        else graphsum_vec_kernel<L, 1, true><<<grid, 256, 0, s>>>(a);
    } else {
        if (u >= 4) graphsum_vec_kernel<L, 4><<<grid, 256, 0, s>>>(a);
        else if (u >= 2) graphsum_vec_kernel<L, 2><<<grid, 256, 0, s>>>(a);
        else graphsum_vec_kernel<L, 1><<<grid, 256, 0, s>>>(a);
    }
}
template <int L>
static void launch_scalar(const GsArgs &a, int nt, hipStream_t s) {
    graphsum_scalar_kernel<L><<<ceil_div(nt, 4), 256, 0, s>>>(a);
}

static int graphsum_impl(gcnhip_ctx *c, const gcnhip_graph *g, const float *in, int ld_in, float *out, int ld_out,
                         int dim, int fuse, int training, float p, uint64_t seed, const uint32_t *d_epoch,
                         uint64_t elem_offset, const uint8_t *keep_mask, const uint32_t *row_bits = nullptr,
                         const uint16_t *in_bf = nullptr, const uint32_t *out_bits = nullptr,
                         const gcnhip_rowset *rs = nullptr, int accumulate = 0, uint32_t *pos_bits = nullptr, int wpr = 0, int scaling = 0,
                         const gcnhip_gs_loss *loss = nullptr) {
    if (!c || !g || (!in && !in_bf) || !out || dim <= 0 || ld_in < dim || ld_out < dim) return -1;
    if (scaling < 0 || scaling > 3) return -1;
    if (accumulate && in_bf) return -1;                     // the bf16 kernel has no accumulating store
    if (rs && rs->owner != g)                                // a subset brings task lists and segment slots of ITS adjacency object
        return gcnhip_fail("gcnhip_graphsum*: the row subset was registered on another adjacency object");
    if (in_bf && (ld_in % 8 != 0 || !aligned16(in_bf))) return -1;
    if (g->n_rows == 0 || (rs && rs->n_tasks == 0)) return 0;
    // the segment scratch of split rows is sized when the object is built (256 columns) or by
    // gcnhip_graph_reserve_width; a launch never allocates, synchronises or touches the object
    if (g->n_slots && g->part_ld < (dim + 7) / 8 * 8)
        return gcnhip_fail("gcnhip_graphsum*: dim is wider than the split-row scratch of this adjacency object; call "
                           "gcnhip_graph_reserve_width(ctx, g, dim) once before the first launch (restricted objects: on them too)");
    GsArgs a;
    a.indptr = g->indptr; a.indices = g->indices; a.coef = g->coef;
    // a registered row subset brings its own compacted task list (same order, same segment slots)
    a.tasks = rs ? rs->tasks : g->tasks; a.n_tasks = rs ? rs->n_tasks : g->n_tasks; a.n_rows = g->n_rows; a.nnz = g->nnz;
    a.table_bytes = (size_t)g->n_cols * ld_in * (in_bf ? 2 : 4);
    const int (*xb)[9] = rs ? rs->bounds : g->bounds;
    const int4 *split_rows = rs ? rs->split_rows : g->split_rows;
    const int n_split_rows = rs ? rs->n_split_rows : g->n_split_rows;
    a.in = in; a.in_bf = in_bf; a.out = out; a.partials = g->partials;
    a.ld_in = ld_in; a.ld_out = ld_out; a.part_ld = g->part_ld; a.dim = dim;
    a.fuse = fuse; a.training = training; a.thr = dropout_threshold(p);
    a.scale = 1 / (1 - p);                                  // module.cpp:212
    a.seed = seed; a.elem_offset = elem_offset; a.d_epoch = d_epoch; a.keep_mask = keep_mask;
    a.row_bits = row_bits;
    a.out_bits = out_bits;
    a.accumulate = accumulate;
    a.pos_bits = pos_bits; a.wpr = wpr;
    a.slot_info = nullptr; a.seg_count = nullptr; a.n_slots_bytes = 0;
    a.post = nullptr;
    const int nt = a.n_tasks ? a.n_tasks : g->n_rows;
    const bool vec = (ld_in % 4 == 0) && (ld_out % 4 == 0) && aligned16(in) && aligned16(out);
    a.xe_truth = nullptr; a.xe_grad = nullptr; a.xe_ld_grad = 0; a.xe_training = 0; a.xe_count = 1.f; a.xe_grad_scale = nullptr; a.xe_terms = nullptr;
    if (loss) {
        const int w4 = (dim + 3) / 4 * 4;
        if (in_bf || !vec || dim > 64 || fuse || pos_bits || !loss->truth || !loss->row_terms || loss->count <= 0 || ((uintptr_t)loss->row_terms & 7))
            return gcnhip_fail("gcnhip_graphsum_ex: the loss epilogue needs f32 rows that are 16-byte aligned, dim <= 64, no relu_dropout, truth, row_terms and count > 0");
        if (loss->training && (!loss->grad || loss->ld_grad < w4 || loss->ld_grad % 4 != 0 || !aligned16(loss->grad)))
            return gcnhip_fail("gcnhip_graphsum_ex: the loss epilogue writes whole 16-byte quads of the gradient row: ld_grad % 4 == 0, ld_grad >= round_up(dim, 4)");
        a.xe_truth = loss->truth; a.xe_grad = loss->training ? loss->grad : nullptr; a.xe_ld_grad = loss->ld_grad; a.xe_training = loss->training ? 1 : 0;
        a.xe_count = (float)loss->count; a.xe_grad_scale = loss->grad_row_scale; a.xe_terms = loss->row_terms;
    }
    if (scaling) {
        // the factored operator runs in the 16-byte-row f32 kernel only (what the model's layouts always are)
        if (in_bf || !vec) return gcnhip_fail("gcnhip_graphsum_ex: scaling != 0 needs f32 rows that are 16-byte aligned (ld % 4 == 0)");
        a.coef = nullptr;
        a.post = scaling == 1 ? g->dinv_row : (scaling == 2 ? g->dinv2_row : nullptr);
    }
    if (pos_bits && !(fuse && !in_bf && vec && dim % 32 == 0 && wpr * 32 >= dim))
        return gcnhip_fail("gcnhip_graphsum_relu_dropout_bits: needs the fused epilogue, 16-byte aligned rows, dim % 32 == 0 and words_per_row * 32 >= dim");
    const int d4 = (dim + 3) / 4;
    a.n_slices = 1;
    if (in_bf) {
        const int d8 = (dim + 7) / 8;                       // 16-byte pieces per row
        if (d8 <= 1) launch_bf16<1>(a, xb, c->stream);
        else if (d8 <= 2) launch_bf16<2>(a, xb, c->stream);
        else if (d8 <= 4) launch_bf16<4>(a, xb, c->stream);
        else if (d8 % 16 == 0) launch_bf16<16>(a, xb, c->stream);   // 128-column (two-line) slices as in the f32 kernel: 305 -> 322 epochs/s
        else launch_bf16<8>(a, xb, c->stream);               // 64-column (one line) slices, one per XCD group when 8 % slices == 0
    }
    // GCNHIP_GS_FOLD (opt-in, round 3): the vector kernel adds the segments of a split row itself (the last segment to finish
    // does) and no finalize launch follows.  Same bits; measured no faster at 5 K segments and slower as their number grows
    // (every segment wave keeps its slot through a store drain and an atomic round trip: 0.835 vs 0.774 ms at 20 K segments),
    // so the default stays the fire-and-forget partial store plus one 8 us launch.  (Context option gs_fold.)
#ifdef GCNHIP_EXPERIMENTS
    const bool fold = c->opt.gs_fold != 0;
#else
    const bool fold = false;
#endif
    if (!in_bf && vec && n_split_rows && g->seg_count && fold) {
        a.slot_info = g->slot_info; a.seg_count = g->seg_count;
        a.n_slots_bytes = (int)std::min<size_t>((size_t)g->n_slots * g->part_ld * sizeof(float), 0x7FFFFFFFu);
    }
    const int gl = c->opt.gs_l;                             // narrower column slices than 64 floats (8: 32 floats, 4: 16 floats), round 5
    if (in_bf) {
    } else if (vec && !loss && (gl == 8 || (gl == 4 && !pos_bits)) && dim % (gl * 4) == 0 && dim / (gl * 4) > 1 && 8 % (dim / (gl * 4)) == 0) {
        // More slices = a smaller share of the table per XCD's L2 (what a structure-free graph's hub rows need) against more
        // re-reads of the index stream and shorter requests.  Measured per graph (tools/exp_structure.py); never the default.
        // (16-float slices hold half a mask word per lane group: with pos_bits they fall through to the 64-float slices below)
        if (gl == 8) launch_vec<8>(a, xb, c); else launch_vec<4>(a, xb, c);
    } else if (vec && dim % 64 == 0 && 8 % (dim / 64) == 0) {
        // rows of whole 128-byte lines: 64-float (two-line) column slices, one per XCD group, so each XCD's L2 holds
        // 1/slices of the table and the (index, coef) stream is re-read only once per slice.  Measured at Reddit scale
        // with 4 row loads in flight, 32- / 64- / 128-float slices: d = 64: 0.434 / 0.403 / - ms; d = 128: 0.843 / 0.760 /
        // 0.874; d = 256: 2.36 / 1.77 / 1.78; R-MAT scale 21 (HBM regime), d = 128: 5.20 / 4.57 ms.
        launch_vec<16>(a, xb, c);
    } else if (vec) {
        if (d4 <= 1) launch_vec<1>(a, xb, c);
        else if (d4 <= 2) launch_vec<2>(a, xb, c);
        else if (d4 <= 4) launch_vec<4>(a, xb, c);
        else if (d4 <= 8) launch_vec<8>(a, xb, c);
        else if (d4 <= 16) launch_vec<16>(a, xb, c);
        else if (d4 <= 32) launch_vec<32>(a, xb, c);
        else launch_vec<64>(a, xb, c);
    } else {
        if (dim <= 1) launch_scalar<1>(a, nt, c->stream);
        else if (dim <= 2) launch_scalar<2>(a, nt, c->stream);
        else if (dim <= 4) launch_scalar<4>(a, nt, c->stream);
        else if (dim <= 8) launch_scalar<8>(a, nt, c->stream);
        else if (dim <= 16) launch_scalar<16>(a, nt, c->stream);
        else if (dim <= 32) launch_scalar<32>(a, nt, c->stream);
        else launch_scalar<64>(a, nt, c->stream);
    }
    GCNHIP_LAUNCH_CHECK();
    if (n_split_rows && !a.seg_count) {
        graphsum_finalize_kernel<<<n_split_rows, 256, 0, c->stream>>>(a, split_rows, n_split_rows);
        GCNHIP_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" {

int gcnhip_graphsum(gcnhip_ctx *c, const gcnhip_graph *g, const float *in, int ld_in,
                    float *out, int ld_out, int dim) {
    return graphsum_impl(c, g, in, ld_in, out, ld_out, dim, 0, 0, 0.f, 0, nullptr, 0, nullptr);
}

int gcnhip_graphsum_rowmask(gcnhip_ctx *c, const gcnhip_graph *g, const float *in, int ld_in,
                            float *out, int ld_out, int dim, const uint32_t *in_row_bits) {
    return graphsum_impl(c, g, in, ld_in, out, ld_out, dim, 0, 0, 0.f, 0, nullptr, 0, nullptr, in_row_bits);
}

int gcnhip_graphsum_masked(gcnhip_ctx *c, const gcnhip_graph *g, const float *in, int ld_in,
                           float *out, int ld_out, int dim, const uint32_t *in_row_bits, const uint32_t *out_row_bits) {
    return graphsum_impl(c, g, in, ld_in, out, ld_out, dim, 0, 0, 0.f, 0, nullptr, 0, nullptr, in_row_bits, nullptr, out_row_bits);
}

int gcnhip_graphsum_rowset(gcnhip_ctx *c, const gcnhip_graph *g, const gcnhip_rowset *rows, const float *in, int ld_in,
                           float *out, int ld_out, int dim, const uint32_t *in_row_bits) {
    if (!rows) return -1;
    return graphsum_impl(c, g, in, ld_in, out, ld_out, dim, 0, 0, 0.f, 0, nullptr, 0, nullptr, in_row_bits, nullptr, nullptr, rows);
}

int gcnhip_graphsum_relu_dropout(gcnhip_ctx *c, const gcnhip_graph *g, const float *in, int ld_in,
                                 float *out, int ld_out, int dim, int training, float p,
                                 uint64_t seed, const uint32_t *d_epoch, uint64_t elem_offset,
                                 const uint8_t *keep_mask) {
    if (training && !(p >= 0.f && p < 1.f)) return -1;
    return graphsum_impl(c, g, in, ld_in, out, ld_out, dim, 1, training, p, seed, d_epoch, elem_offset, keep_mask);
}

int gcnhip_graphsum_relu_dropout_bits(gcnhip_ctx *c, const gcnhip_graph *g, const float *in, int ld_in,
                                      float *out, int ld_out, int dim, int training, float p,
                                      uint64_t seed, const uint32_t *d_epoch, uint64_t elem_offset,
                                      const uint8_t *keep_mask, uint32_t *pos_bits, int words_per_row) {
    if (training && !(p >= 0.f && p < 1.f)) return -1;
    if (!pos_bits) return -1;
    return graphsum_impl(c, g, in, ld_in, out, ld_out, dim, 1, training, p, seed, d_epoch, elem_offset, keep_mask, nullptr, nullptr, nullptr,
                         nullptr, 0, pos_bits, words_per_row);
}

int gcnhip_graphsum_part(gcnhip_ctx *c, const gcnhip_graph *g, const gcnhip_rowset *rows, const float *in, int ld_in,
                         float *out, int ld_out, int dim, const uint32_t *in_row_bits, int accumulate,
                         int relu_dropout, int training, float p, uint64_t seed, const uint32_t *d_epoch,
                         uint64_t elem_offset, const uint8_t *keep_mask) {
    if (relu_dropout && training && !(p >= 0.f && p < 1.f)) return -1;
    return graphsum_impl(c, g, in, ld_in, out, ld_out, dim, relu_dropout ? 1 : 0, relu_dropout ? training : 0, relu_dropout ? p : 0.f,
                         seed, d_epoch, elem_offset, keep_mask, in_row_bits, nullptr, nullptr, rows, accumulate ? 1 : 0);
}

int gcnhip_graphsum_ex(gcnhip_ctx *c, const gcnhip_graph *g, const gcnhip_gs_opts *o, const float *in, int ld_in, float *out, int ld_out, int dim) {
    if (!o) return -1;
    if (o->relu_dropout && o->training && !(o->p >= 0.f && o->p < 1.f)) return -1;
    if (o->pos_bits && !o->relu_dropout) return -1;
    return graphsum_impl(c, g, in, ld_in, out, ld_out, dim, o->relu_dropout ? 1 : 0, o->relu_dropout ? o->training : 0, o->relu_dropout ? o->p : 0.f,
                         o->seed, o->d_epoch, o->elem_offset, o->keep_mask, o->in_row_bits, nullptr, nullptr, o->rows, o->accumulate ? 1 : 0,
                         o->pos_bits, o->words_per_row, o->scaling, o->loss);
}

#ifndef GCNHIP_EXPERIMENTS
int gcnhip_rowpack_create(gcnhip_ctx *, gcnhip_rowpack **, int, int) { return gcnhip_fail("this entry point is a measured-slower experiment: build the library with `make EXPERIMENTS=1`"); }
int gcnhip_rowpack_destroy(gcnhip_ctx *, gcnhip_rowpack *) { return 0; }
int gcnhip_rowpack_expand(gcnhip_ctx *, const gcnhip_rowpack *, float *, int) { return gcnhip_fail("this entry point is a measured-slower experiment: build the library with `make EXPERIMENTS=1`"); }
int gcnhip_graphsum_packed(gcnhip_ctx *, const gcnhip_graph *, const gcnhip_rowpack *, const float *, int, float *, int) { return gcnhip_fail("this entry point is a measured-slower experiment: build the library with `make EXPERIMENTS=1`"); }
#else
int gcnhip_rowpack_create(gcnhip_ctx *c, gcnhip_rowpack **out, int rows, int cols) {
    if (!c || !out || rows < 0 || cols < 64 || cols % 64 != 0) return -1;
    GCNHIP_TRY(hipSetDevice(c->device));
    gcnhip_rowpack *p = new gcnhip_rowpack();
    p->rows = rows; p->cols = cols; p->halves = cols / 64; p->slots = nullptr;
    const size_t bytes = (size_t)std::max(rows, 1) * p->halves * 128;
    hipError_t e = hipMalloc((void **)&p->slots, bytes);
    if (e != hipSuccess) { delete p; return (int)e; }
    e = hipMemsetAsync(p->slots, 0, bytes, c->stream);      // empty masks: every row reads as zero until it is written
    if (e != hipSuccess) { hipFree(p->slots); delete p; return (int)e; }
    *out = p;
    return 0;
}
int gcnhip_rowpack_destroy(gcnhip_ctx *c, gcnhip_rowpack *p) {
    if (!p) return 0;
    hipSetDevice(c->device);
    if (p->slots) hipFree(p->slots);
    delete p;
    return 0;
}
int gcnhip_rowpack_expand(gcnhip_ctx *c, const gcnhip_rowpack *p, float *dense, int ld) {
    if (!c || !p || !dense || ld < p->cols) return -1;
    const int64_t n = (int64_t)p->rows * p->halves;
    if (n == 0) return 0;
    rowpack_expand_kernel<<<ceil_div(n, 256), 256, 0, c->stream>>>(p->slots, p->halves, n, dense, ld);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

int gcnhip_graphsum_packed(gcnhip_ctx *c, const gcnhip_graph *g, const gcnhip_rowpack *p, const float *dense, int ld_dense,
                           float *out, int ld_out) {
    if (!c || !g || !p || !dense || !out || p->rows != g->n_cols || ld_dense < p->cols || ld_out < p->cols) return -1;
    if (ld_dense % 4 != 0 || ld_out % 4 != 0 || !aligned16(dense) || !aligned16(out)) return -1;
    if (g->n_rows == 0) return 0;
    if (g->n_slots && g->part_ld < p->cols) return -1;
    GsArgs a;
    memset(&a, 0, sizeof a);
    a.indptr = g->indptr; a.indices = g->indices; a.coef = g->coef;
    a.tasks = g->tasks; a.n_tasks = g->n_tasks; a.n_rows = g->n_rows; a.nnz = g->nnz;
    a.in = dense; a.out = out; a.partials = g->partials;
    a.ld_in = ld_dense; a.ld_out = ld_out; a.part_ld = g->part_ld; a.dim = p->cols;
    const int pairs = (p->halves + 1) / 2;                  // a wave covers two halves
    const bool sliced = pairs > 1 && 8 % pairs == 0;
    a.n_slices = sliced ? pairs : 1;
    const int G = 8 / a.n_slices;
    const int lg = G == 8 ? 3 : (G == 4 ? 2 : (G == 2 ? 1 : 0));
    int max_blocks = 1;
    for (int k = 0; k <= 8; k++) a.bounds[k] = g->bounds[lg][k];
    for (int k = 0; k < G; k++) max_blocks = std::max(max_blocks, ceil_div(a.bounds[k + 1] - a.bounds[k], 4));
    graphsum_packed_kernel<<<dim3(max_blocks * 8, sliced ? 1 : pairs), 256, 0, c->stream>>>(a, p->slots, p->halves);
    GCNHIP_LAUNCH_CHECK();
    if (g->n_split_rows) {
        graphsum_finalize_kernel<<<g->n_split_rows, 256, 0, c->stream>>>(a, g->split_rows, g->n_split_rows);
        GCNHIP_LAUNCH_CHECK();
    }
    return 0;
}

#endif  // GCNHIP_EXPERIMENTS

int gcnhip_f32_to_bf16(gcnhip_ctx *c, const float *src, int ld_src, uint16_t *dst, int ld_dst, int64_t rows, int dim) {
    if (!c || !src || !dst || rows < 0 || dim <= 0 || ld_src < dim || ld_dst < dim || ld_dst % 8 != 0 || !aligned16(dst)) return -1;
    if (rows == 0) return 0;
    int64_t blocks = (rows * (ld_dst / 8) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    f32_to_bf16_kernel<<<(int)blocks, 256, 0, c->stream>>>(src, ld_src, dst, ld_dst, rows, dim);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

int gcnhip_graphsum_bf16(gcnhip_ctx *c, const gcnhip_graph *g, const uint16_t *in_bf16, int ld_in,
                         float *out, int ld_out, int dim, const uint32_t *in_row_bits, const gcnhip_rowset *out_rows,
                         int relu_dropout, int training, float p, uint64_t seed, const uint32_t *d_epoch,
                         uint64_t elem_offset, const uint8_t *keep_mask) {
    if (relu_dropout && training && !(p >= 0.f && p < 1.f)) return -1;
    return graphsum_impl(c, g, nullptr, ld_in, out, ld_out, dim, relu_dropout ? 1 : 0, training, relu_dropout ? p : 0.f, seed, d_epoch,
                         elem_offset, keep_mask, in_row_bits, in_bf16, nullptr, out_rows);
}

}  // extern "C"

GCNHIP_DEFINE_PRELOAD(graphsum, graphsum_finalize_kernel)
