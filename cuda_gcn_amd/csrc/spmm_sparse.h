// spmm_sparse.h — SparseMatmul on a SPARSE feature matrix (Cora / Citeseer / Pubmed-like X):
//   forward  H0 = X~ . W1   as a CSR row gather of W1 rows          (src/seq/module.cpp:47-61,  cuda_kernel.cu:100-110)
//   backward dW1 = X~^T . dH0 as a CSC gather of dH0 rows            (src/seq/module.cpp:63-77; the reference's CUDA backward,
//                                                                     cuda_kernel.cu:112-122, is a racy scatter)
// Bound: gather bandwidth — per stored value one index, one value and one h-float row of the other operand.  Round 4
// rewrote both kernels after the shipped ISA showed ONE row load in flight per wave (`global_load_dwordx4 ...
// s_waitcnt vmcnt(0)` per iteration: the load sat under a per-lane predicate inside `#pragma unroll 4`, the defect
// graphsum.hip was cured of in round 2):
//  * sp_gather_chunk issues SP_U row loads per lane group before the first is used, from a wave-uniform loop, so no load
//    carries a predicate; the tail of a chunk takes the same batch with idle lanes re-reading the chunk's first row and
//    keeping their sum (same order of every lane group's sum as before: the forward is bit-identical to round 3's);
//  * the backward no longer gives a whole column to ONE wave (Pubmed: 500 waves walking ~2 000 entries each on a 256-CU
//    chip, 62 us): a column is a task of NW waves (1, 4 or 16 by the mean column length), each wave sums a contiguous
//    share and the workgroup adds the NW partials IN WAVE ORDER through LDS — deterministic, no atomics; columns longer
//    than the segment length become several tasks whose partial rows a fold launch adds in order (only then);
//  * spmm_csr_fwd_lds_kernel: W1 staged once per workgroup into LDS when it fits (north_star's "LDS staging of the
//    feature tile"), rows then gathered by ds_read — taken when measured faster (launch site).
#pragma once
#include "common.h"

struct DropSpec {
    int on, thr;
    float scale;
    uint64_t seed, off;
    const uint32_t *d_epoch;
    const uint8_t *keep_mask;
};

__device__ inline float drop_scale(const DropSpec &d, uint64_t e, uint32_t epoch) {
    if (!d.on) return 1.f;
    const bool keep = d.keep_mask ? d.keep_mask[e] != 0 : keep1(d.off + e, epoch, d.seed, d.thr);
    return keep ? d.scale : 0.f;
}

constexpr int SP_U = 4;        // row loads in flight per lane group

template <int V>
__device__ __forceinline__ void sp_load_row(const float *p, float out[V]) {
    if (V == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        out[0] = v.x; out[1 % V] = v.y; out[2 % V] = v.z; out[3 % V] = v.w;
    } else {
        out[0] = p[0];
    }
}

// One chunk of <= 64 stored values whose (row index, coefficient) pairs sit in the wave's lanes:
// acc[0..V) += sum_k c_k * rows[idx_k * ld + 0..V), lane group g taking entries g, g + G, ... in order.
// `rows` already points at this lane's first column.
template <int L, int V, int U>
__device__ __forceinline__ void sp_gather_chunk(const float *rows, int ld, int my_idx, float my_c, int cnt, int g, float acc[V]) {
    constexpr int G = WAVE / L;
    const int iters = (cnt + G - 1) / G;
    int k = 0;
    for (; (k + U) * G <= cnt; k += U) {           // wave-uniform: every lane group has U real entries
        float v[U][V], cc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int src = (k + u) * G + g;
            const int j = __shfl(my_idx, src, WAVE);
            cc[u] = __shfl(my_c, src, WAVE);
            sp_load_row<V>(rows + (size_t)j * ld, v[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < V; i++) acc[i] += cc[u] * v[u][i];
    }
    const int j_safe = __shfl(my_idx, 0, WAVE);
    for (; k < iters; k += U) {                    // the tail: same batch, lanes without an entry read the chunk's first row
        float v[U][V], cc[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int src = (k + u) * G + g;
            const int j = __shfl(my_idx, src & 63, WAVE);
            cc[u] = __shfl(my_c, src & 63, WAVE);
            on[u] = src < cnt;
            sp_load_row<V>(rows + (size_t)(on[u] ? j : j_safe) * ld, v[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < V; i++) {
                const float n = acc[i] + cc[u] * v[u][i];
                acc[i] = on[u] ? n : acc[i];
            }
    }
}

// ------------------------------------------------------------ sparse forward
// one wave per row of X; L lanes (V floats each) per row of W, G = 64 / L rows of W per wave instruction
struct SpFwdArgs {
    const int *indptr, *indices;
    const float *vals, *w;
    float *out;
    int n_rows, ld_w, ld_out, p;
    DropSpec d;
    int relu;                       // store max(x, 0) (module.cpp:175-185 folded into the producer)
    int w_floats;                   // LDS form: floats of W to stage (n_cols * ld_w)
    int rows_per_wave;              // consecutive rows a wave of spmm_csr_fwd_kernel walks (1 .. 32)
    int nnz_bytes;                  // size of indices[] and vals[] in bytes (bounds of the narrow-row kernel's buffer loads)
    int n_slices;                   // SLICED kernel: p = n_slices * L * 4 columns, one slice of L * 4 per XCD group (1, 2, 4 or 8)
};

template <int L, bool VEC>
__device__ __forceinline__ void sp_fwd_row(const SpFwdArgs &a, const float *w, int row, int lane, uint32_t epoch) {
    constexpr int V = VEC ? 4 : 1;
    const int g = lane / L, l = lane % L;
    const int e0 = a.indptr[row], e1 = a.indptr[row + 1];
    for (int cb = 0; cb < a.p; cb += L * V) {
        const int col0 = cb + l * V;
        const bool active = col0 < a.p;
        const float *wp = w + (active ? col0 : 0);          // lanes past the last column read (and discard) the first columns
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; i++) acc[i] = 0.f;
        for (int base = e0; base < e1; base += WAVE) {
            const int cnt = min(WAVE, e1 - base);
            int my_idx = 0;
            float my_v = 0.f;
            if (lane < cnt) {
                my_idx = a.indices[base + lane];
                my_v = a.vals[base + lane] * drop_scale(a.d, (uint64_t)(base + lane), epoch);
            }
            sp_gather_chunk<L, V, SP_U>(wp, a.ld_w, my_idx, my_v, cnt, g, acc);
        }
#pragma unroll
        for (int i = 0; i < V; i++)
#pragma unroll
            for (int m = L; m < WAVE; m <<= 1) acc[i] += __shfl_xor(acc[i], m, WAVE);
        if (g == 0 && active) {
            float *o = a.out + (size_t)row * a.ld_out + col0;
#pragma unroll
            for (int i = 0; i < V; i++)
                if (col0 + i < a.p) o[i] = (a.relu && !(acc[i] > 0.f)) ? 0.f : acc[i];
        }
    }
}

// The default forward: a wave takes a.rows_per_wave consecutive rows.  Their row pointers arrive with one load, and while the
// W rows of one 64-value chunk are being gathered the (index, value) pairs of the NEXT chunk — of the same row or of the
// wave's next row — are already on their way: per row one exposed round trip (the gather) instead of three dependent ones
// (row pointers -> pairs -> W rows), which is what a 50-value row costs when a wave lives for one row (measured: 2 M rows of
// 50 values, h = 16: 0.97 ms with a wave per row).  Same lane groups, same order: bit-identical to sp_fwd_row.
// SLICED (round 5): a W that does not fit an XCD's 4 MiB L2 (F x h x 4 bytes: 25.6 MB at F = 50 000, h = 128) is gathered from
// the Infinity Cache at the uniformly-random-row rate whatever the kernel does (8.8 TB/s of W rows, 6.0 ms for 2 M rows of 50
// values: verdict r04's "fabric traffic 25 x B_sp").  As in the aggregation kernel (graphsum.hip) the columns are cut into
// slices of L * 4 floats and each slice is bound to the workgroups of one XCD group (blockIdx % 8), which also takes a
// contiguous share of the rows: an XCD's L2 then sees 1 / n_slices of W — 6.4 MB of 25.6 at 32-float slices — at the price of
// reading the (index, value) stream n_slices times.  Measured with the aggregation kernel standing in (tools/exp_spmm_gather.py):
// 5.06 / 3.83 / 7.4 ms at 64- / 32- / 16-float slices against 6.04.  Another summation order per row (8 lane groups instead
// of 2 at h = 128): inside the summation bound, not the bits of the unsliced kernel.
template <int L, bool VEC, bool SLICED = false>
__global__ __launch_bounds__(256) void spmm_csr_fwd_kernel(SpFwdArgs a) {
    constexpr int V = VEC ? 4 : 1;
    const int lane = threadIdx.x & 63;
    const int K = a.rows_per_wave;                            // 1 .. 32
    int row0, cslice = 0;
    if (SLICED) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int G = 8 / a.n_slices, g_id = xcd / a.n_slices;     // XCD groups that share the rows; which share
        cslice = xcd % a.n_slices;
        const int units = (a.n_rows + K - 1) / K;
        const int lo = (int)((int64_t)units * g_id / G), hi = (int)((int64_t)units * (g_id + 1) / G);
        const int u = lo + q * 4 + (threadIdx.x >> 6);
        if (u >= hi) return;
        row0 = u * K;
    } else {
        row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * K;
    }
    if (row0 >= a.n_rows) return;
    const uint32_t epoch = (a.d.on && a.d.d_epoch) ? *a.d.d_epoch : 0u;
    if (!SLICED && a.p > L * V) {                             // more than one column pass (p > 256): the plain walk
        for (int r = row0; r < min(a.n_rows, row0 + K); r++) sp_fwd_row<L, VEC>(a, a.w, r, lane, epoch);
        return;
    }
    const int nr = min(K, a.n_rows - row0);
    const int ip = a.indptr[row0 + min(lane, nr)];
    const int g = lane / L, l = lane % L;
    const int col0 = (SLICED ? cslice * L * V : 0) + l * V;
    const bool active = col0 < a.p;
    const float *wp = a.w + (active ? col0 : 0);
    int r = 0;
    int rb = __builtin_amdgcn_readlane(ip, 0), re = __shfl(ip, 1, WAVE);
    re = __builtin_amdgcn_readfirstlane(re);
    int base = rb;
    int cnt = min(WAVE, re - base);
    int cur_idx = 0;
    float cur_v = 0.f;
    if (lane < cnt) {
        cur_idx = a.indices[base + lane];
        cur_v = a.vals[base + lane] * drop_scale(a.d, (uint64_t)(base + lane), epoch);
    }
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; i++) acc[i] = 0.f;
    for (;;) {
        // the step after this one: the row's next chunk, or the first chunk of the wave's next row
        const bool row_done = base + WAVE >= re;
        int nrow = r, nrb = rb, nre = re, nbase = base + WAVE;
        if (row_done) {
            nrow = r + 1;
            if (nrow < nr) {
                nrb = __builtin_amdgcn_readfirstlane(__shfl(ip, nrow, WAVE));
                nre = __builtin_amdgcn_readfirstlane(__shfl(ip, nrow + 1, WAVE));
                nbase = nrb;
            }
        }
        const bool have_next = !row_done || nrow < nr;
        int nxt_idx = 0, ncnt = 0;
        float nxt_v = 0.f;
        if (have_next) {
            ncnt = min(WAVE, nre - nbase);
            if (lane < ncnt) {
                nxt_idx = a.indices[nbase + lane];
                nxt_v = a.vals[nbase + lane] * drop_scale(a.d, (uint64_t)(nbase + lane), epoch);
            }
        }
        if (cnt > 0) sp_gather_chunk<L, V, SP_U>(wp, a.ld_w, cur_idx, cur_v, cnt, g, acc);
        if (row_done) {
#pragma unroll
            for (int i = 0; i < V; i++)
#pragma unroll
                for (int m = L; m < WAVE; m <<= 1) acc[i] += __shfl_xor(acc[i], m, WAVE);
            if (g == 0 && active) {
                float *o = a.out + (size_t)(row0 + r) * a.ld_out + col0;
#pragma unroll
                for (int i = 0; i < V; i++)
                    if (col0 + i < a.p) o[i] = (a.relu && !(acc[i] > 0.f)) ? 0.f : acc[i];
            }
#pragma unroll
            for (int i = 0; i < V; i++) acc[i] = 0.f;
        }
        if (!have_next) break;
        r = nrow; rb = nrb; re = nre; base = nbase; cnt = ncnt; cur_idx = nxt_idx; cur_v = nxt_v;
    }
}

#ifdef GCNHIP_EXPERIMENTS   // W staged in LDS: built, bit-identical, measured slower wherever W fits
// W staged in LDS once per workgroup (16 waves); the workgroup then walks rows wave by wave.  16-byte aligned rows only.
template <int L>
__global__ __launch_bounds__(1024) void spmm_csr_fwd_lds_kernel(SpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sp_ws[];
    for (int i = threadIdx.x * 4; i < a.w_floats; i += 1024 * 4)
        *reinterpret_cast<float4 *>(sp_ws + i) = *reinterpret_cast<const float4 *>(a.w + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t epoch = (a.d.on && a.d.d_epoch) ? *a.d.d_epoch : 0u;
    for (int row = blockIdx.x * 16 + (threadIdx.x >> 6); row < a.n_rows; row += gridDim.x * 16)
        sp_fwd_row<L, true>(a, sp_ws, row, lane, epoch);
}
#endif  // GCNHIP_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------------------------
// The narrow-row kernels (16-byte aligned rows of at most 64 floats: L <= 16 lanes per row — hidden 16 is the reference's
// default).  Round 4 measurement: 2 M rows of 50 values at h = 16 ran at the same 0.95 ms whether W came from L2, from the
// Infinity Cache or from LDS, with 1 or 16 rows per wave: the bound was neither the gather nor a latency chain but the
// LDS crossbar — every `__shfl` is a ds_bpermute, and a 50-value row cost 8 of them to hand (index, value) pairs to the
// lane groups plus 16 for the butterfly over 16 groups x 4 accumulators.  Here
//  * lane group g owns entries g*L .. g*L+L-1 of a 64-entry chunk and READS them itself as one 16-byte buffer load per
//    array (the L lanes of a group read the same address; out-of-range reads return 0 by the buffer's bounds check): no
//    shuffle hands anything out;
//  * dropout decisions are made once per entry in the coalesced layout (lane = entry) and shared as a wave-uniform 64-bit
//    ballot, so no lane group repeats a Philox block;
//  * the sum over the lane groups is a HALVING exchange (each step a lane keeps half of its values and sends the other
//    half: 2 + 1 bpermutes for 4 accumulators) finished by DPP row rotations for the steps inside a 16-lane row: 3
//    bpermutes per row instead of 24.
// A row's sum is then associated differently than in the general kernels above (entries contiguous per lane group instead
// of strided; tree over groups in another order): within the tests' summation-order bound, deterministic run to run.
typedef unsigned int sp_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int sp_v2u __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ void sp_load_entries(__amdgpu_buffer_rsrc_t rs, int first, unsigned out[L]) {
    const int off = first * 4;
    if constexpr (L == 1) {
        out[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
    } else if constexpr (L == 2) {
        const sp_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
        out[0] = v[0]; out[1] = v[1];
    } else {
#pragma unroll
        for (int q = 0; q < L / 4; q++) {
            const sp_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16 * q, 0, 0);
            out[4 * q] = v[0]; out[4 * q + 1] = v[1]; out[4 * q + 2] = v[2]; out[4 * q + 3] = v[3];
        }
    }
}

// acc[0..4) += sum over this lane group's entries of the chunk (entry e = g*L + k is real when e < cnt)
template <int L>
__device__ __forceinline__ void sp_gather_entries(const float *rows, int ld, const unsigned idx[L], const unsigned val[L], int cnt, int g,
                                                  uint64_t keep, float scale, bool drop, float acc[4]) {
    constexpr int U = L < 4 ? L : 4;
#pragma unroll
    for (int k0 = 0; k0 < L; k0 += U) {
        float4 v[U];
        float c[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = g * L + k0 + u;
            on[u] = e < cnt;
            const float x = __uint_as_float(val[k0 + u]);
            c[u] = drop ? (((keep >> e) & 1ull) ? x * scale : 0.f) : x;
            v[u] = *reinterpret_cast<const float4 *>(rows + (size_t)(on[u] ? (int)idx[k0 + u] : 0) * ld);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc[0] = on[u] ? acc[0] + c[u] * v[u].x : acc[0];
            acc[1] = on[u] ? acc[1] + c[u] * v[u].y : acc[1];
            acc[2] = on[u] ? acc[2] + c[u] * v[u].z : acc[2];
            acc[3] = on[u] ? acc[3] + c[u] * v[u].w : acc[3];
        }
    }
}

template <int CTRL>
__device__ __forceinline__ float sp_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}

// Sum of x[0..4) over the G = 64/L lane groups.  On return x[0..nv) are totals of columns l*4 + colsel .. + nv - 1 of the
// row, and `writer` names the lanes that hold them exactly once.
template <int L>
__device__ __forceinline__ void sp_reduce_groups(float x[4], int lane, int &nv, int &colsel, bool &writer) {
    nv = 4; colsel = 0; writer = true;
    // halving exchanges across the high lane bits while a lane still holds more than one value
#pragma unroll
    for (int m = 32; m >= L && m >= 16; m >>= 1) {
        const bool upper = (lane & m) != 0;
        if (nv == 4) {
            const float s0 = upper ? x[0] : x[2], s1 = upper ? x[1] : x[3];
            const float r0 = __shfl_xor(s0, m, WAVE), r1 = __shfl_xor(s1, m, WAVE);
            x[0] = (upper ? x[2] : x[0]) + r0;
            x[1] = (upper ? x[3] : x[1]) + r1;
            colsel += upper ? 2 : 0; nv = 2;
        } else {
            const float s0 = upper ? x[0] : x[1];
            const float r0 = __shfl_xor(s0, m, WAVE);
            x[0] = (upper ? x[1] : x[0]) + r0;
            colsel += upper ? 1 : 0; nv = 1;
        }
    }
    // the steps inside a 16-lane row (L <= 8): one value left, all-reduce by DPP — only the writer lanes' copy is stored
    if constexpr (L <= 8) { x[0] += sp_dpp<0x128>(x[0]); writer = writer && !(lane & 8); }              // row_ror:8
    if constexpr (L <= 4) { x[0] += sp_dpp<0x124>(x[0]); writer = writer && !(lane & 4); }              // row_ror:4 (after ror 8: all four)
    if constexpr (L <= 2) { x[0] += sp_dpp<0x4E>(x[0]); writer = writer && !(lane & 2); }               // quad_perm [2,3,0,1]
    if constexpr (L <= 1) { x[0] += sp_dpp<0xB1>(x[0]); writer = writer && !(lane & 1); }               // quad_perm [1,0,3,2]
}

// forward: rows_per_wave consecutive rows per wave, the next chunk's entries requested before the current chunk's rows
template <int L>
__device__ __forceinline__ void sp_fwd_q_rows(const SpFwdArgs &a, const float *w, int row0, int nr, int lane, uint32_t epoch) {
    const int ip = a.indptr[row0 + min(lane, nr)];
    const int g = lane / L, l = lane % L;
    const int col0 = l * 4;
    const float *wp = w + (col0 < a.p ? col0 : 0);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void *)a.indices, 0, a.nnz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)a.vals, 0, a.nnz_bytes, 0x00020000);
    auto decisions = [&](int base, int cnt) __attribute__((always_inline)) -> uint64_t {
        if (!a.d.on) return ~0ull;
        bool k = false;
        if (lane < cnt) k = a.d.keep_mask ? a.d.keep_mask[base + lane] != 0 : keep1(a.d.off + (uint64_t)(base + lane), epoch, a.d.seed, a.d.thr);
        return __ballot(k);
    };
    int r = 0;
    int rb = __builtin_amdgcn_readfirstlane(__shfl(ip, 0, WAVE)), re = __builtin_amdgcn_readfirstlane(__shfl(ip, 1, WAVE));
    int base = rb, cnt = min(WAVE, re - base);
    unsigned ci[L], cv[L], ni[L], nv_[L];
    sp_load_entries<L>(rs_i, base + g * L, ci);
    sp_load_entries<L>(rs_v, base + g * L, cv);
    uint64_t ckeep = decisions(base, cnt), nkeep = ~0ull;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (;;) {
        const bool row_done = base + WAVE >= re;
        int nrow = r, nre = re, nbase = base + WAVE;
        if (row_done) {
            nrow = r + 1;
            if (nrow < nr) {
                nbase = __builtin_amdgcn_readfirstlane(__shfl(ip, nrow, WAVE));
                nre = __builtin_amdgcn_readfirstlane(__shfl(ip, nrow + 1, WAVE));
            }
        }
        const bool have_next = !row_done || nrow < nr;
        int ncnt = 0;
        if (have_next) {
            ncnt = min(WAVE, nre - nbase);
            sp_load_entries<L>(rs_i, nbase + g * L, ni);
            sp_load_entries<L>(rs_v, nbase + g * L, nv_);
            nkeep = decisions(nbase, ncnt);
        }
        if (cnt > 0) sp_gather_entries<L>(wp, a.ld_w, ci, cv, cnt, g, ckeep, a.d.scale, a.d.on != 0, acc);
        if (row_done) {
            int nv, colsel; bool writer;
            sp_reduce_groups<L>(acc, lane, nv, colsel, writer);
            if (writer) {
                float *o = a.out + (size_t)(row0 + r) * a.ld_out + col0 + colsel;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (i < nv && col0 + colsel + i < a.p) o[i] = (a.relu && !(acc[i] > 0.f)) ? 0.f : acc[i];
            }
            acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
        }
        if (!have_next) break;
        r = nrow; re = nre; base = nbase; cnt = ncnt; ckeep = nkeep;
#pragma unroll
        for (int k = 0; k < L; k++) { ci[k] = ni[k]; cv[k] = nv_[k]; }
    }
}

template <int L>
__global__ __launch_bounds__(256) void spmm_csr_fwd_q_kernel(SpFwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int K = a.rows_per_wave;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * K;
    if (row0 >= a.n_rows) return;
    const uint32_t epoch = (a.d.on && a.d.d_epoch) ? *a.d.d_epoch : 0u;
    sp_fwd_q_rows<L>(a, a.w, row0, min(K, a.n_rows - row0), lane, epoch);
}

#ifdef GCNHIP_EXPERIMENTS
// ... with W staged in LDS once per (persistent) workgroup: the W rows then cost no L2 traffic at all
template <int L>
__global__ __launch_bounds__(1024) void spmm_csr_fwd_q_lds_kernel(SpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sp_wq[];
    for (int i = threadIdx.x * 4; i < a.w_floats; i += 1024 * 4)
        *reinterpret_cast<float4 *>(sp_wq + i) = *reinterpret_cast<const float4 *>(a.w + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int K = a.rows_per_wave;
    const uint32_t epoch = (a.d.on && a.d.d_epoch) ? *a.d.d_epoch : 0u;
    for (int row0 = (blockIdx.x * 16 + (threadIdx.x >> 6)) * K; row0 < a.n_rows; row0 += gridDim.x * 16 * K)
        sp_fwd_q_rows<L>(a, sp_wq, row0, min(K, a.n_rows - row0), lane, epoch);
}
#endif  // GCNHIP_EXPERIMENTS

// ----------------------------------------------------------- sparse backward
// A task = (column j of X, entries [q0, q1) of its CSC list, partial slot or -1): dW[j, :] (or the slot's partial row)
// = sum_q X~[pos q] * dout[row q, :].  NW waves per task.
struct SpBwdArgs {
    const int4 *tasks;
    int n_tasks;
    const int *csc_row, *csc_pos;
    const float *vals, *dout;
    const float *csc_val;           // the stored values in CSC order (a copy of the pristine X made once): saves the dependent
                                    // vals[pos] round trip; NULL when the caller multiplies a value array of its own
    float *dw, *partials;
    int ld_dout, ld_dw, part_ld, p;
    int nnz_bytes;                  // size of csc_row[] / csc_val[] in bytes (narrow-row kernel's buffer loads)
    DropSpec d;
};

template <int L, bool VEC, int NW>
__global__ __launch_bounds__(NW == 1 ? 256 : NW * 64) void spmm_csc_bwd_kernel(SpBwdArgs a) {
    constexpr int G = WAVE / L;
    constexpr int V = VEC ? 4 : 1;
    __shared__ float part[NW > 1 ? NW * L * V : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = NW == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (t >= a.n_tasks) return;                              // NW == 1 only (wave-uniform); wider tasks: grid == n_tasks
    const int4 tk = a.tasks[t];
    const int col = tk.x, slot = tk.w;
    int r0 = tk.y, r1 = tk.z;
    if (NW > 1) {                                            // this wave's contiguous share, whole 64-entry chunks
        const int per = (((tk.z - tk.y + NW - 1) / NW) + 63) & ~63;
        r0 = min(tk.z, tk.y + wave * per);
        r1 = min(tk.z, r0 + per);
    }
    const int g = lane / L, l = lane % L;
    const uint32_t epoch = (a.d.on && a.d.d_epoch) ? *a.d.d_epoch : 0u;
    for (int cb = 0; cb < a.p; cb += L * V) {
        const int col0 = cb + l * V;
        const bool active = col0 < a.p;
        const float *dp = a.dout + (active ? col0 : 0);
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; i++) acc[i] = 0.f;
        // the pairs of chunk k+1 are requested before the rows of chunk k are gathered
        auto fetch = [&](int base, int &row_out, float &v_out) __attribute__((always_inline)) {
            row_out = 0; v_out = 0.f;
            const int cnt = min(WAVE, r1 - base);
            if (lane < cnt) {
                row_out = a.csc_row[base + lane];
                const int pos = a.csc_pos[base + lane];
                v_out = (a.csc_val ? a.csc_val[base + lane] : a.vals[pos]) * drop_scale(a.d, (uint64_t)pos, epoch);
            }
        };
        int cur_row = 0, nxt_row = 0;
        float cur_v = 0.f, nxt_v = 0.f;
        if (r0 < r1) fetch(r0, cur_row, cur_v);
        for (int base = r0; base < r1; base += WAVE) {
            const int cnt = min(WAVE, r1 - base);
            if (base + WAVE < r1) fetch(base + WAVE, nxt_row, nxt_v);
            sp_gather_chunk<L, V, SP_U>(dp, a.ld_dout, cur_row, cur_v, cnt, g, acc);
            cur_row = nxt_row; cur_v = nxt_v;
        }
#pragma unroll
        for (int i = 0; i < V; i++)
#pragma unroll
            for (int m = L; m < WAVE; m <<= 1) acc[i] += __shfl_xor(acc[i], m, WAVE);
        if (NW > 1) {                                        // the NW partials, added in wave order
            if (g == 0)
#pragma unroll
                for (int i = 0; i < V; i++) part[(wave * L + l) * V + i] = acc[i];
            __syncthreads();
            if (wave == 0 && g == 0) {
#pragma unroll
                for (int i = 0; i < V; i++) {
                    float s = part[l * V + i];
                    for (int w = 1; w < NW; w++) s += part[(w * L + l) * V + i];
                    acc[i] = s;
                }
            }
            __syncthreads();                                 // `part` is reused by the next column block
        }
        if ((NW == 1 || wave == 0) && g == 0 && active) {
            float *o = slot < 0 ? a.dw + (size_t)col * a.ld_dw + col0 : a.partials + (size_t)slot * a.part_ld + col0;
#pragma unroll
            for (int i = 0; i < V; i++)
                if (col0 + i < a.p) o[i] = acc[i];
        }
    }
}

// backward, narrow rows (see the narrow-row forward): the values come from the CSC-ordered copy (csc_val)
template <int L, int NW>
__global__ __launch_bounds__(NW == 1 ? 256 : NW * 64) void spmm_csc_bwd_q_kernel(SpBwdArgs a) {
    __shared__ float part[NW > 1 ? NW * L * 4 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = NW == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (t >= a.n_tasks) return;                              // NW == 1 only (wave-uniform)
    const int4 tk = a.tasks[t];
    const int col = tk.x, slot = tk.w;
    int r0 = tk.y, r1 = tk.z;
    if (NW > 1) {
        const int per = (((tk.z - tk.y + NW - 1) / NW) + 63) & ~63;
        r0 = min(tk.z, tk.y + wave * per);
        r1 = min(tk.z, r0 + per);
    }
    const int g = lane / L, l = lane % L;
    const int col0 = l * 4;
    const uint32_t epoch = (a.d.on && a.d.d_epoch) ? *a.d.d_epoch : 0u;
    const float *dp = a.dout + (col0 < a.p ? col0 : 0);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)a.csc_row, 0, a.nnz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)a.csc_val, 0, a.nnz_bytes, 0x00020000);
    auto decisions = [&](int base) __attribute__((always_inline)) -> uint64_t {
        if (!a.d.on) return ~0ull;
        bool k = false;
        if (lane < min(WAVE, r1 - base)) {
            const int pos = a.csc_pos[base + lane];
            k = a.d.keep_mask ? a.d.keep_mask[pos] != 0 : keep1(a.d.off + (uint64_t)pos, epoch, a.d.seed, a.d.thr);
        }
        return __ballot(k);
    };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned ci[L], cv[L], ni[L], nv_[L];
    uint64_t ckeep = ~0ull, nkeep = ~0ull;
    if (r0 < r1) {
        sp_load_entries<L>(rs_r, r0 + g * L, ci);
        sp_load_entries<L>(rs_v, r0 + g * L, cv);
        ckeep = decisions(r0);
    }
    for (int base = r0; base < r1; base += WAVE) {
        const int cnt = min(WAVE, r1 - base);
        if (base + WAVE < r1) {
            sp_load_entries<L>(rs_r, base + WAVE + g * L, ni);
            sp_load_entries<L>(rs_v, base + WAVE + g * L, nv_);
            nkeep = decisions(base + WAVE);
        }
        sp_gather_entries<L>(dp, a.ld_dout, ci, cv, cnt, g, ckeep, a.d.scale, a.d.on != 0, acc);
        ckeep = nkeep;
#pragma unroll
        for (int k = 0; k < L; k++) { ci[k] = ni[k]; cv[k] = nv_[k]; }
    }
    int nv, colsel; bool writer;
    sp_reduce_groups<L>(acc, lane, nv, colsel, writer);
    float *o = slot < 0 ? a.dw + (size_t)col * a.ld_dw : a.partials + (size_t)slot * a.part_ld;
    if (NW == 1) {
        if (writer)
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < nv && col0 + colsel + i < a.p) o[col0 + colsel + i] = acc[i];
    } else {                                                 // the NW partials of a column, added in wave order
        if (writer)
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < nv) part[wave * L * 4 + col0 + colsel + i] = acc[i];
        __syncthreads();
        if (threadIdx.x < L * 4 && (int)threadIdx.x < a.p) {
            float s = part[threadIdx.x];
            for (int w = 1; w < NW; w++) s += part[w * L * 4 + threadIdx.x];
            o[threadIdx.x] = s;
        }
    }
}

// columns that were cut into several tasks: dW[col, :] = partial rows first .. first + n - 1, added in order
__global__ void spmm_bwd_fold_kernel(const int4 *split_cols, int n_split, const float *partials, int part_ld, float *dw, int ld_dw, int p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_split * p) return;
    const int4 sc = split_cols[i / p];
    const int c = i % p;
    float s = partials[(size_t)sc.y * part_ld + c];
    for (int k = 1; k < sc.z; k++) s += partials[(size_t)(sc.y + k) * part_ld + c];
    dw[(size_t)sc.x * ld_dw + c] = s;
}
