// common.h — internals shared by the gfx950 kernels of libgcnhip.so.
// Wave = 64 lanes everywhere in this directory (CDNA4); nothing here is
// written for 32-wide warps.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gcnhip_driver.h"
#include "../../include/gcnhip_experimental.h"

#define GCNHIP_TRY(expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return (int)_e;              \
    } while (0)
#define GCNHIP_LAUNCH_CHECK() GCNHIP_TRY(hipGetLastError())

constexpr int WAVE = 64;

// detail text for a -1 return (gcnhip_last_error); returns -1 so that a check can `return gcnhip_fail("...")`
int gcnhip_fail(const char *detail);

// Options of a context (gcnhip_ctx_set_option / _get_option).  Each starts from the environment variable
// GCNHIP_<NAME IN CAPITALS>, read ONCE when the context is created — never at launch time — so a call's behaviour does not
// depend on what the process environment holds at that moment.  Most are A/B aids for measurements DESIGN.md records.
struct GcnOptions {
    int gs_pipe;            // 1: persistent index-prefetching aggregation kernel (measured slower)
    int gs_u;               // > 0: row loads in flight per lane group of the aggregation (0: by table size)
    int gs_nt;              // 1: non-temporal row loads in the sliced aggregation (measured slower)
    int gs_fold;            // 1: split rows are summed inside the aggregation launch (same bits; no faster)
    int gs_l;               // 8 / 4: column slices of 32 / 16 floats in the XCD-sliced aggregation (default 0: 64 floats)
    int gemm_tiles;         // 1: first-layer forward by the tile kernels instead of the persistent form
    int gemm_bf16x3;        // dense first-layer products (p = 128) from three bf16 planes on the bf16 MFMA pipe: 2 (default) on, also on a
                            // co-running stream; 1 on, the co-running stream keeps the f32 tiles; 0 off: the exact-f32 MFMA kernels
    int gemm_w4;            // 1: four-wave forward tiles
    int cls_abl;            // measurement aid (tools/bench_class.py): class_bf16x3.h kernels without 1: matrix work, 2: stores, 4: the column-wise operand loads
    int cls_wgs;            // measurement aid: workgroups per CU of the class-layer forward (0: default)
    int gemm_lane_waves;    // waves per workgroup of the bf16x3 first-layer forward on a co-running context: 8 (default) or 4 (measured: no gain)
    int gemm_lane_wgs;      // workgroups (= CUs) of that launch; 0: one per CU
    int spmm_slices;        // 1 (default): sparse X, W past an XCD's L2, h % 32 == 0: XCD-bound 32-float column slices of W; 0: the unsliced row kernel
    int cls_fwd;            // 1 (default): H1.W2 through class_bf16x3.h when gemm_bf16x3 >= 1; 0: through the f32-MFMA row stream
    int gemm_persist_bwd;   // 1: persistent first-layer weight gradient (measured slower)
    int dbg_linear;         // timing experiment only: the persistent forward reads X as if tile-major (wrong results)
    int xent_finalize;      // 1: the loss's final reduction as its own launch
    int xent_wave;          // 1: wave-per-row loss kernel for any width
    int adam_sum_launch;    // 1: Adam's sum of squares by a second launch
    int atb_cap_mb;         // split-K slab budget of A^T.B in MiB
    int rs_wgs;             // > 0: workgroups per CU of the row-streaming GEMM
    int spmm_lds;           // 1: sparse forward with W staged in LDS whenever it fits (measured slower: opt-in)
    int spmm_general;       // 1: narrow rows also take the general (shuffle-based) sparse kernels; -1: narrow kernels at any size (A/B, tests)
    int spmm_rows;          // > 0: rows per wave of the sparse forward (default: by row count, 1 .. 8)
    int spmm_nw;            // > 0: waves per column task of the sparse weight gradient (1, 4, 16), read by gcnhip_feat_create
    int split_edges;        // >= 16: segment length of split rows, read by gcnhip_graph_create*
};
struct GcnOptionEntry { const char *name; int GcnOptions::*field; int dflt; };
extern const GcnOptionEntry GCN_OPTION_TABLE[];
extern const int GCN_OPTION_COUNT;

struct gcnhip_ctx {
    int device;
    GcnOptions opt;
    hipStream_t stream;
    bool own_stream;
    int n_cu;
    // scratch for block-level partial reductions (xent, sumsq, adam sumsq)
    float *red_f;       // [RED_SLOTS * 4]
    int32_t *red_i;     // [RED_SLOTS * 4]
    uint32_t *ticket;   // arrival counters for last-block reductions: [0] the loss kernels, [1] Adam's sum of squares; zero between launches
    // gcnhip_metrics_record_with_next_loss: the next loss launch on this context writes the ring row itself
    bool rec_armed = false;
    float *rec_ring = nullptr; int rec_capacity = 0, rec_slot = 0; const uint32_t *rec_epoch = nullptr; const float *rec_sumsq = nullptr;
    // split-K slabs for the dense weight-gradient GEMMs
    float *slab;
    size_t slab_bytes;
    int slab_n;         // partial slabs the last dense weight-gradient product left there (split ranges, or workgroups of the persistent form)
    // packed weight image of the persistent forward GEMM (dense_persist.h), sized once at creation
    float *wpack;
    size_t wpack_bytes;
    int corun;          // gcnhip_ctx_set_corun: this stream's kernels are meant to share the chip with another stream's
};
constexpr size_t WPACK_BYTES = (size_t)2 << 20;    // K <= 4096 at p = 128
constexpr int RED_SLOTS = 4096;

// a registered subset of the rows of an adjacency object: its own compacted task list, in the order of the
// object's current row schedule (rebuilt whenever that changes).  Segment slots are the full schedule's.
struct gcnhip_graph;
struct gcnhip_rowset {
    const gcnhip_graph *owner;    // the adjacency object the subset was registered on (its task lists and segment slots)
    std::vector<uint32_t> bits;   // host copy: bit r = row r is in the subset
    int n_tasks;
    int4 *tasks;
    int n_split_rows;
    int4 *split_rows;
    int bounds[4][9];
};

struct gcnhip_graph {
    int n_rows, n_cols, nnz;
    int *indptr;        // [n_rows+1]
    int *indices;       // [nnz]
    float *coef;        // [nnz]
    // Â = D^-1/2 (A+I) D^-1/2 factored instead of stored per edge (gcnhip_graphsum_ex, scaling != 0):
    // dinv = (float)(1/sqrt(deg)), dinv2 = (float)(1/deg), deg = the FULL graph's degrees (a restricted object copies its parent's)
    float *dinv_row, *dinv2_row;   // [n_rows]
    float *dinv_col, *dinv2_col;   // [n_cols]
    // long-row splitting (rows above SPLIT_EDGES are cut into segments)
    int split_edges_opt; // the creating context's split_edges option (0: by size)
    int n_tasks;        // number of (row, e0, e1) tasks; 0 => one task per row
    int4 *tasks;        // {row, e_begin, e_end, partial_slot or -1}
    int n_split_rows;   // rows that own partial slots
    int4 *split_rows;   // {row, first_slot, n_slots, 0}
    float *partials;    // [n_slots * part_ld]
    int part_ld;
    int n_slots;
    // in-kernel segment sum (graphsum.hip): per segment slot {first slot of its row, number of segments}; per split row
    // and column slice an arrival counter (index first_slot * 8 + slice), zero between launches
    int2 *slot_info;    // [n_slots]
    uint32_t *seg_count; // [n_slots * 8]
    // task ranges of equal edge count for 1, 2, 4 or 8 XCD groups: bounds[log2 G][g] .. bounds[log2 G][g+1]
    int bounds[4][9];
    int *tmp_col_deg;   // only during construction
    std::vector<int> *h_indptr;   // host copy of the row pointers (the schedule can be rebuilt)
    std::vector<int4> *h_tasks, *h_srows;   // host copies of the full schedule (row subsets are cut from them)
    std::vector<gcnhip_rowset *> *rowsets;   // owned
};

// packed rows (dense_kernels.h "packed rows"): one 128-byte slot per (row, 64-column half)
struct gcnhip_rowpack {
    int rows, cols, halves;
    uint32_t *slots;    // [rows x halves x 32]
};

struct gcnhip_feat {
    int n_rows, n_cols;
    int64_t nnz;
    bool dense;
    int *indptr;        // [n_rows+1] (kept for dense too: row r starts at r*n_cols)
    int *indices;       // [nnz]; NULL when dense
    float *values;      // [nnz] pristine X
    float *values_pad;  // dense X whose rows are not 16-byte aligned: a copy with row stride ld_pad (multiple of 4)
    int ld_pad;
    // CSC view for the weight gradient (sparse X only)
    int *csc_ptr;       // [n_cols+1]
    int *csc_row;       // [nnz] source row of each entry
    int *csc_pos;       // [nnz] position jj in CSR order (selects value + dropout decision)
    float *csc_val;     // [nnz] values[csc_pos[q]]: the pristine values in CSC order (one dependent load less per entry)
    uint32_t *keep_bits; // [ceil(nnz/32)+1] input-dropout decisions of the current call (dense path)
    int keep_layout;     // how the last producer laid keep_bits out: 0 = flat (bit e & 31 of word e >> 5), 1 = chunk-major (dense_bf16x3.h)
    // the weight gradient's task list over the CSC view (sparse X, spmm_sparse.h): a column is one task of bwd_nw waves, a
    // column longer than the segment length several tasks whose partial rows a fold launch adds in order
    int4 *bwd_tasks;    // {column, q_begin, q_end, partial slot or -1}
    int n_bwd_tasks, bwd_nw;
    int4 *bwd_split;    // {column, first slot, segments, 0} for the columns that were cut
    int n_bwd_split, n_bwd_slots;
    float *bwd_partials; // [n_bwd_slots * bwd_part_ld], sized by the first backward call that needs it (and on a wider call)
    int bwd_part_ld;
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

// ---- dropout decisions: bit-sliced Bernoulli on top of Philox ----------------
// The reference keeps an element iff (int)RAND() >= int(p * 0x7fffffff) with a
// 31-bit uniform (src/seq/module.cpp:211-215).  Here the uniform has 16 bits:
//     keep(j)  <=>  U16(j) >= thr16,   thr16 = round(p * 65536)
// and U16 is never materialised: its bit planes are 32-lane words, so one
// Philox block decides 32 elements per plane, and only the planes down to the
// lowest set bit of thr16 are needed (p = 0.5 -> thr16 = 0x8000 -> ONE plane:
// 128 decisions per Philox call instead of 4).
//   element j: block c = j >> 7, group g = (j >> 5) & 3, bit b = j & 31
//   plane i (1 = MSB .. 16): P_i = Philox4x32-10(ctr = {lo32 c, hi32 c, epoch, i}, key = seed)[g]
//   ge = ~0; for i = n_planes .. 1: ge = bit(thr16, 16 - i) ? (P_i & ge) : (P_i | ge)
//   keep(j) = bit b of ge;   n_planes = 16 - ctz(thr16);  thr16 == 0 keeps everything.
__host__ __device__ inline int dropout_threshold(float p) {
    int t = (int)(p * 65536.0f + 0.5f);
    return t < 0 ? 0 : (t > 65535 ? 65535 : t);
}

// the 32 keep decisions of group g of block c
__device__ inline uint32_t keep_word(uint64_t c, int g, uint32_t epoch, uint64_t seed, int thr16) {
    if (thr16 == 0) return 0xFFFFFFFFu;
    const int n_planes = 16 - (__ffs(thr16) - 1);
    uint32_t ge = 0xFFFFFFFFu;
    for (int i = n_planes; i >= 1; i--) {
        uint32_t r[4];
        philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), epoch, (uint32_t)i, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        const uint32_t P = g == 0 ? r[0] : (g == 1 ? r[1] : (g == 2 ? r[2] : r[3]));
        ge = (thr16 >> (16 - i) & 1) ? (P & ge) : (P | ge);
    }
    return ge;
}
__device__ inline bool keep1(uint64_t j, uint32_t epoch, uint64_t seed, int thr16) {
    return (keep_word(j >> 7, (int)(j >> 5) & 3, epoch, seed, thr16) >> (j & 31)) & 1u;
}
// V (1, 2 or 4) consecutive elements starting at j, j % V == 0: bit s set = keep element j + s
template <int V>
__device__ inline uint32_t keepv(uint64_t j, uint32_t epoch, uint64_t seed, int thr16) {
    return (keep_word(j >> 7, (int)(j >> 5) & 3, epoch, seed, thr16) >> (j & 31)) & ((1u << V) - 1u);
}
// the 4 elements of quad q (elements 4q .. 4q+3)
__device__ inline uint32_t keep4(uint64_t quad, uint32_t epoch, uint64_t seed, int thr16) {
    return keepv<4>(quad << 2, epoch, seed, thr16);
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

// Each .hip file is its own code object inside the library, and the runtime loads a code object when the first kernel from it
// is launched — 6.8 ms for spmm.hip's, 4.7 ms for matmul.hip's, 1 ms for xent.hip's: 12 ms that landed inside the first
// training epoch of `gcn-hip` (15 ms against 3.2).  gcnhip_ctx_create asks for one kernel's attributes from every file
// instead (once per process), so the load happens with the rest of the set-up.
__attribute__((visibility("hidden"))) int gcnhip_preload_elementwise();
__attribute__((visibility("hidden"))) int gcnhip_preload_graphsum();
__attribute__((visibility("hidden"))) int gcnhip_preload_matmul();
__attribute__((visibility("hidden"))) int gcnhip_preload_spmm();
__attribute__((visibility("hidden"))) int gcnhip_preload_xent();
#define GCNHIP_DEFINE_PRELOAD(NAME, KERNEL)                                      \
    int gcnhip_preload_##NAME() {                                                \
        hipFuncAttributes attr;                                                  \
        return (int)hipFuncGetAttributes(&attr, (const void *)(KERNEL));         \
    }
static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
