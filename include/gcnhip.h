/*
 * gcnhip.h — C-ABI of libgcnhip.so: the MI355X (gfx950) kernels of the two-layer
 * GCN training path, one entry point per kernel-launching wrapper of the
 * reference (paths below are relative to the reference tree).
 *
 * This is the drop-in boundary.  The reference has no FFI: its GPU backend is a
 * set of host wrappers (src/cuda/cuda_module.cu, cuda_gcn.cu, cuda_variable.cu)
 * that size a grid and launch a __global__ kernel on raw device pointers
 * (src/cuda/cuda_kernel.cuh:22-90).  Each function here replaces one of those
 * wrapper bodies; a `Hip*` module calls it where the `CUDA*` module launches.
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - every function returns 0 on success, else a hipError_t (or -1 for an
 *    argument error); nothing aborts or throws.  gcnhip_error_string() names it.
 *    (The reference prints and exits: CUDA_CHECK, src/cuda/cuda_kernel.cuh:11-18 —
 *    that policy belongs to the caller; see GCNHIP_CHECK in host/hip_check.h.)
 *  - all launches are asynchronous on the context's stream; only functions
 *    documented as "synchronises" wait.  Scratch lives in the context / graph /
 *    feature objects; the few buffers whose size depends on an op's arguments
 *    (split-K slabs, split-row partials) are sized on the first call that needs
 *    them (that call synchronises once).  After one warm-up call no op allocates,
 *    so every op is hipGraph-capturable.
 *  - all matrices are row-major f32 with an explicit leading dimension `ld`
 *    (floats).  The reference's layout is ld == number of columns.  Rows that
 *    are 16-byte aligned (ld % 4 == 0 and a 16-byte aligned base) take the
 *    vectorised kernels; any other ld takes a slower scalar-load kernel with the
 *    same results.
 *  - indices are int32 like the reference (src/seq/sparse.h:12-17).
 *  - limits (argument error -1 beyond them): gcnhip_xent_fwd / gcnhip_accuracy up to 256 classes;
 *    gcnhip_matmul_* keep the small operand in LDS (gfx950: 160 KiB per CU), i.e. inner dimension
 *    n <= 736 for p <= 48 output columns (and any n for the backward); element counts below 2^31.
 */
#ifndef GCNHIP_H
#define GCNHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gcnhip_ctx gcnhip_ctx;       /* device + stream + scratch */
typedef struct gcnhip_graph gcnhip_graph;
typedef struct gcnhip_rowset gcnhip_rowset;
typedef struct gcnhip_rowpack gcnhip_rowpack; /* a mostly-zero matrix stored as packed rows (see gcnhip_matmul_bwd_packed) */   /* a registered subset of an adjacency object's rows */   /* prepared adjacency (CUDASparseIndex of the graph) */
typedef struct gcnhip_feat gcnhip_feat;     /* prepared feature matrix (CUDASparseIndex of X + its values) */

/* ---- context / runtime (replaces the implicit null stream + CUDA_CHECK) ---- */
int  gcnhip_device_count(int *count);
/* stream == NULL: the context creates and owns a non-blocking stream.
 * Otherwise `stream` is a hipStream_t owned by the caller. */
int  gcnhip_ctx_create(gcnhip_ctx **ctx, int device, void *stream);
int  gcnhip_ctx_destroy(gcnhip_ctx *ctx);
/* Hint: the launches of this context are meant to run BESIDE another stream's kernels (HipGCN's validation lane next to
 * the training pass).  Ops that have a whole-chip persistent form (the dense first-layer GEMM: one 512-thread workgroup
 * with 150 KB of LDS per CU for the whole launch) then take their tiled form, which leaves wave slots and registers to the
 * neighbour: measured on the two-stream epoch, 294 epochs/s with the persistent form on the lane against 297 with tiles.
 * Results do not depend on the hint beyond the order of floating-point sums of the two forms (each within the tested bound). */
int gcnhip_ctx_set_corun(gcnhip_ctx *ctx, int on);
/* Options of a context, by name (argument error -1 for an unknown name; gcnhip_last_error() says so).  Every option starts
 * from the environment variable GCNHIP_<NAME IN CAPITALS>, which is read ONCE, when the context is created: no entry point
 * consults the environment at call time, so a call's behaviour is a function of its arguments and its context.  Most are
 * A/B aids behind measurements DESIGN.md records; results are the same bits unless the option's line says otherwise:
 *   gs_u (0: by table size; 1/2/4 row loads in flight per lane group of the aggregation), gs_fold (1: split rows summed inside
 *   the aggregation launch), gs_pipe, gs_nt (measured-slower aggregation variants), gemm_tiles / gemm_w4 / gemm_persist_bwd
 *   (first-layer GEMM forms), xent_finalize / adam_sum_launch (final reductions as their own launches), xent_wave,
 *   atb_cap_mb, rs_wgs, spmm_lds (1: the sparse forward stages W in LDS whenever
 *   it fits — measured slower, opt-in), spmm_rows (rows per wave of the sparse forward; 0: by row count), spmm_general (1: narrow
 *   rows take the general sparse kernels too — a differently associated f32 sum), spmm_nw and split_edges (read by gcnhip_feat_create / gcnhip_graph_create*: set them BEFORE building objects). */
int  gcnhip_ctx_set_option(gcnhip_ctx *ctx, const char *name, int value);
int  gcnhip_ctx_get_option(const gcnhip_ctx *ctx, const char *name, int *value);
int  gcnhip_ctx_sync(gcnhip_ctx *ctx);                 /* synchronises the stream */
void *gcnhip_ctx_stream(gcnhip_ctx *ctx);
const char *gcnhip_error_string(int code);
/* detail of the calling thread's most recent -1 (argument error) where the library has one to give, else "" */
const char *gcnhip_last_error(void);
const char *gcnhip_version(void);
/* 1 when the library was built with `make EXPERIMENTS=1`: the variants DESIGN.md records as built, bit-identical and SLOWER
 * (packed rows: gcnhip_rowpack_*, gcnhip_matmul_bwd_packed, gcnhip_graphsum_packed; the options gs_pipe, gs_nt, gs_fold,
 * gemm_persist_bwd, spmm_lds, dbg_linear) are compiled in.  The default build leaves them out: those entry points then
 * return -1 (gcnhip_last_error() says why) and those options are ignored. */
int gcnhip_experiments(void);

/* ---- memory (CUDAVariable ctor/dtor/zero: src/cuda/cuda_variable.cu:3-31) ---- */
int gcnhip_malloc(gcnhip_ctx *ctx, void **ptr, size_t bytes);
int gcnhip_free(gcnhip_ctx *ctx, void *ptr);
int gcnhip_memset_async(gcnhip_ctx *ctx, void *ptr, int byte, size_t bytes);
int gcnhip_h2d(gcnhip_ctx *ctx, void *dst, const void *src, size_t bytes);   /* synchronises */
int gcnhip_d2h(gcnhip_ctx *ctx, void *dst, const void *src, size_t bytes);   /* synchronises */
int gcnhip_d2d_async(gcnhip_ctx *ctx, void *dst, const void *src, size_t bytes);
/* Read-back without stalling the producer: page-locked host memory and a device-to-host copy that is only ENQUEUED on
 * ctx's stream (wait for it with an event recorded behind it + gcnhip_event_sync).  HipGCN::run() uses them to
 * print epoch e's line while epochs e+1.. are already running: behind a group of epochs, the metrics rows of the group are
 * copied on the producing stream and an event recorded that the host waits on.  (The reference's CUDA path instead blocks
 * on a cudaMemcpy of the whole logits matrix per accuracy call, src/cuda/cuda_gcn.cu:100-120.) */
int gcnhip_host_alloc(void **ptr, size_t bytes);
int gcnhip_host_free(void *ptr);
int gcnhip_d2h_async(gcnhip_ctx *ctx, void *dst_pinned, const void *src, size_t bytes);

/* ---- adjacency (CUDASparseIndex(const SparseIndex&): cuda_variable.cu:55-64) --
 * Copies the CSR of n_rows rows to the device and precomputes, once, the per
 * edge coefficient the reference recomputes for every edge of every call
 * (src/seq/module.cpp:91-93, src/cuda/cuda_kernel.cu:136-138):
 *     coef(e) = (float)(1.0 / sqrtf((float)(deg(src) * deg(dst))))
 * deg(src) = row length; deg(dst) = col_deg[dst] (col_deg == NULL: the row
 * length of row dst; then n_cols must equal n_rows).  The product is formed in
 * 64-bit (the reference's int product overflows above 46 340, module.cpp:92).
 * col_deg lets a row block of a partitioned graph name global degrees. */
int gcnhip_graph_create(gcnhip_ctx *ctx, gcnhip_graph **g, const int *h_indptr, const int *h_indices,
                        int n_rows, int n_cols, const int *h_col_deg);
/* As above, with a locality hint: h_row_group[r] >= 0 names the community of row r (any small integer
 * key; the host passes the node's label).  Rows of one group are scheduled together — group-major, heavy rows
 * first inside a group — so the neighbour rows they share stay in the XCD's L2 while the group is processed.
 * Results are those of gcnhip_graph_create bit for bit: no node is renamed and the order of every row's own
 * sum is unchanged; only the order in which rows are computed differs.  NULL = no hint. */
int gcnhip_graph_create_grouped(gcnhip_ctx *ctx, gcnhip_graph **g, const int *h_indptr, const int *h_indices,
                                int n_rows, int n_cols, const int *h_col_deg, const int *h_row_group);
/* Replace the row schedule of a prepared adjacency (synchronises the context).  The aggregation computes one
 * row per wave; WHICH rows are in flight together decides what the caches hold and whether bandwidth-bound hub rows
 * overlap with the overhead-bound tail of short rows.  Every schedule gives bit-identical results.
 *   mode 0: descending degree (what gcnhip_graph_create builds)
 *   mode 1: h_row_group-major, descending degree inside a group (= gcnhip_graph_create_grouped)
 *   mode 2: descending-degree rank dealt round-robin into n_groups groups (every group has the same degree mix),
 *           group-major; measured on an R-MAT graph with a 1 GiB table: 6.9 -> 5.3 ms at d = 128
 * The host (HipGCN) times the candidates once per dataset and keeps the fastest. */
int gcnhip_graph_set_schedule(gcnhip_ctx *ctx, gcnhip_graph *g, int mode, const int *h_row_group, int n_groups);
/* Widest aggregation (columns) this object will be asked for.  Rows cut into segments need a scratch row per
 * segment; it is sized for 256 columns when the object is built and grown HERE (synchronises the context) —
 * never inside gcnhip_graphsum*, which return -1 for a wider call on an object that has split rows (and leave a
 * message naming this function in gcnhip_last_error()).  Objects made by gcnhip_graph_create_restricted copy the
 * parent's width when they are created: call this on the parent FIRST, or on each restricted object as well.
 * So a launch allocates nothing, may be
 * captured into a hipGraph at any time, and treats the object as read-only apart from that scratch: two streams
 * may share one object only if their aggregations never overlap in time (HipGCN gives each lane its own). */
int gcnhip_graph_reserve_width(gcnhip_ctx *ctx, gcnhip_graph *g, int max_dim);
int gcnhip_graph_destroy(gcnhip_ctx *ctx, gcnhip_graph *g);
/* device pointers of the prepared arrays (tests, diagnostics) */
int gcnhip_graph_arrays(const gcnhip_graph *g, const int **d_indptr, const int **d_indices,
                        const float **d_coef, int *n_rows, int *nnz);

/* ---- GraphSum (CUDAGraphSum::forward/backward: cuda_module.cu:75-103;
 *      kernels cuda_kernel.cu:126-162; CPU: src/seq/module.cpp:83-119) --------
 * out[r, 0:dim] = sum_e coef(e) * in[indices[e], 0:dim] over row r's edges.
 * Forward and backward are this same operator (the reference relies on a
 * symmetric adjacency, module.cpp:95).  `in` has g->n_cols rows. */
int gcnhip_graphsum(gcnhip_ctx *ctx, const gcnhip_graph *g, const float *in, int ld_in,
                    float *out, int ld_out, int dim);
/* Same operator when whole rows of `in` are known to be zero: bit j of in_row_bits (n_cols bits,
 * word j >> 5) == 0 promises that row j is all zero, and the kernel does not read it.  The backward
 * of the output layer is the case: dZ is zero for every node outside the training split
 * (module.cpp:129-133), so a third of Reddit's gathers (and 95 % of Cora's) are skipped with
 * identical results. */
int gcnhip_graphsum_rowmask(gcnhip_ctx *ctx, const gcnhip_graph *g, const float *in, int ld_in,
                            float *out, int ld_out, int dim, const uint32_t *in_row_bits);
/* ... and when only some rows of `out` are ever read: bit r of out_row_bits (n_rows bits) == 0 means row r is
 * not computed and its memory is left untouched.  The last aggregation of a forward is the case: the loss and the
 * accuracy read only rows whose node is in the scored split (CrossEntropyLoss::forward, module.cpp:131-133;
 * GCN::get_accuracy, gcn.cpp:86-88), i.e. 66 % of Reddit's rows in a training forward and 10 % in a validation
 * forward.  Every computed row is bit-identical to gcnhip_graphsum's.  Either mask may be NULL. */
int gcnhip_graphsum_masked(gcnhip_ctx *ctx, const gcnhip_graph *g, const float *in, int ld_in,
                           float *out, int ld_out, int dim, const uint32_t *in_row_bits, const uint32_t *out_row_bits);
/* The same for a subset that is known in advance (the three splits of a dataset): gcnhip_graph_add_rowset cuts a
 * compacted task list for the rows with bit r set in h_row_bits (host, n_rows bits) out of the object's row schedule
 * — no wave is launched for a row outside it (the device mask above launches every wave and retires the unwanted ones:
 * at 10 % wanted rows that is 4x slower than the compacted list).  The subset belongs to the object: it follows
 * gcnhip_graph_set_schedule and is freed by gcnhip_graph_destroy; passing it with any other adjacency object is an
 * argument error (-1).  Results as gcnhip_graphsum_masked. */
int gcnhip_graph_add_rowset(gcnhip_ctx *ctx, gcnhip_graph *g, const uint32_t *h_row_bits, gcnhip_rowset **rows);
int gcnhip_rowset_size(const gcnhip_rowset *rows, int *n_tasks);
int gcnhip_graphsum_rowset(gcnhip_ctx *ctx, const gcnhip_graph *g, const gcnhip_rowset *rows, const float *in, int ld_in,
                           float *out, int ld_out, int dim, const uint32_t *in_row_bits);
/* When the zero rows of an aggregation's input are known for good (the backward of the output layer: dZ is zero for every
 * node outside the training split, module.cpp:129-133), the edges that point at them can be left out of the operator
 * instead of being masked at every launch: gcnhip_graph_create_restricted builds a second adjacency object with the
 * edges of `parent` whose SOURCE row (column index) has bit j set in h_col_bits (host, n_cols bits), the parent's
 * coefficients (degrees of the full graph, module.cpp:91-93) and the parent's current row order.  Aggregating an input
 * that is zero outside the set through it gives the same sum with the zero terms absent (the remaining terms may be
 * added in a different order: within the f32 bound of the tests, not bit-identical to the masked launch).  At Reddit
 * scale the masked class-width backward takes 0.39 ms, the restricted operator 0.27 ms (a third of the edges gone and
 * no predicate on the loads).  The object is independent of the parent: destroy it with gcnhip_graph_destroy. */
int gcnhip_graph_create_restricted(gcnhip_ctx *ctx, gcnhip_graph **out, const gcnhip_graph *parent, const uint32_t *h_col_bits);
/* A second, independent object with the parent's edges, coefficients and CURRENT row order and its own task lists and split-row
 * scratch — for a second stream that aggregates through the same adjacency at the same time (HipGCN's validation lane).  Device
 * copies only: the host preparation of gcnhip_graph_create (per-row neighbour sort: 0.4 s at Reddit scale) is not repeated.
 * Row subsets are not copied: register them on the clone.  Synchronises the context. */
int gcnhip_graph_clone(gcnhip_ctx *ctx, gcnhip_graph **out, const gcnhip_graph *parent);
/* One PART of an aggregation whose edges were split over two operators with the same rows (both made by
 * gcnhip_graph_create_restricted from one parent with complementary column sets).  The row-partitioned epoch uses it
 * to start on the edges that point at this rank's own rows while the rows of the other ranks are still in flight on the
 * exchange stream, then adds the remaining edges (xGMI note):
 *     accumulate == 0:  out[r,:]  = sum over the operator's edges            (the first part)
 *     accumulate != 0:  out[r,:]  = out[r,:] + that sum                      (the second part; rows with no edge keep their value)
 * with every option of the other entry points: rows (NULL or a subset registered on THIS g), in_row_bits (NULL or as in
 * gcnhip_graphsum_rowmask), and relu_dropout != 0 = the epilogue of gcnhip_graphsum_relu_dropout applied AFTER the
 * addition, i.e. by the last part only.  The sum of a row is then (terms of part 1) + (terms of part 2): the same real
 * number as gcnhip_graphsum on the parent, associated differently — within the f32 bound of the tests, not bit-identical. */
int gcnhip_graphsum_part(gcnhip_ctx *ctx, const gcnhip_graph *g, const gcnhip_rowset *rows, const float *in, int ld_in,
                         float *out, int ld_out, int dim, const uint32_t *in_row_bits, int accumulate,
                         int relu_dropout, int training, float p, uint64_t seed, const uint32_t *d_epoch,
                         uint64_t elem_offset, const uint8_t *keep_mask);
/* Fused epilogue used by the first layer: GraphSum, then ReLU
 * (module.cpp:175-185), then Dropout (module.cpp:207-221) on the same rows.
 * training == 0: ReLU only.  The dropout decision for element (r, c) is
 * keep(seed, *d_epoch, elem_offset + r*dim + c) (see gcnhip_dropout_fwd), or
 * keep_mask[r*dim + c] != 0 when keep_mask != NULL.  No mask is stored:
 * backward recovers it as out > 0 (gcnhip_relu_dropout_bwd). */
int gcnhip_graphsum_relu_dropout(gcnhip_ctx *ctx, const gcnhip_graph *g, const float *in, int ld_in,
                                 float *out, int ld_out, int dim, int training, float p,
                                 uint64_t seed, const uint32_t *d_epoch, uint64_t elem_offset,
                                 const uint8_t *keep_mask);
/* The same, also leaving the mask its backward needs as ONE BIT per element: bit (c & 31) of
 * pos_bits[r * words_per_row + (c >> 5)] = (out[r, c] > 0) after ReLU and dropout (the reference keeps a bool array for the
 * ReLU and an int array for the Dropout, module.cpp:166-173, 196-205: 5 bytes per element).  The bits are assembled from
 * the lanes that store the row, so they cost no extra pass; gcnhip_matmul_bwd_fused_bits reads them instead of re-reading
 * the activations (119 MB per epoch at Reddit scale).  Needs dim % 32 == 0, 16-byte aligned rows and
 * words_per_row * 32 >= dim; rows of a registered subset are not supported (the hidden layer computes every row). */
int gcnhip_graphsum_relu_dropout_bits(gcnhip_ctx *ctx, const gcnhip_graph *g, const float *in, int ld_in,
                                      float *out, int ld_out, int dim, int training, float p,
                                      uint64_t seed, const uint32_t *d_epoch, uint64_t elem_offset,
                                      const uint8_t *keep_mask, uint32_t *pos_bits, int words_per_row);

/* ---- the factored operator (round 4) ------------------------------------------------------------------------------
 * A^ = D^-1/2 (A + I) D^-1/2.  The reference multiplies every gathered row by a per-EDGE coefficient
 * 1/sqrt(deg(src) deg(dst)) (module.cpp:91-93), and so do the entry points above — which makes the coefficient array a
 * second stream beside the indices: 94 MB per launch at Reddit scale, measured at 6 % (hidden width) to 10 % (class width)
 * of the launch (tools/gather_peak.py --coef; profiles/r04_gather_peak.json).  The same operator FACTORED needs no per-edge
 * number at all:  (A^ x)[r] = dinv[r] * sum_e (dinv[col(e)] * x[col(e)]),  dinv = 1/sqrt(deg).
 * gcnhip_graphsum_ex with scaling != 0 computes  out[r] = post[r] * sum_e in[col(e)]  where the CALLER has stored
 * dinv[col] * x[col] in `in` (every producer on the training path has a row-wise epilogue or a value array that takes the
 * factor for free: HipGCN, host/gcn.cpp "factored").  scaling: 0 = per-edge coefficients (identical to the entry points
 * above), 1 = post = dinv[row], 2 = post = dinv[row]^2 (= 1/deg: the result is already dinv-scaled for the NEXT
 * aggregation), 3 = no post factor (the consumer folds dinv[row] in, e.g. through a pre-scaled feature matrix).
 * Same real numbers as the reference's operator; each term carries two more f32 roundings (dinv[r] * (dinv[c] * x)
 * instead of coef * x), inside the summation-order bound the parity tests use.  f32 rows, 16-byte aligned.
 * The other fields are the options of gcnhip_graphsum_rowset / _rowmask / _part / _relu_dropout_bits, all optional. */
typedef struct {
    const gcnhip_rowset *rows;          /* NULL: every row */
    const uint32_t *in_row_bits;        /* NULL, or as gcnhip_graphsum_rowmask */
    int accumulate;                     /* as gcnhip_graphsum_part: out = out + sum (then post, then the epilogue) */
    int relu_dropout, training;         /* the fused epilogue of gcnhip_graphsum_relu_dropout */
    float p;
    uint64_t seed;
    const uint32_t *d_epoch;
    uint64_t elem_offset;
    const uint8_t *keep_mask;
    uint32_t *pos_bits;                 /* NULL, or as gcnhip_graphsum_relu_dropout_bits (needs relu_dropout) */
    int words_per_row;
    int scaling;                        /* see above */
    const struct gcnhip_gs_loss *loss;  /* NULL, or the loss epilogue below (round 5) */
} gcnhip_gs_opts;
/* Loss epilogue of the aggregation that produces the logits (round 5; CrossEntropyLoss::forward, src/seq/module.cpp:124-161,
 * and GCN::get_accuracy, gcn.cpp:83-96, inside the launch that computes Z = A^.Z0).  After its shuffle reduce a wave holds
 * the whole logit row (dim <= 64) in one lane group: max, sum of exp (left to right, the reference's order), the row's loss
 * term, the accuracy test and — training — the gradient row (softmax - onehot) / count [* grad_row_scale[r]] are computed
 * there; the launch writes `out` (the logits) as always, the gradient row, and row_terms[2r] = loss term, row_terms[2r+1] =
 * 1.f when no logit is above the true one.  gcnhip_xent_from_row_terms then adds the terms of a row list in the order
 * gcnhip_xent_fwd_rows adds them: same bits as that entry point on the stored logits, without reading the logits again.
 * Rows with truth < 0 get a zero gradient row and zero terms.  Needs f32 rows, 16-byte aligned, dim <= 64, no relu_dropout. */
typedef struct gcnhip_gs_loss {
    const int32_t *truth;               /* [rows of out] */
    float *grad; int ld_grad;           /* gradient rows (training != 0) */
    int training, count;                /* as gcnhip_xent_fwd_rows (count > 0) */
    const float *grad_row_scale;        /* NULL, or as gcnhip_xent_fwd_rows_scaled */
    float *row_terms;                   /* [2 * rows of out] */
} gcnhip_gs_loss;
int gcnhip_graphsum_ex(gcnhip_ctx *ctx, const gcnhip_graph *g, const gcnhip_gs_opts *opts, const float *in, int ld_in,
                       float *out, int ld_out, int dim);
/* device pointers of the factor arrays of a prepared adjacency: dinv / dinv^2 per row ([n_rows]) and per column ([n_cols]);
 * degrees are those of the full graph also for objects made by gcnhip_graph_create_restricted */
int gcnhip_graph_scales(const gcnhip_graph *g, const float **d_dinv_row, const float **d_dinv2_row,
                        const float **d_dinv_col, const float **d_dinv2_col);
/* values of row r of a feature object *= d_row_scale[r] (all of its internal copies; synchronises): X -> D^-1/2 X */
int gcnhip_feat_scale_rows(gcnhip_ctx *ctx, gcnhip_feat *f, const float *d_row_scale);

/* ---- features (CUDASparseIndex of X + the input CUDAVariable) ----------------
 * CSR of X with n_rows rows and n_cols (= input_dim) columns.  A matrix whose
 * every row holds exactly the columns 0..n_cols-1 in order is detected as
 * dense and its index array is not kept on the device.  A column-major index
 * (CSC) is built once so the weight gradient is a gather, not the racy
 * scatter of cuda_kernel.cu:112-122. */
int gcnhip_feat_create(gcnhip_ctx *ctx, gcnhip_feat **f, const int *h_indptr, const int *h_indices,
                       const float *h_values, int n_rows, int n_cols);
/* Aggregate-first evaluation.  Without dropout the first layer is linear in X: ReLU(A^.(X.W1)) = ReLU((A^.X).W1)
 * (src/seq/gcn.cpp:23-41 with Dropout skipped, module.cpp:208), and A^.X does not change from epoch to epoch.
 * This builds the feature object of A^.X once (dense X only; x has g->n_cols rows — every column of g); an
 * evaluation forward then runs gcnhip_spmm_fwd_relu on it and needs NO hidden-width aggregation (and, with
 * several GPUs, no exchange before the hidden layer).  Same result up to the rounding of a reassociated f32 sum.
 * Training cannot use it: its X~ changes with every epoch's dropout decisions. */
int gcnhip_feat_create_aggregated(gcnhip_ctx *ctx, gcnhip_feat **f, gcnhip_graph *g, const gcnhip_feat *x);
int gcnhip_feat_destroy(gcnhip_ctx *ctx, gcnhip_feat *f);
int gcnhip_feat_is_dense(const gcnhip_feat *f);
/* device pointer, nnz floats: the pristine X.  READ-ONLY: the object keeps derived copies of these values (a padded image for
 * the MFMA kernels, the CSC-ordered copy the sparse weight gradient reads) that only gcnhip_feat_scale_rows keeps in step —
 * a caller that wants modified values passes its own buffer as `vals` to gcnhip_spmm_fwd / _bwd instead (the modular path
 * does: set_input copies, Dropout mutates the copy, gcn.cpp:23,73-76). */
const float *gcnhip_feat_values(const gcnhip_feat *f);
int64_t gcnhip_feat_nnz(const gcnhip_feat *f);

/* ---- SparseMatmul (CUDASparseMatmul: cuda_module.cu:42-70; kernels
 *      cuda_kernel.cu:100-122; CPU: src/seq/module.cpp:47-77) ------------------
 * forward : out[i, 0:p] = sum_jj vals[jj] * w[col(jj), 0:p]
 * backward: dw[col(jj), 0:p] += dout[i, 0:p] * vals[jj]   (dw is overwritten)
 * `vals` is the device value array to use (nnz floats): pass
 * gcnhip_feat_values(f), or a buffer that gcnhip_dropout_fwd has modified the
 * way the reference's input Dropout modifies v0 (gcn.cpp:23).
 * Fused input dropout (replaces that Dropout module + the per-epoch
 * set_input copy of cuda_gcn.cu:81-83): p_drop > 0 applies
 * vals[jj] * (keep ? 1/(1-p) : 0) on the fly, keep as in gcnhip_dropout_fwd
 * with element index nnz_offset + jj, or keep_mask[jj] when non-NULL.  The
 * backward must be called with the same arguments to see the same X~. */
int gcnhip_spmm_fwd(gcnhip_ctx *ctx, const gcnhip_feat *f, const float *vals, const float *w, int ld_w,
                    float *out, int ld_out, int p, float p_drop, uint64_t seed, const uint32_t *d_epoch,
                    uint64_t nnz_offset, const uint8_t *keep_mask);
/* forward without dropout, ReLU (module.cpp:175-185, keep = x > 0) applied when the result is stored */
int gcnhip_spmm_fwd_relu(gcnhip_ctx *ctx, const gcnhip_feat *f, const float *vals, const float *w, int ld_w,
                         float *out, int ld_out, int p);
/* Evaluation forward of BOTH layers' products in one launch (round 5): z0[m x p2] = ReLU(X . w)[m x p] . w2[p x p2], the hidden
 * matrix never stored (SparseMatmul::forward + ReLU + Matmul::forward, module.cpp:47-61, 175-185, 11-22, for a forward whose
 * hidden activations nobody reads afterwards: GCN::eval with the aggregate-first feature object).  The first product is
 * computed transposed on the bf16 pipe (three-plane splits, as gcnhip_spmm_fwd does at p = 128), which leaves a row's features
 * in one lane's accumulators — the operand layout of the second product.  Available for a dense X, p = 128, p2 <= 64,
 * 16-byte aligned rows of z0 and option gemm_bf16x3 != 0; otherwise returns GCNHIP_NOT_AVAILABLE (nothing launched: call
 * gcnhip_spmm_fwd_relu and gcnhip_matmul_fwd instead).  Results inside the f32 summation bound of the two-call form, not its bits. */
#define GCNHIP_NOT_AVAILABLE (-2)
int gcnhip_spmm_fwd_relu_matmul(gcnhip_ctx *ctx, const gcnhip_feat *f, const float *vals, const float *w, int ld_w, int p,
                                const float *w2, int ld_w2, int p2, float *z0, int ld_z0);
int gcnhip_spmm_bwd(gcnhip_ctx *ctx, const gcnhip_feat *f, const float *vals, const float *dout, int ld_dout,
                    float *dw, int ld_dw, int p, float p_drop, uint64_t seed, const uint32_t *d_epoch,
                    uint64_t nnz_offset, const uint8_t *keep_mask);
/* The dense weight gradient (dense X, p > 64) is a split-K product: n_splits row ranges of rows_per_split rows each
 * write a partial [n_cols x p] slab, and an ordered sum of the slabs gives dW (no atomics: the same bits every run).
 * The three steps are also callable one by one, so that a caller whose dOut arrives in row blocks (the hidden layer's
 * backward aggregation, computed block by block on another stream) can start on the first blocks while the rest is
 * still being produced:  _plan reports the ranges (n_splits == 0: this shape does not take the split-K path, use
 * gcnhip_spmm_bwd);  _part computes splits [split_begin, split_end) into the context's slabs — rows
 * [split_begin * rows_per_split, min(n_rows, split_end * rows_per_split)) of X and dOut are all it reads;
 * make_decisions != 0 (re)generates the input-dropout decisions first, once per backward (0: the decisions a forward or an
 * earlier part made with the same p_drop / seed / epoch / nnz_offset are still in the feature object; their bit layout is the
 * library's own — flat or a word per row and 32 columns, by kernel family — and a part that needs the other one re-derives it
 * from the same arguments);  _finish sums the slabs of the same context.  gcnhip_spmm_bwd is exactly _part(0, n_splits, 1) + _finish. */
int gcnhip_spmm_bwd_plan(const gcnhip_ctx *ctx, const gcnhip_feat *f, int p, int *rows_per_split, int *n_splits);
int gcnhip_spmm_bwd_part(gcnhip_ctx *ctx, const gcnhip_feat *f, const float *vals, const float *dout, int ld_dout, int p,
                         float p_drop, uint64_t seed, const uint32_t *d_epoch, uint64_t nnz_offset, const uint8_t *keep_mask,
                         int split_begin, int split_end, int make_decisions);
int gcnhip_spmm_bwd_finish(gcnhip_ctx *ctx, const gcnhip_feat *f, float *dw, int ld_dw, int p);

/* ---- opt-in storage format: bfloat16 gathered tables (beyond the reference) ----------
 * gcnhip_f32_to_bf16 rounds rows of f32 to bf16 (nearest even; NaN kept) into a table with row stride ld_dst
 * (multiple of 8, 16-byte aligned; columns dim..ld_dst-1 are written as zero).  gcnhip_graphsum_bf16 is
 * GraphSum reading that table: coef, the running sum and `out` are f32, so the ONLY difference to
 * gcnhip_graphsum* is the rounding of the gathered values — on a table that holds bf16-representable numbers
 * the two agree bit for bit.  A row of d values is 2d bytes: half the cache lines per edge.
 * in_row_bits (optional) as in gcnhip_graphsum_rowmask, out_rows (optional) as in gcnhip_graphsum_rowset; relu_dropout != 0 selects the fused epilogue of
 * gcnhip_graphsum_relu_dropout with the arguments that follow. */
int gcnhip_f32_to_bf16(gcnhip_ctx *ctx, const float *src, int ld_src, uint16_t *dst, int ld_dst, int64_t rows, int dim);
int gcnhip_graphsum_bf16(gcnhip_ctx *ctx, const gcnhip_graph *g, const uint16_t *in_bf16, int ld_in,
                         float *out, int ld_out, int dim, const uint32_t *in_row_bits, const gcnhip_rowset *out_rows,
                         int relu_dropout, int training, float p, uint64_t seed, const uint32_t *d_epoch,
                         uint64_t elem_offset, const uint8_t *keep_mask);

/* ---- Matmul (CUDAMatmul: cuda_module.cu:8-34; kernels cuda_kernel.cu:6-96;
 *      CPU: src/seq/module.cpp:11-42) -------------------------------------------
 * forward : c[m x p] = a[m x n] . b[n x p]
 * backward: da = dc . b^T (assigned); db = a^T . dc (assigned) */
int gcnhip_matmul_fwd(gcnhip_ctx *ctx, const float *a, int lda, const float *b, int ldb,
                      float *c, int ldc, int m, int n, int p);
int gcnhip_matmul_bwd(gcnhip_ctx *ctx, const float *a, int lda, const float *b, int ldb,
                      const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                      int m, int n, int p);
/* da only, with the ReLU+Dropout backward fused into the store:
 * da[i,j] = (h[i,j] > 0) ? scale * (dc . b^T)[i,j] : 0, h = the forward output
 * of gcnhip_graphsum_relu_dropout (module.cpp:187-194, 223-233). */
int gcnhip_matmul_bwd_fused(gcnhip_ctx *ctx, const float *a, int lda, const float *b, int ldb,
                            const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                            int m, int n, int p, float relu_dropout_scale);
/* ... with the mask taken from the bits gcnhip_graphsum_relu_dropout_bits left (same result bit for bit: the bit IS h > 0) */
int gcnhip_matmul_bwd_fused_bits(gcnhip_ctx *ctx, const float *a, int lda, const float *b, int ldb,
                                 const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                                 int m, int n, int p, float relu_dropout_scale, const uint32_t *pos_bits, int words_per_row);

/* every form of the fused backward in one call, plus an optional factor per row of da:  db = a^T . dc when db != NULL;
 * da[r,:] = mask . (relu_dropout_scale * d_da_row_scale[r]) . (dc . b^T)[r,:], mask = pos_bits when given (a may then be
 * NULL if db is NULL too), else a > 0; d_da_row_scale == NULL: factor 1 */
int gcnhip_matmul_bwd_ex(gcnhip_ctx *ctx, const float *a, int lda, const float *b, int ldb,
                         const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                         int m, int n, int p, float relu_dropout_scale, const uint32_t *pos_bits, int words_per_row,
                         const float *d_da_row_scale);

/* (The packed-dH1 entry points — gcnhip_rowpack_*, gcnhip_matmul_bwd_packed, gcnhip_graphsum_packed: built, bit-identical,
 * measured slower, compiled only by `make EXPERIMENTS=1` — are declared in gcnhip_experimental.h, not in this header.) */

/* Multi-GPU form of the same backward.  dH1 = mask . (dZ0 . W2^T) is cheap to recompute and 128 floats wide,
 * while its inputs are 48 floats (dZ0) and 1 bit per element (mask = H1 > 0): ranks all-gather those and each
 * rebuilds dH1 for every row.  gcnhip_pack_positive writes bit (c & 31) of bits[r*words_per_row + (c >> 5)] =
 * (h[r,c] > 0); gcnhip_matmul_bwd_da_bits computes da[m x n] = bit ? scale * (dc . b^T) : 0 for all m rows. */
int gcnhip_pack_positive(gcnhip_ctx *ctx, const float *h, int ld, int n_rows, int dim, uint32_t *bits, int words_per_row);
int gcnhip_matmul_bwd_da_bits(gcnhip_ctx *ctx, const float *b, int ldb, const float *dc, int lddc,
                              float *da, int ldda, int m, int n, int p,
                              const uint32_t *h_pos_bits, int words_per_row, float scale);

/* Row packing for the halo exchange (new: the reference is single-GPU).  dst[i, 0:ld_words] = src[d_rows[i], 0:ld_words]
 * for i < n; rows are arrays of 4-byte words that are moved, not interpreted (f32 rows, bf16 rows, mask words). */
int gcnhip_gather_rows(gcnhip_ctx *ctx, const float *src, int ld_words, const int *d_rows, int n, float *dst);

/* ---- ReLU (CUDAReLU: cuda_module.cu:164-186; cuda_kernel.cu:204-219) ---------
 * mask: one byte per element, written only when training (module.cpp:180). */
int gcnhip_relu_fwd(gcnhip_ctx *ctx, float *x, uint8_t *mask, int64_t n, int training);
int gcnhip_relu_bwd(gcnhip_ctx *ctx, float *grad, const uint8_t *mask, int64_t n);
/* the same on a [rows x cols] matrix with leading dimension ld (the model's padded layouts, e.g. 40 columns at
 * ld 48): element (r, c) is the reference's flat element r*cols + c — masks stay dense [rows*cols] */
int gcnhip_relu_fwd_2d(gcnhip_ctx *ctx, float *x, int ld, int rows, int cols, uint8_t *mask, int training);
int gcnhip_relu_bwd_2d(gcnhip_ctx *ctx, float *grad, int ld, int rows, int cols, const uint8_t *mask);

/* ---- Dropout (CUDADropout: cuda_module.cu:201-227; cuda_kernel.cu:223-240;
 *      CPU: src/seq/module.cpp:207-233) -----------------------------------------
 * x[i] *= keep(i) ? 1/(1-p) : 0;  mask[i] = keep(i) when mask != NULL.
 * keep(i) <=> U16(j) >= thr16 with j = elem_offset + i, thr16 = round(p * 65536) — the reference's
 * test (int)RAND() >= int(p * 0x7fffffff) (module.cpp:211-215) at 16-bit resolution.  U16 is generated
 * bit-sliced from a counter-based generator: with c = j >> 7, g = (j >> 5) & 3, b = j & 31 and
 *     P_i = Philox4x32-10(counter = {lo32 c, hi32 c, epoch, i}, key = {lo32 seed, hi32 seed})[g]
 * for planes i = 1 (MSB) .. n = 16 - ctz(thr16):  ge = ~0; for i = n..1: ge = bit(thr16, 16-i) ?
 * (P_i & ge) : (P_i | ge);  keep = bit b of ge  (p = 0.5 needs one plane: 128 decisions per Philox
 * block; thr16 == 0 keeps everything).  epoch = d_epoch ? *d_epoch : 0 is a DEVICE word, so a captured
 * graph replays with fresh masks.  This replaces the reference's sequential xorshift128+
 * (rand.cpp:17-28) and its 1024 shared curand states (cuda_kernel.cu:229), and is invariant to how
 * rows are partitioned across GPUs.  For bit-parity runs against the CPU path pass the reference's
 * own decisions in keep_in (one byte per element); then the generator is not used. */
int gcnhip_dropout_fwd(gcnhip_ctx *ctx, float *x, int32_t *mask, int64_t n, float p,
                       uint64_t seed, const uint32_t *d_epoch, uint64_t elem_offset,
                       const uint8_t *keep_in);
int gcnhip_dropout_bwd(gcnhip_ctx *ctx, float *grad, const int32_t *mask, int64_t n, float p);
/* [rows x cols] with leading dimension ld; the decision for (r, c) is keep(elem_offset + r*cols + c) or
 * keep_in[r*cols + c]; mask is dense [rows*cols] */
int gcnhip_dropout_fwd_2d(gcnhip_ctx *ctx, float *x, int ld, int rows, int cols, int32_t *mask, float p,
                          uint64_t seed, const uint32_t *d_epoch, uint64_t elem_offset, const uint8_t *keep_in);
int gcnhip_dropout_bwd_2d(gcnhip_ctx *ctx, float *grad, int ld, int rows, int cols, const int32_t *mask, float p);
/* backward of the fused ReLU+Dropout: grad[i] = h[i] > 0 ? scale * grad[i] : 0 */
int gcnhip_relu_dropout_bwd(gcnhip_ctx *ctx, float *grad, int ld_grad, const float *h, int ld_h,
                            int n_rows, int dim, float scale);

/* ---- CrossEntropyLoss + accuracy (CUDACrossEntropyLoss::forward:
 *      cuda_module.cu:117-147, kernels cuda_kernel.cu:166-200; accuracy:
 *      cuda_gcn.cu:100-120; CPU: module.cpp:124-161, gcn.cpp:83-96) ------------
 * One pass over the labelled rows (truth[i] >= 0): max-shift, sum-exp, loss_i,
 * and — because the max is already in hand — the accuracy test "no logit above
 * the true one" (ties count as correct).  training: grad = (softmax - onehot)
 * / count for labelled rows, 0 for the others.  count > 0 is the number of
 * labelled rows (the caller knows it from the split); count == 0 makes the
 * kernel count first (second launch), as the reference does.
 * shift_in_place != 0 also writes logits -= max like the reference
 * (module.cpp:140).  Results land in d_result[4] = {loss_sum, count,
 * correct, total} (floats; exact integers up to 2^24 rows per GPU are kept
 * in the two int words d_result_i[2] = {correct, total}); nothing is copied
 * to the host: read them with gcnhip_d2h or let gcnhip_metrics_* do it. */
int gcnhip_xent_fwd(gcnhip_ctx *ctx, float *logits, int ld, float *grad, int ld_grad,
                    const int32_t *truth, int n_rows, int num_classes, int training,
                    int count, int shift_in_place, float *d_result, int32_t *d_result_i);
/* The same over a list of rows known to be labelled (d_rows[n_listed], ascending row ids of one split; count = the
 * split's size over all ranks): only those rows of logits are read and only those rows of grad are written — the
 * caller guarantees that every other row of grad is already zero (it is after allocation, and stays so because the
 * training split never changes).  Losses are added in list order per wave, so the result equals gcnhip_xent_fwd's
 * to rounding, not bit for bit. */
int gcnhip_xent_fwd_rows(gcnhip_ctx *ctx, float *logits, int ld, float *grad, int ld_grad,
                         const int32_t *truth, const int32_t *d_rows, int n_listed, int num_classes, int training,
                         int count, int shift_in_place, float *d_result, int32_t *d_result_i);
/* ... with row r of grad multiplied by d_grad_row_scale[r] (NULL: as above) — the factored aggregation's input dinv . dZ */
int gcnhip_xent_fwd_rows_scaled(gcnhip_ctx *ctx, float *logits, int ld, float *grad, int ld_grad,
                                const int32_t *truth, const int32_t *d_rows, int n_listed, int num_classes, int training,
                                int count, int shift_in_place, float *d_result, int32_t *d_result_i, const float *d_grad_row_scale);
/* The end of gcnhip_xent_fwd_rows from the per-row terms a loss epilogue left (gcnhip_gs_loss above): d_result / d_result_i
 * exactly as gcnhip_xent_fwd_rows(_scaled) would have written them from the stored logits (same per-lane order, same block
 * partials, same final reduction — bit for bit), an armed metrics record included. */
int gcnhip_xent_from_row_terms(gcnhip_ctx *ctx, const float *d_row_terms, const int32_t *truth, const int32_t *d_rows, int n_listed,
                               float *d_result, int32_t *d_result_i);
/* accuracy alone (cuda_gcn.cu:100-120 without the 38 MB D2H) */
int gcnhip_accuracy(gcnhip_ctx *ctx, const float *logits, int ld, const int32_t *truth,
                    int n_rows, int num_classes, int32_t *d_result_i);
/* truth[i] = split[i] == s ? label[i] : -1   (cuda_kernel.cu:283-288) */
int gcnhip_set_truth(gcnhip_ctx *ctx, int32_t *truth, const int32_t *split, const int32_t *label,
                     int n, int s);
/* *d_out = sum x[i]^2   (the thrust transform+reduce of cuda_gcn.cu:122-134) */
int gcnhip_sumsq(gcnhip_ctx *ctx, const float *x, int64_t n, float *d_out);

/* ---- Adam (CUDAAdam::step: cuda_module.cu:253-263; cuda_kernel.cu:270-281;
 *      CPU: src/seq/optim.cpp:24-37) ---------------------------------------------
 * One launch for up to 4 variables.  step_size is
 * lr*sqrtf(1-beta2^t)/(1-beta1^t) computed by the caller (optim.cpp:26), or
 * read from d_step_sizes[*d_epoch] when d_step_sizes != NULL (graph replay).
 * "(1.0 - beta)" is evaluated in double as in the reference.  When
 * d_sumsq != NULL the kernel also leaves sum(w0^2) of the UPDATED variable 0
 * there (the L2 penalty the next loss report needs, gcn.cpp:98-105). */
typedef struct {
    float *w, *g, *m, *v;
    int64_t n;
    int decay;
} gcnhip_adam_var;
int gcnhip_adam_step(gcnhip_ctx *ctx, const gcnhip_adam_var *vars, int n_vars, float step_size,
                     const float *d_step_sizes, const uint32_t *d_epoch,
                     float beta1, float beta2, float eps, float weight_decay, float *d_sumsq);
/* The same update, and the epoch word advanced behind it in the same launch (the block that finishes last, after every
 * block has read the word): *d_epoch_done = e, *d_epoch_counter = e + 1 with e the counter's value during the launch.
 * The training pass of epoch e + 1 then reads its epoch from d_epoch_counter without a launch of its own
 * (gcnhip_counter_add), while an evaluation of epoch e's weights names its metrics row through d_epoch_done.
 * d_epoch (the index into d_step_sizes) may be d_epoch_counter itself or NULL. */
int gcnhip_adam_step_advance(gcnhip_ctx *ctx, const gcnhip_adam_var *vars, int n_vars, float step_size,
                             const float *d_step_sizes, const uint32_t *d_epoch,
                             float beta1, float beta2, float eps, float weight_decay, float *d_sumsq,
                             uint32_t *d_epoch_counter, uint32_t *d_epoch_done);

/* ---- small device utilities for graph-replayed epochs -------------------------- */
int gcnhip_counter_add(gcnhip_ctx *ctx, uint32_t *d_counter, uint32_t inc);
/* metrics ring [capacity][4 slots][8 floats]: slot `slot_in_row` (0..3) of row `*d_epoch % capacity`
 * = {loss_sum, count, correct, total, sumsq, epoch, 0, 0}.  correct/total come from d_result_i when it is
 * non-NULL, else from d_result[2..3] (the all-reduced floats of a multi-GPU run). */
int gcnhip_metrics_record(gcnhip_ctx *ctx, float *d_ring, int capacity, int slot_in_row,
                          const uint32_t *d_epoch, const float *d_result, const int32_t *d_result_i,
                          const float *d_sumsq);
/* The same row, written by the NEXT gcnhip_xent_fwd / gcnhip_xent_fwd_rows launched on ctx, from that launch's own
 * final reduction (the block that finishes last adds the block partials in block order and then fills the row: no
 * launch for the final sum, none for the copy).  One-shot: the loss call disarms it.  For a run whose d_result needs
 * no all-reduce before it is reported (one GPU); correct/total are taken from the launch's counts, as
 * gcnhip_metrics_record does with d_result_i == NULL.  Reference: the four scalars CUDACrossEntropyLoss leaves for
 * GCN::train_epoch / eval to read back (src/cuda/cuda_module.cu, src/seq/gcn.cpp:107-128). */
int gcnhip_metrics_record_with_next_loss(gcnhip_ctx *ctx, float *d_ring, int capacity, int slot_in_row,
                                         const uint32_t *d_epoch, const float *d_sumsq);

/* ---- hipGraph capture of a launch sequence (small graphs are launch-bound: ~25 kernels of a few
 *      microseconds per epoch).  Everything the ops read that changes from epoch to epoch lives in
 *      device memory (epoch word, step-size table, metrics ring), so one captured epoch replays
 *      correctly any number of times.  Ops must have run once eagerly before (scratch is sized on
 *      first use). */
int gcnhip_capture_begin(gcnhip_ctx *ctx);
int gcnhip_capture_end(gcnhip_ctx *ctx, void **graph_exec);
int gcnhip_graph_launch(gcnhip_ctx *ctx, void *graph_exec);
int gcnhip_graph_exec_destroy(void *graph_exec);

/* ---- timing (replaces the host chrono timers that the CUDA path leaves
 *      unsynchronised) ---------------------------------------------- */
int gcnhip_event_create(void **ev);
/* an event used only to order streams (gcnhip_event_record + gcnhip_stream_wait_event): no timestamp is taken, so
 * gcnhip_event_elapsed_ms must not be asked of it */
int gcnhip_event_create_sync(void **ev);
int gcnhip_event_destroy(void *ev);
int gcnhip_event_record(gcnhip_ctx *ctx, void *ev);
int gcnhip_event_elapsed_ms(void *start, void *stop, float *ms);   /* synchronises on stop */
int gcnhip_event_sync(void *ev);                                   /* the host waits for the work recorded before ev */
/* make the context's stream wait (on the device) for work recorded before `ev` on another context's stream */
int gcnhip_stream_wait_event(gcnhip_ctx *ctx, void *ev);

#ifdef __cplusplus
}
#endif
#endif
