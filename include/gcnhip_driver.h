/*
 * gcnhip_driver.h — entry points of libgcnhip.so beyond the 1:1 surface of gcnhip.h: the protocol of THIS repository's host
 * driver (cuda_gcn_amd/host: HipGCN and its Hip* modules).
 *
 * gcnhip.h holds one entry point per kernel-launching wrapper of the reference — what a maintainer binding the library
 * into the reference's own CUDA* modules calls.  Everything here is an optimisation of the same
 * operators that the host driver uses: fused epilogues (ReLU / dropout / mask bits / loss in the aggregation; ReLU and the
 * second product in the first-layer GEMM), operators restricted to row subsets or column sets, the factored
 * coefficients, split-K parts of the weight gradient, bf16 table storage, the fused backward of the class layer, and
 * the device-side bookkeeping of graph-replayed epochs (counters, metrics ring, capture).  Each declaration cites the
 * reference lines whose work it takes over; conventions (return codes, streams, layouts) are those of gcnhip.h.
 * tests/test_abi_cpu.py checks that the library exports every symbol of both headers.
 */
#ifndef GCNHIP_DRIVER_H
#define GCNHIP_DRIVER_H
#include "gcnhip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gcnhip_rowset gcnhip_rowset;   /* a registered subset of an adjacency object's rows */
typedef struct gcnhip_rowpack gcnhip_rowpack; /* a mostly-zero matrix stored as packed rows (gcnhip_experimental.h) */

/* Hint: the launches of this context are meant to run BESIDE another stream's kernels (HipGCN's validation lane next to
 * the training pass).  Ops that have a whole-chip persistent form (the dense first-layer GEMM: one 512-thread workgroup
 * with 150 KB of LDS per CU for the whole launch) then take their tiled form, which leaves wave slots and registers to the
 * neighbour: measured on the two-stream epoch, 294 epochs/s with the persistent form on the lane against 297 with tiles.
 * Results do not depend on the hint beyond the order of floating-point sums of the two forms (each within the tested bound). */
int gcnhip_ctx_set_corun(gcnhip_ctx *ctx, int on);
/* Read-back without stalling the producer: page-locked host memory and a device-to-host copy that is only ENQUEUED on
 * ctx's stream (wait for it with an event recorded behind it + gcnhip_event_sync).  HipGCN::run() uses them to
 * print epoch e's line while epochs e+1.. are already running: behind a group of epochs, the metrics rows of the group are
 * copied on the producing stream and an event recorded that the host waits on.  (The reference's CUDA path instead blocks
 * on a cudaMemcpy of the whole logits matrix per accuracy call, src/cuda/cuda_gcn.cu:100-120.) */
int gcnhip_host_alloc(void **ptr, size_t bytes);
int gcnhip_host_free(void *ptr);
int gcnhip_d2h_async(gcnhip_ctx *ctx, void *dst_pinned, const void *src, size_t bytes);
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

/* Aggregate-first evaluation.  Without dropout the first layer is linear in X: ReLU(A^.(X.W1)) = ReLU((A^.X).W1)
 * (src/seq/gcn.cpp:23-41 with Dropout skipped, module.cpp:208), and A^.X does not change from epoch to epoch.
 * This builds the feature object of A^.X once (dense X only; x has g->n_cols rows — every column of g); an
 * evaluation forward then runs gcnhip_spmm_fwd_relu on it and needs NO hidden-width aggregation (and, with
 * several GPUs, no exchange before the hidden layer).  Same result up to the rounding of a reassociated f32 sum.
 * Training cannot use it: its X~ changes with every epoch's dropout decisions. */
int gcnhip_feat_create_aggregated(gcnhip_ctx *ctx, gcnhip_feat **f, gcnhip_graph *g, const gcnhip_feat *x);
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

/* backward of the fused ReLU+Dropout: grad[i] = h[i] > 0 ? scale * grad[i] : 0 */
int gcnhip_relu_dropout_bwd(gcnhip_ctx *ctx, float *grad, int ld_grad, const float *h, int ld_h,
                            int n_rows, int dim, float scale);

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

#ifdef __cplusplus
}
#endif
#endif
