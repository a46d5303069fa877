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
 * This header is the 1:1 surface: what such a binding calls.  The fused, restricted, row-subset,
 * mask-bit, factored and graph-replay variants that THIS repository's host driver (cuda_gcn_amd/host) adds on top are
 * its private protocol and live in gcnhip_driver.h; measured-slower experiments in gcnhip_experimental.h.
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
typedef struct gcnhip_graph gcnhip_graph;   /* prepared adjacency (CUDASparseIndex of the graph) */
typedef struct gcnhip_feat gcnhip_feat;     /* prepared feature matrix (CUDASparseIndex of X + its values) */

/* ---- context / runtime (replaces the implicit null stream + CUDA_CHECK) ---- */
int  gcnhip_device_count(int *count);
/* stream == NULL: the context creates and owns a non-blocking stream.
 * Otherwise `stream` is a hipStream_t owned by the caller. */
int  gcnhip_ctx_create(gcnhip_ctx **ctx, int device, void *stream);
int  gcnhip_ctx_destroy(gcnhip_ctx *ctx);
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

/* ---- features (CUDASparseIndex of X + the input CUDAVariable) ----------------
 * CSR of X with n_rows rows and n_cols (= input_dim) columns.  A matrix whose
 * every row holds exactly the columns 0..n_cols-1 in order is detected as
 * dense and its index array is not kept on the device.  A column-major index
 * (CSC) is built once so the weight gradient is a gather, not the racy
 * scatter of cuda_kernel.cu:112-122. */
int gcnhip_feat_create(gcnhip_ctx *ctx, gcnhip_feat **f, const int *h_indptr, const int *h_indices,
                       const float *h_values, int n_rows, int n_cols);
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
int gcnhip_spmm_bwd(gcnhip_ctx *ctx, const gcnhip_feat *f, const float *vals, const float *dout, int ld_dout,
                    float *dw, int ld_dw, int p, float p_drop, uint64_t seed, const uint32_t *d_epoch,
                    uint64_t nnz_offset, const uint8_t *keep_mask);
/* ---- Matmul (CUDAMatmul: cuda_module.cu:8-34; kernels cuda_kernel.cu:6-96;
 *      CPU: src/seq/module.cpp:11-42) -------------------------------------------
 * forward : c[m x p] = a[m x n] . b[n x p]
 * backward: da = dc . b^T (assigned); db = a^T . dc (assigned) */
int gcnhip_matmul_fwd(gcnhip_ctx *ctx, const float *a, int lda, const float *b, int ldb,
                      float *c, int ldc, int m, int n, int p);
int gcnhip_matmul_bwd(gcnhip_ctx *ctx, const float *a, int lda, const float *b, int ldb,
                      const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                      int m, int n, int p);
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
