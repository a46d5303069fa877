/* gcnhip_experimental.h — entry points of libgcnhip.so that are NOT part of the drop-in surface (include/gcnhip.h).
 *
 * Variants that were built, are bit-identical to the default path and measured SLOWER on MI355X (lab notebook).
 * They are compiled only by `make EXPERIMENTS=1` (gcnhip_experiments() == 1); in the default library these symbols exist
 * and return -1 with a message, so that a host built against this header still links.  Their tests skip without the
 * experiments build.  The context options gs_pipe, gs_nt, gs_fold, gemm_persist_bwd and spmm_lds belong to the same set.
 */
#ifndef GCNHIP_EXPERIMENTAL_H
#define GCNHIP_EXPERIMENTAL_H
#include "gcnhip_driver.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Packed dH1 (exact).  ReLU and dropout zero about three quarters of dH1 = mask . (dZ0 . W2^T), at positions known
 * from H1, and its only reader is the hidden layer's backward aggregation (module.cpp:103-119), which pays per
 * 128-byte line gathered.  gcnhip_matmul_bwd_packed is gcnhip_matmul_bwd_fused writing every 64-column half of a row
 * as ONE 128-byte slot (64-bit mask + the masked-in f32 values in column order, at most 30); a half with more
 * masked-in columns is written to da_dense as usual and its slot holds only the mask.  gcnhip_graphsum_packed gathers
 * from the slots (falling back to `dense` for halves that did not fit): half the lines per edge, and — same lane
 * groups, same order of the non-zero terms — bit-identical to gcnhip_graphsum on the dense matrix.  cols % 64 == 0.
 * gcnhip_rowpack_expand rebuilds the dense image in place (tests, introspection). */
int gcnhip_rowpack_create(gcnhip_ctx *ctx, gcnhip_rowpack **p, int rows, int cols);
int gcnhip_rowpack_destroy(gcnhip_ctx *ctx, gcnhip_rowpack *p);
int gcnhip_rowpack_expand(gcnhip_ctx *ctx, const gcnhip_rowpack *p, float *dense, int ld);
int gcnhip_matmul_bwd_packed(gcnhip_ctx *ctx, const float *a, int lda, const float *b, int ldb,
                             const float *dc, int lddc, float *da_dense, int ldda, gcnhip_rowpack *pack,
                             float *db, int lddb, int m, int n, int p, float relu_dropout_scale);
int gcnhip_graphsum_packed(gcnhip_ctx *ctx, const gcnhip_graph *g, const gcnhip_rowpack *p, const float *dense, int ld_dense,
                           float *out, int ld_out);

#ifdef __cplusplus
}
#endif
#endif
