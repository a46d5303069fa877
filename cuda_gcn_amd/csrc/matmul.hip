// matmul.hip — Matmul forward / backward (src/seq/module.cpp:11-42; the
// reference's 32x32 shared-memory tiles: src/cuda/cuda_kernel.cu:6-96, whose
// dB kernel runs 8 blocks that each loop over all N rows serially).
// All three products are HBM-bound here (N x 128 and N x 41 operands, 21 KB of
// weights): the small operand sits in LDS, the long one streams once through
// exact-f32 MFMA tiles (dense_kernels.h).
#include "dense_kernels.h"
#include "class_bf16x3.h"

extern "C" {

int gcnhip_matmul_fwd(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb,
                      float *cc, int ldc, int m, int n, int p) {
    if (!c || !a || !b || !cc || m < 0 || n <= 0 || p <= 0 || lda < n || ldb < p || ldc < p) return -1;
    if (m == 0) return 0;
    // option gemm_bf16x3 (round 5): hidden width 128, at most 64 classes — the product from three bf16 planes per operand
    if (c->opt.gemm_bf16x3 >= 1 && c->opt.cls_fwd && cls_fwd_fits(a, lda, cc, ldc, m, n, p)) return launch_class_fwd(c, a, lda, b, ldb, cc, ldc, m, p);
    return launch_rowstream(c, a, lda, b, ldb, 0, cc, ldc, m, n, p, nullptr, 0, 1.f);
}

// dA and dB of the class layer in one launch (class_bf16x3.h)
static int launch_class_bwd(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb, const float *dc, int lddc, float *da, int ldda,
                            float *db, int lddb, int m, int p, float scale, const uint32_t *bits, const float *rowscale) {
    ClsBwdArgs k;
    k.dz = dc; k.lddz = lddc; k.h1 = a; k.ldh = lda; k.w2 = b; k.ldw = ldb; k.da = da; k.ldda = ldda; k.bits = bits;
    k.rowscale = rowscale; k.scale = scale; k.p_ld = (p + 3) / 4 * 4; k.m = m; k.p = p; k.n_rb = ceil_div(m, 32);
    const int nks = (p + 15) / 16;
    int grid = ceil_div(k.n_rb, CLS_BWD_PAIRS);
    if (grid > c->n_cu) grid = c->n_cu;                      // one workgroup of 8 waves per CU, every wave keeps its share of dW2 in registers
    const int rc = ensure_slab(c, (size_t)grid * 128 * k.p_ld * sizeof(float));
    if (rc) return rc;
    k.slab = c->slab;
    // (the attribute is set per launch, as the other kernels with more than 64 KB of LDS do: it belongs to the function on the
    //  CURRENT device, and a process may drive several — gcn-hip with GCN_GPUS=N runs one host thread per GPU)
#define CLS_BWD(...)                                                                                                              \
    do {                                                                                                                          \
        GCNHIP_TRY(hipFuncSetAttribute((const void *)class_bwd_bf16x3_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, CLS_BWD_LDS)); \
        class_bwd_bf16x3_kernel<__VA_ARGS__><<<grid, 512, CLS_BWD_LDS, c->stream>>>(k);                                          \
    } while (0)
    const int abl = c->opt.cls_abl & 7;                       // measurement aid (tools/bench_class.py), p = 33..48 only
    if (abl && nks == 3) {
        if (abl == 1) CLS_BWD(3, 1); else if (abl == 2) CLS_BWD(3, 2); else if (abl == 3) CLS_BWD(3, 3); else CLS_BWD(3, 4);
    } else {
        switch (nks) {
            case 1: CLS_BWD(1); break;
            case 2: CLS_BWD(2); break;
            case 3: CLS_BWD(3); break;
            default: CLS_BWD(4); break;
        }
    }
#undef CLS_BWD
    GCNHIP_LAUNCH_CHECK();
    launch_slab_reduce(k.slab, grid, 128, p, k.p_ld, db, lddb, c->stream);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

static int matmul_bwd_impl(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb,
                           const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                           int m, int n, int p, int fused, float scale) {
    if (!c || !a || !b || !dc || m < 0 || n <= 0 || p <= 0 || lda < n || ldb < p || lddc < p) return -1;
    if ((da && ldda < n) || (db && lddb < p)) return -1;
    if (m == 0) {
        if (db) for (int j = 0; j < n; j++) GCNHIP_TRY(hipMemsetAsync(db + (size_t)j * lddb, 0, p * sizeof(float), c->stream));
        return 0;
    }
    if (db) {                                 // db = a^T . dc, module.cpp:35 (a is still the forward input here)
        const int rc = launch_atb(c, a, lda, dc, lddc, db, lddb, m, n, p, 0, 0.f, 0, nullptr, 0, nullptr);
        if (rc) return rc;
    }
    if (da) {                                 // da = dc . b^T, module.cpp:34,37
        const int rc = launch_rowstream(c, dc, lddc, b, ldb, 1, da, ldda, m, p, n, fused ? a : nullptr, lda, scale);
        if (rc) return rc;
    }
    return 0;
}

int gcnhip_matmul_bwd(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb,
                      const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                      int m, int n, int p) {
    return matmul_bwd_impl(c, a, lda, b, ldb, dc, lddc, da, ldda, db, lddb, m, n, p, 0, 1.f);
}

int gcnhip_matmul_bwd_fused(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb,
                            const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                            int m, int n, int p, float relu_dropout_scale) {
    return matmul_bwd_impl(c, a, lda, b, ldb, dc, lddc, da, ldda, db, lddb, m, n, p, 1, relu_dropout_scale);
}

// gcnhip_matmul_bwd_fused with the mask read from bits (written by gcnhip_graphsum_relu_dropout_bits) instead of from the
// activations: db still needs a (= H1), da does not
int gcnhip_matmul_bwd_fused_bits(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb,
                                 const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                                 int m, int n, int p, float relu_dropout_scale, const uint32_t *pos_bits, int words_per_row) {
    if (!c || !a || !b || !dc || !da || !pos_bits || m < 0 || n <= 0 || p <= 0 || lda < n || ldb < p || lddc < p || ldda < n) return -1;
    if (db && lddb < p) return -1;
    if (words_per_row * 32 < n) return -1;
    if (m == 0) return matmul_bwd_impl(c, a, lda, b, ldb, dc, lddc, nullptr, 0, db, lddb, m, n, p, 0, 1.f);
    if (db) {
        const int rc = launch_atb(c, a, lda, dc, lddc, db, lddb, m, n, p, 0, 0.f, 0, nullptr, 0, nullptr);
        if (rc) return rc;
    }
    return launch_rowstream(c, dc, lddc, b, ldb, 1, da, ldda, m, p, n, nullptr, 0, relu_dropout_scale, pos_bits, words_per_row);
}

// the same with da leaving as packed rows (dense_kernels.h): da_dense receives only the halves that do not fit a slot
#ifndef GCNHIP_EXPERIMENTS
int gcnhip_matmul_bwd_packed(gcnhip_ctx *, const float *, int, const float *, int, const float *, int, float *, int, gcnhip_rowpack *,
                             float *, int, int, int, int, float) {
    return gcnhip_fail("gcnhip_matmul_bwd_packed is a measured-slower experiment: build the library with `make EXPERIMENTS=1`");
}
#else
int gcnhip_matmul_bwd_packed(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb,
                             const float *dc, int lddc, float *da_dense, int ldda, gcnhip_rowpack *pack,
                             float *db, int lddb, int m, int n, int p, float relu_dropout_scale) {
    if (!c || !a || !b || !dc || !da_dense || !pack || m < 0 || n <= 0 || p <= 0 || lda < n || ldb < p || lddc < p || ldda < n) return -1;
    if (pack->rows != m || pack->cols != n) return -1;
    if (db && lddb < p) return -1;
    if (m == 0) return matmul_bwd_impl(c, a, lda, b, ldb, dc, lddc, nullptr, 0, db, lddb, m, n, p, 0, 1.f);
    if (db) {
        const int rc = launch_atb(c, a, lda, dc, lddc, db, lddb, m, n, p, 0, 0.f, 0, nullptr, 0, nullptr);
        if (rc) return rc;
    }
    return launch_rowstream(c, dc, lddc, b, ldb, 1, da_dense, ldda, m, p, n, a, lda, relu_dropout_scale, nullptr, 0, pack->slots, pack->halves);
}
#endif

// Every form of the fused backward behind one call, with an optional factor per row of da (the factored aggregation,
// gcnhip_graphsum_ex: dH1' = dinv^2 . dH1):  db = a^T . dc when db != NULL (a is then required);
// da[r, :] = mask . (relu_dropout_scale * da_row_scale[r]) . (dc . b^T)[r, :], mask = pos_bits when given, else a > 0.
int gcnhip_matmul_bwd_ex(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb,
                         const float *dc, int lddc, float *da, int ldda, float *db, int lddb,
                         int m, int n, int p, float relu_dropout_scale, const uint32_t *pos_bits, int words_per_row,
                         const float *d_da_row_scale) {
    if (!c || !b || !dc || !da || m < 0 || n <= 0 || p <= 0 || ldb < p || lddc < p || ldda < n) return -1;
    if ((db || !pos_bits) && (!a || lda < n)) return -1;
    if (db && lddb < p) return -1;
    if (pos_bits && words_per_row * 32 < n) return -1;
    if (!(relu_dropout_scale > 0.f)) return -1;
    if (m == 0) {
        if (db) for (int j = 0; j < n; j++) GCNHIP_TRY(hipMemsetAsync(db + (size_t)j * lddb, 0, p * sizeof(float), c->stream));
        return 0;
    }
    if (db && c->opt.gemm_bf16x3 >= 1 && cls_bwd_fits(a, lda, dc, lddc, da, ldda, pos_bits, words_per_row, m, n, p))
        return launch_class_bwd(c, a, lda, b, ldb, dc, lddc, da, ldda, db, lddb, m, p, relu_dropout_scale, pos_bits, d_da_row_scale);
    if (db) {
        const int rc = launch_atb(c, a, lda, dc, lddc, db, lddb, m, n, p, 0, 0.f, 0, nullptr, 0, nullptr);
        if (rc) return rc;
    }
    return launch_rowstream(c, dc, lddc, b, ldb, 1, da, ldda, m, p, n, pos_bits ? nullptr : a, lda, relu_dropout_scale, pos_bits, words_per_row,
                            nullptr, 0, d_da_row_scale);
}

// da for ALL m rows from a bit mask instead of the forward activations (multi-GPU: every rank rebuilds the
// whole dH1 from the gathered dZ0 and 1 bit per element of H1, instead of gathering dH1 itself)
int gcnhip_matmul_bwd_da_bits(gcnhip_ctx *c, const float *b, int ldb, const float *dc, int lddc,
                              float *da, int ldda, int m, int n, int p,
                              const uint32_t *h_pos_bits, int words_per_row, float scale) {
    if (!c || !b || !dc || !da || !h_pos_bits || m < 0 || n <= 0 || p <= 0 || ldb < p || lddc < p || ldda < n) return -1;
    if (words_per_row * 32 < n) return -1;
    if (m == 0) return 0;
    return launch_rowstream(c, dc, lddc, b, ldb, 1, da, ldda, m, p, n, nullptr, 0, scale, h_pos_bits, words_per_row);
}

}  // extern "C"

GCNHIP_DEFINE_PRELOAD(matmul, (gemm_atb_kernel<4, 4>))
