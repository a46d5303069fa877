/*
 * gcn_oracle.h — CPU restatement of the reference's sequential GCN path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it, and only as the checker / the timed CPU baseline.  The product
 * (libgcnhip.so, libgcnhost.so, gcn-hip) never links or loads this file.
 *
 * Parity status: PINNED.  tests/test_oracle_pin.py checks this restatement
 * bit-for-bit against outputs of the reference's own objects (oracle/_ref,
 * built from /root/reference by oracle/Makefile) and against the committed
 * fixtures in tests/golden/ that those objects produced.
 *
 * Every function cites the reference lines it restates
 * (paths relative to /root/reference).  All arithmetic keeps the reference's
 * operand types and evaluation order (float vs double promotions included) so
 * that, compiled with the reference's flags (-O3, no -march, no fast-math),
 * the results are bit-identical.
 */
#ifndef GCN_ORACLE_H
#define GCN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG: src/seq/rand.cpp:5-28, src/seq/rand.h:6-11 ------------------- */
#define OR_RAND_MAX 0x7fffffff
void     or_rand_seed_time(unsigned t);             /* srand(t); state={rand(),rand()}  rand.cpp:6-15 */
void     or_rand_set_state(uint64_t s0, uint64_t s1);
void     or_rand_get_state(uint64_t *s0, uint64_t *s1);
uint32_t or_rand_next(void);                        /* xorshift128plus, rand.cpp:17-28 */

/* ---- Variable helpers: src/seq/variable.cpp:11-18 ----------------------- */
void or_glorot(float *w, int size, int in_size, int out_size);

/* ---- Modules: src/seq/module.cpp ---------------------------------------- */
/* Matmul  c[m x p] = a[m x n] . b[n x p]                     module.cpp:11-22 */
void or_matmul_fwd(const float *a, const float *b, float *c, int m, int n, int p);
/* a_grad = c_grad . b^T (assign), b_grad = a^T . c_grad      module.cpp:24-42 */
void or_matmul_bwd(const float *a, const float *b, const float *c_grad,
                   float *a_grad, float *b_grad, int m, int n, int p);
/* SparseMatmul  c[i,:] = sum_jj val[jj] * b[idx[jj],:]       module.cpp:47-61 */
void or_spmm_fwd(const int *indptr, const int *indices, int n_rows,
                 const float *val, const float *b, float *c, int p);
/* b_grad[idx[jj],:] += c_grad[i,:] * val[jj]; b has n rows   module.cpp:63-77 */
void or_spmm_bwd(const int *indptr, const int *indices, int n_rows,
                 const float *val, const float *c_grad, float *b_grad, int n, int p);
/* GraphSum forward and backward are the same operator        module.cpp:83-119 */
void or_graphsum(const int *indptr, const int *indices, int n_rows,
                 const float *in, float *out, int dim);
/* wide-degree variant of the coefficient (64-bit degree product instead of module.cpp:92's int product):
 * process-wide switch, off by default; bit-identical wherever no product reaches 2^31 */
void or_set_wide_degree(int on);
int  or_get_wide_degree(void);
long or_graphsum_overflowing_edges(const int *indptr, const int *indices, int n_rows);
/* the same operator for a subset of source rows (out[k,:] = row rows[k]); returns how many of the
 * selected rows contain an edge whose `int` degree product (module.cpp:91-93) would overflow */
int or_graphsum_rows(const int *indptr, const int *indices, const int *rows, int n_sel,
                     const float *in, float *out, int dim);
/* CrossEntropyLoss::forward; grad may be NULL (eval)         module.cpp:124-161
 * logits are shifted in place exactly as the reference does. */
void or_xent_fwd(float *logits, float *grad, const int *truth,
                 int n_rows, int num_classes, int training, float *loss);
/* ReLU; mask is one byte per element                         module.cpp:175-194 */
void or_relu_fwd(float *x, unsigned char *mask, int n, int training);
void or_relu_bwd(float *grad, const unsigned char *mask, int n);
/* Dropout; mask may be NULL (input without grad)             module.cpp:207-233 */
void or_dropout_fwd(float *x, int *mask, int n, float p, int training);
void or_dropout_bwd(float *grad, const int *mask, int n, float p);

/* ---- Adam: src/seq/optim.cpp:6-37 ---------------------------------------- */
typedef struct {
    float lr, beta1, beta2, eps, weight_decay;
} or_adam_params;
or_adam_params or_adam_default(void);               /* optim.cpp:6-8 */
/* one variable, given the already-incremented step_count     optim.cpp:24-37 */
void or_adam_step_var(float *w, const float *g, float *m, float *v, int n,
                      int decay, int step_count, const or_adam_params *ap);

/* ---- GCN driver: src/seq/gcn.cpp ------------------------------------------ */
typedef struct {
    int num_nodes, input_dim, hidden_dim, output_dim;
    float dropout, learning_rate, weight_decay;
    int epochs, early_stopping;
} or_params;
or_params or_params_default(void);                  /* gcn.cpp:9-11 */

typedef struct {
    /* graph CSR with the self loop first in every row (parser.cpp:20-46) */
    int *g_indptr, *g_indices; int g_nnz;
    /* feature CSR + values (parser.cpp:52-92) */
    int *f_indptr, *f_indices; float *f_val; int f_nnz;
    int *split, *label;
    int n_split, n_label;
} or_data;

/* Parser: src/common/parser.cpp:11-118.  root is the directory that holds
 * <name>.graph/.split/.svmlight ("data/" in the reference).  Returns 0 on
 * success, -1 if a file cannot be opened.  Sets num_nodes/input_dim/output_dim. */
int  or_parse(const char *root, const char *name, or_params *p, or_data *d);
void or_data_free(or_data *d);

typedef struct or_gcn or_gcn;
/* GCN::GCN (gcn.cpp:13-66).  seed_time >= 0: seed the RNG like
 * init_rand_state() would with time(NULL)==seed_time; < 0: keep the current
 * RNG state (caller already seeded). */
or_gcn *or_gcn_create(or_params p, const or_data *d, long seed_time);
void    or_gcn_destroy(or_gcn *g);
void    or_gcn_train_epoch(or_gcn *g, float *loss, float *acc);   /* gcn.cpp:107-118 */
void    or_gcn_eval(or_gcn *g, int split, float *loss, float *acc); /* gcn.cpp:120-128 */
/* GCN::run (gcn.cpp:130-158): prints the reference's lines to stdout.  If
 * trace != NULL it receives 4 floats per epoch run (train_loss, train_acc,
 * val_loss, val_acc) then test_loss, test_acc; returns epochs run. */
int     or_gcn_run(or_gcn *g, float *trace, int quiet);
/* accessors for tests: variable k as in gcn.cpp:21-54 (0 input,1 H0,2 W1,
 * 3 H1,4 Z0,5 W2,6 Z) */
float  *or_gcn_var_data(or_gcn *g, int k, int *size);
float  *or_gcn_var_grad(or_gcn *g, int k, int *size);
/* wall-clock seconds accumulated per reference timer id (timer.h:5-20) */
double  or_timer_total(int id);
void    or_timer_reset(void);

#ifdef __cplusplus
}
#endif
#endif
