/*
 * ref_shim.cpp — C-callable shim over the REFERENCE's own objects.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with
 * the reference's sources *where they lie* under /root/reference (see
 * oracle/Makefile, target `ref`) into oracle/_ref/libref.so.  It can only be BUILT
 * in the build container (the reference's sources never travel and none is
 * copied into this repo); the built library is git-ignored but does travel
 * to the GPU box with the repo snapshot, like our own .so files.  It is used to
 *   (1) pin oracle/gcn_oracle.c bit-for-bit against the reference,
 *   (2) generate the fixtures committed under tests/golden/, and
 *   (3) be the timed CPU baseline of bench.py (cpu_baseline.kind "reference")
 * and by nothing else: tests/ and bench.py's cpu_baseline leg are its only loaders.
 *
 * No reference source is modified: time(NULL) in src/seq/rand.cpp:7 is
 * redirected at link time (-Wl,--wrap=time) to __wrap_time below, and the
 * private members of GCN are reached with the usual test-only
 * `#define class struct` / `#define private public` around the include.
 */
#include <cstdint>
#include <cstring>
#include <ctime>
#include <vector>
#include <utility>
#include <string>
#include <sstream>
#include <iostream>
#include <fstream>

/* GCN declares its members with the default (private) access of `class`; every
 * standard header gcn.h pulls in is already included above, so the keyword
 * swap below touches the reference's own class declarations only. */
#define class struct
#define private public
#include "gcn.h"
#undef private
#undef class
#include "module.h"
#include "optim.h"
#include "rand.h"
#include "variable.h"
#include "sparse.h"

static long g_fake_time = 0;
extern "C" time_t __wrap_time(time_t *t) {
    if (t) *t = (time_t)g_fake_time;
    return (time_t)g_fake_time;
}

namespace {
void fill(std::vector<float> &v, const float *src) { if (!v.empty()) memcpy(v.data(), src, v.size() * sizeof(float)); }
void take(const std::vector<float> &v, float *dst) { if (!v.empty()) memcpy(dst, v.data(), v.size() * sizeof(float)); }
SparseIndex make_csr(const int *indptr, const int *indices, int n_rows) {
    SparseIndex s;
    s.indptr.assign(indptr, indptr + n_rows + 1);
    s.indices.assign(indices, indices + indptr[n_rows]);
    return s;
}
}

extern "C" {

void ref_rand_seed_time(long t) { g_fake_time = t; init_rand_state(); }
void ref_rand_set_state(uint64_t s0, uint64_t s1) { rand_state[0] = s0; rand_state[1] = s1; }
void ref_rand_get_state(uint64_t *s0, uint64_t *s1) { *s0 = rand_state[0]; *s1 = rand_state[1]; }
uint32_t ref_rand_next(void) { return RAND(); }

void ref_glorot(float *w, int size, int in_size, int out_size) {
    Variable v(size);
    v.glorot(in_size, out_size);
    take(v.data, w);
}

void ref_matmul_fwd(const float *a, const float *b, float *c, int m, int n, int p) {
    Variable A(m * n), B(n * p), C(m * p);
    fill(A.data, a); fill(B.data, b);
    Matmul mod(&A, &B, &C, m, n, p);
    mod.forward(true);
    take(C.data, c);
}

void ref_matmul_bwd(const float *a, const float *b, const float *c_grad,
                    float *a_grad, float *b_grad, int m, int n, int p) {
    Variable A(m * n), B(n * p), C(m * p);
    fill(A.data, a); fill(B.data, b); fill(C.grad, c_grad);
    Matmul mod(&A, &B, &C, m, n, p);
    mod.backward();
    take(A.grad, a_grad); take(B.grad, b_grad);
}

void ref_spmm_fwd(const int *indptr, const int *indices, int n_rows,
                  const float *val, const float *b, float *c, int n, int p) {
    SparseIndex sp = make_csr(indptr, indices, n_rows);
    Variable A(indptr[n_rows], false), B(n * p), C(n_rows * p);
    fill(A.data, val); fill(B.data, b);
    SparseMatmul mod(&A, &B, &C, &sp, n_rows, n, p);
    mod.forward(true);
    take(C.data, c);
}

void ref_spmm_bwd(const int *indptr, const int *indices, int n_rows,
                  const float *val, const float *c_grad, float *b_grad, int n, int p) {
    SparseIndex sp = make_csr(indptr, indices, n_rows);
    Variable A(indptr[n_rows], false), B(n * p), C(n_rows * p);
    fill(A.data, val); fill(C.grad, c_grad);
    SparseMatmul mod(&A, &B, &C, &sp, n_rows, n, p);
    mod.backward();
    take(B.grad, b_grad);
}

void ref_graphsum_fwd(const int *indptr, const int *indices, int n_rows,
                      const float *in, float *out, int dim) {
    SparseIndex g = make_csr(indptr, indices, n_rows);
    Variable I(n_rows * dim), O(n_rows * dim);
    fill(I.data, in);
    GraphSum mod(&I, &O, &g, dim);
    mod.forward(true);
    take(O.data, out);
}

void ref_graphsum_bwd(const int *indptr, const int *indices, int n_rows,
                      const float *out_grad, float *in_grad, int dim) {
    SparseIndex g = make_csr(indptr, indices, n_rows);
    Variable I(n_rows * dim), O(n_rows * dim);
    fill(O.grad, out_grad);
    GraphSum mod(&I, &O, &g, dim);
    mod.backward();
    take(I.grad, in_grad);
}

void ref_xent_fwd(float *logits, float *grad, int *truth, int n_rows, int num_classes,
                  int training, float *loss) {
    Variable L(n_rows * num_classes);
    fill(L.data, logits);
    CrossEntropyLoss mod(&L, truth, loss, num_classes);
    mod.forward(training != 0);
    take(L.data, logits);
    if (training && grad) take(L.grad, grad);
}

/* ReLU forward then (optionally) backward through the same module object */
void ref_relu(float *x, float *grad, int n, int training, int do_backward) {
    Variable V(n);
    fill(V.data, x);
    if (grad) fill(V.grad, grad);
    ReLU mod(&V);
    mod.forward(training != 0);
    take(V.data, x);
    if (do_backward && grad) { mod.backward(); take(V.grad, grad); }
}

/* Dropout forward then (optionally) backward; consumes the global RNG */
void ref_dropout(float *x, float *grad, int n, float p, int training, int requires_grad, int do_backward) {
    Variable V(n, requires_grad != 0);
    fill(V.data, x);
    if (requires_grad && grad) fill(V.grad, grad);
    Dropout mod(&V, p);
    mod.forward(training != 0);
    take(V.data, x);
    if (do_backward && requires_grad && grad) { mod.backward(); take(V.grad, grad); }
}

/* k Adam steps on one variable; grads holds k*n floats (one gradient per step) */
void ref_adam_steps(float *w, const float *grads, int n, int k, int decay,
                    float lr, float weight_decay) {
    Variable V(n);
    fill(V.data, w);
    AdamParams ap = AdamParams::get_default();
    ap.lr = lr; ap.weight_decay = weight_decay;
    Adam opt({{&V, decay != 0}}, ap);
    for (int s = 0; s < k; s++) {
        memcpy(V.grad.data(), grads + (size_t)s * n, n * sizeof(float));
        opt.step();
    }
    take(V.data, w);
}

/* ---- whole model ---------------------------------------------------------- */
struct ref_model {
    GCNData data;
    GCN *gcn;
};

void *ref_gcn_create(int num_nodes, int input_dim, int hidden_dim, int output_dim,
                     float dropout, float lr, float wd, int epochs, int early_stopping,
                     const int *g_indptr, const int *g_indices,
                     const int *f_indptr, const int *f_indices, const float *f_val,
                     const int *split, const int *label, long seed_time) {
    ref_model *m = new ref_model;
    m->data.graph = make_csr(g_indptr, g_indices, num_nodes);
    m->data.feature_index = make_csr(f_indptr, f_indices, num_nodes);
    m->data.feature_value.assign(f_val, f_val + f_indptr[num_nodes]);
    m->data.split.assign(split, split + num_nodes);
    m->data.label.assign(label, label + num_nodes);
    GCNParams p = GCNParams::get_default();
    p.num_nodes = num_nodes; p.input_dim = input_dim; p.hidden_dim = hidden_dim; p.output_dim = output_dim;
    p.dropout = dropout; p.learning_rate = lr; p.weight_decay = wd; p.epochs = epochs; p.early_stopping = early_stopping;
    g_fake_time = seed_time;
    m->gcn = new GCN(p, &m->data);
    return m;
}
void ref_gcn_destroy(void *h) { ref_model *m = (ref_model *)h; delete m->gcn; delete m; }
void ref_gcn_train_epoch(void *h, float *loss, float *acc) {
    auto r = ((ref_model *)h)->gcn->train_epoch(); *loss = r.first; *acc = r.second;
}
void ref_gcn_eval(void *h, int split, float *loss, float *acc) {
    auto r = ((ref_model *)h)->gcn->eval(split); *loss = r.first; *acc = r.second;
}
void ref_gcn_run(void *h) { ((ref_model *)h)->gcn->run(); }
int ref_gcn_var_size(void *h, int k) { return (int)((ref_model *)h)->gcn->variables[k].data.size(); }
void ref_gcn_var_data(void *h, int k, float *dst) { take(((ref_model *)h)->gcn->variables[k].data, dst); }
void ref_gcn_var_grad(void *h, int k, float *dst) { take(((ref_model *)h)->gcn->variables[k].grad, dst); }

/* Parser over <cwd>/data/<name>.* (root is hard-coded in parser.cpp:12).
 * Returns sizes; arrays are copied out by ref_parse_copy. */
static GCNData g_parsed; static GCNParams g_parsed_params;
}

#include "parser.h"
extern "C" {
int ref_parse(const char *name, int *num_nodes, int *input_dim, int *output_dim,
              int *g_nnz, int *f_nnz, int *n_split, int *n_label) {
    g_parsed = GCNData();
    g_parsed_params = GCNParams::get_default();
    Parser parser(&g_parsed_params, &g_parsed, name);
    if (!parser.parse()) return -1;
    *num_nodes = g_parsed_params.num_nodes; *input_dim = g_parsed_params.input_dim; *output_dim = g_parsed_params.output_dim;
    *g_nnz = (int)g_parsed.graph.indices.size(); *f_nnz = (int)g_parsed.feature_index.indices.size();
    *n_split = (int)g_parsed.split.size(); *n_label = (int)g_parsed.label.size();
    return 0;
}
void ref_parse_copy(int *g_indptr, int *g_indices, int *f_indptr, int *f_indices, float *f_val,
                    int *split, int *label) {
    auto cp = [](const std::vector<int> &v, int *d) { if (!v.empty()) memcpy(d, v.data(), v.size() * sizeof(int)); };
    cp(g_parsed.graph.indptr, g_indptr); cp(g_parsed.graph.indices, g_indices);
    cp(g_parsed.feature_index.indptr, f_indptr); cp(g_parsed.feature_index.indices, f_indices);
    take(g_parsed.feature_value, f_val);
    cp(g_parsed.split, split); cp(g_parsed.label, label);
}
}
