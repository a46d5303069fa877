/*
 * gcn_oracle.c — CPU restatement of the reference's sequential GCN path.
 * TEST INFRASTRUCTURE ONLY (see gcn_oracle.h).  Parity status: PINNED against
 * oracle/_ref (the reference's own objects) and tests/golden/.
 *
 * Build with the reference's flags (Makefile:6 of the reference): -O3, no
 * -march, no -ffast-math.  Do not "clean up" mixed float/double expressions
 * here: each one mirrors the promotion the reference's C++ performs.
 */
#define _POSIX_C_SOURCE 200809L
#include "gcn_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <time.h>

/* ------------------------------------------------------------------ timers */
/* ids follow src/common/timer.h:5-20 */
enum { T_TRAIN = 0, T_TEST, T_MM_FW, T_MM_BW, T_SP_FW, T_SP_BW, T_GS_FW, T_GS_BW,
       T_LOSS, T_RELU_FW, T_RELU_BW, T_DROP_FW, T_DROP_BW, T_NUM };
static double t_sum[T_NUM], t_t0[T_NUM];
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void t_start(int id) { t_t0[id] = now_s(); }
static double t_stop(int id) { double d = now_s() - t_t0[id]; t_sum[id] += d; return d; }
double or_timer_total(int id) { return (id >= 0 && id < T_NUM) ? t_sum[id] : 0.0; }
void or_timer_reset(void) { memset(t_sum, 0, sizeof t_sum); }

/* --------------------------------------------------------------------- RNG */
static uint64_t rstate[2];

/* src/seq/rand.cpp:6-15 — srand(time), then draw pairs until both non-zero */
void or_rand_seed_time(unsigned t) {
    int x = 0, y = 0;
    srand(t);
    while (x == 0 || y == 0) {
        x = rand();
        y = rand();
    }
    rstate[0] = (uint64_t)x;
    rstate[1] = (uint64_t)y;
}
void or_rand_set_state(uint64_t s0, uint64_t s1) { rstate[0] = s0; rstate[1] = s1; }
void or_rand_get_state(uint64_t *s0, uint64_t *s1) { *s0 = rstate[0]; *s1 = rstate[1]; }

/* src/seq/rand.cpp:17-28 — xorshift128+, output masked to 31 bits */
uint32_t or_rand_next(void) {
    uint64_t t = rstate[0];
    const uint64_t s = rstate[1];
    rstate[0] = s;
    t ^= t << 23;
    t ^= t >> 17;
    t ^= s ^ (s >> 26);
    rstate[1] = t;
    return (uint32_t)((t + s) & 0x7fffffff);
}

/* src/seq/variable.cpp:11-18.  float(RAND()) / MY_RAND_MAX is a float
 * division by (float)0x7fffffff; "- 0.5" happens in double; the product
 * "rand * range * 2" is float. */
void or_glorot(float *w, int size, int in_size, int out_size) {
    float range = sqrtf(6.0f / (in_size + out_size));
    for (int i = 0; i < size; i++) {
        const float r = (float)((double)((float)or_rand_next() / (float)OR_RAND_MAX) - 0.5);
        w[i] = r * range * 2;
    }
}

/* ------------------------------------------------------------------ Matmul */
/* src/seq/module.cpp:11-22 — i, j, k order; c zeroed first */
void or_matmul_fwd(const float *a, const float *b, float *c, int m, int n, int p) {
    t_start(T_MM_FW);
    for (long i = 0; i < (long)m * p; i++) c[i] = 0;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            const float aij = a[(long)i * n + j];
            for (int k = 0; k < p; k++)
                c[(long)i * p + k] += aij * b[(long)j * p + k];
        }
    t_stop(T_MM_FW);
}

/* src/seq/module.cpp:24-42 — a_grad assigned from a running float tmp,
 * b_grad accumulated over i in row order */
void or_matmul_bwd(const float *a, const float *b, const float *c_grad,
                   float *a_grad, float *b_grad, int m, int n, int p) {
    t_start(T_MM_BW);
    for (long i = 0; i < (long)m * n; i++) a_grad[i] = 0;
    for (long i = 0; i < (long)n * p; i++) b_grad[i] = 0;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            float tmp = 0;
            const float aij = a[(long)i * n + j];
            for (int k = 0; k < p; k++) {
                const float g = c_grad[(long)i * p + k];
                tmp += g * b[(long)j * p + k];
                b_grad[(long)j * p + k] += g * aij;
            }
            a_grad[(long)i * n + j] = tmp;
        }
    t_stop(T_MM_BW);
}

/* ------------------------------------------------------------ SparseMatmul */
/* src/seq/module.cpp:47-61 */
void or_spmm_fwd(const int *indptr, const int *indices, int n_rows,
                 const float *val, const float *b, float *c, int p) {
    t_start(T_SP_FW);
    for (long i = 0; i < (long)n_rows * p; i++) c[i] = 0;
    for (int i = 0; i < n_rows; i++)
        for (int jj = indptr[i]; jj < indptr[i + 1]; jj++) {
            const int j = indices[jj];
            const float v = val[jj];
            for (int k = 0; k < p; k++)
                c[(long)i * p + k] += v * b[(long)j * p + k];
        }
    t_stop(T_SP_FW);
}

/* src/seq/module.cpp:63-77 — scatter-add into rows of b_grad */
void or_spmm_bwd(const int *indptr, const int *indices, int n_rows,
                 const float *val, const float *c_grad, float *b_grad, int n, int p) {
    t_start(T_SP_BW);
    for (long i = 0; i < (long)n * p; i++) b_grad[i] = 0;
    for (int i = 0; i < n_rows; i++)
        for (int jj = indptr[i]; jj < indptr[i + 1]; jj++) {
            const int j = indices[jj];
            const float v = val[jj];
            for (int k = 0; k < p; k++)
                b_grad[(long)j * p + k] += c_grad[(long)i * p + k] * v;
        }
    t_stop(T_SP_BW);
}

/* ---------------------------------------------------------------- GraphSum */
/* src/seq/module.cpp:83-101 (forward) and :103-119 (backward): the same
 * row-gather.  coef: int product of the two row lengths -> float for sqrtf,
 * "1.0 / x" divides in double, the result narrows to float.
 *
 * WIDE-DEGREE VARIANT (or_set_wide_degree(1); off by default).  The reference multiplies the two row lengths as `int`
 * (module.cpp:91-93): undefined behaviour once the product reaches 2^31, i.e. for a node of degree > 46 340 (its own self
 * loop already: R-MAT scale 20 has a hub of degree 64 619, whose product wraps negative and turns the hub's row into NaN).
 * SURVEY App. C records the fix the HIP path makes: the product in 64 bits.  With the variant on, the product is formed as
 * `long` and converted to float exactly as the int product would be — int -> float and long -> float round the same way,
 * so every coefficient whose product stays below 2^31 is bit-identical (tests/test_oracle_pin.py proves it against
 * oracle/_ref on every fixture) and only the overflowing ones differ. */
static int g_wide_degree = 0;
void or_set_wide_degree(int on) { g_wide_degree = on ? 1 : 0; }
int or_get_wide_degree(void) { return g_wide_degree; }

static inline float gs_coef(const int *indptr, int src, int dst) {
    const int a = indptr[src + 1] - indptr[src], b = indptr[dst + 1] - indptr[dst];
    if (g_wide_degree) return (float)(1.0 / (double)sqrtf((float)((long)a * (long)b)));
    return (float)(1.0 / (double)sqrtf((float)(a * b)));
}

/* number of stored edges whose int degree product is not representable (>= 2^31): 0 means or_graphsum is the
 * reference's defined behaviour on this graph and the two variants agree bit for bit */
long or_graphsum_overflowing_edges(const int *indptr, const int *indices, int n_rows) {
    long n = 0;
    for (int src = 0; src < n_rows; src++)
        for (int e = indptr[src]; e < indptr[src + 1]; e++) {
            const int dst = indices[e];
            if ((long)(indptr[src + 1] - indptr[src]) * (long)(indptr[dst + 1] - indptr[dst]) >= 2147483648L) n++;
        }
    return n;
}

void or_graphsum(const int *indptr, const int *indices, int n_rows,
                 const float *in, float *out, int dim) {
    for (long i = 0; i < (long)n_rows * dim; i++) out[i] = 0;
    for (int src = 0; src < n_rows; src++)
        for (int e = indptr[src]; e < indptr[src + 1]; e++) {
            const int dst = indices[e];
            const float coef = gs_coef(indptr, src, dst);
            for (int j = 0; j < dim; j++)
                out[(long)src * dim + j] += coef * in[(long)dst * dim + j];
        }
}

/* The same loop body as or_graphsum (src/seq/module.cpp:88-99) for a chosen subset of source rows:
 * out[k,:] = row rows[k] of the full result.  For graphs too large to evaluate whole on one core
 * (R-MAT 2^21+ nodes).  The reference multiplies the two row lengths as `int` (module.cpp:91-93); the
 * product overflows (undefined behaviour) once it reaches 2^31, so a caller comparing against an
 * implementation with a 64-bit product must choose rows whose products all stay below 2^31 —
 * returns the number of selected rows that violate this (their output is still written). */
int or_graphsum_rows(const int *indptr, const int *indices, const int *rows, int n_sel,
                     const float *in, float *out, int dim) {
    int overflowing = 0;
    for (long i = 0; i < (long)n_sel * dim; i++) out[i] = 0;
    for (int k = 0; k < n_sel; k++) {
        const int src = rows[k];
        int bad = 0;
        for (int e = indptr[src]; e < indptr[src + 1]; e++) {
            const int dst = indices[e];
            const long wide = (long)(indptr[src + 1] - indptr[src]) * (long)(indptr[dst + 1] - indptr[dst]);
            if (wide >= 2147483648L) bad = 1;
            const float coef = gs_coef(indptr, src, dst);
            for (int j = 0; j < dim; j++)
                out[(long)k * dim + j] += coef * in[(long)dst * dim + j];
        }
        overflowing += bad;
    }
    return overflowing;
}

/* -------------------------------------------------------- CrossEntropyLoss */
/* src/seq/module.cpp:124-161 */
void or_xent_fwd(float *logits, float *grad, const int *truth,
                 int n_rows, int num_classes, int training, float *loss) {
    t_start(T_LOSS);
    float total_loss = 0;
    int count = 0;
    const long total = (long)n_rows * num_classes;
    if (training)
        for (long i = 0; i < total; i++) grad[i] = 0;
    for (int i = 0; i < n_rows; i++) {
        if (truth[i] < 0) continue;
        count++;
        float *logit = &logits[(long)i * num_classes];
        float max_logit = -1e30, sum_exp = 0;
        for (int j = 0; j < num_classes; j++)
            max_logit = fmaxf(max_logit, logit[j]);
        for (int j = 0; j < num_classes; j++) {
            logit[j] -= max_logit;
            sum_exp += expf(logit[j]);
        }
        total_loss += logf(sum_exp) - logit[truth[i]];
        if (training) {
            for (int j = 0; j < num_classes; j++) {
                float prob = expf(logit[j]) / sum_exp;
                grad[(long)i * num_classes + j] = prob;
            }
            /* "-= 1.0": double subtraction, narrowed on store */
            grad[(long)i * num_classes + truth[i]] =
                (float)((double)grad[(long)i * num_classes + truth[i]] - 1.0);
        }
    }
    *loss = total_loss / count;          /* count == 0 -> NaN, as the reference */
    if (training)
        for (long i = 0; i < total; i++) grad[i] /= count;
    t_stop(T_LOSS);
}

/* -------------------------------------------------------------------- ReLU */
/* src/seq/module.cpp:175-185 */
void or_relu_fwd(float *x, unsigned char *mask, int n, int training) {
    t_start(T_RELU_FW);
    for (int i = 0; i < n; i++) {
        int keep = x[i] > 0;
        if (training) mask[i] = (unsigned char)keep;
        if (!keep) x[i] = 0;
    }
    t_stop(T_RELU_FW);
}
/* src/seq/module.cpp:187-194 */
void or_relu_bwd(float *grad, const unsigned char *mask, int n) {
    t_start(T_RELU_BW);
    for (int i = 0; i < n; i++)
        if (!mask[i]) grad[i] = 0;
    t_stop(T_RELU_BW);
}

/* ----------------------------------------------------------------- Dropout */
/* src/seq/module.cpp:207-221 — one RNG draw per element even when p == 0 */
void or_dropout_fwd(float *x, int *mask, int n, float p, int training) {
    if (!training) return;
    t_start(T_DROP_FW);
    const int threshold = (int)(p * OR_RAND_MAX);   /* float * int -> float -> int */
    float scale = 1 / (1 - p);
    for (int i = 0; i < n; i++) {
        int keep = (int)or_rand_next() >= threshold;
        x[i] *= keep ? scale : 0;
        if (mask) mask[i] = keep;
    }
    t_stop(T_DROP_FW);
}
/* src/seq/module.cpp:223-233 */
void or_dropout_bwd(float *grad, const int *mask, int n, float p) {
    if (!mask) return;
    t_start(T_DROP_BW);
    float scale = 1 / (1 - p);
    for (int i = 0; i < n; i++)
        grad[i] *= mask[i] ? scale : 0;
    t_stop(T_DROP_BW);
}

/* -------------------------------------------------------------------- Adam */
or_adam_params or_adam_default(void) {              /* src/seq/optim.cpp:6-8 */
    or_adam_params a = {0.001, 0.9, 0.999, 1e-8, 0.0};
    return a;
}
/* src/seq/optim.cpp:24-37.  "(1.0 - beta)" is a double; the sums are formed
 * in double and narrowed when stored into m / v. */
void or_adam_step_var(float *w, const float *g, float *m, float *v, int n,
                      int decay, int step_count, const or_adam_params *ap) {
    float step_size = ap->lr * sqrtf(1 - powf(ap->beta2, step_count)) /
                      (1 - powf(ap->beta1, step_count));
    for (int i = 0; i < n; i++) {
        float grad = g[i];
        if (decay) grad += ap->weight_decay * w[i];
        m[i] = (float)((double)(ap->beta1 * m[i]) + (1.0 - (double)ap->beta1) * (double)grad);
        v[i] = (float)((double)(ap->beta2 * v[i]) +
                       (1.0 - (double)ap->beta2) * (double)grad * (double)grad);
        w[i] -= step_size * m[i] / (sqrtf(v[i]) + ap->eps);
    }
}

/* ------------------------------------------------------------------ Parser */
typedef struct { int *p; long n, cap; } ivec;
typedef struct { float *p; long n, cap; } fvec;
static void iv_push(ivec *v, int x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (int *)realloc(v->p, v->cap * sizeof(int)); }
    v->p[v->n++] = x;
}
static void fv_push(fvec *v, float x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (float *)realloc(v->p, v->cap * sizeof(float)); }
    v->p[v->n++] = x;
}

/* getline-then-eof test of parser.cpp:25-28: a final line with no '\n' is
 * dropped.  Returns 1 and the line (without '\n') if one complete line was
 * read. */
static int next_line(FILE *f, char **buf, size_t *cap) {
    ssize_t len = getline(buf, cap, f);
    if (len <= 0) return 0;
    if ((*buf)[len - 1] != '\n') return 0;       /* hit EOF inside the line */
    (*buf)[len - 1] = 0;
    return 1;
}

/* istream >> int: skip blanks, optional sign, digits; no digits => fail */
static int scan_int(const char **s, int *out) {
    const char *q = *s;
    while (*q && isspace((unsigned char)*q)) q++;
    char *end;
    long v = strtol(q, &end, 10);
    if (end == q) return 0;
    *out = (int)v;
    *s = end;
    return 1;
}

or_params or_params_default(void) {                 /* src/seq/gcn.cpp:9-11 */
    or_params p = {2708, 1433, 16, 7, 0.5, 0.01, 5e-4, 100, 0};
    return p;
}

int or_parse(const char *root, const char *name, or_params *p, or_data *d) {
    char path[3][4096];
    static const char *ext[3] = {".graph", ".split", ".svmlight"};
    FILE *f[3];
    memset(d, 0, sizeof *d);
    for (int i = 0; i < 3; i++) {
        snprintf(path[i], sizeof path[i], "%s%s%s", root, name, ext[i]);
        f[i] = fopen(path[i], "r");
    }
    if (!f[0] || !f[1] || !f[2]) {                  /* parser.cpp:48-50,111 */
        for (int i = 0; i < 3; i++) if (f[i]) fclose(f[i]);
        return -1;
    }
    char *line = NULL; size_t cap = 0;

    /* parseGraph, parser.cpp:20-46: self loop first, then neighbours */
    ivec gi = {0}, gp = {0};
    iv_push(&gp, 0);
    int node = 0;
    while (next_line(f[0], &line, &cap)) {
        iv_push(&gi, node);
        iv_push(&gp, gp.p[gp.n - 1] + 1);
        node++;
        const char *s = line; int nb;
        while (scan_int(&s, &nb)) { iv_push(&gi, nb); gp.p[gp.n - 1] += 1; }
    }
    p->num_nodes = node;
    printf("Parse Graph Succeeded.\n");

    /* parseNode, parser.cpp:52-92 */
    ivec fi = {0}, fp = {0}, lab = {0}; fvec fv = {0};
    iv_push(&fp, 0);
    int max_idx = 0, max_label = 0;
    while (next_line(f[2], &line, &cap)) {
        iv_push(&fp, fp.p[fp.n - 1]);
        const char *s = line;
        int label = -1;
        /* blank line: the stream's sentry fails and label keeps -1; a
         * non-numeric token: C++11 num_get stores 0 and sets failbit */
        const char *q = s; while (*q && isspace((unsigned char)*q)) q++;
        int ok = 0;
        if (*q) { ok = scan_int(&s, &label); if (!ok) label = 0; }
        iv_push(&lab, label);
        if (!ok) continue;
        if (label > max_label) max_label = label;
        for (;;) {
            while (*s && isspace((unsigned char)*s)) s++;
            if (!*s) break;
            /* token "k:v" (parser.cpp:73-82) */
            char *end;
            long k = strtol(s, &end, 10);
            const char *t = end;
            float v = 0;
            if (*t && !isspace((unsigned char)*t)) {
                t++;                                /* the ':' */
                v = strtof(t, &end);
                t = end;
            }
            while (*t && !isspace((unsigned char)*t)) t++;   /* rest of the token */
            s = t;
            fv_push(&fv, v);
            iv_push(&fi, (int)k);
            fp.p[fp.n - 1] += 1;
            if ((int)k > max_idx) max_idx = (int)k;
        }
    }
    p->input_dim = max_idx + 1;
    p->output_dim = max_label + 1;
    printf("Parse Node Succeeded.\n");

    /* parseSplit, parser.cpp:94-103 */
    ivec sp = {0};
    while (next_line(f[1], &line, &cap)) iv_push(&sp, (int)strtol(line, NULL, 10));
    printf("Parse Split Succeeded.\n");

    free(line);
    for (int i = 0; i < 3; i++) fclose(f[i]);
    d->g_indptr = gp.p; d->g_indices = gi.p; d->g_nnz = (int)gi.n;
    d->f_indptr = fp.p; d->f_indices = fi.p; d->f_val = fv.p; d->f_nnz = (int)fi.n;
    d->split = sp.p; d->n_split = (int)sp.n;
    d->label = lab.p; d->n_label = (int)lab.n;
    return 0;
}

void or_data_free(or_data *d) {
    free(d->g_indptr); free(d->g_indices); free(d->f_indptr); free(d->f_indices);
    free(d->f_val); free(d->split); free(d->label);
    memset(d, 0, sizeof *d);
}

/* --------------------------------------------------------------- GCN model */
struct or_gcn {
    or_params params;
    const or_data *data;
    /* variables in construction order, gcn.cpp:21-54 */
    float *vdata[7], *vgrad[7];
    int vsize[7];
    unsigned char *relu_mask;
    int *drop_mask;                 /* hidden dropout only; input has no grad */
    int *truth;
    float loss;
    /* Adam, gcn.cpp:62-65 */
    or_adam_params adam;
    int step_count;
    float *m[2], *v[2];
};

static float *zalloc_f(long n) { return (float *)calloc(n > 0 ? n : 1, sizeof(float)); }

or_gcn *or_gcn_create(or_params p, const or_data *d, long seed_time) {
    or_gcn *g = (or_gcn *)calloc(1, sizeof *g);
    if (seed_time >= 0) or_rand_seed_time((unsigned)seed_time);   /* gcn.cpp:14 */
    g->params = p;
    g->data = d;
    const long N = p.num_nodes, F = p.input_dim, H = p.hidden_dim, C = p.output_dim;
    const long sizes[7] = {d->f_nnz, N * H, F * H, N * H, N * C, H * C, N * C};
    for (int k = 0; k < 7; k++) {
        g->vsize[k] = (int)sizes[k];
        g->vdata[k] = zalloc_f(sizes[k]);
        g->vgrad[k] = (k == 0) ? NULL : zalloc_f(sizes[k]);      /* gcn.cpp:21: input has no grad */
    }
    /* RNG draw order: all of W1 (gcn.cpp:30), then all of W2 (gcn.cpp:49) */
    or_glorot(g->vdata[2], g->vsize[2], p.input_dim, p.hidden_dim);
    or_glorot(g->vdata[5], g->vsize[5], p.hidden_dim, p.output_dim);
    g->relu_mask = (unsigned char *)calloc(N * H > 0 ? N * H : 1, 1);
    g->drop_mask = (int *)calloc(N * H > 0 ? N * H : 1, sizeof(int));
    g->truth = (int *)calloc(N > 0 ? N : 1, sizeof(int));
    g->adam = or_adam_default();
    g->adam.lr = p.learning_rate;
    g->adam.weight_decay = p.weight_decay;
    g->step_count = 0;
    g->m[0] = zalloc_f(sizes[2]); g->v[0] = zalloc_f(sizes[2]);
    g->m[1] = zalloc_f(sizes[5]); g->v[1] = zalloc_f(sizes[5]);
    return g;
}

void or_gcn_destroy(or_gcn *g) {
    if (!g) return;
    for (int k = 0; k < 7; k++) { free(g->vdata[k]); free(g->vgrad[k]); }
    free(g->relu_mask); free(g->drop_mask); free(g->truth);
    for (int k = 0; k < 2; k++) { free(g->m[k]); free(g->v[k]); }
    free(g);
}

float *or_gcn_var_data(or_gcn *g, int k, int *size) { if (size) *size = g->vsize[k]; return g->vdata[k]; }
float *or_gcn_var_grad(or_gcn *g, int k, int *size) { if (size) *size = g->vgrad[k] ? g->vsize[k] : 0; return g->vgrad[k]; }

static void set_input(or_gcn *g) {                  /* gcn.cpp:73-76 */
    memcpy(g->vdata[0], g->data->f_val, (size_t)g->vsize[0] * sizeof(float));
}
static void set_truth(or_gcn *g, int split) {       /* gcn.cpp:78-81 */
    for (int i = 0; i < g->params.num_nodes; i++)
        g->truth[i] = g->data->split[i] == split ? g->data->label[i] : -1;
}
static float get_accuracy(or_gcn *g) {              /* gcn.cpp:83-96 */
    int wrong = 0, total = 0;
    const int C = g->params.output_dim;
    const float *out = g->vdata[6];
    for (int i = 0; i < g->params.num_nodes; i++) {
        if (g->truth[i] < 0) continue;
        total++;
        float truth_logit = out[(long)i * C + g->truth[i]];
        for (int j = 0; j < C; j++)
            if (out[(long)i * C + j] > truth_logit) { wrong++; break; }
    }
    return (float)(total - wrong) / total;
}
static float get_l2_penalty(or_gcn *g) {            /* gcn.cpp:98-105: W1 only */
    float l2 = 0;
    for (int i = 0; i < g->vsize[2]; i++) {
        float x = g->vdata[2][i];
        l2 += x * x;
    }
    return g->params.weight_decay * l2 / 2;
}

/* the eight modules in list order, gcn.cpp:23-59 */
static void forward_all(or_gcn *g, int training) {
    const or_params *p = &g->params;
    const or_data *d = g->data;
    const int N = p->num_nodes;
    or_dropout_fwd(g->vdata[0], NULL, g->vsize[0], p->dropout, training);
    or_spmm_fwd(d->f_indptr, d->f_indices, N, g->vdata[0], g->vdata[2], g->vdata[1], p->hidden_dim);
    t_start(T_GS_FW);
    or_graphsum(d->g_indptr, d->g_indices, N, g->vdata[1], g->vdata[3], p->hidden_dim);
    t_stop(T_GS_FW);
    or_relu_fwd(g->vdata[3], g->relu_mask, g->vsize[3], training);
    or_dropout_fwd(g->vdata[3], g->drop_mask, g->vsize[3], p->dropout, training);
    or_matmul_fwd(g->vdata[3], g->vdata[5], g->vdata[4], N, p->hidden_dim, p->output_dim);
    t_start(T_GS_FW);
    or_graphsum(d->g_indptr, d->g_indices, N, g->vdata[4], g->vdata[6], p->output_dim);
    t_stop(T_GS_FW);
    or_xent_fwd(g->vdata[6], g->vgrad[6], g->truth, N, p->output_dim, training, &g->loss);
}

static void backward_all(or_gcn *g) {               /* gcn.cpp:114-115, reverse order */
    const or_params *p = &g->params;
    const or_data *d = g->data;
    const int N = p->num_nodes;
    t_start(T_GS_BW);
    or_graphsum(d->g_indptr, d->g_indices, N, g->vgrad[6], g->vgrad[4], p->output_dim);
    t_stop(T_GS_BW);
    or_matmul_bwd(g->vdata[3], g->vdata[5], g->vgrad[4], g->vgrad[3], g->vgrad[5],
                  N, p->hidden_dim, p->output_dim);
    or_dropout_bwd(g->vgrad[3], g->drop_mask, g->vsize[3], p->dropout);
    or_relu_bwd(g->vgrad[3], g->relu_mask, g->vsize[3]);
    t_start(T_GS_BW);
    or_graphsum(d->g_indptr, d->g_indices, N, g->vgrad[3], g->vgrad[1], p->hidden_dim);
    t_stop(T_GS_BW);
    or_spmm_bwd(d->f_indptr, d->f_indices, N, g->vdata[0], g->vgrad[1], g->vgrad[2],
                p->input_dim, p->hidden_dim);
}

void or_gcn_train_epoch(or_gcn *g, float *loss, float *acc) {   /* gcn.cpp:107-118 */
    set_input(g);
    set_truth(g, 1);
    forward_all(g, 1);
    float train_loss = g->loss + get_l2_penalty(g);
    float train_acc = get_accuracy(g);
    backward_all(g);
    g->step_count++;                                /* optim.cpp:25 */
    or_adam_step_var(g->vdata[2], g->vgrad[2], g->m[0], g->v[0], g->vsize[2], 1, g->step_count, &g->adam);
    or_adam_step_var(g->vdata[5], g->vgrad[5], g->m[1], g->v[1], g->vsize[5], 0, g->step_count, &g->adam);
    *loss = train_loss; *acc = train_acc;
}

void or_gcn_eval(or_gcn *g, int split, float *loss, float *acc) {   /* gcn.cpp:120-128 */
    set_input(g);
    set_truth(g, split);
    forward_all(g, 0);
    *loss = g->loss + get_l2_penalty(g);
    *acc = get_accuracy(g);
}

int or_gcn_run(or_gcn *g, float *trace, int quiet) {            /* gcn.cpp:130-158 */
    const or_params *p = &g->params;
    float *hist = (float *)malloc(sizeof(float) * (p->epochs > 0 ? p->epochs : 1));
    int epoch = 1, ran = 0;
    for (; epoch <= p->epochs; epoch++) {
        float tl, ta, vl, va;
        t_start(T_TRAIN);
        or_gcn_train_epoch(g, &tl, &ta);
        or_gcn_eval(g, 2, &vl, &va);
        double dt = t_stop(T_TRAIN);
        if (!quiet)
            printf("epoch=%d train_loss=%.5f train_acc=%.5f val_loss=%.5f val_acc=%.5f time=%.5f\n",
                   epoch, tl, ta, vl, va, dt);
        if (trace) { trace[4 * ran + 0] = tl; trace[4 * ran + 1] = ta; trace[4 * ran + 2] = vl; trace[4 * ran + 3] = va; }
        hist[ran++] = vl;
        if (p->early_stopping > 0 && epoch >= p->early_stopping) {
            float recent = 0.0;
            for (int i = epoch - p->early_stopping; i < epoch; i++) recent += hist[i];
            if (vl > recent / p->early_stopping) {
                if (!quiet) printf("Early stopping...\n");
                break;
            }
        }
    }
    if (!quiet) printf("total training time=%.5f\n", t_sum[T_TRAIN]);
    float sl, sa;
    t_start(T_TEST);
    or_gcn_eval(g, 3, &sl, &sa);
    double dt = t_stop(T_TEST);
    if (!quiet) printf("test_loss=%.5f test_acc=%.5f time=%.5f\n", sl, sa, dt);
    if (trace) { trace[4 * ran] = sl; trace[4 * ran + 1] = sa; }
    free(hist);
    return ran;
}
