// capi.cpp — extern "C" surface of the host library (include/gcnhost.h).
#include "gcnhost.h"
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "gcn.h"
#include "cluster.h"
#include "hip_check.h"
#include "parser.h"

static thread_local std::string g_err;

struct gcnhost_model {
    GCNData data;
    HipGCN *gcn = nullptr;
};
struct gcnhost_dataset {
    GCNData data;
};

#define API_TRY(...)                                    \
    try {                                               \
        __VA_ARGS__;                                    \
        return 0;                                       \
    } catch (const GcnHipFailure &e) {                  \
        g_err = e.what();                               \
        return e.code ? e.code : -1;                    \
    } catch (const std::exception &e) {                 \
        g_err = e.what();                               \
        return -1;                                      \
    }

extern "C" {

const char *gcnhost_last_error(void) { return g_err.c_str(); }

gcnhost_params gcnhost_params_default(void) {
    GCNParams d = GCNParams::get_default();
    gcnhost_params p;
    memcpy(&p, &d, sizeof p);
    return p;
}

int gcnhost_nccl_unique_id(char id[GCNHOST_NCCL_ID_BYTES]) { return rccl_get_unique_id(id); }

int gcnhost_model_create(gcnhost_model **out, const gcnhost_params *p,
                         const int *g_indptr, const int *g_indices,
                         const int *f_indptr, const int *f_indices, const float *f_val,
                         const int *split, const int *label,
                         long seed, int device, int flags, int rank, int world, const char *nccl_id,
                         gcnhost_allgather_fn host_ag, gcnhost_allreduce_fn host_ar, void *host_user) {
    if (!out || !p || !g_indptr || !g_indices || !f_indptr || !f_val || !split || !label) { g_err = "null argument"; return -1; }
    API_TRY({
        gcnhost_model *m = new gcnhost_model();
        const int N = p->num_nodes;
        m->data.graph.indptr.assign(g_indptr, g_indptr + N + 1);
        m->data.graph.indices.assign(g_indices, g_indices + g_indptr[N]);
        m->data.feature_index.indptr.assign(f_indptr, f_indptr + N + 1);
        if (f_indices) m->data.feature_index.indices.assign(f_indices, f_indices + f_indptr[N]);
        m->data.feature_value.assign(f_val, f_val + f_indptr[N]);
        m->data.split.assign(split, split + N);
        m->data.label.assign(label, label + N);
        GCNParams gp;
        static_assert(sizeof(GCNParams) == sizeof(gcnhost_params), "params layout");
        memcpy(&gp, p, sizeof gp);
        HipGCNOptions o;
        o.device = device; o.seed = seed; o.flags = flags; o.rank = rank; o.world = world; o.nccl_id = nccl_id;
        o.host_allgather = host_ag; o.host_allreduce = host_ar; o.host_user = host_user;
        o = HipGCNOptions::from_environment(o);                 // every HIPGCN_* variable, read once (host/options.cpp)
        try {
            m->gcn = new HipGCN(gp, &m->data, o);
        } catch (...) {
            delete m;
            throw;
        }
        *out = m;
    })
}

int gcnhost_model_destroy(gcnhost_model *m) {
    if (!m) return 0;
    API_TRY({ delete m->gcn; delete m; })
}
int gcnhost_model_train_epoch(gcnhost_model *m, float *loss, float *acc) {
    API_TRY({ auto r = m->gcn->train_epoch(); *loss = r.first; *acc = r.second; })
}
int gcnhost_model_eval(gcnhost_model *m, int split, float *loss, float *acc) {
    API_TRY({ auto r = m->gcn->eval(split); *loss = r.first; *acc = r.second; })
}
int gcnhost_model_run_epochs(gcnhost_model *m, int n, float *trace) { API_TRY({ m->gcn->run_epochs(n, trace); }) }
int gcnhost_model_run(gcnhost_model *m) { API_TRY({ m->gcn->run(); }) }
int gcnhost_model_sync(gcnhost_model *m) { API_TRY({ m->gcn->sync(); }) }

int gcnhost_model_exchange(gcnhost_model *m, int *halo, int64_t *recv_rows, int64_t *send_rows, int *table_rows, double *halo_share) {
    API_TRY({
        const ExchangePlan &x = m->gcn->exchange_plan();
        if (halo) *halo = x.halo ? 1 : 0;
        if (recv_rows) *recv_rows = x.world > 1 ? x.recv_total() : 0;
        if (send_rows) *send_rows = x.world > 1 ? x.send_total() : 0;
        if (table_rows) *table_rows = x.table_rows;
        if (halo_share) *halo_share = x.halo_share;
    })
}
int gcnhost_model_info(gcnhost_model *m, int *rank, int *world, int *row_start, int *local_rows, int64_t *local_edges) {
    API_TRY({
        if (rank) *rank = m->gcn->rank();
        if (world) *world = m->gcn->world();
        if (row_start) *row_start = m->gcn->row_start();
        if (local_rows) *local_rows = m->gcn->local_rows();
        if (local_edges) *local_edges = m->gcn->n_edges_local();
    })
}
int gcnhost_model_row_ids(gcnhost_model *m, int *ids, int *renumbered) {
    API_TRY({
        const std::vector<int> &order = m->gcn->node_order();
        const int r0 = m->gcn->row_start(), n = m->gcn->local_rows();
        if (renumbered) *renumbered = order.empty() ? 0 : 1;
        if (ids)
            for (int r = 0; r < n; r++) ids[r] = order.empty() ? r0 + r : order[(size_t)r0 + r];
    })
}
int gcnhost_model_row_scale(gcnhost_model *m, float *dinv, int *factored) {
    API_TRY({
        if (factored) *factored = m->gcn->factored() ? 1 : 0;
        if (dinv) {
            std::vector<float> v;
            m->gcn->row_scale(v);
            if (!v.empty()) memcpy(dinv, v.data(), v.size() * sizeof(float));
        }
    })
}
int gcnhost_model_schedule(gcnhost_model *m, int *mode, int *n_groups) {
    API_TRY({
        if (mode) *mode = m->gcn->schedule_mode();
        if (n_groups) *n_groups = m->gcn->schedule_groups();
    })
}
int gcnhost_model_transport(gcnhost_model *m, int *ranks, char name[32]) {
    API_TRY({
        if (ranks) *ranks = m->gcn->transport_ranks();
        if (name) { strncpy(name, m->gcn->transport(), 31); name[31] = 0; }
    })
}
int gcnhost_model_slice_floats(gcnhost_model *m, int *floats) {
    API_TRY({ if (floats) *floats = m->gcn->schedule_slice_floats(); })
}
int gcnhost_model_get_var(gcnhost_model *m, int k, int grad, float *out, int *rows, int *cols) {
    API_TRY({
        std::vector<float> v;
        int r, c;
        m->gcn->get_var(k, grad != 0, v, &r, &c);
        if (rows) *rows = r;
        if (cols) *cols = c;
        if (out) memcpy(out, v.data(), v.size() * sizeof(float));
    })
}
int gcnhost_model_set_weights(gcnhost_model *m, const float *w1, const float *w2) { API_TRY({ m->gcn->set_weights(w1, w2); }) }
int gcnhost_model_timer(gcnhost_model *m, int id, double *seconds, long *count) {
    if (id < 0 || id >= __NUM_TMR) { g_err = "bad timer id"; return -1; }
    API_TRY({ *seconds = m->gcn->timer_total((timer_instance)id, count); })
}
int gcnhost_model_timers_reset(gcnhost_model *m) { API_TRY({ m->gcn->timers_reset(); }) }
int gcnhost_model_set_timers(gcnhost_model *m, int on) { API_TRY({ m->gcn->set_timers(on != 0); }) }

int gcnhost_dataset_load(gcnhost_dataset **out, const char *root, const char *name, gcnhost_params *p) {
    API_TRY({
        gcnhost_dataset *d = new gcnhost_dataset();
        GCNParams gp;
        memcpy(&gp, p, sizeof gp);
        Parser parser(&gp, &d->data, name, root ? root : "");
        if (!parser.parse()) { delete d; throw std::runtime_error(std::string("Cannot read input: ") + name); }
        memcpy(p, &gp, sizeof gp);
        *out = d;
    })
}
int gcnhost_dataset_arrays(gcnhost_dataset *d, const int **g_indptr, const int **g_indices, int64_t *g_nnz,
                           const int **f_indptr, const int **f_indices, const float **f_val, int64_t *f_nnz,
                           const int **split, int64_t *n_split, const int **label, int64_t *n_label) {
    if (!d) return -1;
    *g_indptr = d->data.graph.indptr.data(); *g_indices = d->data.graph.indices.data(); *g_nnz = (int64_t)d->data.graph.indices.size();
    *f_indptr = d->data.feature_index.indptr.data(); *f_indices = d->data.feature_index.indices.data();
    *f_val = d->data.feature_value.data(); *f_nnz = (int64_t)d->data.feature_value.size();
    *split = d->data.split.data(); *n_split = (int64_t)d->data.split.size();
    *label = d->data.label.data(); *n_label = (int64_t)d->data.label.size();
    return 0;
}
int gcnhost_dataset_save_binary(gcnhost_dataset *d, const gcnhost_params *p, const char *path) {
    GCNParams gp;
    memcpy(&gp, p, sizeof gp);
    return Parser::save_binary(path, gp, d->data) ? 0 : -1;
}
int gcnhost_dataset_free(gcnhost_dataset *d) { delete d; return 0; }

// The HALO exchange (RCCL: grouped ncclSend/ncclRecv straight into the table segments) on a small synthetic graph:
// node i points at i+1, i+3, i-1 and i + N/2 (mod N), so every rank needs rows of its neighbour ranks and of the rank
// half-way round; every table row must come back holding its global id.  Any transport.
static void halo_round_trip(gcnhip_ctx *ctx, Comm *comm, int rank, int world) {
    const int N = world * 96, ld = 8;
    std::vector<int> gp(N + 1), gi;
    for (int i = 0; i < N; i++) {
        gp[i] = (int)gi.size();
        gi.push_back(i);
        for (int off : {1, 3, N / 2, N - 1}) gi.push_back((i + off) % N);
    }
    gp[N] = (int)gi.size();
    const RowPartition part = make_partition(gp.data(), N, world);
    const ExchangePlan plan = make_exchange_plan(gp.data(), gi.data(), N, part, rank, /*HALO*/ 2);
    ExchangeBuffers xb;
    exchange_buffers_create(ctx, plan, ld, &xb);
    std::vector<float> tab((size_t)plan.table_rows * ld, -1.f);
    for (int r = 0; r < plan.n_local; r++)
        for (int k = 0; k < ld; k++) tab[(size_t)r * ld + k] = (float)(part.start[rank] + r) + 0.125f * k;
    void *dt = nullptr;
    try {
        GCNHIP_CHECK(gcnhip_malloc(ctx, &dt, tab.size() * sizeof(float)));
        GCNHIP_CHECK(gcnhip_h2d(ctx, dt, tab.data(), tab.size() * sizeof(float)));
        for (int it = 0; it < 3; it++) comm->exchange_rows(plan, xb, (float *)dt, ld);
        GCNHIP_CHECK(gcnhip_d2h(ctx, tab.data(), dt, tab.size() * sizeof(float)));
    } catch (...) {
        gcnhip_free(ctx, dt);
        exchange_buffers_destroy(&xb);
        throw;
    }
    gcnhip_free(ctx, dt);
    exchange_buffers_destroy(&xb);
    for (int t = 0; t < plan.table_rows; t++)
        for (int k = 0; k < ld; k++)
            if (tab[(size_t)t * ld + k] != (float)plan.table_global[t] + 0.125f * k)
                throw GcnHipFailure(-1, "halo exchange self-test: a table row came back with the wrong content");
}

// RCCL round trip with `world` ranks (one per process; world == 1: a single-rank communicator): communicator
// init from the shared unique id, in-place all-gather of distinct blocks, all-reduce, the halo exchange (grouped
// ncclSend/ncclRecv through an ExchangePlan's send lists, world > 1), the validation lane's split communicator on a
// second stream, and collectives alternating between the two (turnstile order).
int gcnhost_rccl_selftest_world(int device, int rank, int world, const char *nccl_id) {
    if (world < 1 || rank < 0 || rank >= world || !nccl_id) { g_err = "bad rank/world/id"; return -1; }
    API_TRY({
        gcnhip_ctx *ctx = nullptr;
        GCNHIP_CHECK(gcnhip_ctx_create(&ctx, device, nullptr));
        {
            std::unique_ptr<Comm> comm(make_rccl_comm(ctx, rank, world, nccl_id));
            const size_t B = 1024;
            std::vector<float> h(B * world, -1.f);
            for (size_t i = 0; i < B; i++) h[rank * B + i] = (float)(rank * 1000) + (float)i * 0.5f;
            void *d;
            GCNHIP_CHECK(gcnhip_malloc(ctx, &d, h.size() * sizeof(float)));
            GCNHIP_CHECK(gcnhip_h2d(ctx, d, h.data(), h.size() * sizeof(float)));
            comm->allgather_rows((float *)d, B);
            std::vector<float> back(h.size());
            GCNHIP_CHECK(gcnhip_d2h(ctx, back.data(), d, back.size() * sizeof(float)));
            for (int q = 0; q < world; q++)
                for (size_t i = 0; i < B; i++)
                    if (back[q * B + i] != (float)(q * 1000) + (float)i * 0.5f)
                        throw GcnHipFailure(-1, "RCCL self-test: all-gather block mismatch");
            comm->allreduce_sum((float *)d, B);        // block 0 of every rank is now identical: sum = world * value
            GCNHIP_CHECK(gcnhip_d2h(ctx, back.data(), d, B * sizeof(float)));
            for (size_t i = 0; i < B; i++)
                if (back[i] != (float)world * ((float)i * 0.5f)) throw GcnHipFailure(-1, "RCCL self-test: all-reduce mismatch");
            gcnhip_free(ctx, d);
            if (world > 1) halo_round_trip(ctx, comm.get(), rank, world);
            void *sa;
            GCNHIP_CHECK(gcnhip_malloc(ctx, &sa, 256 * world * sizeof(float)));
            GCNHIP_CHECK(gcnhip_memset_async(ctx, sa, 0, 256 * world * sizeof(float)));
            float *scratch_a = (float *)sa;
            double v[2] = {3.0, 4.0 + rank};
            comm->allreduce_sum_host(v, 2);
            if (v[0] != 3.0 * world || v[1] != 4.0 * world + world * (world - 1) / 2.0)
                throw GcnHipFailure(-1, "RCCL self-test: host reduction");
            // the validation lane's communicator (ncclCommSplit) on a second stream
            gcnhip_ctx *ctx2 = nullptr;
            GCNHIP_CHECK(gcnhip_ctx_create(&ctx2, device, nullptr));
            {
                std::unique_ptr<Comm> comm2(comm->clone_for(ctx2));
                double w2[1] = {5.0};
                comm2->allreduce_sum_host(w2, 1);
                if (w2[0] != 5.0 * world) throw GcnHipFailure(-1, "RCCL self-test: split communicator");
                // collectives alternating between the two lanes (serialised by the turnstile event)
                void *d2;
                GCNHIP_CHECK(gcnhip_malloc(ctx2, &d2, 256 * world * sizeof(float)));
                GCNHIP_CHECK(gcnhip_memset_async(ctx2, d2, 0, 256 * world * sizeof(float)));
                for (int it = 0; it < 4; it++) {
                    comm->allreduce_sum(scratch_a, 256);
                    comm2->allgather_rows((float *)d2, 256);
                    comm2->allreduce_sum((float *)d2, 256);
                    comm->allgather_rows(scratch_a, 256);
                }
                GCNHIP_CHECK(gcnhip_ctx_sync(ctx));
                GCNHIP_CHECK(gcnhip_ctx_sync(ctx2));
                gcnhip_free(ctx2, d2);
            }
            gcnhip_ctx_destroy(ctx2);
            gcnhip_free(ctx, sa);
        }
        gcnhip_ctx_destroy(ctx);
    })
}

// the same halo round trip through the host-staged transport (tests: ranks as threads or gloo processes sharing a GPU)
int gcnhost_halo_selftest_host(int device, int rank, int world, gcnhost_allgather_fn ag, gcnhost_allreduce_fn ar, void *user) {
    if (world < 2 || rank < 0 || rank >= world || !ag || !ar) { g_err = "bad rank/world/callbacks"; return -1; }
    API_TRY({
        gcnhip_ctx *ctx = nullptr;
        GCNHIP_CHECK(gcnhip_ctx_create(&ctx, device, nullptr));
        try {
            std::unique_ptr<Comm> comm(make_host_comm(ctx, rank, world, ag, ar, user));
            halo_round_trip(ctx, comm.get(), rank, world);
        } catch (...) {
            gcnhip_ctx_destroy(ctx);
            throw;
        }
        gcnhip_ctx_destroy(ctx);
    })
}

// Device time per collective on the stream, HIP events around `iters` back-to-back calls (after two warm-up calls): the
// in-place all-gather of `block_floats` floats per rank and the all-reduce of `reduce_floats` floats, as the epoch issues
// them.  With one rank (what a single-GPU box can run) this is the launch + kernel floor of a collective — a LOWER bound on
// what a peer adds; tools/comm_model.py takes it instead of an assumed latency.
int gcnhost_rccl_collective_us(int device, int rank, int world, const char *nccl_id, long block_floats, long reduce_floats, int iters,
                               double *us_allgather, double *us_allreduce) {
    if (world < 1 || rank < 0 || rank >= world || !nccl_id || block_floats < 1 || reduce_floats < 1 || iters < 1) { g_err = "bad arguments"; return -1; }
    API_TRY({
        gcnhip_ctx *ctx = nullptr;
        GCNHIP_CHECK(gcnhip_ctx_create(&ctx, device, nullptr));
        {
            std::unique_ptr<Comm> comm(make_rccl_comm(ctx, rank, world, nccl_id));
            void *d = nullptr, *e0 = nullptr, *e1 = nullptr;
            const size_t n = std::max((size_t)block_floats * world, (size_t)reduce_floats);
            GCNHIP_CHECK(gcnhip_malloc(ctx, &d, n * sizeof(float)));
            GCNHIP_CHECK(gcnhip_memset_async(ctx, d, 0, n * sizeof(float)));
            GCNHIP_CHECK(gcnhip_event_create(&e0));
            GCNHIP_CHECK(gcnhip_event_create(&e1));
            float ms = 0.f;
            for (int which = 0; which < 2; which++) {
                for (int i = 0; i < 2 + iters; i++) {
                    if (i == 2) GCNHIP_CHECK(gcnhip_event_record(ctx, e0));
                    if (which == 0) comm->allgather_rows((float *)d, (size_t)block_floats);
                    else comm->allreduce_sum((float *)d, (size_t)reduce_floats);
                }
                GCNHIP_CHECK(gcnhip_event_record(ctx, e1));
                GCNHIP_CHECK(gcnhip_event_elapsed_ms(e0, e1, &ms));
                if (which == 0 && us_allgather) *us_allgather = 1e3 * ms / iters;
                if (which == 1 && us_allreduce) *us_allreduce = 1e3 * ms / iters;
            }
            gcnhip_event_destroy(e0);
            gcnhip_event_destroy(e1);
            gcnhip_free(ctx, d);
        }
        gcnhip_ctx_destroy(ctx);
    })
}

int gcnhost_rccl_selftest(int device) {
    char id[GCN_NCCL_ID_BYTES];
    const int rc = rccl_get_unique_id(id);
    if (rc) { g_err = "ncclGetUniqueId failed"; return rc; }
    return gcnhost_rccl_selftest_world(device, 0, 1, id);
}

int gcnhost_partition(const int *g_indptr, int n_rows, int world, int *start, int *rows_max) {
    if (!g_indptr || !start || world < 1) return -1;
    RowPartition p = make_partition(g_indptr, n_rows, world);
    for (int q = 0; q <= world; q++) start[q] = p.start[q];
    if (rows_max) *rows_max = p.rows_max;
    return 0;
}
int gcnhost_local_graph(const int *g_indptr, const int *g_indices, int n_rows, int world, int rank,
                        int *indptr, int *indices, int *col_deg, int *n_local, int *n_cols, int64_t *nnz_local) {
    if (!g_indptr || !g_indices || world < 1 || rank < 0 || rank >= world) return -1;
    const RowPartition part = make_partition(g_indptr, n_rows, world);
    const LocalGraph lg = build_local_graph(g_indptr, g_indices, n_rows, part, rank);
    if (n_local) *n_local = lg.n_rows;
    if (n_cols) *n_cols = lg.n_cols;
    if (nnz_local) *nnz_local = (int64_t)lg.indices.size();
    if (indptr) memcpy(indptr, lg.indptr.data(), lg.indptr.size() * sizeof(int));
    if (indices && !lg.indices.empty()) memcpy(indices, lg.indices.data(), lg.indices.size() * sizeof(int));
    if (col_deg) memcpy(col_deg, lg.col_deg.data(), lg.col_deg.size() * sizeof(int));
    return 0;
}
struct gcnhost_plan {
    RowPartition part;
    ExchangePlan plan;
    LocalGraph graph;
};
int gcnhost_plan_create(gcnhost_plan **out, const int *g_indptr, const int *g_indices, int n_rows, int world, int rank, int mode) {
    if (!out || !g_indptr || !g_indices || world < 1 || rank < 0 || rank >= world || mode < 0 || mode > 2) return -1;
    API_TRY({
        gcnhost_plan *p = new gcnhost_plan();
        p->part = make_partition(g_indptr, n_rows, world);
        p->plan = make_exchange_plan(g_indptr, g_indices, n_rows, p->part, rank, mode);
        p->graph = build_table_graph(g_indptr, g_indices, n_rows, p->part, p->plan);
        *out = p;
    })
}
int gcnhost_plan_info(const gcnhost_plan *p, int *halo, int *n_local, int *table_rows, int *own_offset, int *rows_max,
                      double *halo_share, int64_t *nnz_local, int64_t *n_recv, int64_t *n_send) {
    if (!p) return -1;
    if (halo) *halo = p->plan.halo ? 1 : 0;
    if (n_local) *n_local = p->plan.n_local;
    if (table_rows) *table_rows = p->plan.table_rows;
    if (own_offset) *own_offset = p->plan.own_offset;
    if (rows_max) *rows_max = p->plan.rows_max;
    if (halo_share) *halo_share = p->plan.halo_share;
    if (nnz_local) *nnz_local = (int64_t)p->graph.indices.size();
    if (n_recv) *n_recv = (int64_t)p->plan.recv_rows.size();
    if (n_send) *n_send = (int64_t)p->plan.send_rows.size();
    return 0;
}
int gcnhost_plan_arrays(const gcnhost_plan *p, const int **recv_off, const int **recv_rows, const int **send_off, const int **send_rows,
                        const int **table_global, const int **indptr, const int **indices, const int **col_deg) {
    if (!p) return -1;
    if (recv_off) *recv_off = p->plan.recv_off.data();
    if (recv_rows) *recv_rows = p->plan.recv_rows.data();
    if (send_off) *send_off = p->plan.send_off.data();
    if (send_rows) *send_rows = p->plan.send_rows.data();
    if (table_global) *table_global = p->plan.table_global.data();
    if (indptr) *indptr = p->graph.indptr.data();
    if (indices) *indices = p->graph.indices.data();
    if (col_deg) *col_deg = p->graph.col_deg.data();
    return 0;
}
int gcnhost_plan_free(gcnhost_plan *p) { delete p; return 0; }

int gcnhost_choose_node_order(const int *g_indptr, const int *g_indices, int n_rows, int world, int force, int *order, int *renumbered,
                              double *ids_share, int64_t *ids_recv_rows, double *new_share, int64_t *new_recv_rows, int64_t *allgather_rows) {
    if (!g_indptr || !g_indices || n_rows < 0 || world < 1) return -1;
    API_TRY({
        StructureGroups sg;
        const OrderCost ids = exchange_cost(g_indptr, g_indices, n_rows, world);
        if ((force || ids.halo_share > 0.75) && n_rows >= 4096) sg = structure_groups(g_indptr, g_indices, n_rows);
        const NodeOrderChoice ch = choose_node_order(g_indptr, g_indices, n_rows, world, sg.useful ? sg.group.data() : nullptr, force != 0);
        if (renumbered) *renumbered = ch.order.empty() ? 0 : 1;
        if (order)
            for (int k = 0; k < n_rows; k++) order[k] = ch.order.empty() ? k : ch.order[k];
        if (ids_share) *ids_share = ch.ids.halo_share;
        if (ids_recv_rows) *ids_recv_rows = ch.ids.recv_rows_max;
        if (new_share) *new_share = ch.chosen.halo_share;
        if (new_recv_rows) *new_recv_rows = ch.chosen.recv_rows_max;
        if (allgather_rows) *allgather_rows = (int64_t)(world - 1) * ch.ids.rows_max;
    })
}

int gcnhost_structure_groups(const int *g_indptr, const int *g_indices, int n_rows, int *group, int *n_groups, int *sweeps,
                             double *largest_share, int *useful) {
    if (!g_indptr || !g_indices || n_rows < 0 || !group) return -1;
    API_TRY({
        const StructureGroups sg = structure_groups(g_indptr, g_indices, n_rows);
        if (n_rows) memcpy(group, sg.group.data(), (size_t)n_rows * sizeof(int));
        if (n_groups) *n_groups = sg.n_groups;
        if (sweeps) *sweeps = sg.sweeps;
        if (largest_share) *largest_share = sg.largest_share;
        if (useful) *useful = sg.useful ? 1 : 0;
    })
}

int gcnhost_glorot(float *w, int size, int in_size, int out_size, long seed, int skip_draws) {
    HostRng rng;
    rng.seed_time((unsigned)seed);
    for (int i = 0; i < skip_draws; i++) rng.next();
    Variable v(size, false);
    v.glorot(in_size, out_size, rng);
    memcpy(w, v.data.data(), (size_t)size * sizeof(float));
    return 0;
}
int gcnhost_host_masks(uint8_t *keep, int64_t n, float p, long seed, int64_t skip_draws) {
    HostRng rng;
    rng.seed_time((unsigned)seed);
    for (int64_t i = 0; i < skip_draws; i++) rng.next();
    const int thr = (int)(p * MY_RAND_MAX);
    for (int64_t i = 0; i < n; i++) keep[i] = (int)rng.next() >= thr;
    return 0;
}

}  // extern "C"
