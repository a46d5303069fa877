#include "comm.h"
#include <cstring>
#include <memory>
#include <vector>
#include <rccl/rccl.h>
#include "hip_check.h"

#define NCCL_CHECK(expr)                                                                        \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != ncclSuccess) {                                                                \
            char _buf[512];                                                                     \
            snprintf(_buf, sizeof _buf, "RCCL_ASSERT: %s %s %d", ncclGetErrorString(_r), __FILE__, __LINE__); \
            throw GcnHipFailure(1000 + (int)_r, _buf);                                          \
        }                                                                                       \
    } while (0)

static_assert(sizeof(ncclUniqueId) == GCN_NCCL_ID_BYTES, "ncclUniqueId size");

int rccl_get_unique_id(char id[GCN_NCCL_ID_BYTES]) {
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return 1000 + (int)r;
    memcpy(id, &u, sizeof u);
    return 0;
}

void exchange_buffers_create(gcnhip_ctx *ctx, const ExchangePlan &plan, int max_ld_words, ExchangeBuffers *out) {
    out->ctx = ctx;
    out->max_ld_words = (size_t)max_ld_words;
    if (!plan.halo) return;
    void *p;
    const size_t n = plan.send_rows.size();
    GCNHIP_CHECK(gcnhip_malloc(ctx, &p, (n ? n : 1) * sizeof(int)));
    out->d_send_rows = (int *)p;
    if (n) GCNHIP_CHECK(gcnhip_h2d(ctx, p, plan.send_rows.data(), n * sizeof(int)));
    GCNHIP_CHECK(gcnhip_malloc(ctx, &p, (n ? n : 1) * (size_t)max_ld_words * sizeof(float)));
    out->d_send_buf = (float *)p;
}
void exchange_buffers_destroy(ExchangeBuffers *b) {
    if (!b || !b->ctx) return;
    gcnhip_free(b->ctx, b->d_send_rows);
    gcnhip_free(b->ctx, b->d_send_buf);
    *b = ExchangeBuffers();
}

ExchangeLane::ExchangeLane(gcnhip_ctx *main_ctx, int device, Comm *parent, const ExchangePlan &plan, int max_ld_words, bool timers_on)
    : main(main_ctx) {
    GCNHIP_CHECK(gcnhip_ctx_create(&ctx, device, nullptr));
    try {
        comm.reset(parent->clone_for(ctx));
        exchange_buffers_create(ctx, plan, max_ld_words, &xbuf);
        timers.reset(new DeviceTimers(ctx));
        timers->enabled = timers_on;
        for (int i = 0; i < 32; i++) {
            void *e = nullptr;
            GCNHIP_CHECK(gcnhip_event_create_sync(&e));
            events.push_back(e);
        }
    } catch (...) {
        for (void *e : events) gcnhip_event_destroy(e);
        timers.reset();
        exchange_buffers_destroy(&xbuf);
        comm.reset();
        gcnhip_ctx_destroy(ctx);
        throw;
    }
}

ExchangeLane::~ExchangeLane() {
    gcnhip_ctx_sync(ctx);
    for (void *e : events) gcnhip_event_destroy(e);
    timers.reset();
    exchange_buffers_destroy(&xbuf);
    comm.reset();
    gcnhip_ctx_destroy(ctx);
}

void *ExchangeLane::next_event() {
    void *e = events[next];
    next = (next + 1) % events.size();
    return e;
}

void *ExchangeLane::begin(const ExchangePlan &plan, float *table, int ld_words) {
    void *ev_in = next_event(), *ev_done = next_event();
    GCNHIP_CHECK(gcnhip_event_record(main, ev_in));          // this rank's block of the table is complete behind this point
    GCNHIP_CHECK(gcnhip_stream_wait_event(ctx, ev_in));
    timers->start(TMR_COMM);
    comm->exchange_rows(plan, xbuf, table, ld_words);
    timers->stop(TMR_COMM);
    GCNHIP_CHECK(gcnhip_event_record(ctx, ev_done));
    return ev_done;
}

void ExchangeLane::wait(void *ev) { GCNHIP_CHECK(gcnhip_stream_wait_event(main, ev)); }

namespace {

// Two communicators of one process (training lane + validation lane) must never have collectives in
// flight at the same time: RCCL kernels wait for their peers, and two of them queued in different orders
// on different GPUs can wait for each other.  Every rank runs the same host program, so the enqueue
// order is the same everywhere; the turnstile event makes the device execute the collectives in that
// order too — the other lane's compute still overlaps, which is what the second lane is for.
struct Turnstile {
    void *ev = nullptr;
    bool armed = false;
    Turnstile() { GCNHIP_CHECK(gcnhip_event_create_sync(&ev)); }
    ~Turnstile() { gcnhip_event_destroy(ev); }
};

struct RcclComm : Comm {
    gcnhip_ctx *ctx;
    ncclComm_t comm;
    int r, w;
    float *scratch = nullptr;       // device staging for the init-time host reductions
    std::shared_ptr<Turnstile> turn;
    void enter() {
        if (turn.use_count() > 1 && turn->armed) GCNHIP_CHECK(gcnhip_stream_wait_event(ctx, turn->ev));
    }
    void leave() {
        if (turn.use_count() > 1) { GCNHIP_CHECK(gcnhip_event_record(ctx, turn->ev)); turn->armed = true; }
    }
    RcclComm(gcnhip_ctx *c, int rank, int world, const char *id) : ctx(c), r(rank), w(world) {
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        NCCL_CHECK(ncclCommInitRank(&comm, world, u, rank));
        alloc_scratch();
        turn = std::make_shared<Turnstile>();
    }
    RcclComm(gcnhip_ctx *c, int rank, int world, const RcclComm &parent) : ctx(c), r(rank), w(world), turn(parent.turn) {
        NCCL_CHECK(ncclCommSplit(parent.comm, 0, rank, &comm, nullptr));
        alloc_scratch();
    }
    void alloc_scratch() {
        void *p;
        GCNHIP_CHECK(gcnhip_malloc(ctx, &p, 64 * sizeof(float)));
        scratch = (float *)p;
    }
    Comm *clone_for(gcnhip_ctx *other) override { return new RcclComm(other, r, w, *this); }
    ~RcclComm() override {
        gcnhip_ctx_sync(ctx);
        ncclCommDestroy(comm);
        gcnhip_free(ctx, scratch);
    }
    int rank() const override { return r; }
    int size() const override { return w; }
    const char *transport() const override { return "rccl"; }
    int transport_ranks() const override {
        int n = 0;
        NCCL_CHECK(ncclCommCount(comm, &n));
        return n;
    }
    void allgather_rows(float *base, size_t block) override {
        // in place: sendbuff == recvbuff + rank * count
        enter();
        NCCL_CHECK(ncclAllGather(base + block * r, base, block, ncclFloat, comm, (hipStream_t)gcnhip_ctx_stream(ctx)));
        leave();
    }
    void exchange_rows(const ExchangePlan &plan, ExchangeBuffers &bufs, float *table, int ld) override {
        if (!plan.halo) { allgather_rows(table, (size_t)plan.rows_max * ld); return; }
        if ((size_t)ld > bufs.max_ld_words) throw GcnHipFailure(-1, "exchange_rows: row wider than the packing buffer");
        // pack: one gather kernel for all peers (send_rows is grouped by destination)
        GCNHIP_CHECK(gcnhip_gather_rows(ctx, table + (size_t)plan.own_offset * ld, ld, bufs.d_send_rows, (int)plan.send_rows.size(), bufs.d_send_buf));
        enter();
        hipStream_t st = (hipStream_t)gcnhip_ctx_stream(ctx);
        NCCL_CHECK(ncclGroupStart());
        for (int q = 0; q < w; q++) {
            if (q == r) continue;
            const size_t ns = (size_t)(plan.send_off[q + 1] - plan.send_off[q]) * ld, nr = (size_t)(plan.recv_off[q + 1] - plan.recv_off[q]) * ld;
            if (ns) NCCL_CHECK(ncclSend(bufs.d_send_buf + (size_t)plan.send_off[q] * ld, ns, ncclFloat, q, comm, st));
            if (nr) NCCL_CHECK(ncclRecv(table + ((size_t)plan.n_local + plan.recv_off[q]) * ld, nr, ncclFloat, q, comm, st));
        }
        NCCL_CHECK(ncclGroupEnd());
        leave();
    }
    void allreduce_sum(float *buf, size_t n) override {
        enter();
        NCCL_CHECK(ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, comm, (hipStream_t)gcnhip_ctx_stream(ctx)));
        leave();
    }
    void allreduce_sum_host(double *vals, int n) override {
        // counts and small scalars: exact in f32 up to 2^24, which bounds num_nodes here
        std::vector<float> f(n);
        for (int i = 0; i < n; i++) f[i] = (float)vals[i];
        GCNHIP_CHECK(gcnhip_h2d(ctx, scratch, f.data(), n * sizeof(float)));
        allreduce_sum(scratch, n);
        GCNHIP_CHECK(gcnhip_d2h(ctx, f.data(), scratch, n * sizeof(float)));
        for (int i = 0; i < n; i++) vals[i] = f[i];
    }
};

struct HostComm : Comm {
    gcnhip_ctx *ctx;
    int r, w;
    gcn_host_allgather_fn ag;
    gcn_host_allreduce_fn ar;
    void *user;
    std::vector<float> stage;
    std::vector<double> dstage;
    HostComm(gcnhip_ctx *c, int rank, int world, gcn_host_allgather_fn a, gcn_host_allreduce_fn b, void *u)
        : ctx(c), r(rank), w(world), ag(a), ar(b), user(u) {}
    int rank() const override { return r; }
    int size() const override { return w; }
    const char *transport() const override { return "host callbacks"; }
    void allgather_rows(float *base, size_t block) override {
        stage.resize(block * w);
        GCNHIP_CHECK(gcnhip_d2h(ctx, stage.data() + block * r, base + block * r, block * sizeof(float)));
        ag(user, stage.data(), block);
        // the other ranks' blocks only: this rank's own block is what it sent, and another stream of this rank may be
        // reading it right now (the exchange lane runs beside the aggregation of the own columns) — do not rewrite it
        if (r > 0) GCNHIP_CHECK(gcnhip_h2d(ctx, base, stage.data(), block * r * sizeof(float)));
        if (r + 1 < w) GCNHIP_CHECK(gcnhip_h2d(ctx, base + block * (r + 1), stage.data() + block * (r + 1), block * (size_t)(w - r - 1) * sizeof(float)));
    }
    // HALO through a host-staged all-gather of whole (padded) blocks: every rank's block reaches the host, the rows
    // of the plan's peer segments are picked from it.  Same table as the point-to-point exchange, so the layout,
    // the column remap and everything downstream of it are what the tests exercise.
    void exchange_rows(const ExchangePlan &plan, ExchangeBuffers &, float *table, int ld) override {
        if (!plan.halo) { allgather_rows(table, (size_t)plan.rows_max * ld); return; }
        const size_t block = (size_t)plan.rows_max * ld;
        stage.assign(block * w, 0.f);
        if (plan.n_local)
            GCNHIP_CHECK(gcnhip_d2h(ctx, stage.data() + block * r, table + (size_t)plan.own_offset * ld, (size_t)plan.n_local * ld * sizeof(float)));
        ag(user, stage.data(), block);
        std::vector<float> halo(plan.recv_rows.size() * (size_t)ld);
        for (int q = 0; q < w; q++)
            for (int k = plan.recv_off[q]; k < plan.recv_off[q + 1]; k++)
                memcpy(&halo[(size_t)k * ld], &stage[block * q + (size_t)plan.recv_rows[k] * ld], (size_t)ld * sizeof(float));
        if (!halo.empty())
            GCNHIP_CHECK(gcnhip_h2d(ctx, table + (size_t)plan.n_local * ld, halo.data(), halo.size() * sizeof(float)));
    }
    void allreduce_sum(float *buf, size_t n) override {
        stage.resize(n);
        dstage.resize(n);
        GCNHIP_CHECK(gcnhip_d2h(ctx, stage.data(), buf, n * sizeof(float)));
        for (size_t i = 0; i < n; i++) dstage[i] = stage[i];
        ar(user, dstage.data(), n);
        for (size_t i = 0; i < n; i++) stage[i] = (float)dstage[i];
        GCNHIP_CHECK(gcnhip_h2d(ctx, buf, stage.data(), n * sizeof(float)));
    }
    void allreduce_sum_host(double *vals, int n) override { ar(user, vals, (size_t)n); }
    Comm *clone_for(gcnhip_ctx *other) override { return new HostComm(other, r, w, ag, ar, user); }
};

}  // namespace

Comm *make_rccl_comm(gcnhip_ctx *ctx, int rank, int world, const char id[GCN_NCCL_ID_BYTES]) {
    return new RcclComm(ctx, rank, world, id);
}
Comm *make_host_comm(gcnhip_ctx *ctx, int rank, int world, gcn_host_allgather_fn ag, gcn_host_allreduce_fn ar, void *user) {
    return new HostComm(ctx, rank, world, ag, ar, user);
}
