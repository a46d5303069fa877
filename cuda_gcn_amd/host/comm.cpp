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

namespace {

// Two communicators of one process (training lane + validation lane) must never have collectives in
// flight at the same time: RCCL kernels wait for their peers, and two of them queued in different orders
// on different GPUs can wait for each other.  Every rank runs the same host program, so the enqueue
// order is the same everywhere; the turnstile event makes the device execute the collectives in that
// order too — the other lane's compute still overlaps, which is what the second lane is for.
struct Turnstile {
    void *ev = nullptr;
    bool armed = false;
    Turnstile() { GCNHIP_CHECK(gcnhip_event_create(&ev)); }
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
    void allgather_rows(float *base, size_t block) override {
        // in place: sendbuff == recvbuff + rank * count
        enter();
        NCCL_CHECK(ncclAllGather(base + block * r, base, block, ncclFloat, comm, (hipStream_t)gcnhip_ctx_stream(ctx)));
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
    void allgather_rows(float *base, size_t block) override {
        stage.resize(block * w);
        GCNHIP_CHECK(gcnhip_d2h(ctx, stage.data() + block * r, base + block * r, block * sizeof(float)));
        ag(user, stage.data(), block);
        GCNHIP_CHECK(gcnhip_h2d(ctx, base, stage.data(), block * w * sizeof(float)));
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
