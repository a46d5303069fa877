// comm.h — the two collectives the row-partitioned epoch needs.
// The reference has no distributed layer; this is new work for 1..8 MI355X of
// one node: one process (or thread) per GPU, RCCL over xGMI.
//   allgather_rows: every rank contributes its block of an [world*block] f32
//                   buffer IN PLACE (block r lives at base + r*block_elems);
//   allreduce_sum : in-place f32 sum (weight gradients + loss/accuracy scalars).
// Both are enqueued on the context's stream: no host synchronisation.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
#include "gcnhip_driver.h"
#include "partition.h"
#include "timer.h"

// device-side companions of an ExchangePlan for ONE context / stream (each lane has its own, so the lanes'
// exchanges never share a send buffer): the send lists and a packing buffer.  Empty for ALLGATHER plans.
struct ExchangeBuffers {
    gcnhip_ctx *ctx = nullptr;
    int *d_send_rows = nullptr;
    float *d_send_buf = nullptr;      // [send_total x max_ld_words]
    size_t max_ld_words = 0;
};
void exchange_buffers_create(gcnhip_ctx *ctx, const ExchangePlan &plan, int max_ld_words, ExchangeBuffers *out);
void exchange_buffers_destroy(ExchangeBuffers *b);

struct Comm {
    virtual ~Comm() {}
    virtual int rank() const = 0;
    virtual int size() const = 0;
    // what moves the bytes ("rccl", "host callbacks", "none") and how many ranks THAT layer counts (RCCL: ncclCommCount of
    // the communicator) — so that a benchmark record states that the collectives really spanned the ranks it claims
    virtual const char *transport() const { return "none"; }
    virtual int transport_ranks() const { return size(); }
    virtual void allgather_rows(float *base, size_t block_elems) = 0;
    // Complete a gathered table [plan.table_rows x ld_words 4-byte words] whose own block (plan.own_offset ..
    // + n_local) this rank has just written: ALLGATHER plans -> allgather_rows; HALO plans -> pack the rows each
    // peer needs, exchange point to point, receive straight into the peer segments.  Bytes are moved, not interpreted.
    virtual void exchange_rows(const ExchangePlan &plan, ExchangeBuffers &bufs, float *table, int ld_words) = 0;
    virtual void allreduce_sum(float *buf, size_t n) = 0;
    virtual void allreduce_sum_host(double *vals, int n) = 0;   // init-time scalars (synchronises)
    // a second communicator over the same ranks whose collectives run on another context's stream
    // (the validation lane); every rank must call it at the same point
    virtual Comm *clone_for(gcnhip_ctx *ctx) = 0;
};

// Asynchronous exchanges (HIPGCN_OVERLAP_EXCHANGE, SURVEY §8e "overlap local-block SpMM with arrival of remote blocks"):
// a second stream of this rank with its own communicator (clone_for) and packing buffers.  begin() makes that stream
// wait for everything enqueued so far on the main stream (the producer of the rows this rank contributes), enqueues the
// exchange there and returns an event; the main stream goes on with work that needs only this rank's own rows and
// waits for the event (wait()) in front of the first kernel that reads a row of another rank.  Every rank runs the same
// host program, so the collectives of the two communicators are enqueued in the same order everywhere (and, under RCCL,
// executed in that order: the turnstile of comm.cpp).
struct ExchangeLane {
    gcnhip_ctx *main = nullptr;           // the stream whose work the exchange follows and feeds (not owned)
    gcnhip_ctx *ctx = nullptr;            // the exchange stream (owned)
    std::unique_ptr<Comm> comm;
    ExchangeBuffers xbuf;
    std::unique_ptr<DeviceTimers> timers; // TMR_COMM on the exchange stream
    std::vector<void *> events;           // a ring: an event is reused long after its last waiter was enqueued
    size_t next = 0;
    ExchangeLane(gcnhip_ctx *main_ctx, int device, Comm *parent, const ExchangePlan &plan, int max_ld_words, bool timers_on);
    ~ExchangeLane();
    ExchangeLane(const ExchangeLane &) = delete;
    void *begin(const ExchangePlan &plan, float *table, int ld_words);
    void wait(void *ev);
private:
    void *next_event();
};

// world == 1: nothing to exchange
struct SelfComm : Comm {
    int rank() const override { return 0; }
    int size() const override { return 1; }
    void allgather_rows(float *, size_t) override {}
    void exchange_rows(const ExchangePlan &, ExchangeBuffers &, float *, int) override {}
    void allreduce_sum(float *, size_t) override {}
    void allreduce_sum_host(double *, int) override {}
    Comm *clone_for(gcnhip_ctx *) override { return new SelfComm(); }
};

// Timing aid: rank r of `world` with collectives that do nothing.  The rank computes exactly what it would in
// a real run (its row block, its share of every kernel) on whatever the gather buffers hold, so per-rank
// compute time can be measured on one GPU; losses and weights are meaningless.  Never used by bench.py's number.
struct NullComm : Comm {
    int r, w;
    NullComm(int rank, int world) : r(rank), w(world) {}
    int rank() const override { return r; }
    int size() const override { return w; }
    void allgather_rows(float *, size_t) override {}
    void exchange_rows(const ExchangePlan &, ExchangeBuffers &, float *, int) override {}
    void allreduce_sum(float *, size_t) override {}
    void allreduce_sum_host(double *, int) override {}
    Comm *clone_for(gcnhip_ctx *) override { return new NullComm(r, w); }
};

// RCCL (librccl; "nccl" API) on the context's stream
#define GCN_NCCL_ID_BYTES 128
int rccl_get_unique_id(char id[GCN_NCCL_ID_BYTES]);
Comm *make_rccl_comm(gcnhip_ctx *ctx, int rank, int world, const char id[GCN_NCCL_ID_BYTES]);

// Host-staged comm through caller-supplied callbacks (D2H -> callback -> H2D).
// Diagnostic / test transport: lets N ranks be driven by any host-side
// collective (torch.distributed gloo in tests/) without RCCL, e.g. two ranks
// sharing one GPU.  Never used for timing.
typedef void (*gcn_host_allgather_fn)(void *user, float *host_full, size_t block_elems);   // in place on host
typedef void (*gcn_host_allreduce_fn)(void *user, double *host_buf, size_t n);
Comm *make_host_comm(gcnhip_ctx *ctx, int rank, int world, gcn_host_allgather_fn ag, gcn_host_allreduce_fn ar, void *user);
