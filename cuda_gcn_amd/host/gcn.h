// gcn.h — the model driver.  GCNParams / GCNData are the reference's types
// (src/seq/gcn.h:9-22); HipGCN has the shape of GCN / CUDAGCN
// (src/seq/gcn.h:24-44, src/cuda/cuda_gcn.cuh:11-34): constructed from
// (GCNParams, GCNData*), driven by run().  Everything between the two lives on
// the GPU; per epoch the host reads back 2 x 8 floats.
#pragma once
#include <memory>
#include <utility>
#include <vector>
#include "comm.h"
#include "module.h"
#include "optim.h"
#include "partition.h"
#include "sparse.h"
#include "timer.h"
#include "variable.h"

struct GCNParams {
    int num_nodes, input_dim, hidden_dim, output_dim;
    float dropout, learning_rate, weight_decay;
    int epochs, early_stopping;
    static GCNParams get_default();          // {2708,1433,16,7, 0.5, 0.01, 5e-4, 100, 0} (gcn.cpp:9-11)
};

class GCNData {
public:
    SparseIndex feature_index, graph;
    std::vector<int> split;
    std::vector<int> label;
    std::vector<float> feature_value;
};

enum HipGCNFlags {
    HIPGCN_MODULAR = 1,       // one module per reference module instead of the fused epilogues
    HIPGCN_HOST_MASKS = 2,    // parity mode: dropout decisions from the reference's host RNG stream
    HIPGCN_TIMERS = 4,        // record device-event timers per op
    HIPGCN_NO_GRAPH = 8,      // never replay epochs from a captured hipGraph
    HIPGCN_EVAL_LANE = 16,    // validation forward on a second stream, overlapped with the next training epoch (opt-in; with
                              // several GPUs it adds a second communicator: not the default until measured on such a node)
    HIPGCN_NO_EVAL_LANE = 32, // never
    HIPGCN_NO_REPLICATE_L1 = 64, // multi-GPU: all-gather H0 instead of computing X.W1 for all rows on every rank
    HIPGCN_REPLICATE_L1 = 128,   // ... or force the replication (default: 2-4 GPUs replicate, 8 gather)
    HIPGCN_GATHER_DH1 = 256,     // multi-GPU: all-gather dH1 (128 wide) instead of dZ0 (48 wide) + 1 bit per element of H1
    HIPGCN_BF16_TABLES = 2048,   // opt-in, beyond the reference: GraphSum gathers bfloat16 copies of its inputs (f32 accumulate)
    HIPGCN_ALL_ROWS = 4096,      // compute every row of the logits (default: only rows of the scored split, which is all the loss and accuracy read)
    HIPGCN_NO_AGG_FIRST_EVAL = 8192, // evaluation forwards keep the reference's order A^.(X.W1) instead of (A^.X).W1 with A^.X built once
    HIPGCN_EXCHANGE_ALLGATHER = 16384, // multi-GPU: always all-gather whole row blocks before an aggregation
    HIPGCN_EXCHANGE_HALO = 32768,      // ... or always exchange only the rows some local edge points at (default: decided per graph)
    HIPGCN_BWD_PIPELINE = 262144,      // opt-in: the hidden layer's backward aggregation in row blocks, each block's share of the first
                                       // layer's weight gradient on a second stream (same bits; measured slower)
    HIPGCN_NO_LABEL_HINT = 524288,     // never use the dataset's labels as row groups of the aggregation's schedule (groups are then
                                       // looked for in the graph itself, cluster.h)
    HIPGCN_MASKED_BWD = 131072,        // the output layer's backward masks the known-zero rows of dZ at every launch instead of
                                       // aggregating through an operator that has lost the edges pointing at them
    HIPGCN_PACKED_DH1 = 65536,         // opt-in: dH1 reaches the hidden layer's backward gather as packed rows (same bits; measured slower)
    HIPGCN_NULL_COMM = 1024,     // world > 1 without transport: collectives are no-ops (per-rank compute timing only)
    HIPGCN_NO_ROW_GROUPS = 512,  // keep the aggregation's plain descending-degree row schedule (no timing of alternatives)
    HIPGCN_OVERLAP_EXCHANGE = 1048576,  // multi-GPU: exchanges on their own stream; each aggregation starts on the edges that point at
                                        // this rank's own rows while the other ranks' rows arrive, then adds the rest (the order of a
                                        // row's sum then depends on the partition: float tolerance, not bit-identity, across P)
    HIPGCN_STRUCTURE_PARTITION = 2097152, // multi-GPU: rank blocks formed from groups found in the graph (cluster.h) instead of
                                          // contiguous id ranges, when that shrinks the neediest rank's halo (default: decided per graph)
    HIPGCN_ID_PARTITION = 4194304,        // ... never
    HIPGCN_EDGE_COEF = 16777216,          // aggregate with the reference's per-edge coefficients 1/sqrt(deg deg) (module.cpp:91-93) instead of
                                          // the factored form dinv[r] * sum(dinv[c] * x[c]) (default on the fused f32 path: no coefficient
                                          // stream beside the indices, -6..10 % per aggregation; same real numbers, two more roundings per term)
    HIPGCN_SYNC_EPOCHS = 8388608,         // run(): wait for every epoch before the next is enqueued (the reference's loop, gcn.cpp:133-151:
                                          // `time=` is then that epoch's own latency).  Default: epochs are enqueued ahead of the line being
                                          // printed whenever no decision depends on a printed number (early_stopping == 0)
};

struct HipGCNOptions {
    int device = 0;
    long seed = 0;            // plays time(NULL) of src/seq/rand.cpp:7
    int flags = 0;
    Comm *comm = nullptr;     // NULL: single GPU.  Not owned unless own_comm.
    bool own_comm = false;
    // multi-GPU: RCCL unique id (GCN_NCCL_ID_BYTES) and rank/world when comm == NULL and world > 1
    int rank = 0, world = 1;
    const char *nccl_id = nullptr;
    // or a host-staged transport (tests: torch.distributed gloo through callbacks)
    gcn_host_allgather_fn host_allgather = nullptr;
    gcn_host_allreduce_fn host_allreduce = nullptr;
    void *host_user = nullptr;

    // ---- switches that have no flag bit.  HipGCN itself never reads the environment: the two places that build a model
    // from outside (main.cpp, capi.cpp) call from_environment() ONCE and hand the result in.
    bool verbose = false;                 // HIPGCN_VERBOSE: where the model build's wall time goes, run-loop statistics (stderr)
    int exchange = -1;                    // HIPGCN_EXCHANGE: -1 unset, 0 auto (per graph), 1 allgather, 2 halo
    bool structure_groups = true;         // HIPGCN_NO_STRUCTURE_GROUPS clears: never search the graph for row groups
    bool eval_fusion = true;              // HIPGCN_NO_EVAL_FUSION clears: evaluation forwards store the hidden matrix and run H1.W2 as its own launch
    bool loss_epilogue = true;            // HIPGCN_NO_LOSS_EPILOGUE clears: the loss kernel reads the stored logits (round 4) instead of riding in the class-width aggregation's epilogue
    bool mask_bits = true;                // HIPGCN_NO_MASK_BITS clears: the Matmul backward re-reads H1 instead of one bit per element
    bool loss_records_metrics = true;     // HIPGCN_RECORD_LAUNCH clears: the metrics row gets a launch of its own (A/B)
    int bwd_chunks = 4;                   // HIPGCN_BWD_CHUNKS: row blocks of the opt-in backward pipeline
    bool readback_stream = false;         // HIPGCN_READBACK_STREAM: run()'s read-back copies on a stream of their own (measured slower)
    int readback_group = 0;               // HIPGCN_READBACK_GROUP: epochs per read-back group (0: calibrated)
    // HIPGCN_SCHEDULE=degree|label|dealt[-G]|structure: pin the aggregation's row schedule instead of timing the candidates at
    // load (-1: timed).  A pinned run launches no tuning kernels, so a kernel-trace profile of it holds the epochs' launches only.
    int schedule = -1, schedule_groups = 256;
    bool fold_training = false;           // HIPGCN_FOLD_TRAINING (experiments build): the TRAINING context's aggregations add split rows' segments
                                          // inside the launch (context option gs_fold) — no finalize launch to queue behind the validation lane's kernels
    bool slice_tuning = true;             // HIPGCN_NO_SLICE_TUNING clears: the hidden-width aggregation keeps 64-float column slices
                                          // (default: chosen per graph by a rule on its structure, HipGCN::choose_slice_width)
    // HIPGCN_GEMM=f32|bf16x3: arithmetic of the dense first-layer products (0: exact-f32 MFMA, 1: three-plane bf16 split on the
    // bf16 MFMA pipe, same f32 error bound; -1: the library's default)
    int gemm = -1;

    // `base` with every HIPGCN_* variable of the process environment applied (flag variables OR their bit in)
    static HipGCNOptions from_environment(HipGCNOptions base);
};

class HipGCN {
public:
    GCNParams params;
    HipGCN(GCNParams params, GCNData *data, const HipGCNOptions &opt = HipGCNOptions());
    ~HipGCN();
    HipGCN(const HipGCN &) = delete;

    // gcn.cpp:130-158, same output lines.  With early stopping (or HIPGCN_SYNC_EPOCHS) the loop is the reference's: enqueue
    // one epoch, wait, print, decide.  Otherwise nothing the host prints feeds back into the run, so epochs are enqueued
    // ahead of the line being printed and their metrics are copied back behind them in groups of consecutive epochs
    // (1 when an epoch takes milliseconds, up to READBACK_GROUP_MAX when it takes tens of microseconds; run_pipelined):
    // `time=` is then the interval between consecutive group completions / the group's size and `total training time`
    // their sum = the wall time of the whole loop.
    void run();
    std::pair<float, float> train_epoch();                    // synchronises to return (loss, acc)
    std::pair<float, float> eval(int current_split);
    // enqueue n x (train_epoch + eval(2)) with no host synchronisation in between, then read back
    // 4 floats per epoch into trace (may be NULL)
    void run_epochs(int n, float *trace);
    void sync();

    // introspection for tests / bench
    int rank() const { return env.comm->rank(); }
    int schedule_mode() const { return sched_mode; }
    int schedule_groups() const { return sched_groups; }
    int schedule_slice_floats() const { return slice_floats; }  // column-slice width the hidden-width aggregation was tuned to (64 or 32)
    int world() const { return env.comm->size(); }
    const char *transport() const { return env.comm->transport(); }
    int transport_ranks() const { return env.comm->transport_ranks(); }
    int local_rows() const { return n_local; }
    int row_start() const { return part.start[env.comm->rank()]; }
    const RowPartition &partition() const { return part; }
    // multi-GPU: the model may renumber the nodes before it partitions them (partition.h, choose_node_order): row r of
    // this rank is then node node_order()[row_start() + r] of the dataset it was given.  Empty: the ids were kept.
    const std::vector<int> &node_order() const { return node_order_; }
    const char *node_order_name() const { return node_order_name_; }
    const ExchangePlan &exchange_plan() const { return xplan; }
    // variable k as in gcn.cpp:21-54 (1 H0, 2 W1, 3 H1, 4 Z0, 5 W2, 6 Z); rows x cols floats, this rank's rows.
    // FACTORED FORM (default on the fused f32 path; factored() tells, HIPGCN_EDGE_COEF restores the reference's values): the
    // matrices an aggregation gathers from are stored pre-multiplied by dinv = 1/sqrt(deg) of their row, so get_var returns
    // dinv.H0 (1), dinv.H1 (3), dinv.Z0 (4) and, as gradients, dinv.dZ (6), dinv.dH1 (3), dZ0/dinv (4), dH0/dinv (1); the
    // logits Z (6), the weights and their gradients are the reference's own.  row_scale() returns dinv for this rank's rows.
    // PARTIAL-ROW CONTRACT of variable 6 (and 4 on an evaluation forward): by default the last aggregation of a forward
    // computes only the rows of the split being scored — all the loss and the accuracy read (module.cpp:131-133,
    // gcn.cpp:86-88) — so after train_epoch() only rows of the training split hold this epoch's logits, after eval(s)
    // only rows of split s; the other rows keep whatever an earlier forward left (or the zeros of the allocation).  The
    // reference fills every row on every forward: construct with HIPGCN_ALL_ROWS to get that.
    void get_var(int k, bool grad, std::vector<float> &out, int *rows, int *cols);
    bool factored() const { return factored_; }
    void row_scale(std::vector<float> &dinv);                 // 1/sqrt(deg) of this rank's rows (deg of the full graph, self loop included)
    void set_weights(const float *w1, const float *w2);       // [F x h], [h x C] row-major
    DeviceTimers &device_timers() { return *timers; }
    double timer_total(timer_instance t, long *count);        // both lanes
    void timers_reset();
    // switch the per-op device-event timers on or off (both lanes).  While they are on, run_epochs does not
    // replay the captured epoch (event records per op are not part of it).
    void set_timers(bool on);
    long n_edges_local() const { return nnzA_local; }

private:
    GCNData *data;
    std::unique_ptr<GCNData> renumbered;                       // the dataset in node_order_ (owned), when the ids were not kept
    std::vector<int> node_order_;
    const char *node_order_name_ = "ids";
    void renumber_nodes(int world);
    HipEnv env;
    std::unique_ptr<Comm> owned_comm;
    std::unique_ptr<DeviceTimers> timers;
    RowPartition part;
    ExchangePlan xplan;                                        // layout of gathered tables, send/receive lists
    ExchangeBuffers xbuf;
    int n_local = 0;
    long nnzA_local = 0;
    int flags = 0;
    bool factored_ = false;
    void apply_factored_scales();                              // X, A^.X and the replicated X of this rank -> D^-1/2 . (them)
    int device_ = 0;
    HipGCNOptions opt_;                                        // the switches of this model (a copy; comm pointers not used after init)
    const float *eval_vals = nullptr;
    HostRng rng;

    gcnhip_graph *graph = nullptr;
    gcnhip_feat *feat = nullptr;
    // multi-GPU: the first-layer product is replicated (every rank multiplies ALL rows of X by W1): one GEMM
    // of N x F x h per forward instead of an all-gather of N x h floats over xGMI
    gcnhip_feat *feat_full = nullptr;
    gcnhip_graph *graph_l1 = nullptr;                          // this rank's rows, GLOBAL column ids
    // Aggregate-first evaluation (dense X, fused mode): A^.X of this rank's rows, built once.  An evaluation forward is
    // then ReLU((A^.X).W1) — one GEMM, no hidden-width aggregation and no exchange before the hidden layer.
    gcnhip_feat *feat_agg = nullptr;
    const float *agg_vals = nullptr;
    std::vector<Module *> eval_modules;                        // [0] owned (the GEMM on A^.X); the rest are modules[2..]
    bool h1_from_fused_eval = false;                           // the last forward on the main stream kept its hidden matrix in registers (get_var(3) rebuilds it)
    void build_agg_first_eval();
    const float *full_vals = nullptr;
    bool replicate_l1 = false;
    bool rebuild_dh1 = false;                                  // multi-GPU backward: gather dZ0 + mask bits, rebuild dH1 everywhere
    uint32_t *d_pos_bits = nullptr;                            // [table_rows * wpr]
    gcnhip_rowpack *dh1_pack = nullptr;                        // dH1 as packed rows (single GPU, hidden % 64 == 0)
    std::vector<std::unique_ptr<HipVariable>> variables;       // index = reference variable number
    HipVariable *input = nullptr, *output = nullptr;
    const float *input_vals = nullptr;                         // what SparseMatmul reads
    std::vector<Module *> modules;
    std::unique_ptr<HipAdam> optimizer;

    float *gradbuf = nullptr;                                  // [W1.grad | W2.grad | result(4)] one all-reduce
    size_t gradbuf_elems = 0;
    float *d_result = nullptr;
    int32_t *d_result_i = nullptr;
    int32_t *d_truth[4] = {};                                  // per split code 1..3
    int32_t *cur_truth = nullptr;
    std::unique_ptr<BackwardPipeline> bwd_pipe;                 // hidden-layer backward aggregation || dW1 product, in row blocks
    void build_bwd_pipeline(HipSparseMatmul *sm, HipGraphSum *gs);
    void destroy_bwd_pipeline();
    gcnhip_graph *graph_bwd_out = nullptr;                     // `graph` without the edges whose source is outside the training split
    // HIPGCN_OVERLAP_EXCHANGE: `graph` and `graph_bwd_out` cut by column owner (own rows / other ranks' rows), the split
    // subsets of the last aggregation on both halves, and the exchange stream
    gcnhip_graph *graph_loc = nullptr, *graph_rem = nullptr, *graph_bwd_loc = nullptr, *graph_bwd_rem = nullptr;
    gcnhip_rowset *split_rows_loc[4] = {}, *split_rows_rem[4] = {};
    gcnhip_rowset *cur_out_rows_loc = nullptr, *cur_out_rows_rem = nullptr;
    std::unique_ptr<ExchangeLane> xlane;
    void build_overlap();
    void wire_overlap(HipGraphSum *gs, bool output_layer);
    std::vector<uint32_t> h_train_bits;
    uint32_t *d_train_bits = nullptr;                          // bit per (padded) node: in the training split
    const uint32_t *bwd_bits = nullptr;
    gcnhip_rowset *split_rows[4] = {};                         // rows of `graph` whose node is in split s (all the loss reads); owned by graph
    gcnhip_rowset *cur_out_rows = nullptr;                     // follows set_truth; NULL with HIPGCN_ALL_ROWS
    int32_t *d_split_list[4] = {};                             // local row ids of split s, ascending (the loss walks only these)
    int split_local_n[4] = {};
    int32_t *cur_rows = nullptr;
    int cur_rows_n = 0;
    int split_count[4] = {};
    int cur_count = 0;
    float *d_ring = nullptr;
    static constexpr int RING = 1024;
    uint8_t *d_keep0 = nullptr, *d_keep1 = nullptr;
    std::vector<uint8_t> h_keep0, h_keep1;
    long keep0_first = 0;                                      // global nnz index of h_keep0[0]
    long epochs_done = 0;                                      // training passes enqueued = *env.d_epoch once their Adam launches have run
    void *epoch_graph = nullptr;                               // captured train_epoch + eval(2)
    bool enqueue_epoch_replay();                               // one epoch from the captured hipGraph (captures it on first use); false: not replayable
    // run(): read-back of an epoch's metrics row without stalling the producer streams
    static constexpr int PIPELINE_DEPTH = 4;                   // read-back groups in flight
    static constexpr int READBACK_GROUP_MAX = 64;              // epochs per read-back group, at most (RING is a multiple)
    static constexpr int READBACK_CALIBRATION = 16;            // epochs read back one by one before the group size is set
    static constexpr double READBACK_GROUP_SECONDS = 2e-3;     // a group spans about this long
    struct Readback {
        gcnhip_ctx *ctx = nullptr;                             // its own stream (NULL: copies go on the producer's)
        float *host = nullptr;                                 // pinned [PIPELINE_DEPTH][READBACK_GROUP_MAX][32]: whole ring rows
        void *ev_ready[PIPELINE_DEPTH] = {}, *ev_copied[PIPELINE_DEPTH] = {};
    };
    std::unique_ptr<Readback> readback;
    void readback_create(bool own_stream);
    void readback_destroy();
    void readback_enqueue(long first_epoch_index, int n_epochs, int slot, gcnhip_ctx *producer);
    void run_synchronous();
    void run_pipelined();
    void report_test();

    // Validation lane.  eval(e) reads only the weights Adam(e) wrote, and train(e+1) needs the same
    // weights and nothing from eval(e): the two are independent until Adam(e+1).  With more than one GPU
    // each of them alternates compute with an all-gather, so running eval(e) on its own stream /
    // communicator / activation buffers lets one lane compute while the other communicates.
    struct EvalLane {
        HipEnv env;
        ExchangeBuffers xbuf;
        std::unique_ptr<Comm> comm;
        std::unique_ptr<DeviceTimers> timers;
        gcnhip_graph *graph = nullptr;                         // own split-row scratch
        gcnhip_graph *graph_l1 = nullptr;
        std::unique_ptr<HipVariable> H0, H1, Z0, Z;
        std::vector<Module *> modules;
        float *d_result = nullptr;
        int32_t *d_result_i = nullptr;
        int32_t *truth = nullptr;
        gcnhip_rowset *split_rows[4] = {};                     // the lane has its own adjacency object
        gcnhip_rowset *out_rows = nullptr;
        int32_t *rows = nullptr;
        int rows_n = 0;
        int count = 0;
        void *ev_weights = nullptr, *ev_done = nullptr;        // Adam(e) -> eval(e);  eval(e) -> Adam(e+1)
        void *ev_fork = nullptr;                               // one GPU: training GEMM done -> the validation pass may start
        bool pending = false;
        long epoch_word = -1;                                  // host shadow of *env.d_epoch (starts at 0xFFFFFFFF)
    };
    std::unique_ptr<EvalLane> lane;
    void init(const HipGCNOptions &opt);
    void release();                                            // frees everything that exists; safe on a half-built object
    void destroy_lane();
    void build_eval_lane();
    void eval_on_lane(int current_split);
    void lane_begin(int current_split);
    void lane_end(int current_split);
    void eval_then_train_zipped(int current_split);
    void train_begin();
    void train_end();

    // row schedule of the aggregation (gcnhip_graph_set_schedule): candidates timed once, fastest kept
    int sched_mode = 0, sched_groups = 0, slice_floats = 64;
    bool labels_assortative = false;
    std::vector<int> structure_group;                          // per node: group found in the graph (cluster.h); empty: none useful
    int structure_n_groups = 0;
    void tune_schedule();
    void choose_slice_width();
    void apply_schedule(gcnhip_ctx *ctx, gcnhip_graph *g);
    void add_split_rowsets(gcnhip_ctx *ctx, gcnhip_graph *g, gcnhip_rowset *out[4]);
    void build_modules();
    void set_truth(int current_split);
    void host_masks_for_epoch();
    void train_epoch_async();
    void eval_async(int current_split);
    std::pair<float, float> read_metrics(long epoch_index, int slot);
};
