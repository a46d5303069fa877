/*
 * gcnhost.h — C entry points of libgcnhost.so: the C++ host side of the GCN
 * training path (HipGCN and its Hip* modules, cuda_gcn_amd/host/) for callers
 * that are not C++ (bench.py, tests, __graft_entry__.smoke()).
 *
 * The host mirrors the reference's driver: GCN(GCNParams, GCNData*) + run()
 * (src/seq/gcn.h:24-44; CUDA twin src/cuda/cuda_gcn.cuh:11-34).  It reaches
 * the GPU only through include/gcnhip.h (and RCCL for more than one GPU).
 * Every function returns 0 on success, else the failing HIP/RCCL code (or -1);
 * gcnhost_last_error() gives the message.
 */
#ifndef GCNHOST_H
#define GCNHOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef struct gcnhost_model gcnhost_model;

/* GCNParams of the reference (src/seq/gcn.h:9-14), same field order */
typedef struct {
    int num_nodes, input_dim, hidden_dim, output_dim;
    float dropout, learning_rate, weight_decay;
    int epochs, early_stopping;
} gcnhost_params;

enum {
    GCNHOST_MODULAR = 1,      /* one module per reference module (no fused epilogues) */
    GCNHOST_HOST_MASKS = 2,   /* dropout decisions replayed from the reference's host RNG (parity runs) */
    GCNHOST_TIMERS = 4,       /* per-op device-event timers */
    GCNHOST_NO_GRAPH = 8,     /* run_epochs never replays a captured hipGraph */
    GCNHOST_EVAL_LANE = 16,   /* validation forward on a second stream, overlapped with the next training epoch */
    GCNHOST_NO_EVAL_LANE = 32, /* never (default: on when world > 1) */
    GCNHOST_NO_REPLICATE_L1 = 64, /* multi-GPU: all-gather H0 instead of replicating the first-layer product */
    GCNHOST_REPLICATE_L1 = 128,   /* force the replication (default: on for 2-4 GPUs, off for 8) */
    GCNHOST_GATHER_DH1 = 256,     /* multi-GPU backward: all-gather dH1 instead of dZ0 + mask bits */
    GCNHOST_NO_ROW_GROUPS = 512,  /* keep the plain descending-degree row schedule (no load-time timing of alternatives) */
    GCNHOST_BF16_TABLES = 2048,   /* opt-in, beyond the reference: the aggregation gathers bfloat16 copies of H0, Z0, dZ, dH1 (f32 sums) */
    GCNHOST_ALL_ROWS = 4096,      /* compute every row of the logits (default: the last aggregation computes only the rows of the
                                     scored split — all that loss and accuracy read; env HIPGCN_ALL_ROWS=1 does the same) */
    GCNHOST_NO_AGG_FIRST_EVAL = 8192, /* evaluation forwards keep the reference's order A^.(X.W1); default for a dense X: (A^.X).W1 with
                                         A^.X built once (no hidden-width aggregation in eval; env HIPGCN_NO_AGG_FIRST_EVAL=1 does the same) */
    GCNHOST_EXCHANGE_ALLGATHER = 16384, /* multi-GPU: always all-gather whole row blocks before an aggregation */
    GCNHOST_EXCHANGE_HALO = 32768,      /* ... or always exchange only the needed rows, peer to peer (default: decided per graph;
                                           env HIPGCN_EXCHANGE=halo|allgather) */
    GCNHOST_PACKED_DH1 = 65536,       /* opt-in: dH1 travels to the hidden layer's backward gather as packed rows (bit-identical to the
                                         dense gather, measured slower on gfx950; env HIPGCN_PACKED_DH1=1 does the same) */
    GCNHOST_BWD_PIPELINE = 262144,    /* opt-in: hidden-layer backward aggregation in row blocks, each block's share of the first layer's
                                         weight gradient on a second stream (bit-identical; measured slower on one MI355X; env
                                         HIPGCN_BWD_PIPELINE=1; HIPGCN_BWD_CHUNKS sets the number of blocks, default 4) */
    GCNHOST_NO_LABEL_HINT = 524288,   /* the labels are never used as row groups of the aggregation's schedule; groups are looked for in the
                                         graph itself (modularity local moving) and used if they time faster (env HIPGCN_NO_LABEL_HINT=1) */
    GCNHOST_MASKED_BWD = 131072,      /* the output layer's backward masks the rows of dZ outside the training split at every launch
                                         (default: aggregates through gcnhip_graph_create_restricted's operator; env HIPGCN_MASKED_BWD=1) */
    GCNHOST_NULL_COMM = 1024      /* timing aid: rank r of world > 1 with no-op collectives (per-rank compute time; numbers meaningless) */
};

#define GCNHOST_NCCL_ID_BYTES 128
typedef void (*gcnhost_allgather_fn)(void *user, float *host_full, size_t block_elems);
typedef void (*gcnhost_allreduce_fn)(void *user, double *host_buf, size_t n);

const char *gcnhost_last_error(void);
gcnhost_params gcnhost_params_default(void);               /* gcn.cpp:9-11 */
int gcnhost_nccl_unique_id(char id[GCNHOST_NCCL_ID_BYTES]);   /* rank 0; broadcast it to the others */

/* Build the model from the whole dataset in the reference's in-memory layout
 * (GCNData, src/seq/gcn.h:16-22: adjacency CSR with the self loop first,
 * feature CSR + values, split, label); each rank keeps its own row block.
 * f_indices may be NULL for a dense X (every row = columns 0..input_dim-1).
 * seed plays time(NULL) of src/seq/rand.cpp:7.  world > 1: nccl_id (from
 * gcnhost_nccl_unique_id) selects RCCL; or pass host callbacks for a
 * host-staged transport (tests). */
int gcnhost_model_create(gcnhost_model **m, const gcnhost_params *p,
                         const int *g_indptr, const int *g_indices,
                         const int *f_indptr, const int *f_indices, const float *f_val,
                         const int *split, const int *label,
                         long seed, int device, int flags,
                         int rank, int world, const char *nccl_id,
                         gcnhost_allgather_fn host_ag, gcnhost_allreduce_fn host_ar, void *host_user);
int gcnhost_model_destroy(gcnhost_model *m);

int gcnhost_model_train_epoch(gcnhost_model *m, float *loss, float *acc);      /* gcn.cpp:107-118; synchronises */
int gcnhost_model_eval(gcnhost_model *m, int split, float *loss, float *acc);  /* gcn.cpp:120-128; synchronises */
/* n x (train_epoch + eval(2)) enqueued back to back, one synchronisation at the
 * end; trace (may be NULL) gets train_loss, train_acc, val_loss, val_acc per epoch */
int gcnhost_model_run_epochs(gcnhost_model *m, int n, float *trace);
int gcnhost_model_run(gcnhost_model *m);                                       /* gcn.cpp:130-158, prints the reference's lines */
int gcnhost_model_sync(gcnhost_model *m);

/* introspection */
/* multi-GPU: how this rank's gathered tables are completed (halo = 1: per-peer send lists; 0: whole-block all-gather),
 * rows received / sent per exchange, rows of a table, and the share of remote rows the neediest rank reads */
int gcnhost_model_exchange(gcnhost_model *m, int *halo, int64_t *recv_rows, int64_t *send_rows, int *table_rows, double *halo_share);
int gcnhost_model_info(gcnhost_model *m, int *rank, int *world, int *row_start, int *local_rows, int64_t *local_edges);
/* the aggregation's row schedule this rank timed as fastest: 0 descending degree (also when not tuned), 1 label-major,
 * 2 degree rank dealt into n_groups groups, 3 group-major over n_groups groups found in the graph (modularity local moving) */
/* Several GPUs: the model may renumber the nodes by structure before partitioning them (rank blocks are contiguous in the
 * node order; HIPGCN_ID_PARTITION keeps the ids, HIPGCN_STRUCTURE_PARTITION forces the renumbering).  ids[r] (local_rows
 * entries) = the node of the caller's dataset that local row r of this rank is; *renumbered = 0 when the ids were kept. */
int gcnhost_model_row_ids(gcnhost_model *m, int *ids, int *renumbered);
/* factored aggregation (gcn.h, HIPGCN_EDGE_COEF restores the reference's per-edge coefficients): *factored says whether
 * gcnhost_model_get_var returns the gathered matrices pre-multiplied by dinv = 1/sqrt(deg) of their row (variables 1, 3, 4;
 * the logits, weights and weight gradients are never scaled); dinv [local_rows] receives that factor (may be NULL) */
int gcnhost_model_row_scale(gcnhost_model *m, float *dinv, int *factored);
int gcnhost_model_schedule(gcnhost_model *m, int *mode, int *n_groups);
/* the layer that moves this model's rows between ranks ("rccl", "host callbacks", "none") and the number of ranks THAT layer
 * counts (RCCL: ncclCommCount of the model's communicator) */
int gcnhost_model_transport(gcnhost_model *m, int *ranks, char name[32]);
/* column-slice width (floats) the hidden-width aggregation was tuned to at load: 64 (two slices of a 128-wide row) or 32 */
int gcnhost_model_slice_floats(gcnhost_model *m, int *floats);
/* variable k of gcn.cpp:21-54 (1 H0, 2 W1, 3 H1, 4 Z0, 5 W2, 6 Z): this rank's rows, row-major rows x cols.
 * out == NULL: only report the shape. */
/* Variable k of the reference's list (gcn.cpp:21-54: 1 H0, 2 W1, 3 H1, 4 Z0, 5 W2, 6 Z), this rank's rows, row-major rows x cols
 * (call with out == NULL for the shape).  Variable 6 holds CURRENT logits only for the rows of the split scored by the last
 * forward (train_epoch: split 1; eval(s): split s) unless the model was built with GCNHOST flag ALL_ROWS (4096): the
 * default path does not compute rows nobody reads; the reference fills every row. */
int gcnhost_model_get_var(gcnhost_model *m, int k, int grad, float *out, int *rows, int *cols);
int gcnhost_model_set_weights(gcnhost_model *m, const float *w1, const float *w2);
/* device-event timer `id` (host/timer.h, ids of src/common/timer.h:5-20 plus 13 Adam, 14 comm,
 * 15 GraphSum at the hidden width): accumulated seconds and number of intervals */
int gcnhost_model_timer(gcnhost_model *m, int id, double *seconds, long *count);
int gcnhost_model_timers_reset(gcnhost_model *m);
/* switch the per-op timers on/off after construction (synchronises).  While on, run_epochs runs eagerly instead
 * of replaying its captured hipGraph: time the headline with them off, collect the breakdown in a separate pass */
int gcnhost_model_set_timers(gcnhost_model *m, int on);

/* the text loader (src/common/parser.cpp) — fills caller-visible arrays owned by the returned handle */
typedef struct gcnhost_dataset gcnhost_dataset;
int gcnhost_dataset_load(gcnhost_dataset **d, const char *root, const char *name, gcnhost_params *p);
int gcnhost_dataset_arrays(gcnhost_dataset *d, const int **g_indptr, const int **g_indices, int64_t *g_nnz,
                           const int **f_indptr, const int **f_indices, const float **f_val, int64_t *f_nnz,
                           const int **split, int64_t *n_split, const int **label, int64_t *n_label);
int gcnhost_dataset_save_binary(gcnhost_dataset *d, const gcnhost_params *p, const char *path);
int gcnhost_dataset_free(gcnhost_dataset *d);

/* one-rank RCCL round trip on `device` (communicator init, in-place all-gather, all-reduce,
 * destroy): checks that the RCCL this process loaded works before a multi-GPU job relies on it */
int gcnhost_rccl_selftest(int device);
/* the same with `world` ranks, one process per rank, all given the id rank 0 got from gcnhost_nccl_unique_id:
 * in-place all-gather of distinct blocks, all-reduce, the halo exchange (an ExchangePlan's send lists moved by grouped
 * ncclSend/ncclRecv, every table row checked), split communicator, alternating lanes — values checked */
int gcnhost_rccl_selftest_world(int device, int rank, int world, const char *nccl_id);
/* device time (microseconds, HIP events) per in-place all-gather of block_floats floats per rank and per all-reduce of
 * reduce_floats floats, back to back on the stream; with world == 1 the launch + kernel floor of a collective */
int gcnhost_rccl_collective_us(int device, int rank, int world, const char *nccl_id, long block_floats, long reduce_floats, int iters,
                               double *us_allgather, double *us_allreduce);
/* the halo round trip of that self-test through the host-staged transport (the callbacks of gcnhost_model_create) */
int gcnhost_halo_selftest_host(int device, int rank, int world, gcnhost_allgather_fn host_allgather,
                               gcnhost_allreduce_fn host_allreduce, void *host_user);

/* host-only helpers, callable without a GPU (CPU tests) */
int gcnhost_partition(const int *g_indptr, int n_rows, int world, int *start /* [world+1] */, int *rows_max);
/* rank's row block with columns rewritten to padded all-gather positions (what HipGCN feeds to
 * gcnhip_graph_create when world > 1).  Call once with NULL arrays for the sizes. */
int gcnhost_local_graph(const int *g_indptr, const int *g_indices, int n_rows, int world, int rank,
                        int *indptr, int *indices, int *col_deg, int *n_local, int *n_cols, int64_t *nnz_local);
/* How rank `rank` of `world` completes the tables an aggregation gathers from (host/partition.h): mode 0 decides
 * per graph between an all-gather of whole row blocks and a halo exchange of only the rows some local edge points
 * at (1 / 2 force one).  Gives the table layout, the per-peer send and receive lists and the rank's row block of the
 * adjacency with its columns rewritten to table rows — what HipGCN hands to gcnhip_graph_create.  Host only. */
typedef struct gcnhost_plan gcnhost_plan;
int gcnhost_plan_create(gcnhost_plan **p, const int *g_indptr, const int *g_indices, int n_rows, int world, int rank, int mode);
int gcnhost_plan_info(const gcnhost_plan *p, int *halo, int *n_local, int *table_rows, int *own_offset, int *rows_max,
                      double *halo_share, int64_t *nnz_local, int64_t *n_recv, int64_t *n_send);
/* recv_off/send_off: [world+1]; recv_rows: peer-local row ids per segment; send_rows: local row ids per destination;
 * table_global: [table_rows] global node id (-1 = padding); indptr/indices/col_deg: the local adjacency */
int gcnhost_plan_arrays(const gcnhost_plan *p, const int **recv_off, const int **recv_rows, const int **send_off, const int **send_rows,
                        const int **table_global, const int **indptr, const int **indices, const int **col_deg);
int gcnhost_plan_free(gcnhost_plan *p);
int gcnhost_glorot(float *w, int size, int in_size, int out_size, long seed, int skip_draws);
int gcnhost_host_masks(uint8_t *keep, int64_t n, float p, long seed, int64_t skip_draws);
/* Graph500 R-MAT graph (a,b,c = .57,.19,.19) of 2^scale nodes and edge_factor * 2^scale sampled pairs, symmetrised,
 * duplicates and self pairs dropped, in the layout the reference's loader produces (src/common/parser.cpp:20-46:
 * CSR with the self loop first, neighbours ascending).  BASELINE configs[4] = scale 22, edge_factor 16.  The arrays
 * are malloc'ed here; release them with gcnhost_free_array.  Deterministic in (scale, edge_factor, seed). */
int gcnhost_rmat_graph(int scale, int edge_factor, uint64_t seed, int **indptr, int **indices, int64_t *nnz);
void gcnhost_free_array(void *p);
/* The node order a `world`-rank model would renumber the dataset with (host/partition.h, choose_node_order): order[new] =
 * old (identity when the ids are kept), and the rows the neediest rank receives per exchange under the ids and under the
 * chosen order, next to what the padded all-gather moves.  Host only. */
int gcnhost_choose_node_order(const int *g_indptr, const int *g_indices, int n_rows, int world, int force, int *order, int *renumbered,
                              double *ids_share, int64_t *ids_recv_rows, double *new_share, int64_t *new_recv_rows, int64_t *allgather_rows);
/* Row groups for the aggregation's schedule found in the graph itself (Louvain local moving from singletons with a size
 * bound, asynchronous in a fixed node order, host/cluster.h): group[i] in 0 .. *n_groups-1, largest group first.  *useful == 0: the graph has no such
 * structure (everything collapsed into one group, or nothing merged) and HipGCN would not try the grouping.  What
 * HipGCN runs when the labels are not assortative on the graph (or GCNHOST_NO_LABEL_HINT is set); deterministic. */
int gcnhost_structure_groups(const int *g_indptr, const int *g_indices, int n_rows, int *group, int *n_groups, int *sweeps,
                             double *largest_share, int *useful);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
