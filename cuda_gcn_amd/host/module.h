// module.h — the op boundary of the reference, kept: an abstract Module with
// forward(bool training) / backward() whose operands are wired at construction
// as raw non-owning pointers (src/seq/module.h:6-76; CUDA twin
// src/cuda/cuda_module.cuh:11-84).  The Hip* modules below hold no kernels:
// each forward/backward is one or two calls into the C-ABI of libgcnhip.so.
//
// Two ways to build the model from them (gcn.cpp):
//  * modular: the reference's eight modules one for one (Dropout, SparseMatmul,
//    GraphSum, ReLU, Dropout, Matmul, GraphSum, CrossEntropyLoss);
//  * fused (default): the input Dropout folds into SparseMatmul, ReLU+Dropout
//    into GraphSum's store epilogue and into Matmul's backward epilogue.
#pragma once
#include <vector>
#include <cstdint>
#include "comm.h"
#include "gcnhip_driver.h"
#include "gcnhip_experimental.h"   // the opt-in packed-dH1 path (HIPGCN_PACKED_DH1, experiments build only)
#include "timer.h"
#include "variable.h"

// what every module needs from the model: context, collectives, RNG keys, the
// device epoch word (dropout stream + Adam step index) and the device timers
struct HipEnv {
    gcnhip_ctx *ctx = nullptr;
    Comm *comm = nullptr;
    const ExchangePlan *plan = nullptr;      // layout of gathered tables + how remote rows arrive (partition.h)
    ExchangeBuffers *xbuf = nullptr;         // this context's packing buffers for HALO plans
    DeviceTimers *timers = nullptr;
    uint64_t seed = 0;
    uint32_t *d_epoch = nullptr;
    uint32_t *d_epoch_done = nullptr;      // main context only: the epoch whose update ran last (written with d_epoch + 1 by the Adam launch)
    // parity mode: decisions generated on the host with the reference's RNG
    const uint8_t *keep_input = nullptr;     // [nnz of the X the forward multiplies: local rows, or all rows when replicated]
    const uint8_t *keep_input_bwd = nullptr; // [local nnz of X] (same decisions, this rank's slice)
    const uint8_t *keep_hidden = nullptr;    // [local rows * hidden]
    // opt-in: GraphSum gathers bfloat16 copies of its inputs (f32 accumulate); beyond the reference's f32 path
    bool bf16_tables = false;
    // HIPGCN_OVERLAP_EXCHANGE: exchanges run on their own stream (comm.h, ExchangeLane) while this stream works on the
    // edges that point at this rank's own rows
    ExchangeLane *xlane = nullptr;
    void *pos_bits_ready = nullptr;          // event of the in-flight exchange of the H1 > 0 bits (forward -> Matmul backward)
};

// The hidden layer's backward aggregation hands dH0 to the weight gradient dW1 = X~^T . dH0 in row blocks: the
// aggregation of block k+1 (gather-bound) runs beside the split-K product of block k (MFMA-bound) on a second stream.
// Same kernels on the same rows and the same split ranges as the one-stream order: not a bit changes.  OPT-IN
// (HIPGCN_BWD_PIPELINE): on one MI355X it measured slower than the one-stream order (gcn.cpp, build_modules).
struct BackwardPipeline {
    gcnhip_ctx *side = nullptr;              // the second stream (it owns the split-K slabs of the product)
    std::vector<gcnhip_rowset *> blocks;     // row blocks, registered on the hidden layer's adjacency object
    std::vector<int> cuts;                   // block k = splits [cuts[k], cuts[k+1]) of gcnhip_spmm_bwd_plan
    std::vector<void *> ev_block;            // main stream: block k of dH0 is complete
    void *ev_done = nullptr;                 // side stream: dW1 is complete
    bool armed = false;                      // the backward in progress went through the pipeline
};

class Module {
public:
    virtual void forward(bool) = 0;
    virtual void backward() = 0;
    virtual ~Module() {}
};

class HipMatmul : public Module {
    HipEnv *env;
    HipVariable *a, *b, *c;
    int m, n, p;
    float fused_bwd_scale;          // > 0: da = (a > 0) ? scale * da : 0 (ReLU+Dropout backward folded in)
public:
    // multi-GPU: instead of all-gathering da (n floats per row) the ranks all-gather dc (p floats per row) and
    // one bit per element of a > 0, and every rank rebuilds da for ALL rows (gcnhip_matmul_bwd_da_bits)
    const uint32_t *pos_bits_full = nullptr;    // [all_rows x wpr], gathered by the producer of `a`
    int wpr = 0, all_rows = 0;
    // single GPU, fused backward: da leaves as packed rows (gcnhip_matmul_bwd_packed); a->grad keeps only the halves
    // that do not fit a slot.  The consumer is the GraphSum that owns the same pack.
    gcnhip_rowpack *da_pack = nullptr;
    // single GPU, fused backward: the ReLU/dropout mask as one bit per element, left by the producer of `a`
    // (gcnhip_graphsum_relu_dropout_bits): da does not read `a` again
    const uint32_t *mask_bits = nullptr;
    int mask_wpr = 0;
    // factored aggregation (gcnhip_graphsum_ex): da leaves multiplied by dinv^2 of its row — da_row_scale for this rank's
    // rows, da_row_scale_full for the rows of the gathered table (rebuild_da)
    const float *da_row_scale = nullptr, *da_row_scale_full = nullptr;
    HipMatmul(HipEnv *env, HipVariable *a, HipVariable *b, HipVariable *c, int m, int n, int p, float fused_bwd_scale = 0.f);
    // Evaluation, round 5: c = ReLU(X . w1) . b in ONE launch with the producer of `a` (gcnhip_spmm_fwd_relu_matmul) — `a`, the
    // hidden matrix, is then never stored.  Returns false when that form is not available for these shapes / options (nothing
    // launched); on success this module's next forward() is a no-op.
    bool fused_eval_forward(gcnhip_feat *sp, const float *vals, HipVariable *w1, int p1);
private:
    bool skip_forward_once = false;
    void rebuild_da(int first_row, int n_rows);     // da rows [first_row, first_row + n_rows) of the table from dc + mask bits
public:
    void forward(bool) override;
    void backward() override;
};

class HipSparseMatmul : public Module {
    HipEnv *env;
    const float *const *vals;       // address of the pointer to the value array in use
    HipVariable *b, *c;
    gcnhip_feat *sp;
    int m, n, p;
    float fused_dropout;            // > 0: input dropout applied on the fly (training only)
    uint64_t nnz_offset;            // global index of this rank's first stored value
    bool last_training = false;
    bool fwd_decisions_valid = false;   // sp's keep-bit array holds the decisions of the last training forward
public:
    // replicated forward (multi-GPU): X of ALL rows, so c->full is computed here and never gathered
    gcnhip_feat *sp_full = nullptr;
    const float *const *vals_full = nullptr;
    bool relu_out = false;          // evaluation on A^.X: ReLU when the product is stored (forward(false) only)
    HipMatmul *fuse_next = nullptr; // relu_out: the Matmul that consumes `c` and nothing else does — both products in one launch
    bool hidden_not_stored = false; // the last forward was that fused launch: `c` does not hold this forward's hidden matrix
    void forward_stored();          // the evaluation forward as its own launch, `c` stored (what get_var(3) asks for after a fused one)
    BackwardPipeline *pipe = nullptr;       // set: the producer of c->grad may run backward_part/_finish block by block
    void backward_part(int block);
    void backward_finish();
    HipSparseMatmul(HipEnv *env, const float *const *vals, HipVariable *b, HipVariable *c, gcnhip_feat *sp,
                    int m, int n, int p, float fused_dropout, uint64_t nnz_offset);
    void forward(bool) override;
    void backward() override;
};

class HipCrossEntropyLoss;
class HipGraphSum : public Module {
    HipEnv *env;
    HipVariable *in, *out;
    gcnhip_graph *graph;
    int dim;
    float fused_relu_dropout;       // >= 0: ReLU (+ dropout with this p when training) in the store epilogue
    uint64_t elem_offset;           // global element index of this rank's first output element
public:
    HipCrossEntropyLoss *loss = nullptr;            // the loss module that reads `out`: its arithmetic rides in this launch's epilogue
    gcnhip_graph *fwd_graph_replicated = nullptr;   // global column ids: forward reads a replicated `in` without a gather
    // multi-GPU, first layer: after a training forward publish bit = (out > 0) of this rank's rows to every rank;
    // in exchange backward() finds out->full_grad already complete (rebuilt locally) and gathers nothing
    uint32_t *pos_bits_full = nullptr;
    int wpr = 0;
    bool out_grad_complete = false;
    // single GPU: the same bits for this GPU's own backward (HipMatmul::mask_bits), written by the store epilogue
    uint32_t *mask_bits_out = nullptr;
    // rows of out->grad known to be zero (bit = 0) are not gathered in backward(); NULL: none known
    const uint32_t *const *bwd_row_bits = nullptr;
    const gcnhip_graph *bwd_graph = nullptr;              // the operator without the edges that point at known-zero rows of out->grad
                                                          // (gcnhip_graph_create_restricted); replaces the row mask when set
    // rows of `out` that the consumer reads in forward() (bit = 1); the others are not computed.  NULL: all rows.
    // The last aggregation sets it: loss and accuracy read only rows of the scored split (module.cpp:131-133).
    gcnhip_rowset *const *fwd_out_rows = nullptr;
    // backward(): out->grad arrives as packed rows (written by HipMatmul::backward into the same pack)
    gcnhip_rowpack *out_grad_pack = nullptr;              // a subset registered on `graph` (gcnhip_graph_add_rowset)
    // HIPGCN_OVERLAP_EXCHANGE: `graph` cut in two by the owner of the column (gcnhip_graph_create_restricted, twice): the
    // edges that point at this rank's own rows, and the rest.  forward()/backward() then start the exchange on the
    // exchange stream, aggregate through `loc` meanwhile, wait, and add the terms of `rem` (gcnhip_graphsum_part).
    // bwd_split_*: the same cut of bwd_graph.  The row subsets of the last aggregation exist per operator.
    const gcnhip_graph *split_loc = nullptr, *split_rem = nullptr;
    const gcnhip_graph *bwd_split_loc = nullptr, *bwd_split_rem = nullptr;
    gcnhip_rowset *const *fwd_out_rows_loc = nullptr, *const *fwd_out_rows_rem = nullptr;
    // backward(): in->grad is produced block by block and handed to this consumer's weight gradient (BackwardPipeline)
    BackwardPipeline *pipe = nullptr;
    HipSparseMatmul *pipe_consumer = nullptr;
    // gcnhip_graphsum_ex scaling of the forward / backward aggregation: 0 = the reference's per-edge coefficients; factored
    // model: hidden layer forward 2 (result x dinv^2: already scaled for the class-width aggregation), class layer forward 1
    // (true logits), every backward 3 (the consumer's operand carries the factor)
    int fwd_scaling = 0, bwd_scaling = 0;
    HipGraphSum(HipEnv *env, HipVariable *in, HipVariable *out, gcnhip_graph *graph, int dim,
                float fused_relu_dropout = -1.f, uint64_t elem_offset = 0);
    ~HipGraphSum() override;
    void forward(bool) override;
    void backward() override;
private:
    // bf16 storage mode: one table [padded rows x ld_bf], used by forward (copy of `in`) and backward (copy of out->grad)
    uint16_t *bf_table = nullptr;
    int ld_bf = 0;
    size_t bf_rows = 0;
    uint16_t *table();
    static size_t full_rows(const HipVariable *v, bool grad);
};

class HipCrossEntropyLoss : public Module {
    HipEnv *env;
    HipVariable *logits;
    int32_t *const *truth;          // address of the current truth pointer (set_truth switches splits)
    const int *count;               // labelled rows of the current split, all ranks
    float *d_result;                // {loss_sum, count, correct, total}
    int32_t *d_result_i;
    int num_classes;
    bool shift_in_place;
public:
    // rows of the current split on this rank (ascending) and their number: only they are visited (the other rows'
    // gradients are zero from allocation on).  NULL: every row is visited, as the reference does.
    int32_t *const *rows_list = nullptr;
    const int *rows_n = nullptr;
    const float *grad_row_scale = nullptr;      // factored aggregation: the gradient rows leave multiplied by dinv of their row
    // Loss epilogue (round 5, gcnhip_gs_loss): the aggregation that produces `logits` computes each scored row's loss term,
    // accuracy flag and gradient row while the row is still in its wave's registers (HipGraphSum::loss points here and
    // fills epilogue_opts()); forward() then only adds the terms — same bits as the loss kernel on the stored logits.
    // row_terms: [2 x rows] floats owned by this module (NULL: the loss kernel reads the logits, as before).
    float *row_terms = nullptr;
    bool terms_fresh = false;                   // the producing launch of this forward wrote row_terms
    bool epilogue_opts(bool training, gcnhip_gs_loss *o) const;   // false: not applicable (no row list, empty split)
    ~HipCrossEntropyLoss() override;
    HipCrossEntropyLoss(HipEnv *env, HipVariable *logits, int32_t *const *truth, const int *count,
                        float *d_result, int32_t *d_result_i, int num_classes, bool shift_in_place);
    void forward(bool) override;
    void backward() override {}
};

class HipReLU : public Module {
    HipEnv *env;
    HipVariable *in;
    uint8_t *mask;
public:
    HipReLU(HipEnv *env, HipVariable *in);
    ~HipReLU() override;
    void forward(bool) override;
    void backward() override;
};

class HipDropout : public Module {
    HipEnv *env;
    HipVariable *in;
    int32_t *mask;
    float p;
    uint64_t key_tweak, elem_offset;
    const uint8_t *const *keep_in;
public:
    HipDropout(HipEnv *env, HipVariable *in, float p, uint64_t key_tweak, uint64_t elem_offset, const uint8_t *const *keep_in);
    ~HipDropout() override;
    void forward(bool) override;
    void backward() override;
};

// distinct Philox keys for the two dropout sites
constexpr uint64_t KEY_INPUT_DROPOUT = 0x9E3779B97F4A7C15ull;
constexpr uint64_t KEY_HIDDEN_DROPOUT = 0xD1B54A32D192ED03ull;
