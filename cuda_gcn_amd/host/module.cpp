#include "module.h"
#include <algorithm>
#include "hip_check.h"

// ------------------------------------------------------------------- Matmul
HipMatmul::HipMatmul(HipEnv *env, HipVariable *a, HipVariable *b, HipVariable *c, int m, int n, int p, float s)
    : env(env), a(a), b(b), c(c), m(m), n(n), p(p), fused_bwd_scale(s) {}

bool HipMatmul::fused_eval_forward(gcnhip_feat *sp, const float *vals, HipVariable *w1, int p1) {
    if (p1 != n) return false;
    const int rc = gcnhip_spmm_fwd_relu_matmul(env->ctx, sp, vals, w1->data, w1->ld, p1, b->data, b->ld, p, c->data, c->ld);
    if (rc == GCNHIP_NOT_AVAILABLE) return false;
    GCNHIP_CHECK(rc);
    skip_forward_once = true;
    return true;
}

void HipMatmul::forward(bool) {
    if (skip_forward_once) { skip_forward_once = false; return; }     // the producer's launch has written c already
    env->timers->start(TMR_MATMUL_FW);
    GCNHIP_CHECK(gcnhip_matmul_fwd(env->ctx, a->data, a->ld, b->data, b->ld, c->data, c->ld, m, n, p));
    env->timers->stop(TMR_MATMUL_FW);
}

void HipMatmul::backward() {
    env->timers->start(TMR_MATMUL_BW);
    if (pos_bits_full && env->xlane) {
        // the exchange of dc runs on the exchange stream beside everything that needs only this rank's rows: db, and da
        // of the own block; da of the other ranks' rows follows when their dc (and their mask bits, in flight since the
        // forward) have arrived
        void *ev = env->xlane->begin(*env->plan, c->full_grad, c->ld);
        GCNHIP_CHECK(gcnhip_matmul_bwd(env->ctx, a->data, a->ld, b->data, b->ld, c->grad, c->ld,
                                       nullptr, a->ld, b->grad, b->ld, m, n, p));
        const int own = env->plan->own_offset;
        rebuild_da(own, m);
        env->xlane->wait(ev);
        if (env->pos_bits_ready) { env->xlane->wait(env->pos_bits_ready); env->pos_bits_ready = nullptr; }
        rebuild_da(0, own);
        rebuild_da(own + m, all_rows - own - m);
    } else if (pos_bits_full) {
        // db from this rank's rows; da for every row of every rank from the gathered dc and mask bits
        GCNHIP_CHECK(gcnhip_matmul_bwd(env->ctx, a->data, a->ld, b->data, b->ld, c->grad, c->ld,
                                       nullptr, a->ld, b->grad, b->ld, m, n, p));
        env->timers->stop(TMR_MATMUL_BW);
        env->timers->start(TMR_COMM);
        env->comm->exchange_rows(*env->plan, *env->xbuf, c->full_grad, c->ld);
        env->timers->stop(TMR_COMM);
        env->timers->start(TMR_MATMUL_BW);
        rebuild_da(0, all_rows);
    } else if (fused_bwd_scale > 0.f && da_pack)
        GCNHIP_CHECK(gcnhip_matmul_bwd_packed(env->ctx, a->data, a->ld, b->data, b->ld, c->grad, c->ld,
                                              a->grad, a->ld, da_pack, b->grad, b->ld, m, n, p, fused_bwd_scale));
    else if (fused_bwd_scale > 0.f)             // mask from the bits the aggregation left, else from a > 0; da rows x dinv^2 when factored
        GCNHIP_CHECK(gcnhip_matmul_bwd_ex(env->ctx, a->data, a->ld, b->data, b->ld, c->grad, c->ld, a->grad, a->ld, b->grad, b->ld, m, n, p,
                                          fused_bwd_scale, mask_bits, mask_bits ? mask_wpr : 0, da_row_scale));
    else
        GCNHIP_CHECK(gcnhip_matmul_bwd(env->ctx, a->data, a->ld, b->data, b->ld, c->grad, c->ld,
                                       a->grad, a->ld, b->grad, b->ld, m, n, p));
    env->timers->stop(TMR_MATMUL_BW);
}

void HipMatmul::rebuild_da(int first, int rows) {
    if (rows <= 0) return;
    GCNHIP_CHECK(gcnhip_matmul_bwd_ex(env->ctx, nullptr, 0, b->data, b->ld, c->full_grad + (size_t)first * c->ld, c->ld,
                                      a->full_grad + (size_t)first * a->ld, a->ld, nullptr, 0, rows, n, p, fused_bwd_scale,
                                      pos_bits_full + (size_t)first * wpr, wpr, da_row_scale_full ? da_row_scale_full + first : nullptr));
}

// ------------------------------------------------------------- SparseMatmul
HipSparseMatmul::HipSparseMatmul(HipEnv *env, const float *const *vals, HipVariable *b, HipVariable *c, gcnhip_feat *sp,
                                 int m, int n, int p, float fd, uint64_t off)
    : env(env), vals(vals), b(b), c(c), sp(sp), m(m), n(n), p(p), fused_dropout(fd), nnz_offset(off) {}

void HipSparseMatmul::forward(bool training) {
    last_training = training;
    // a training forward through `sp` itself (not the replicated all-rows object) leaves this epoch's dropout decisions
    // in sp's bit array; any other forward through sp (evaluation: no dropout) leaves them untouched
    if (training) fwd_decisions_valid = fused_dropout > 0.f && !sp_full && !relu_out && gcnhip_feat_is_dense(sp);
    env->timers->start(TMR_SPMATMUL_FW);
    const float pd = training ? fused_dropout : 0.f;
    hidden_not_stored = false;
    if (relu_out) {
        if (training) throw GcnHipFailure(-1, "HipSparseMatmul: the ReLU epilogue is an evaluation-only form");
        if (fuse_next && fuse_next->fused_eval_forward(sp, *vals, b, p)) hidden_not_stored = true;
        else GCNHIP_CHECK(gcnhip_spmm_fwd_relu(env->ctx, sp, *vals, b->data, b->ld, c->data, c->ld, p));
    } else if (sp_full)        // every row of the product, with global element indices for the dropout stream
        GCNHIP_CHECK(gcnhip_spmm_fwd(env->ctx, sp_full, *vals_full, b->data, b->ld, c->full, c->ld, p, pd,
                                     env->seed ^ KEY_INPUT_DROPOUT, env->d_epoch, 0,
                                     pd > 0.f ? env->keep_input : nullptr));
    else
        GCNHIP_CHECK(gcnhip_spmm_fwd(env->ctx, sp, *vals, b->data, b->ld, c->data, c->ld, p, pd,
                                     env->seed ^ KEY_INPUT_DROPOUT, env->d_epoch, nnz_offset,
                                     pd > 0.f ? env->keep_input : nullptr));
    env->timers->stop(TMR_SPMATMUL_FW);
}

void HipSparseMatmul::forward_stored() {
    if (!relu_out) throw GcnHipFailure(-1, "HipSparseMatmul::forward_stored: an evaluation-only form");
    GCNHIP_CHECK(gcnhip_spmm_fwd_relu(env->ctx, sp, *vals, b->data, b->ld, c->data, c->ld, p));
    hidden_not_stored = false;
}

void HipSparseMatmul::backward_part(int k) {
    const float pd = last_training ? fused_dropout : 0.f;
    GCNHIP_CHECK(gcnhip_spmm_bwd_part(pipe->side, sp, *vals, c->grad, c->ld, p, pd, env->seed ^ KEY_INPUT_DROPOUT, env->d_epoch,
                                      nnz_offset, pd > 0.f ? env->keep_input_bwd : nullptr, pipe->cuts[k], pipe->cuts[k + 1], k == 0));
}

void HipSparseMatmul::backward_finish() {
    GCNHIP_CHECK(gcnhip_spmm_bwd_finish(pipe->side, sp, b->grad, b->ld, p));
    GCNHIP_CHECK(gcnhip_event_record(pipe->side, pipe->ev_done));
}

void HipSparseMatmul::backward() {
    if (pipe && pipe->armed) {                  // the producer of c->grad has already run this product on the second stream
        GCNHIP_CHECK(gcnhip_stream_wait_event(env->ctx, pipe->ev_done));
        pipe->armed = false;
        return;
    }
    env->timers->start(TMR_SPMATMUL_BW);
    const float pd = last_training ? fused_dropout : 0.f;     // the same X~ the forward saw (module.cpp:72)
    int rps = 0, n_splits = 0;
    GCNHIP_CHECK(gcnhip_spmm_bwd_plan(env->ctx, sp, p, &rps, &n_splits));
    if (n_splits > 0 && pd > 0.f && fwd_decisions_valid) {
        // the dropout decisions of this epoch's X~ are still in the object's bit array: this rank's forward wrote them
        // (same object, same seed / epoch word / offset) and nothing has redrawn them since — no second generation pass
        GCNHIP_CHECK(gcnhip_spmm_bwd_part(env->ctx, sp, *vals, c->grad, c->ld, p, pd, env->seed ^ KEY_INPUT_DROPOUT, env->d_epoch,
                                          nnz_offset, env->keep_input_bwd, 0, n_splits, 0));
        GCNHIP_CHECK(gcnhip_spmm_bwd_finish(env->ctx, sp, b->grad, b->ld, p));
    } else {
        GCNHIP_CHECK(gcnhip_spmm_bwd(env->ctx, sp, *vals, c->grad, c->ld, b->grad, b->ld, p, pd,
                                     env->seed ^ KEY_INPUT_DROPOUT, env->d_epoch, nnz_offset,
                                     pd > 0.f ? env->keep_input_bwd : nullptr));
    }
    env->timers->stop(TMR_SPMATMUL_BW);
}

// ----------------------------------------------------------------- GraphSum
HipGraphSum::HipGraphSum(HipEnv *env, HipVariable *in, HipVariable *out, gcnhip_graph *graph, int dim, float frd, uint64_t off)
    : env(env), in(in), out(out), graph(graph), dim(dim), fused_relu_dropout(frd), elem_offset(off) {}

HipGraphSum::~HipGraphSum() { if (bf_table) gcnhip_free(env->ctx, bf_table); }

size_t HipGraphSum::full_rows(const HipVariable *v, bool grad) {
    const float *full = grad ? v->full_grad : v->full;
    return full ? v->full_elems / v->ld : (size_t)v->rows;
}

// bf16 rows: one 128-byte line for up to 64 columns, whole lines above (d = 128: 2 lines instead of 4;
// 41 classes: 1 line instead of 2)
uint16_t *HipGraphSum::table() {
    if (!bf_table) {
        ld_bf = dim <= 8 ? 8 : (dim <= 16 ? 16 : (dim <= 32 ? 32 : (dim + 63) / 64 * 64));
        bf_rows = std::max(full_rows(in, false), full_rows(out, true));
        void *p;
        GCNHIP_CHECK(gcnhip_malloc(env->ctx, &p, bf_rows * ld_bf * sizeof(uint16_t)));
        GCNHIP_CHECK(gcnhip_memset_async(env->ctx, p, 0, bf_rows * ld_bf * sizeof(uint16_t)));
        bf_table = (uint16_t *)p;
    }
    return bf_table;
}

void HipGraphSum::forward(bool training) {
    // rows of `in` named by this rank's columns live on other ranks: gather them first — unless every
    // rank computed all of `in` itself (replicated first-layer product)
    gcnhip_graph *graph = this->graph;
    const int world = env->comm->size();
    const gcnhip_rowset *out_rows = fwd_out_rows ? *fwd_out_rows : nullptr;
    const bool replicated = in->replicated && fwd_graph_replicated;
    if (replicated) graph = fwd_graph_replicated;
    bool bits_written = false;
    if (env->bf16_tables) {
        // the table travels (and is gathered) as bfloat16: half the lines per edge, half the bytes per all-gather
        uint16_t *tab = table();
        env->timers->start(TMR_GRAPHSUM_FW);
        if (replicated) {
            GCNHIP_CHECK(gcnhip_f32_to_bf16(env->ctx, in->full, in->ld, tab, ld_bf, (int64_t)full_rows(in, false), dim));
        } else {
            const size_t own = world > 1 ? (size_t)env->plan->own_offset : 0;
            GCNHIP_CHECK(gcnhip_f32_to_bf16(env->ctx, in->data, in->ld, tab + own * ld_bf, ld_bf, in->rows, dim));
            if (world > 1) {
                env->timers->start(TMR_COMM);
                env->comm->exchange_rows(*env->plan, *env->xbuf, reinterpret_cast<float *>(tab), ld_bf / 2);   // bytes are moved, not interpreted
                env->timers->stop(TMR_COMM);
            }
        }
        if (dim > 64) env->timers->start(TMR_GRAPHSUM_WIDE);
        const bool fused = fused_relu_dropout >= 0.f;
        GCNHIP_CHECK(gcnhip_graphsum_bf16(env->ctx, graph, tab, ld_bf, out->data, out->ld, dim, nullptr, out_rows, fused ? 1 : 0, training ? 1 : 0,
                                          fused ? fused_relu_dropout : 0.f, env->seed ^ KEY_HIDDEN_DROPOUT, env->d_epoch, elem_offset,
                                          training ? env->keep_hidden : nullptr));
        if (dim > 64) env->timers->stop(TMR_GRAPHSUM_WIDE);
        env->timers->stop(TMR_GRAPHSUM_FW);
    } else if (env->xlane && !replicated && world > 1) {
        // exchange on its own stream; meanwhile the edges that point at this rank's own rows, then the others on top.
        // (A rank that owns no rows has no operators to cut — split_loc is NULL — but takes part in the exchange all the same:
        // it runs on the lane's communicator on EVERY rank.)
        void *ev = env->xlane->begin(*env->plan, in->full, in->ld);
        const gcnhip_rowset *rows_loc = fwd_out_rows_loc ? *fwd_out_rows_loc : nullptr, *rows_rem = fwd_out_rows_rem ? *fwd_out_rows_rem : nullptr;
        const bool fused = fused_relu_dropout >= 0.f;
        env->timers->start(TMR_GRAPHSUM_FW);
        if (dim > 64) env->timers->start(TMR_GRAPHSUM_WIDE);
        if (split_loc) {
            gcnhip_gs_opts o1 = {};
            o1.rows = rows_loc; o1.scaling = fwd_scaling ? 3 : 0;     // the first part leaves the raw sum
            GCNHIP_CHECK(gcnhip_graphsum_ex(env->ctx, split_loc, &o1, in->full, in->ld, out->data, out->ld, dim));
        }
        env->xlane->wait(ev);
        if (split_loc) {
            gcnhip_gs_opts o2 = {};
            o2.rows = rows_rem; o2.accumulate = 1; o2.scaling = fwd_scaling;
            o2.relu_dropout = fused ? 1 : 0; o2.training = training ? 1 : 0; o2.p = fused ? fused_relu_dropout : 0.f;
            o2.seed = env->seed ^ KEY_HIDDEN_DROPOUT; o2.d_epoch = env->d_epoch; o2.elem_offset = elem_offset;
            o2.keep_mask = training ? env->keep_hidden : nullptr;
            GCNHIP_CHECK(gcnhip_graphsum_ex(env->ctx, split_rem, &o2, in->full, in->ld, out->data, out->ld, dim));
        }
        if (dim > 64) env->timers->stop(TMR_GRAPHSUM_WIDE);
        env->timers->stop(TMR_GRAPHSUM_FW);
    } else {
        if (!replicated && world > 1) {
            env->timers->start(TMR_COMM);
            env->comm->exchange_rows(*env->plan, *env->xbuf, in->full, in->ld);
            env->timers->stop(TMR_COMM);
        }
        const float *src = in->full ? in->full : in->data;
        env->timers->start(TMR_GRAPHSUM_FW);
        if (dim > 64) env->timers->start(TMR_GRAPHSUM_WIDE);
        // the mask of this layer's backward as bits, straight from the store epilogue: this GPU's own (single GPU), or this
        // rank's block of the table every rank completes (several GPUs; saves the gcnhip_pack_positive pass below)
        uint32_t *bits_here = nullptr;
        if (training && fused_relu_dropout >= 0.f && dim % 32 == 0)
            bits_here = mask_bits_out ? mask_bits_out : (pos_bits_full ? pos_bits_full + (size_t)env->plan->own_offset * wpr : nullptr);
        {
            gcnhip_gs_opts o = {};
            o.scaling = fwd_scaling;
            if (fused_relu_dropout >= 0.f) {
                o.relu_dropout = 1; o.training = training ? 1 : 0; o.p = fused_relu_dropout;
                o.seed = env->seed ^ KEY_HIDDEN_DROPOUT; o.d_epoch = env->d_epoch; o.elem_offset = elem_offset;
                o.keep_mask = training ? env->keep_hidden : nullptr;
                if (bits_here) { o.pos_bits = bits_here; o.words_per_row = mask_bits_out ? (dim + 31) / 32 : wpr; bits_written = true; }
            } else {
                o.rows = out_rows;
            }
            gcnhip_gs_loss lo = {};
            if (loss && fused_relu_dropout < 0.f && loss->epilogue_opts(training, &lo)) { o.loss = &lo; loss->terms_fresh = true; }
            GCNHIP_CHECK(gcnhip_graphsum_ex(env->ctx, graph, &o, src, in->ld, out->data, out->ld, dim));
        }
        if (dim > 64) env->timers->stop(TMR_GRAPHSUM_WIDE);
        env->timers->stop(TMR_GRAPHSUM_FW);
    }
    if (pos_bits_full && training) {
        uint32_t *mine = pos_bits_full + (size_t)env->plan->own_offset * wpr;
        if (!bits_written) GCNHIP_CHECK(gcnhip_pack_positive(env->ctx, out->data, out->ld, out->rows, dim, mine, wpr));
        if (env->xlane) {
            // nobody reads the other ranks' bits before the Matmul backward: the exchange stream delivers them meanwhile
            if (env->pos_bits_ready) env->xlane->wait(env->pos_bits_ready);        // (a forward whose backward never ran)
            env->pos_bits_ready = env->xlane->begin(*env->plan, reinterpret_cast<float *>(pos_bits_full), wpr);
        } else {
            env->timers->start(TMR_COMM);
            env->comm->exchange_rows(*env->plan, *env->xbuf, reinterpret_cast<float *>(pos_bits_full), wpr);   // bytes are moved, not interpreted
            env->timers->stop(TMR_COMM);
        }
    }
}

void HipGraphSum::backward() {
    // same operator on the gradients (symmetric adjacency, module.cpp:103-119); out->grad is gathered,
    // unless every rank has already rebuilt all of it
    const int world = env->comm->size();
    const uint32_t *row_bits = bwd_row_bits && !bwd_graph ? *bwd_row_bits : nullptr;
    const gcnhip_graph *graph = bwd_graph ? bwd_graph : this->graph;
    if (env->bf16_tables) {
        uint16_t *tab = table();
        env->timers->start(TMR_GRAPHSUM_BW);
        if (world > 1 && out_grad_complete) {
            GCNHIP_CHECK(gcnhip_f32_to_bf16(env->ctx, out->full_grad, out->ld, tab, ld_bf, (int64_t)full_rows(out, true), dim));
        } else {
            const size_t own = world > 1 ? (size_t)env->plan->own_offset : 0;
            GCNHIP_CHECK(gcnhip_f32_to_bf16(env->ctx, out->grad, out->ld, tab + own * ld_bf, ld_bf, out->rows, dim));
            if (world > 1) {
                env->timers->start(TMR_COMM);
                env->comm->exchange_rows(*env->plan, *env->xbuf, reinterpret_cast<float *>(tab), ld_bf / 2);
                env->timers->stop(TMR_COMM);
            }
        }
        if (dim > 64) env->timers->start(TMR_GRAPHSUM_WIDE);
        GCNHIP_CHECK(gcnhip_graphsum_bf16(env->ctx, graph, tab, ld_bf, in->grad, in->ld, dim, row_bits, nullptr, 0, 0, 0.f, 0, nullptr, 0, nullptr));
        if (dim > 64) env->timers->stop(TMR_GRAPHSUM_WIDE);
        env->timers->stop(TMR_GRAPHSUM_BW);
        return;
    }
    if (world > 1 && !out_grad_complete && env->xlane) {
        const gcnhip_graph *loc = bwd_split_loc ? bwd_split_loc : split_loc, *rem = bwd_split_rem ? bwd_split_rem : split_rem;
        const uint32_t *bits = bwd_split_loc ? nullptr : row_bits;      // the restricted operators have lost the known-zero rows already
        void *ev = env->xlane->begin(*env->plan, out->full_grad, out->ld);
        env->timers->start(TMR_GRAPHSUM_BW);
        if (dim > 64) env->timers->start(TMR_GRAPHSUM_WIDE);
        gcnhip_gs_opts o = {};
        o.in_row_bits = bits; o.scaling = bwd_scaling;
        if (loc) GCNHIP_CHECK(gcnhip_graphsum_ex(env->ctx, loc, &o, out->full_grad, out->ld, in->grad, in->ld, dim));
        env->xlane->wait(ev);
        o.accumulate = 1;
        if (rem) GCNHIP_CHECK(gcnhip_graphsum_ex(env->ctx, rem, &o, out->full_grad, out->ld, in->grad, in->ld, dim));
        if (dim > 64) env->timers->stop(TMR_GRAPHSUM_WIDE);
        env->timers->stop(TMR_GRAPHSUM_BW);
        return;
    }
    if (world > 1 && !out_grad_complete) {
        env->timers->start(TMR_COMM);
        env->comm->exchange_rows(*env->plan, *env->xbuf, out->full_grad, out->ld);
        env->timers->stop(TMR_COMM);
    }
    const float *src = out->full_grad ? out->full_grad : out->grad;
    env->timers->start(TMR_GRAPHSUM_BW);
    if (dim > 64) env->timers->start(TMR_GRAPHSUM_WIDE);
    if (pipe && pipe_consumer && !env->timers->enabled && !out_grad_pack && !row_bits && !bwd_graph) {
        // (with per-op timers on, every launch runs alone on the main stream: the branches below)
        const size_t nb = pipe->blocks.size();
        for (size_t k = 0; k < nb; k++) {
            gcnhip_gs_opts ob = {};
            ob.rows = pipe->blocks[k]; ob.scaling = bwd_scaling;
            GCNHIP_CHECK(gcnhip_graphsum_ex(env->ctx, graph, &ob, src, out->ld, in->grad, in->ld, dim));
            GCNHIP_CHECK(gcnhip_event_record(env->ctx, pipe->ev_block[k]));
            GCNHIP_CHECK(gcnhip_stream_wait_event(pipe->side, pipe->ev_block[k]));
            pipe_consumer->backward_part((int)k);
        }
        pipe_consumer->backward_finish();
        pipe->armed = true;
    } else if (out_grad_pack)
        GCNHIP_CHECK(gcnhip_graphsum_packed(env->ctx, graph, out_grad_pack, src, out->ld, in->grad, in->ld));
    else {
        gcnhip_gs_opts o = {};
        o.in_row_bits = row_bits; o.scaling = bwd_scaling;
        GCNHIP_CHECK(gcnhip_graphsum_ex(env->ctx, graph, &o, src, out->ld, in->grad, in->ld, dim));
    }
    if (dim > 64) env->timers->stop(TMR_GRAPHSUM_WIDE);
    env->timers->stop(TMR_GRAPHSUM_BW);
}

// --------------------------------------------------------- CrossEntropyLoss
HipCrossEntropyLoss::HipCrossEntropyLoss(HipEnv *env, HipVariable *logits, int32_t *const *truth, const int *count,
                                         float *d_result, int32_t *d_result_i, int num_classes, bool shift)
    : env(env), logits(logits), truth(truth), count(count), d_result(d_result), d_result_i(d_result_i),
      num_classes(num_classes), shift_in_place(shift) {}

HipCrossEntropyLoss::~HipCrossEntropyLoss() { if (row_terms) gcnhip_free(env->ctx, row_terms); }

bool HipCrossEntropyLoss::epilogue_opts(bool training, gcnhip_gs_loss *o) const {
    if (!row_terms || !rows_list || !*rows_list || *count <= 0 || num_classes > 64 || shift_in_place) return false;
    o->truth = *truth; o->grad = logits->grad; o->ld_grad = logits->ld; o->training = training ? 1 : 0; o->count = *count;
    o->grad_row_scale = grad_row_scale; o->row_terms = row_terms;
    return true;
}

void HipCrossEntropyLoss::forward(bool training) {
    env->timers->start(TMR_LOSS_FW);
    if (terms_fresh) {
        terms_fresh = false;
        GCNHIP_CHECK(gcnhip_xent_from_row_terms(env->ctx, row_terms, *truth, *rows_list, *rows_n, d_result, d_result_i));
    } else if (rows_list && *rows_list && *count > 0)
        GCNHIP_CHECK(gcnhip_xent_fwd_rows_scaled(env->ctx, logits->data, logits->ld, logits->grad, logits->ld, *truth, *rows_list, *rows_n,
                                                 num_classes, training ? 1 : 0, *count, shift_in_place ? 1 : 0, d_result, d_result_i,
                                                 grad_row_scale));
    else
    GCNHIP_CHECK(gcnhip_xent_fwd(env->ctx, logits->data, logits->ld, logits->grad, logits->ld, *truth, logits->rows,
                                 num_classes, training ? 1 : 0, *count, shift_in_place ? 1 : 0, d_result, d_result_i));
    env->timers->stop(TMR_LOSS_FW);
}

// --------------------------------------------------------------------- ReLU
HipReLU::HipReLU(HipEnv *env, HipVariable *in) : env(env), in(in), mask(nullptr) {
    void *p;
    GCNHIP_CHECK(gcnhip_malloc(env->ctx, &p, (size_t)in->rows * in->cols));
    mask = (uint8_t *)p;
}
HipReLU::~HipReLU() { gcnhip_free(env->ctx, mask); }
void HipReLU::forward(bool training) {
    env->timers->start(TMR_RELU_FW);
    GCNHIP_CHECK(gcnhip_relu_fwd_2d(env->ctx, in->data, in->ld, in->rows, in->cols, mask, training ? 1 : 0));
    env->timers->stop(TMR_RELU_FW);
}
void HipReLU::backward() {
    env->timers->start(TMR_RELU_BW);
    GCNHIP_CHECK(gcnhip_relu_bwd_2d(env->ctx, in->grad, in->ld, in->rows, in->cols, mask));
    env->timers->stop(TMR_RELU_BW);
}

// ------------------------------------------------------------------ Dropout
HipDropout::HipDropout(HipEnv *env, HipVariable *in, float p, uint64_t key_tweak, uint64_t elem_offset, const uint8_t *const *keep_in)
    : env(env), in(in), mask(nullptr), p(p), key_tweak(key_tweak), elem_offset(elem_offset), keep_in(keep_in) {
    if (in->grad) {                                 // module.cpp:199: a mask only when the input has a gradient
        void *q;
        GCNHIP_CHECK(gcnhip_malloc(env->ctx, &q, (size_t)in->rows * in->cols * sizeof(int32_t)));
        mask = (int32_t *)q;
    }
}
HipDropout::~HipDropout() { if (mask) gcnhip_free(env->ctx, mask); }
void HipDropout::forward(bool training) {
    if (!training) return;                          // module.cpp:208
    env->timers->start(TMR_DROPOUT_FW);
    // element (r, c) of the padded layout is the reference's flat element r*cols + c (RNG stream, masks)
    GCNHIP_CHECK(gcnhip_dropout_fwd_2d(env->ctx, in->data, in->ld, in->rows, in->cols, mask, p, env->seed ^ key_tweak,
                                       env->d_epoch, elem_offset, *keep_in));
    env->timers->stop(TMR_DROPOUT_FW);
}
void HipDropout::backward() {
    if (!mask) return;                              // module.cpp:224
    env->timers->start(TMR_DROPOUT_BW);
    GCNHIP_CHECK(gcnhip_dropout_bwd_2d(env->ctx, in->grad, in->ld, in->rows, in->cols, mask, p));
    env->timers->stop(TMR_DROPOUT_BW);
}
