#include "gcn.h"
#include "cluster.h"
#include <chrono>
#include <deque>
#include <future>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include "hip_check.h"

GCNParams GCNParams::get_default() { return {2708, 1433, 16, 7, 0.5, 0.01, 5e-4, 100, 0}; }

namespace {
template <class T>
T *dev_upload(gcnhip_ctx *ctx, const T *h, size_t n) {
    void *p;
    GCNHIP_CHECK(gcnhip_malloc(ctx, &p, (n ? n : 1) * sizeof(T)));
    if (n) GCNHIP_CHECK(gcnhip_h2d(ctx, p, h, n * sizeof(T)));
    return (T *)p;
}

// Are the labels communities of THIS graph?  Edge homophily (share of stored non-loop edges whose two ends carry
// the same label) against what label frequencies alone would give; the hint is used at twice chance or more.
bool labels_are_assortative(const GCNData &d, int N, int C) {
    if ((int)d.label.size() != N || C <= 1) return false;
    const std::vector<int> &gp = d.graph.indptr, &gi = d.graph.indices;
    std::vector<double> freq(C, 0.0);
    for (int i = 0; i < N; i++) {
        if (d.label[i] < 0 || d.label[i] >= C) return false;
        freq[d.label[i]] += 1.0;
    }
    double chance = 0;
    for (int c = 0; c < C; c++) chance += (freq[c] / N) * (freq[c] / N);
    long same = 0, total = 0;
    for (int i = 0; i < N; i++)
        for (int e = gp[i]; e < gp[i + 1]; e++) {
            if (gi[e] == i) continue;
            total++;
            same += d.label[gi[e]] == d.label[i];
        }
    return total > 0 && (double)same / (double)total >= 2.0 * chance;
}
}  // namespace

HipGCN::HipGCN(GCNParams p, GCNData *input_data, const HipGCNOptions &opt) : params(p), data(input_data), flags(opt.flags) {
    // a constructor that throws runs no destructor: release whatever init() had built before the failure
    try {
        init(opt);
    } catch (...) {
        release();
        throw;
    }
}

void HipGCN::init(const HipGCNOptions &opt) {
    device_ = opt.device;
    opt_ = opt;                 // every switch comes from the options (HipGCNOptions::from_environment for the HIPGCN_* variables)
    // opt.verbose: where the model build's wall time goes (stderr), phase by phase
    const bool verbose = opt.verbose && opt.rank == 0;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "gcn-hip: build %-34s %8.3f s\n", what, std::chrono::duration<double>(now - t_phase).count());
        t_phase = now;
    };
    // argument checks first: nothing is allocated for a request that cannot be served
    if (opt.world > 1 && !opt.comm && !opt.host_allgather && !(flags & HIPGCN_NULL_COMM) && !opt.nccl_id)
        throw GcnHipFailure(-1, "world > 1 needs an RCCL unique id");
    if (params.num_nodes < 1 || params.hidden_dim < 1 || params.output_dim < 1 || params.input_dim < 1)
        throw GcnHipFailure(-1, "HipGCN: empty model dimensions");
    if ((int)data->graph.indptr.size() != params.num_nodes + 1 || (int)data->split.size() != params.num_nodes ||
        (int)data->label.size() != params.num_nodes || (int)data->feature_index.indptr.size() != params.num_nodes + 1)
        throw GcnHipFailure(-1, "HipGCN: GCNData arrays do not match num_nodes");
    GCNHIP_CHECK(gcnhip_ctx_create(&env.ctx, opt.device, nullptr));
    if (opt.gemm >= 0) GCNHIP_CHECK(gcnhip_ctx_set_option(env.ctx, "gemm_bf16x3", opt.gemm ? 2 : 0));   // HIPGCN_GEMM; else the library's default
    if (opt.fold_training) GCNHIP_CHECK(gcnhip_ctx_set_option(env.ctx, "gs_fold", 1));                  // (ignored by a library built without the experiments)
    timers.reset(new DeviceTimers(env.ctx));
    timers->enabled = (flags & HIPGCN_TIMERS) != 0;
    env.timers = timers.get();
    if (opt.comm) {
        env.comm = opt.comm;
        if (opt.own_comm) owned_comm.reset(opt.comm);
    } else if (opt.world > 1 && opt.host_allgather) {
        owned_comm.reset(make_host_comm(env.ctx, opt.rank, opt.world, opt.host_allgather, opt.host_allreduce, opt.host_user));
        env.comm = owned_comm.get();
    } else if (opt.world > 1 && (flags & HIPGCN_NULL_COMM)) {
        owned_comm.reset(new NullComm(opt.rank, opt.world));
        env.comm = owned_comm.get();
    } else if (opt.world > 1) {
        if (!opt.nccl_id) throw GcnHipFailure(-1, "world > 1 needs an RCCL unique id");
        owned_comm.reset(make_rccl_comm(env.ctx, opt.rank, opt.world, opt.nccl_id));
        env.comm = owned_comm.get();
    } else {
        owned_comm.reset(new SelfComm());
        env.comm = owned_comm.get();
    }
    env.seed = (uint64_t)opt.seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    env.bf16_tables = (flags & HIPGCN_BF16_TABLES) != 0;
    // The factored aggregation (gcnhip_graphsum_ex): no per-edge coefficient stream; the gathered matrices are stored
    // pre-multiplied by dinv of their row (the producers fold the factor into a row-wise epilogue or a value array).  The
    // fused f32 path only; HIPGCN_EDGE_COEF restores the reference's per-edge coefficients.
    if ((flags & (HIPGCN_PACKED_DH1 | HIPGCN_BWD_PIPELINE)) && !gcnhip_experiments()) {
        // measured-slower variants live behind the library's compile-time switch (make EXPERIMENTS=1)
        fprintf(stderr, "gcn-hip: HIPGCN_PACKED_DH1 / HIPGCN_BWD_PIPELINE ignored: libgcnhip.so was built without GCNHIP_EXPERIMENTS\n");
        flags &= ~(HIPGCN_PACKED_DH1 | HIPGCN_BWD_PIPELINE);
    }
    factored_ = !(flags & (HIPGCN_MODULAR | HIPGCN_BF16_TABLES | HIPGCN_PACKED_DH1 | HIPGCN_EDGE_COEF));
    const int world = env.comm->size(), rank = env.comm->rank();
    const int N = params.num_nodes, F = params.input_dim, H = params.hidden_dim, C = params.output_dim;

    // ---- node order (several GPUs: by structure when the ids carry no locality), row partition, this rank's slice
    // How remote rows will arrive is decided BEFORE the node order: renumbering exists to turn an all-gather into halo
    // lists, so a run whose exchange is pinned to the all-gather (the flag, HIPGCN_EXCHANGE=allgather, or the default over
    // RCCL until the halo exchange has met a peer) keeps its ids — no second host copy of X, no group search, and rows,
    // dropout decisions and get_var() stay in the dataset's order.
    int exchange_mode = (flags & HIPGCN_EXCHANGE_HALO) ? 2 : ((flags & HIPGCN_EXCHANGE_ALLGATHER) ? 1 : 0);
    {
        if (opt.exchange >= 0) exchange_mode = opt.exchange;
        // Over RCCL the per-graph decision is opt-in (HIPGCN_EXCHANGE=auto|halo, or the flag): the halo exchange is a grouped
        // ncclSend/ncclRecv that has run against real peers only in gcnhost_rccl_selftest_world, so an unasked-for run takes
        // the in-place all-gather.  bench.py's launcher runs that self-test as a throw-away group of ranks and then asks
        // for `auto`.  (Host-staged transports and tests decide per graph as before.)
        const bool over_rccl = world > 1 && !opt.comm && !opt.host_allgather && !(flags & HIPGCN_NULL_COMM);
        if (over_rccl && exchange_mode == 0 && opt.exchange != 0) exchange_mode = 1;
    }
    // Parity mode (HIPGCN_HOST_MASKS) replays the reference's RNG stream in the DATASET's element order
    // (host_masks_for_epoch): a renumbered run would hand node k's decisions to another node, so it keeps its ids too
    // (HIPGCN_STRUCTURE_PARTITION still forces the renumbering; the run is then the parity run of the renumbered dataset).
    const bool may_renumber = world > 1 && !(flags & HIPGCN_ID_PARTITION) &&
                              ((flags & HIPGCN_STRUCTURE_PARTITION) || (exchange_mode != 1 && !(flags & HIPGCN_HOST_MASKS)));
    if (may_renumber) renumber_nodes(world);
    phase("context, comm, node order");
    const std::vector<int> &gp = data->graph.indptr, &gi = data->graph.indices;
    part = make_partition(gp.data(), N, world);
    const int r0 = part.start[rank], r1 = part.start[rank + 1];
    n_local = r1 - r0;
    nnzA_local = (long)gp[r1] - gp[r0];
    labels_assortative = !(flags & (HIPGCN_NO_ROW_GROUPS | HIPGCN_NO_LABEL_HINT)) && labels_are_assortative(*data, N, C);
    // no usable labels: look for row groups in the graph itself (one pass over the edges per sweep, every rank the same
    // result); whether they are used is decided by timing, like every other schedule (tune_schedule)
    // (on a host thread beside the object builds and H2D copies below: one sweep over an R-MAT graph of scale 22 is 3 s, and its
    //  result is not needed before the schedules are timed)
    std::future<StructureGroups> groups_search;
    if (!(flags & HIPGCN_NO_ROW_GROUPS) && !labels_assortative && n_local >= 4096 && opt.structure_groups &&
        (opt.schedule < 0 || opt.schedule == 3 || opt.slice_tuning))   // the slice rule reads the groups whatever picked the schedule
        groups_search = std::async(std::launch::async, [&gp, &gi, N]() { return structure_groups(gp.data(), gi.data(), N); });
    struct JoinGroups {                       // never leave the thread running over a dataset that is being torn down
        std::future<StructureGroups> &f;
        ~JoinGroups() { if (f.valid()) f.wait(); }
    } join_groups{groups_search};
    phase("labels / structure groups");
    xplan = make_exchange_plan(gp.data(), gi.data(), N, part, rank, exchange_mode);
    env.plan = &xplan;
    env.xbuf = &xbuf;
    if (world > 1) {
        const LocalGraph lg = build_table_graph(gp.data(), gi.data(), N, part, xplan);
        GCNHIP_CHECK(gcnhip_graph_create(env.ctx, &graph, lg.indptr.data(), lg.indices.data(), lg.n_rows, lg.n_cols, lg.col_deg.data()));
        GCNHIP_CHECK(gcnhip_graph_reserve_width(env.ctx, graph, std::max(params.hidden_dim, params.output_dim)));
    } else {
        GCNHIP_CHECK(gcnhip_graph_create(env.ctx, &graph, gp.data(), gi.data(), N, N, nullptr));
        GCNHIP_CHECK(gcnhip_graph_reserve_width(env.ctx, graph, std::max(params.hidden_dim, params.output_dim)));
    }
    phase("adjacency object (gcnhip_graph_create)");
    const std::vector<int> &fp = data->feature_index.indptr, &fi = data->feature_index.indices;
    const long f0 = fp[r0], f1 = fp[r1];
    {
        std::vector<int> lp(n_local + 1);
        for (int r = 0; r <= n_local; r++) lp[r] = fp[r0 + r] - fp[r0];
        GCNHIP_CHECK(gcnhip_feat_create(env.ctx, &feat, lp.data(), fi.empty() ? nullptr : fi.data() + f0,
                                        data->feature_value.data() + f0, n_local, F));
    }
    // Replicating X.W1 trades a 119 MB all-gather per forward for 0.4 ms of extra GEMM on every rank.  With 2-4
    // GPUs each rank receives over 1-3 xGMI links (~60 GB/s each) and the GEMM is cheaper; with 8 GPUs seven
    // links feed the gather (~0.3 ms) while the replicated GEMMs would be 45 % of the per-rank compute.
    // (ALLGATHER plans only: a HALO plan moves few rows, and its table is not in global row order)
    replicate_l1 = world > 1 && !xplan.halo && (world <= 4 || (flags & HIPGCN_REPLICATE_L1)) && !(flags & (HIPGCN_NO_REPLICATE_L1 | HIPGCN_MODULAR));
    if (replicate_l1) {
        GCNHIP_CHECK(gcnhip_feat_create(env.ctx, &feat_full, fp.data(), fi.empty() ? nullptr : fi.data(),
                                        data->feature_value.data(), N, F));
        full_vals = gcnhip_feat_values(feat_full);
        std::vector<int> lp(n_local + 1), deg(N);
        for (int r = 0; r <= n_local; r++) lp[r] = gp[r0 + r] - gp[r0];
        for (int j = 0; j < N; j++) deg[j] = gp[j + 1] - gp[j];
        GCNHIP_CHECK(gcnhip_graph_create(env.ctx, &graph_l1, lp.data(), gi.data() + gp[r0], n_local, N, deg.data()));
        GCNHIP_CHECK(gcnhip_graph_reserve_width(env.ctx, graph_l1, std::max(params.hidden_dim, params.output_dim)));
    }
    phase("feature objects (gcnhip_feat_create)");
    // truth per split, once (the reference rebuilds and re-uploads it per call: cuda_gcn.cu:85-97)
    {
        int32_t *d_split = dev_upload(env.ctx, data->split.data() + r0, (size_t)n_local);
        int32_t *d_label = dev_upload(env.ctx, data->label.data() + r0, (size_t)n_local);
        double cnt[4] = {0, 0, 0, 0};
        for (int s = 1; s <= 3; s++) {
            void *t;
            GCNHIP_CHECK(gcnhip_malloc(env.ctx, &t, (size_t)(n_local ? n_local : 1) * sizeof(int32_t)));
            d_truth[s] = (int32_t *)t;
            GCNHIP_CHECK(gcnhip_set_truth(env.ctx, d_truth[s], d_split, d_label, n_local, s));
            for (int i = r0; i < r1; i++) cnt[s] += data->split[i] == s;
        }
        GCNHIP_CHECK(gcnhip_ctx_sync(env.ctx));
        gcnhip_free(env.ctx, d_split);
        gcnhip_free(env.ctx, d_label);
        env.comm->allreduce_sum_host(cnt, 4);
        for (int s = 1; s <= 3; s++) split_count[s] = (int)cnt[s];
    }

    // The last aggregation of a forward computes only the rows the loss and the accuracy read
    // (CrossEntropyLoss::forward skips truth < 0, module.cpp:131-133; get_accuracy, gcn.cpp:86-88):
    // one registered row subset per split.
    if (!(flags & HIPGCN_ALL_ROWS)) add_split_rowsets(env.ctx, graph, split_rows);
    if (!(flags & HIPGCN_MODULAR)) {
        // the loss walks the rows of the scored split only (it skips the others anyway, module.cpp:131-133)
        for (int s = 1; s <= 3; s++) {
            std::vector<int32_t> rows;
            for (int r = 0; r < n_local; r++)
                if (data->split[r0 + r] == s) rows.push_back(r);
            split_local_n[s] = (int)rows.size();
            d_split_list[s] = dev_upload(env.ctx, rows.data(), rows.size());
        }
    }

    // training-split bit per table row (= node, on one GPU): dZ is zero elsewhere, GraphSum's backward skips those rows
    {
        const size_t n_pos = world > 1 ? (size_t)xplan.table_rows : (size_t)N;
        std::vector<uint32_t> bits(n_pos / 32 + 2, 0u);
        for (size_t t = 0; t < n_pos; t++) {
            const int j = world > 1 ? xplan.table_global[t] : (int)t;
            if (j >= 0 && data->split[j] == 1) bits[t >> 5] |= 1u << (t & 31);
        }
        d_train_bits = dev_upload(env.ctx, bits.data(), bits.size());
        bwd_bits = d_train_bits;
        h_train_bits = std::move(bits);
    }

    // ---- variables (numbering of gcn.cpp:21-54)
    variables.resize(7);
    for (auto &v : variables) v.reset(new HipVariable());
    const ExchangePlan *xp = world > 1 ? &xplan : nullptr;
    if (flags & HIPGCN_MODULAR) {
        variables[0]->alloc(env.ctx, 1, (int)(f1 - f0), false);
        input = variables[0].get();
        input_vals = input->data;
    } else {
        input_vals = gcnhip_feat_values(feat);
    }
    if (replicate_l1) variables[1]->alloc_replicated(env.ctx, N, n_local, r0, H, true);   // H0: every row computed here
    else variables[1]->alloc(env.ctx, n_local, H, true, true, false, xp);    // H0: data gathered
    variables[3]->alloc(env.ctx, n_local, H, true, false, true, xp);    // H1: grad gathered
    rebuild_dh1 = world > 1 && !(flags & (HIPGCN_GATHER_DH1 | HIPGCN_MODULAR));
    variables[4]->alloc(env.ctx, n_local, C, true, true, rebuild_dh1, xp);   // Z0: data gathered (+ grad when dH1 is rebuilt)
    variables[6]->alloc(env.ctx, n_local, C, true, false, true, xp);    // Z : grad gathered
    if (world > 1) {
        // widest row (in 4-byte words) an exchange will carry: f32 rows of H / C, mask words, bf16 rows
        const int ldH = variables[1]->ld, ldC = variables[4]->ld;
        exchange_buffers_create(env.ctx, xplan, std::max(std::max(ldH, ldC), (H + 63) / 64 * 64), &xbuf);
    }
    output = variables[6].get();
    HipVariable *W1 = variables[2].get(), *W2 = variables[5].get();
    W1->alloc(env.ctx, F, H, false);
    W2->alloc(env.ctx, H, C, false);
    // both weight gradients and the 4 loss/accuracy scalars share one buffer: one all-reduce per epoch
    gradbuf_elems = W1->elems() + W2->elems() + 4;
    {
        void *q;
        GCNHIP_CHECK(gcnhip_malloc(env.ctx, &q, gradbuf_elems * sizeof(float)));
        GCNHIP_CHECK(gcnhip_memset_async(env.ctx, q, 0, gradbuf_elems * sizeof(float)));
        gradbuf = (float *)q;
        W1->grad = gradbuf; W1->requires_grad = true;
        W2->grad = gradbuf + W1->elems(); W2->requires_grad = true;
        d_result = gradbuf + W1->elems() + W2->elems();
        GCNHIP_CHECK(gcnhip_malloc(env.ctx, &q, 2 * sizeof(int32_t)));
        d_result_i = (int32_t *)q;
        GCNHIP_CHECK(gcnhip_malloc(env.ctx, &q, (size_t)RING * 4 * 8 * sizeof(float)));
        d_ring = (float *)q;
        GCNHIP_CHECK(gcnhip_memset_async(env.ctx, q, 0, (size_t)RING * 4 * 8 * sizeof(float)));
        // two epoch words: [0] the epoch being (or about to be) trained, read by the training pass; [1] the epoch whose update
        // ran last, naming the metrics row of an evaluation on this stream.  Adam's launch moves both (gcnhip_adam_step_advance),
        // so an epoch starts without a counter launch.  Before the first update: 0 and -1 (the row an evaluation of the
        // initial weights has always used).
        GCNHIP_CHECK(gcnhip_malloc(env.ctx, &q, 2 * sizeof(uint32_t)));
        env.d_epoch = (uint32_t *)q;
        env.d_epoch_done = env.d_epoch + 1;
        GCNHIP_CHECK(gcnhip_memset_async(env.ctx, env.d_epoch, 0, sizeof(uint32_t)));
        GCNHIP_CHECK(gcnhip_memset_async(env.ctx, env.d_epoch_done, 0xFF, sizeof(uint32_t)));
    }
    // Glorot with the reference's RNG and draw order: all of W1, then all of W2 (gcn.cpp:30,49)
    rng.seed_time((unsigned)opt.seed);
    {
        Variable h1(F * H), h2(H * C);
        h1.glorot(F, H, rng);
        h2.glorot(H, C, rng);
        W1->upload(h1.data.data());
        W2->upload(h2.data.data());
    }
    if (flags & HIPGCN_HOST_MASKS) {
        keep0_first = replicate_l1 ? 0 : f0;
        h_keep0.resize(replicate_l1 ? (size_t)fp[N] : (size_t)(f1 - f0));
        h_keep1.resize((size_t)n_local * H);
        void *q;
        GCNHIP_CHECK(gcnhip_malloc(env.ctx, &q, h_keep0.size() + 16)); d_keep0 = (uint8_t *)q;
        GCNHIP_CHECK(gcnhip_malloc(env.ctx, &q, h_keep1.size() + 16)); d_keep1 = (uint8_t *)q;
        env.keep_input = d_keep0;
        env.keep_input_bwd = d_keep0 + (f0 - keep0_first);
        env.keep_hidden = d_keep1;
    }
    phase("truth, row subsets, variables, Glorot");
    if (groups_search.valid()) {
        StructureGroups sg = groups_search.get();
        if (sg.useful) { structure_group = std::move(sg.group); structure_n_groups = sg.n_groups; }
        phase("structure groups (waited for)");
    }
    if (!(flags & HIPGCN_NO_ROW_GROUPS)) tune_schedule();
    phase("row schedules timed (tune_schedule)");
    // The output layer's backward aggregates dZ, which is zero outside the training split: the edges that point at those
    // rows leave the operator for good (a third of Reddit's, 95 % of Cora's) — after the row order has been chosen,
    // which the restricted object inherits.
    if (!(flags & HIPGCN_MASKED_BWD) && n_local > 0)
        GCNHIP_CHECK(gcnhip_graph_create_restricted(env.ctx, &graph_bwd_out, graph, h_train_bits.data()));
    // decided from world, flags and the storage format only — the same on every rank: the exchange lane's communicator is
    // an ncclCommSplit, a collective over the parent; a rank that owns no rows still creates it (and skips only the cuts)
    if ((flags & HIPGCN_OVERLAP_EXCHANGE) && world > 1 && !env.bf16_tables) build_overlap();
    build_modules();
    if (!(flags & (HIPGCN_NO_AGG_FIRST_EVAL | HIPGCN_MODULAR)) && gcnhip_feat_is_dense(feat) && n_local > 0) build_agg_first_eval();
    phase("restricted operator, modules, A^.X");
    if (factored_) apply_factored_scales();
    phase("factored scales");
    // opt-in everywhere: with several GPUs the lane brings a second communicator and the turnstile, which must be measured
    // on a multi-GPU node before they may become a default there (bench.py tries both schedules)
    if (!(flags & HIPGCN_NO_EVAL_LANE) && (flags & HIPGCN_EVAL_LANE)) {
        try {
            build_eval_lane();
        } catch (const GcnHipFailure &e) {
            // e.g. an RCCL without ncclCommSplit: every rank fails the same way and falls back to one lane
            fprintf(stderr, "gcn-hip: validation lane disabled (%s)\n", e.what());
            destroy_lane();
        }
    }
    AdamParams ap = AdamParams::get_default();
    ap.lr = params.learning_rate;
    ap.weight_decay = params.weight_decay;
    optimizer.reset(new HipAdam());
    optimizer->init(&env, {{W1, true}, {W2, false}}, ap, params.epochs > 0 ? params.epochs + 8 : 8);   // gcn.cpp:62-65
    GCNHIP_CHECK(gcnhip_ctx_sync(env.ctx));
    phase("validation lane, optimizer");
}

// The first layer of the factored model multiplies D^-1/2 X (and, for evaluation, D^-1/2 (A^ X)): the factor that the
// aggregation's input rows must carry rides in the value arrays, so no GEMM or sparse kernel changes, and the weight
// gradient X'^T . (raw sum) comes out as the reference's X^T . dH0.  Called once, after A^.X has been built from the
// unscaled X.
void HipGCN::apply_factored_scales() {
    const float *dinv_row = nullptr;
    GCNHIP_CHECK(gcnhip_graph_scales(graph, &dinv_row, nullptr, nullptr, nullptr));
    GCNHIP_CHECK(gcnhip_feat_scale_rows(env.ctx, feat, dinv_row));
    if (feat_agg) GCNHIP_CHECK(gcnhip_feat_scale_rows(env.ctx, feat_agg, dinv_row));
    if (feat_full) {                                           // every row of X on every rank: the global degrees are graph_l1's columns
        const float *dinv_all = nullptr;
        GCNHIP_CHECK(gcnhip_graph_scales(graph_l1, nullptr, nullptr, &dinv_all, nullptr));
        GCNHIP_CHECK(gcnhip_feat_scale_rows(env.ctx, feat_full, dinv_all));
    }
}

void HipGCN::row_scale(std::vector<float> &dinv) {
    dinv.assign((size_t)n_local, 1.f);
    const float *d = nullptr;
    GCNHIP_CHECK(gcnhip_graph_scales(graph, &d, nullptr, nullptr, nullptr));
    if (n_local) GCNHIP_CHECK(gcnhip_d2h(env.ctx, dinv.data(), d, dinv.size() * sizeof(float)));
}

// Rank blocks are contiguous ranges of the node order.  When the order the dataset came in forces the all-gather (the
// neediest rank would read more than 75 % of the other ranks' rows) the graph is priced under orders derived from its
// structure (partition.h); if one of them gets by with halo lists, the dataset is renumbered once, here, before anything
// is built from it.  Every rank computes the same answer.
void HipGCN::renumber_nodes(int world) {
    const int N = params.num_nodes;
    const std::vector<int> &gp = data->graph.indptr, &gi = data->graph.indices;
    const bool force = (flags & HIPGCN_STRUCTURE_PARTITION) != 0;
    StructureGroups sg;
    const OrderCost ids = exchange_cost(gp.data(), gi.data(), N, world);
    if ((force || ids.halo_share > 0.75) && N >= 4096) sg = structure_groups(gp.data(), gi.data(), N);
    NodeOrderChoice ch = choose_node_order(gp.data(), gi.data(), N, world, sg.useful ? sg.group.data() : nullptr, force);
    if (ch.order.empty()) return;
    if (env.comm->rank() == 0 && opt_.verbose)
        fprintf(stderr, "gcn-hip: nodes renumbered by %s: neediest rank reads %ld rows per exchange instead of %ld (all-gather: %ld)\n",
                ch.name, ch.chosen.recv_rows_max, ch.ids.recv_rows_max, (long)(world - 1) * ch.ids.rows_max);
    renumbered.reset(new GCNData());
    GCNData &d = *renumbered;
    permute_csr(gp.data(), gi.data(), N, ch.order, d.graph.indptr, d.graph.indices);
    const std::vector<int> &fp = data->feature_index.indptr, &fi = data->feature_index.indices;
    d.feature_index.indptr.assign((size_t)N + 1, 0);
    for (int k = 0; k < N; k++) d.feature_index.indptr[k + 1] = d.feature_index.indptr[k] + (fp[ch.order[k] + 1] - fp[ch.order[k]]);
    d.feature_value.resize(data->feature_value.size());
    if (!fi.empty()) d.feature_index.indices.resize(fi.size());
    d.split.resize(N); d.label.resize(N);
    for (int k = 0; k < N; k++) {
        const int o = ch.order[k];
        const size_t n = (size_t)(fp[o + 1] - fp[o]), dst = (size_t)d.feature_index.indptr[k];
        if (n) memcpy(&d.feature_value[dst], &data->feature_value[(size_t)fp[o]], n * sizeof(float));
        if (n && !fi.empty()) memcpy(&d.feature_index.indices[dst], &fi[(size_t)fp[o]], n * sizeof(int));
        d.split[k] = data->split[o];
        d.label[k] = data->label[o];
    }
    node_order_ = std::move(ch.order);
    node_order_name_ = ch.name;
    data = renumbered.get();
}

// Which rows the aggregation has in flight together decides its speed (what the XCD L2s hold; whether hub rows
// overlap with the tail of short rows) and nothing else: every schedule gives the same bits.  Candidates:
// descending degree; label-major when the labels are communities of this graph (Reddit: subreddits), else group-major over
// groups found in the graph by modularity local moving (cluster.h) when that finds any; degree rank
// dealt into 256 equal-mix groups (graphs with a long tail of short rows, e.g. R-MAT).  Each is timed on the
// hidden-width aggregation of this rank's rows and the fastest is kept for all of this rank's adjacency objects.
void HipGCN::apply_schedule(gcnhip_ctx *ctx, gcnhip_graph *g) {
    const int r0 = part.start[env.comm->rank()];
    if (sched_mode == 3)                                      // groups found in the graph: the same group-major order as labels
        GCNHIP_CHECK(gcnhip_graph_set_schedule(ctx, g, 1, structure_group.data() + r0, 0));
    else
        GCNHIP_CHECK(gcnhip_graph_set_schedule(ctx, g, sched_mode, sched_mode == 1 ? data->label.data() + r0 : nullptr, sched_groups));
}

void HipGCN::add_split_rowsets(gcnhip_ctx *ctx, gcnhip_graph *g, gcnhip_rowset *out[4]) {
    const int r0 = part.start[env.comm->rank()];
    for (int s = 1; s <= 3; s++) {
        std::vector<uint32_t> bits((size_t)n_local / 32 + 2, 0u);
        for (int r = 0; r < n_local; r++)
            if (data->split[r0 + r] == s) bits[r >> 5] |= 1u << (r & 31);
        GCNHIP_CHECK(gcnhip_graph_add_rowset(ctx, g, bits.data(), &out[s]));
    }
}

// Column-slice width of the XCD-sliced hidden-width launch (round 5): 64-float slices (two per 128-wide row: each XCD's L2
// sees half the table) or 32-float slices (four: a quarter of the table per L2, twice the re-reads of the index stream,
// 128-byte requests).  Which wins depends on where the graph's reuse sits (tools/exp_structure.py, bench.py's structure legs):
// with row groups that fit an L2 and hold most of the edges the wide slices do (reddit-syn: 0.76 vs 0.84 ms); on a graph whose
// reuse is its hub rows the narrow ones (reddit-syn-h0: 1.22 vs 1.11 ms; -h03 1.02 vs 0.98; -zipf 0.97 vs 0.89); past the
// Infinity Cache the wide ones again (R-MAT scale 21: 4.57 vs 5.20 ms).  The two widths split a row's sum over 4 or 8 lane
// groups, i.e. associate it differently, so the choice must NOT depend on a timing (two runs of one dataset print the same
// bits): it is a rule on the graph — narrow when the gathered table is cache-resident and less than 45 % of the stored
// edges stay inside a row group (label or found community) whose 256-byte slices fit half an L2 (8192 rows).
void HipGCN::choose_slice_width() {
    const int N = params.num_nodes, H = params.hidden_dim;
    slice_floats = 64;
    int cur_gl = 0;
    GCNHIP_CHECK(gcnhip_ctx_get_option(env.ctx, "gs_l", &cur_gl));
    if (cur_gl == 8 || cur_gl == 4) {                      // preset (GCNHIP_GS_L): report what the launches will use; the lane mirrors it
        const int f = cur_gl * 4;
        if (H % f == 0 && H / f > 1 && H / f <= 8 && 8 % (H / f) == 0) slice_floats = f;
        return;
    }
    if (cur_gl != 0 || !opt_.slice_tuning || H % 64 != 0 || H / 32 > 8 || 8 % (H / 32) != 0) return;
    if ((size_t)N * H * sizeof(float) > ((size_t)256 << 20)) return;                 // HBM regime: wide
    // (the groups the schedule candidates were built from — labels when they are communities of the graph, else what the
    //  group search found — NOT the candidate the timing picked: every schedule gives the same bits, the slice width does not)
    const int *group = labels_assortative ? data->label.data() : (!structure_group.empty() ? structure_group.data() : nullptr);
    double share = 0.0;
    if (group) {
        std::vector<int> size;
        for (int i = 0; i < N; i++) {
            if (group[i] < 0) continue;
            if ((size_t)group[i] >= size.size()) size.resize((size_t)group[i] + 1, 0);
            size[group[i]]++;
        }
        const std::vector<int> &gp = data->graph.indptr, &gi = data->graph.indices;
        long inside = 0, total = 0;
        for (int i = 0; i < N; i++)
            for (int e = gp[i]; e < gp[i + 1]; e++) {
                if (gi[e] == i) continue;
                total++;
                inside += group[i] >= 0 && group[gi[e]] == group[i] && size[group[i]] <= 8192;
            }
        share = total ? (double)inside / (double)total : 0.0;
    }
    slice_floats = share < 0.45 ? 32 : 64;
    GCNHIP_CHECK(gcnhip_ctx_set_option(env.ctx, "gs_l", slice_floats == 32 ? 8 : 0));
    if (opt_.verbose && env.comm->rank() == 0)
        fprintf(stderr, "gcn-hip: hidden-width aggregation: %.0f %% of the edges inside a row group that fits an L2 -> %d-float column slices\n",
                100 * share, slice_floats);
}

void HipGCN::tune_schedule() {
    if (n_local < 4096 && opt_.schedule < 0) return;         // launch-bound graphs: nothing to gain
    const int H = params.hidden_dim;
    gcnhip_graph *g = replicate_l1 ? graph_l1 : graph;       // the layer-1 aggregation, the widest one
    if (opt_.schedule >= 0) {
        // pinned (HIPGCN_SCHEDULE): no candidate is timed, so no tuning launch shares a kernel name with the epochs' launches
        sched_mode = opt_.schedule; sched_groups = opt_.schedule == 2 ? opt_.schedule_groups : 0;
        if (sched_mode == 3 && structure_group.empty()) sched_mode = 0;      // the search found no usable groups
        if (sched_mode == 3) sched_groups = structure_n_groups;
        if (sched_mode != 0) {
            apply_schedule(env.ctx, graph);
            if (graph_l1) apply_schedule(env.ctx, graph_l1);
        }
        choose_slice_width();
        return;
    }
    // the slice width first (a rule on the graph, not a timing): the candidates are timed with the launch the epochs will use
    choose_slice_width();
    HipVariable *in = variables[1].get(), *out = variables[3].get();
    float *src = in->full ? in->full : in->data;
    const size_t src_elems = in->full ? in->full_elems : in->elems();
    GCNHIP_CHECK(gcnhip_memset_async(env.ctx, src, 0, src_elems * sizeof(float)));
    struct Cand { int mode, groups; };
    std::vector<Cand> cands = {{0, 0}, {2, 256}};
    if (labels_assortative) cands.push_back({1, 0});
    if (!structure_group.empty()) cands.push_back({3, structure_n_groups});
    void *e0, *e1;
    GCNHIP_CHECK(gcnhip_event_create(&e0));
    GCNHIP_CHECK(gcnhip_event_create(&e1));
    float best = 0.f;
    Cand pick = cands[0];
    // the candidates are timed with the launch the epoch uses: the factored operator (no coefficient stream, 6-10 % of a
    // launch) on the fused f32 path, the reference's per-edge coefficients otherwise
    gcnhip_gs_opts gso;
    memset(&gso, 0, sizeof gso);
    gso.scaling = factored_ ? 2 : 0;
    bool fresh = true;                                       // g still has the schedule it was built with: descending degree = candidate 0
    for (const Cand &c : cands) {
        sched_mode = c.mode; sched_groups = c.groups;
        if (!(fresh && c.mode == 0)) apply_schedule(env.ctx, g);
        fresh = false;
        float ms = 0.f;
        // two launches warm the caches, size the scratch and let the clock settle, five are timed.  (Until round 6: one and
        // two — on reddit-syn-h0, where the candidates are 5 % apart, the driver's run picked plain degree order, 245 epochs/s,
        // where the same tree had picked dealt-256 an hour earlier, 254.)
        for (int it = 0; it < 7; it++) {
            if (it == 2) GCNHIP_CHECK(gcnhip_event_record(env.ctx, e0));
            GCNHIP_CHECK(gcnhip_graphsum_ex(env.ctx, g, &gso, src, in->ld, out->data, out->ld, H));
        }
        GCNHIP_CHECK(gcnhip_event_record(env.ctx, e1));
        GCNHIP_CHECK(gcnhip_event_elapsed_ms(e0, e1, &ms));
        if (best == 0.f || ms < best) { best = ms; pick = c; }
    }
    const bool g_has_pick = sched_mode == pick.mode && sched_groups == pick.groups;    // the last candidate timed is still applied to g
    sched_mode = pick.mode; sched_groups = pick.groups;
    if (!(g == graph && g_has_pick)) apply_schedule(env.ctx, graph);
    if (graph_l1 && !(g == graph_l1 && g_has_pick)) apply_schedule(env.ctx, graph_l1);
    gcnhip_event_destroy(e0);
    gcnhip_event_destroy(e1);
    GCNHIP_CHECK(gcnhip_memset_async(env.ctx, out->data, 0, out->elems() * sizeof(float)));
}

// The adjacency of this rank cut in two by the owner of the column: edges whose source row is one of this rank's own rows
// (complete as soon as the producer kernel has finished) and edges that need a row of another rank (complete when the
// exchange has finished).  Both halves keep the parent's coefficients and row order.  The same cut of the restricted
// operator of the output layer's backward, and the split subsets of the last aggregation on both halves.
void HipGCN::build_overlap() {
    const size_t n_pos = (size_t)xplan.table_rows;
    std::vector<uint32_t> own(n_pos / 32 + 2, 0u), other(n_pos / 32 + 2, 0u);
    for (size_t t = 0; t < n_pos; t++) {
        const bool mine = (int)t >= xplan.own_offset && (int)t < xplan.own_offset + n_local;
        (mine ? own : other)[t >> 5] |= 1u << (t & 31);
    }
    xlane.reset(new ExchangeLane(env.ctx, device_, env.comm, xplan, (int)xbuf.max_ld_words, timers->enabled));     // collective: every rank
    env.xlane = xlane.get();
    if (n_local == 0) return;             // no rows, no operators to cut: the modules see no split_loc and aggregate nothing
    GCNHIP_CHECK(gcnhip_graph_create_restricted(env.ctx, &graph_loc, graph, own.data()));
    GCNHIP_CHECK(gcnhip_graph_create_restricted(env.ctx, &graph_rem, graph, other.data()));
    if (graph_bwd_out) {
        GCNHIP_CHECK(gcnhip_graph_create_restricted(env.ctx, &graph_bwd_loc, graph_bwd_out, own.data()));
        GCNHIP_CHECK(gcnhip_graph_create_restricted(env.ctx, &graph_bwd_rem, graph_bwd_out, other.data()));
    }
    if (!(flags & HIPGCN_ALL_ROWS)) {
        add_split_rowsets(env.ctx, graph_loc, split_rows_loc);
        add_split_rowsets(env.ctx, graph_rem, split_rows_rem);
    }
}

// hand the halves of the cut operator to an aggregation (output_layer: also the halves of its restricted backward
// operator and the per-half subsets of the scored rows)
void HipGCN::wire_overlap(HipGraphSum *gs, bool output_layer) {
    if (!graph_loc) return;
    gs->split_loc = graph_loc; gs->split_rem = graph_rem;
    if (!output_layer) return;
    gs->bwd_split_loc = graph_bwd_loc; gs->bwd_split_rem = graph_bwd_rem;
    if (!(flags & HIPGCN_ALL_ROWS)) { gs->fwd_out_rows_loc = &cur_out_rows_loc; gs->fwd_out_rows_rem = &cur_out_rows_rem; }
}

void HipGCN::build_modules() {
    const int N = n_local, F = params.input_dim, H = params.hidden_dim, C = params.output_dim;
    const int rank = env.comm->rank();
    const uint64_t nnz_off = (uint64_t)data->feature_index.indptr[part.start[rank]];
    const uint64_t hid_off = (uint64_t)part.start[rank] * H;
    HipVariable *H0 = variables[1].get(), *W1 = variables[2].get(), *H1 = variables[3].get(),
                *Z0 = variables[4].get(), *W2 = variables[5].get(), *Z = variables[6].get();
    const float p = params.dropout;
    static const uint8_t *const no_mask = nullptr;
    if (flags & HIPGCN_MODULAR) {
        // the reference's list, one for one (gcn.cpp:23-59)
        modules.push_back(new HipDropout(&env, input, p, KEY_INPUT_DROPOUT, nnz_off, (flags & HIPGCN_HOST_MASKS) ? &env.keep_input : &no_mask));
        modules.push_back(new HipSparseMatmul(&env, &input_vals, W1, H0, feat, N, F, H, 0.f, nnz_off));
        { auto *gs = new HipGraphSum(&env, H0, H1, graph, H); wire_overlap(gs, false); modules.push_back(gs); }
        modules.push_back(new HipReLU(&env, H1));
        modules.push_back(new HipDropout(&env, H1, p, KEY_HIDDEN_DROPOUT, hid_off, (flags & HIPGCN_HOST_MASKS) ? &env.keep_hidden : &no_mask));
        modules.push_back(new HipMatmul(&env, H1, W2, Z0, N, H, C));
        { auto *gs = new HipGraphSum(&env, Z0, Z, graph, C); gs->bwd_row_bits = &bwd_bits; gs->bwd_graph = graph_bwd_out; gs->fwd_out_rows = &cur_out_rows; wire_overlap(gs, true); modules.push_back(gs); }
        modules.push_back(new HipCrossEntropyLoss(&env, Z, &cur_truth, &cur_count, d_result, d_result_i, C, true));
    } else {
        const float scale = 1 / (1 - p);
        auto *sm = new HipSparseMatmul(&env, &input_vals, W1, H0, feat, N, F, H, p, nnz_off);
        auto *gs = new HipGraphSum(&env, H0, H1, graph, H, p, hid_off);
        auto *mm = new HipMatmul(&env, H1, W2, Z0, N, H, C, scale);
        if (replicate_l1) { sm->sp_full = feat_full; sm->vals_full = &full_vals; gs->fwd_graph_replicated = graph_l1; }
        wire_overlap(gs, false);
        if (factored_) {
            const float *dinv_row = nullptr, *dinv2_row = nullptr, *dinv2_col = nullptr;
            GCNHIP_CHECK(gcnhip_graph_scales(graph, &dinv_row, &dinv2_row, nullptr, &dinv2_col));
            gs->fwd_scaling = 2; gs->bwd_scaling = 3;          // H1' = dropout(relu(dinv^2 . sum)) ; dH0' = raw sum (dW1 = X'^T . dH0')
            mm->da_row_scale = dinv2_row;                      // dH1' = dinv^2 . mask . (T . W2^T)
            mm->da_row_scale_full = dinv2_col;                 // rows of the gathered table (several GPUs: dH1 rebuilt for all of them)
        }
        // Opt-in (single GPU, hidden % 64 == 0, dropout >= 0.3 so that a 64-column half averages <= 22 values against
        // the slot's 30).  dH1 = mask . (dZ0 . W2^T) is ~3/4 zeros at positions known from H1: packed rows halve the
        // lines per edge of the backward gather with identical bits — but on gfx950 the unpacking (per column: rank,
        // LDS read, select, fma) costs more issue slots than the halved gather saves: 1.13 ms against 1.075 ms dense
        // at Reddit scale, so the dense gather stays the default.
        if (env.comm->size() == 1 && H % 64 == 0 && p >= 0.3f && !env.bf16_tables && (flags & HIPGCN_PACKED_DH1)) {
            GCNHIP_CHECK(gcnhip_rowpack_create(env.ctx, &dh1_pack, N, H));
            mm->da_pack = dh1_pack;
            gs->out_grad_pack = dh1_pack;
        }
        // single GPU: the ReLU/dropout mask of H1 leaves the aggregation's store epilogue as one bit per element and the
        // Matmul backward reads those instead of H1 (-119 MB per epoch at Reddit scale); HIPGCN_NO_MASK_BITS: re-read H1
        if (env.comm->size() == 1 && H % 32 == 0 && !env.bf16_tables && !dh1_pack && opt_.mask_bits) {
            const int wpr = H / 32;
            d_pos_bits = dev_upload(env.ctx, std::vector<uint32_t>((size_t)N * wpr, 0u).data(), (size_t)N * wpr);
            gs->mask_bits_out = d_pos_bits;
            mm->mask_bits = d_pos_bits; mm->mask_wpr = wpr;
        }
        if (rebuild_dh1) {
            const int wpr = (H + 31) / 32;
            d_pos_bits = dev_upload(env.ctx, std::vector<uint32_t>((size_t)xplan.table_rows * wpr, 0u).data(),
                                    (size_t)xplan.table_rows * wpr);
            gs->pos_bits_full = d_pos_bits; gs->wpr = wpr; gs->out_grad_complete = true;
            mm->pos_bits_full = d_pos_bits; mm->wpr = wpr; mm->all_rows = xplan.table_rows;
        }
        // Opt-in: measured at Reddit scale, 2 / 4 / 8 blocks: 254 / 240 / 237 epochs/s against 277 on one stream — the
        // split-K product next to the gather takes the gather's wave slots, and each block launch has its own tail.
        if ((flags & HIPGCN_BWD_PIPELINE) && !env.bf16_tables && !dh1_pack) build_bwd_pipeline(sm, gs);
        modules.push_back(sm);
        modules.push_back(gs);
        modules.push_back(mm);
        HipGraphSum *gs_logits = nullptr;
        {
            auto *gs = new HipGraphSum(&env, Z0, Z, graph, C);
            gs->bwd_row_bits = &bwd_bits; gs->bwd_graph = graph_bwd_out; gs->fwd_out_rows = &cur_out_rows;
            if (factored_) { gs->fwd_scaling = 1; gs->bwd_scaling = 3; }     // Z = dinv . sum(Z0') (the true logits); T = raw sum of dZ'
            wire_overlap(gs, true);
            modules.push_back(gs);
            gs_logits = gs;
        }
        auto *ce = new HipCrossEntropyLoss(&env, Z, &cur_truth, &cur_count, d_result, d_result_i, C, false);
        ce->rows_list = &cur_rows; ce->rows_n = &cur_rows_n;
        if (factored_) GCNHIP_CHECK(gcnhip_graph_scales(graph, &ce->grad_row_scale, nullptr, nullptr, nullptr));   // dZ' = dinv . dZ
        // the loss rides in the epilogue of the launch that produces the logits (f32 tables, at most 64 classes; the paths that
        // cut that launch in two — exchange overlap — or gather bf16 tables keep the loss kernel: HipGraphSum::forward decides)
        if (opt_.loss_epilogue && C <= 64 && !env.bf16_tables) {
            ce->row_terms = dev_upload(env.ctx, std::vector<float>((size_t)2 * std::max(N, 1), 0.f).data(), (size_t)2 * std::max(N, 1));
            gs_logits->loss = ce;
        }
        modules.push_back(ce);
    }
}

// Row blocks for the backward pipeline (module.h, BackwardPipeline): the split ranges of the dense weight gradient are cut
// into HIPGCN_BWD_CHUNKS (default 4) runs of whole splits, each registered as a row subset of the adjacency object.
void HipGCN::build_bwd_pipeline(HipSparseMatmul *sm, HipGraphSum *gs) {
    int rps = 0, n_splits = 0;
    GCNHIP_CHECK(gcnhip_spmm_bwd_plan(env.ctx, feat, params.hidden_dim, &rps, &n_splits));
    const int chunks = opt_.bwd_chunks;
    if (n_splits < 2 * chunks || chunks < 2) return;
    bwd_pipe.reset(new BackwardPipeline());
    BackwardPipeline &P = *bwd_pipe;
    GCNHIP_CHECK(gcnhip_ctx_create(&P.side, device_, nullptr));
    GCNHIP_CHECK(gcnhip_event_create_sync(&P.ev_done));
    for (int k = 0; k <= chunks; k++) P.cuts.push_back((int)((int64_t)n_splits * k / chunks));
    for (int k = 0; k < chunks; k++) {
        const int r_lo = std::min(n_local, P.cuts[k] * rps), r_hi = std::min(n_local, P.cuts[k + 1] * rps);
        std::vector<uint32_t> bits((size_t)n_local / 32 + 2, 0u);
        for (int r = r_lo; r < r_hi; r++) bits[r >> 5] |= 1u << (r & 31);
        gcnhip_rowset *rs = nullptr;
        GCNHIP_CHECK(gcnhip_graph_add_rowset(env.ctx, graph, bits.data(), &rs));
        P.blocks.push_back(rs);
        void *ev = nullptr;
        GCNHIP_CHECK(gcnhip_event_create_sync(&ev));
        P.ev_block.push_back(ev);
    }
    sm->pipe = &P;
    gs->pipe = &P;
    gs->pipe_consumer = sm;
}

void HipGCN::destroy_bwd_pipeline() {
    if (!bwd_pipe) return;
    BackwardPipeline &P = *bwd_pipe;
    if (P.side) gcnhip_ctx_sync(P.side);
    for (void *ev : P.ev_block) gcnhip_event_destroy(ev);
    if (P.ev_done) gcnhip_event_destroy(P.ev_done);
    if (P.side) gcnhip_ctx_destroy(P.side);
    bwd_pipe.reset();                       // the row subsets belong to the adjacency object
}

// A^.X for this rank's rows (needs every column of the adjacency and the matching rows of X: with several GPUs
// both are taken from the whole dataset once and released), then the evaluation module list that uses it.
void HipGCN::build_agg_first_eval() {
    const int world = env.comm->size(), rank = env.comm->rank();
    const int N = params.num_nodes, F = params.input_dim, H = params.hidden_dim;
    if (world == 1) {
        GCNHIP_CHECK(gcnhip_feat_create_aggregated(env.ctx, &feat_agg, graph, feat));
    } else {
        const std::vector<int> &gp = data->graph.indptr, &gi = data->graph.indices;
        const std::vector<int> &fp = data->feature_index.indptr, &fi = data->feature_index.indices;
        const int r0 = part.start[rank];
        gcnhip_graph *g_all = graph_l1;
        gcnhip_feat *x_all = feat_full;
        try {
            if (!g_all) {
                std::vector<int> lp(n_local + 1), deg(N);
                for (int r = 0; r <= n_local; r++) lp[r] = gp[r0 + r] - gp[r0];
                for (int j = 0; j < N; j++) deg[j] = gp[j + 1] - gp[j];
                GCNHIP_CHECK(gcnhip_graph_create(env.ctx, &g_all, lp.data(), gi.data() + gp[r0], n_local, N, deg.data()));
            }
            if (!x_all)
                GCNHIP_CHECK(gcnhip_feat_create(env.ctx, &x_all, fp.data(), fi.empty() ? nullptr : fi.data(), data->feature_value.data(), N, F));
            GCNHIP_CHECK(gcnhip_feat_create_aggregated(env.ctx, &feat_agg, g_all, x_all));
        } catch (...) {
            if (g_all && g_all != graph_l1) gcnhip_graph_destroy(env.ctx, g_all);
            if (x_all && x_all != feat_full) gcnhip_feat_destroy(env.ctx, x_all);
            throw;
        }
        if (g_all != graph_l1) gcnhip_graph_destroy(env.ctx, g_all);
        if (x_all != feat_full) gcnhip_feat_destroy(env.ctx, x_all);
    }
    agg_vals = gcnhip_feat_values(feat_agg);
    // H1 = ReLU((A^.X).W1) written straight into variable 3; from there on the training modules' own forward(false)
    auto *sm = new HipSparseMatmul(&env, &agg_vals, variables[2].get(), variables[3].get(), feat_agg, n_local, F, H, 0.f, 0);
    sm->relu_out = true;
    // nothing but H1.W2 reads an evaluation's hidden matrix: both products in one launch, H1 not stored (get_var(3) rebuilds it)
    if (opt_.eval_fusion && !env.bf16_tables) sm->fuse_next = dynamic_cast<HipMatmul *>(modules[2]);
    eval_modules.push_back(sm);
    for (size_t i = 2; i < modules.size(); i++) eval_modules.push_back(modules[i]);
}

void HipGCN::build_eval_lane() {
    const int world = env.comm->size(), rank = env.comm->rank();
    const int N = n_local, F = params.input_dim, H = params.hidden_dim, C = params.output_dim;
    lane.reset(new EvalLane());
    EvalLane &L = *lane;
    HipSparseMatmul *lane_sm = nullptr;
    GCNHIP_CHECK(gcnhip_ctx_create(&L.env.ctx, /*device of the main context*/ device_, nullptr));
    GCNHIP_CHECK(gcnhip_ctx_set_corun(L.env.ctx, 1));           // the lane's kernels share the chip with the training pass
    if (slice_floats == 32 || slice_floats == 16) GCNHIP_CHECK(gcnhip_ctx_set_option(L.env.ctx, "gs_l", slice_floats / 4));   // as on the training context
    if (opt_.gemm >= 0) GCNHIP_CHECK(gcnhip_ctx_set_option(L.env.ctx, "gemm_bf16x3", opt_.gemm ? 2 : 0));
    L.timers.reset(new DeviceTimers(L.env.ctx));
    L.timers->enabled = timers->enabled;
    L.env.timers = L.timers.get();
    L.comm.reset(env.comm->clone_for(L.env.ctx));
    L.env.comm = L.comm.get();
    L.env.plan = &xplan;
    L.env.xbuf = &L.xbuf;
    if (world > 1) exchange_buffers_create(L.env.ctx, xplan, xbuf.max_ld_words, &L.xbuf);
    L.env.seed = env.seed;
    L.env.bf16_tables = env.bf16_tables;
    void *q;
    GCNHIP_CHECK(gcnhip_malloc(L.env.ctx, &q, sizeof(uint32_t)));
    L.env.d_epoch = (uint32_t *)q;
    GCNHIP_CHECK(gcnhip_memset_async(L.env.ctx, q, 0xFF, sizeof(uint32_t)));
    GCNHIP_CHECK(gcnhip_malloc(L.env.ctx, &q, 4 * sizeof(float))); L.d_result = (float *)q;
    GCNHIP_CHECK(gcnhip_malloc(L.env.ctx, &q, 2 * sizeof(int32_t))); L.d_result_i = (int32_t *)q;
    // same adjacency, own scratch for split rows: a device-side clone of the training lane's object, in the row schedule that
    // lane measured as fastest (round 4: rebuilding it from the host lists was 0.67 s of a 1.8 s model build at Reddit scale)
    const std::vector<int> &gp = data->graph.indptr, &gi = data->graph.indices;
    GCNHIP_CHECK(gcnhip_graph_clone(L.env.ctx, &L.graph, graph));
    if (!(flags & HIPGCN_ALL_ROWS)) add_split_rowsets(L.env.ctx, L.graph, L.split_rows);
    const ExchangePlan *xp = world > 1 ? &xplan : nullptr;
    L.H0.reset(new HipVariable()); L.H1.reset(new HipVariable()); L.Z0.reset(new HipVariable()); L.Z.reset(new HipVariable());
    if (feat_agg) {
        // aggregate-first: the lane's hidden layer is one GEMM on A^.X — no H0, no hidden-width aggregation, no exchange
    } else if (replicate_l1) {
        std::vector<int> lp(N + 1), deg(params.num_nodes);
        const int r0 = part.start[rank];
        for (int r = 0; r <= N; r++) lp[r] = gp[r0 + r] - gp[r0];
        for (int j = 0; j < params.num_nodes; j++) deg[j] = gp[j + 1] - gp[j];
        GCNHIP_CHECK(gcnhip_graph_create(L.env.ctx, &L.graph_l1, lp.data(), gi.data() + gp[r0], N, params.num_nodes, deg.data()));
        GCNHIP_CHECK(gcnhip_graph_reserve_width(L.env.ctx, L.graph_l1, std::max(params.hidden_dim, params.output_dim)));
        apply_schedule(L.env.ctx, L.graph_l1);
        L.H0->alloc_replicated(L.env.ctx, params.num_nodes, N, r0, H, false);
    } else {
        L.H0->alloc(L.env.ctx, N, H, false, true, false, xp);
    }
    L.H1->alloc(L.env.ctx, N, H, false);
    L.Z0->alloc(L.env.ctx, N, C, false, true, false, xp);
    L.Z->alloc(L.env.ctx, N, C, false);
    const uint64_t nnz_off = (uint64_t)data->feature_index.indptr[part.start[rank]];
    eval_vals = gcnhip_feat_values(feat);
    if (feat_agg) {
        auto *sm = new HipSparseMatmul(&L.env, &agg_vals, variables[2].get(), L.H1.get(), feat_agg, N, F, H, 0.f, 0);
        sm->relu_out = true;
        lane_sm = sm;
        L.modules.push_back(sm);
    } else {
        auto *sm = new HipSparseMatmul(&L.env, &eval_vals, variables[2].get(), L.H0.get(), feat, N, F, H, 0.f, nnz_off);
        auto *gs = new HipGraphSum(&L.env, L.H0.get(), L.H1.get(), L.graph, H, 0.f, 0);   // ReLU epilogue, no dropout in eval
        if (factored_) gs->fwd_scaling = 2;
        if (replicate_l1) { sm->sp_full = feat_full; sm->vals_full = &full_vals; gs->fwd_graph_replicated = L.graph_l1; }
        L.modules.push_back(sm);
        L.modules.push_back(gs);
    }
    {
        auto *mm = new HipMatmul(&L.env, L.H1.get(), variables[5].get(), L.Z0.get(), N, H, C);
        if (lane_sm && opt_.eval_fusion && !L.env.bf16_tables) lane_sm->fuse_next = mm;      // as on the training context
        L.modules.push_back(mm);
    }
    auto *gs_logits = new HipGraphSum(&L.env, L.Z0.get(), L.Z.get(), L.graph, C);
    gs_logits->fwd_out_rows = &L.out_rows;
    if (factored_) gs_logits->fwd_scaling = 1;
    L.modules.push_back(gs_logits);
    {
        auto *ce = new HipCrossEntropyLoss(&L.env, L.Z.get(), &L.truth, &L.count, L.d_result, L.d_result_i, C, false);
        ce->rows_list = &L.rows; ce->rows_n = &L.rows_n;
        if (opt_.loss_epilogue && C <= 64 && !L.env.bf16_tables) {     // as on the training context
            ce->row_terms = dev_upload(L.env.ctx, std::vector<float>((size_t)2 * std::max(N, 1), 0.f).data(), (size_t)2 * std::max(N, 1));
            gs_logits->loss = ce;
        }
        L.modules.push_back(ce);
    }
    GCNHIP_CHECK(gcnhip_event_create_sync(&L.ev_weights));
    GCNHIP_CHECK(gcnhip_event_create_sync(&L.ev_done));
    GCNHIP_CHECK(gcnhip_event_create_sync(&L.ev_fork));
    GCNHIP_CHECK(gcnhip_ctx_sync(L.env.ctx));
}

// whatever of the lane exists (it may be half built when build_eval_lane threw)
void HipGCN::destroy_lane() {
    if (!lane) return;
    EvalLane &L = *lane;
    if (L.env.ctx) {
        gcnhip_ctx_sync(L.env.ctx);
        for (auto m : L.modules) delete m;
        L.modules.clear();
        L.H0.reset(); L.H1.reset(); L.Z0.reset(); L.Z.reset();
        if (L.graph) gcnhip_graph_destroy(L.env.ctx, L.graph);
        if (L.graph_l1) gcnhip_graph_destroy(L.env.ctx, L.graph_l1);
        gcnhip_free(L.env.ctx, L.d_result); gcnhip_free(L.env.ctx, L.d_result_i); gcnhip_free(L.env.ctx, L.env.d_epoch);
        if (L.ev_weights) gcnhip_event_destroy(L.ev_weights);
        if (L.ev_done) gcnhip_event_destroy(L.ev_done);
        if (L.ev_fork) gcnhip_event_destroy(L.ev_fork);
        L.timers.reset();
        exchange_buffers_destroy(&L.xbuf);
        L.comm.reset();
        gcnhip_ctx_destroy(L.env.ctx);
    }
    lane.reset();
}

HipGCN::~HipGCN() { release(); }

void HipGCN::release() {
    if (!env.ctx) return;
    gcnhip_ctx_sync(env.ctx);
    destroy_lane();
    if (!eval_modules.empty()) delete eval_modules[0];       // the rest are borrowed from `modules`
    eval_modules.clear();
    for (auto m : modules) delete m;
    modules.clear();
    // W1/W2 grads live in gradbuf (interior pointers: never freed through the variable)
    if (variables.size() == 7 && gradbuf) {
        if (variables[2]) variables[2]->grad = nullptr;
        if (variables[5]) variables[5]->grad = nullptr;
    }
    variables.clear();
    optimizer.reset();
    if (epoch_graph) { gcnhip_graph_exec_destroy(epoch_graph); epoch_graph = nullptr; }
    readback_destroy();
    destroy_bwd_pipeline();
    env.xlane = nullptr;
    xlane.reset();                                            // its communicator goes before the parent's
    for (gcnhip_graph *g : {graph_loc, graph_rem, graph_bwd_loc, graph_bwd_rem})
        if (g) gcnhip_graph_destroy(env.ctx, g);
    graph_loc = graph_rem = graph_bwd_loc = graph_bwd_rem = nullptr;
    if (graph_bwd_out) gcnhip_graph_destroy(env.ctx, graph_bwd_out);
    if (graph) gcnhip_graph_destroy(env.ctx, graph);
    if (feat) gcnhip_feat_destroy(env.ctx, feat);
    if (feat_full) gcnhip_feat_destroy(env.ctx, feat_full);
    if (feat_agg) gcnhip_feat_destroy(env.ctx, feat_agg);
    if (graph_l1) gcnhip_graph_destroy(env.ctx, graph_l1);
    for (int s = 1; s <= 3; s++) gcnhip_free(env.ctx, d_truth[s]);
    gcnhip_free(env.ctx, gradbuf);
    gcnhip_free(env.ctx, d_result_i);
    gcnhip_free(env.ctx, d_ring);
    gcnhip_free(env.ctx, env.d_epoch);
    gcnhip_free(env.ctx, d_keep0);
    gcnhip_free(env.ctx, d_keep1);
    gcnhip_free(env.ctx, d_train_bits);
    for (int s = 1; s <= 3; s++) gcnhip_free(env.ctx, d_split_list[s]);
    gcnhip_free(env.ctx, d_pos_bits);
    if (dh1_pack) gcnhip_rowpack_destroy(env.ctx, dh1_pack);
    timers.reset();
    exchange_buffers_destroy(&xbuf);
    owned_comm.reset();
    gcnhip_ctx_destroy(env.ctx);
    env.ctx = nullptr;
}

void HipGCN::sync() {
    if (xlane) GCNHIP_CHECK(gcnhip_ctx_sync(xlane->ctx));
    GCNHIP_CHECK(gcnhip_ctx_sync(env.ctx));
    if (lane) GCNHIP_CHECK(gcnhip_ctx_sync(lane->env.ctx));
}

double HipGCN::timer_total(timer_instance t, long *count) {
    long c0 = 0, c1 = 0, c2 = 0;
    double s = timers->total(t, &c0);
    if (lane) s += lane->timers->total(t, &c1);
    if (xlane) s += xlane->timers->total(t, &c2);             // exchanges on the exchange stream (TMR_COMM)
    if (count) *count = c0 + c1 + c2;
    return s;
}
void HipGCN::timers_reset() {
    timers->reset();
    if (lane) lane->timers->reset();
    if (xlane) xlane->timers->reset();
}

void HipGCN::set_timers(bool on) {
    sync();
    timers->enabled = on;
    if (lane) lane->timers->enabled = on;
    if (xlane) xlane->timers->enabled = on;
}

void HipGCN::set_truth(int s) {                 // gcn.cpp:78-81: here a pointer switch
    cur_truth = d_truth[s];
    cur_count = split_count[s];
    cur_out_rows = split_rows[s];
    cur_out_rows_loc = split_rows_loc[s];
    cur_out_rows_rem = split_rows_rem[s];
    cur_rows = d_split_list[s];
    cur_rows_n = split_local_n[s];
}

// replay the reference's RNG consumption for one training epoch: nnzX draws for
// the input dropout, then N*h draws for the hidden one (module.cpp:207-221;
// order fixed by the module list, gcn.cpp:23,42).  Every rank walks the whole
// stream and keeps its slice, so the decisions do not depend on the partition.
void HipGCN::host_masks_for_epoch() {
    const int thr = (int)(params.dropout * MY_RAND_MAX);
    const int rank = env.comm->rank();
    const long nnz_total = data->feature_index.indptr[params.num_nodes];
    const long f0 = keep0_first;
    for (long i = 0; i < nnz_total; i++) {
        const bool keep = (int)rng.next() >= thr;
        if (i >= f0 && i < f0 + (long)h_keep0.size()) h_keep0[i - f0] = keep;
    }
    const long H = params.hidden_dim, h0 = (long)part.start[rank] * H, total = (long)params.num_nodes * H;
    for (long i = 0; i < total; i++) {
        const bool keep = (int)rng.next() >= thr;
        if (i >= h0 && i < h0 + (long)h_keep1.size()) h_keep1[i - h0] = keep;
    }
    GCNHIP_CHECK(gcnhip_h2d(env.ctx, d_keep0, h_keep0.data(), h_keep0.size()));
    GCNHIP_CHECK(gcnhip_h2d(env.ctx, d_keep1, h_keep1.data(), h_keep1.size()));
}

// One GPU: the loss launch fills the metrics row itself (gcnhip_metrics_record_with_next_loss) — its result needs no
// all-reduce first.  HIPGCN_RECORD_LAUNCH (options.loss_records_metrics = false) keeps the separate launch (A/B).
#define loss_records() (opt_.loss_records_metrics)

void HipGCN::train_begin() {
    // (*env.d_epoch is this epoch's number already: the previous epoch's Adam launch advanced it)
    if (env.comm->size() == 1 && loss_records())   // loss/accuracy of this forward + the L2 term of the weights it uses
        GCNHIP_CHECK(gcnhip_metrics_record_with_next_loss(env.ctx, d_ring, RING, 0, env.d_epoch, optimizer->d_sumsq));
    epochs_done++;
    if (flags & HIPGCN_MODULAR)                  // set_input (gcn.cpp:73-76): device-to-device, never from the host
        GCNHIP_CHECK(gcnhip_d2d_async(env.ctx, input->data, gcnhip_feat_values(feat), (size_t)gcnhip_feat_nnz(feat) * sizeof(float)));
    if (flags & HIPGCN_HOST_MASKS) host_masks_for_epoch();
    set_truth(1);
}

void HipGCN::train_end() {
    if (env.comm->size() > 1) {
        timers->start(TMR_COMM);
        env.comm->allreduce_sum(gradbuf, gradbuf_elems);
        timers->stop(TMR_COMM);
    }
    // loss/accuracy of this forward + the L2 term of the weights it used, then the update
    if (!(env.comm->size() == 1 && loss_records()))
        GCNHIP_CHECK(gcnhip_metrics_record(env.ctx, d_ring, RING, 0, env.d_epoch, d_result, nullptr, optimizer->d_sumsq));
    if (lane && lane->pending) {                    // the previous validation pass still reads W1, W2 and the L2 term
        GCNHIP_CHECK(gcnhip_stream_wait_event(env.ctx, lane->ev_done));
        lane->pending = false;
    }
    optimizer->step();
    if (lane) GCNHIP_CHECK(gcnhip_event_record(env.ctx, lane->ev_weights));
}

void HipGCN::train_epoch_async() {              // gcn.cpp:107-118
    h1_from_fused_eval = false;                 // the training forward stores H1
    train_begin();
    for (auto m : modules) m->forward(true);
    for (int i = (int)modules.size() - 1; i >= 0; i--) modules[i]->backward();
    train_end();
}

// validation forward of the epoch that just finished, on the second stream
void HipGCN::lane_begin(int s) {
    EvalLane &L = *lane;
    GCNHIP_CHECK(gcnhip_stream_wait_event(L.env.ctx, L.ev_weights));
    // the lane's epoch word names the ring row: the epoch whose weights are evaluated, whatever was called in between
    const long want = epochs_done - 1;
    GCNHIP_CHECK(gcnhip_counter_add(L.env.ctx, L.env.d_epoch, (uint32_t)(want - L.epoch_word)));
    L.epoch_word = want;
    L.truth = d_truth[s];
    L.count = split_count[s];
    L.out_rows = L.split_rows[s];
    L.rows = d_split_list[s];
    L.rows_n = split_local_n[s];
    if (L.env.comm->size() == 1 && loss_records())
        GCNHIP_CHECK(gcnhip_metrics_record_with_next_loss(L.env.ctx, d_ring, RING, s == 2 ? 1 : 2, L.env.d_epoch, optimizer->d_sumsq));
}

void HipGCN::lane_end(int s) {
    EvalLane &L = *lane;
    if (L.env.comm->size() > 1) {
        L.timers->start(TMR_COMM);
        L.env.comm->allreduce_sum(L.d_result, 4);
        L.timers->stop(TMR_COMM);
    }
    if (!(L.env.comm->size() == 1 && loss_records()))
        GCNHIP_CHECK(gcnhip_metrics_record(L.env.ctx, d_ring, RING, s == 2 ? 1 : 2, L.env.d_epoch, L.d_result, nullptr, optimizer->d_sumsq));
    GCNHIP_CHECK(gcnhip_event_record(L.env.ctx, L.ev_done));
    L.pending = true;
}

void HipGCN::eval_on_lane(int s) {
    lane_begin(s);
    for (auto m : lane->modules) m->forward(false);
    lane_end(s);
}

// eval(e) on the lane and train(e+1) on the main stream, enqueued stage by stage in alternation.  Both need only
// the weights Adam(e) wrote.  The collectives of the two communicators execute in enqueue order (comm.cpp,
// turnstile), so enqueueing all of eval(e) first would make train(e+1)'s first all-gather wait for the whole
// validation pass; zipped, each collective waits only for the other lane's previous one, and the lanes' compute
// overlaps with each other's exchanges.
void HipGCN::eval_then_train_zipped(int s) {
    lane_begin(s);
    train_begin();
    const size_t nb = lane->modules.size(), na = modules.size();
    if (env.comm->size() == 1) {
        // One GPU: nothing to exchange, so the point of the second stream is to run kernels with different bottlenecks
        // side by side.  Both passes start with the same MFMA-bound GEMM; the validation pass is therefore released
        // only when the training GEMM has finished, and its GEMM then shares the chip with the gather-bound
        // hidden-width aggregation of the training pass.
        modules[0]->forward(true);
        GCNHIP_CHECK(gcnhip_event_record(env.ctx, lane->ev_fork));
        GCNHIP_CHECK(gcnhip_stream_wait_event(lane->env.ctx, lane->ev_fork));
        for (size_t i = 0; i < nb; i++) lane->modules[i]->forward(false);
        lane_end(s);
        for (size_t i = 1; i < na; i++) modules[i]->forward(true);
        for (int i = (int)na - 1; i >= 0; i--) modules[i]->backward();
        train_end();
        return;
    }
    for (size_t i = 0; i < std::max(na, nb); i++) {
        if (i < nb) lane->modules[i]->forward(false);
        if (i < na) modules[i]->forward(true);
    }
    lane_end(s);
    for (int i = (int)na - 1; i >= 0; i--) modules[i]->backward();
    train_end();
}

void HipGCN::eval_async(int s) {                // gcn.cpp:120-128
    if (flags & HIPGCN_MODULAR)
        GCNHIP_CHECK(gcnhip_d2d_async(env.ctx, input->data, gcnhip_feat_values(feat), (size_t)gcnhip_feat_nnz(feat) * sizeof(float)));
    set_truth(s);
    const bool in_loss = env.comm->size() == 1 && loss_records();
    // (an evaluation on this stream scores the weights of the last update: the row of env.d_epoch_done, not of the epoch to come)
    if (in_loss) GCNHIP_CHECK(gcnhip_metrics_record_with_next_loss(env.ctx, d_ring, RING, s == 2 ? 1 : 2, env.d_epoch_done, optimizer->d_sumsq));
    for (auto m : eval_modules.empty() ? modules : eval_modules) m->forward(false);
    h1_from_fused_eval = !eval_modules.empty() && static_cast<HipSparseMatmul *>(eval_modules[0])->hidden_not_stored;
    if (env.comm->size() > 1) {
        timers->start(TMR_COMM);
        env.comm->allreduce_sum(d_result, 4);
        timers->stop(TMR_COMM);
    }
    if (!in_loss) GCNHIP_CHECK(gcnhip_metrics_record(env.ctx, d_ring, RING, s == 2 ? 1 : 2, env.d_epoch_done, d_result, nullptr, optimizer->d_sumsq));
}

std::pair<float, float> HipGCN::read_metrics(long epoch_index, int slot) {
    float row[8];
    if (lane) GCNHIP_CHECK(gcnhip_ctx_sync(lane->env.ctx));
    const uint32_t e = (uint32_t)epoch_index;   // epoch_index == -1 (eval before any training) wraps like the device word
    GCNHIP_CHECK(gcnhip_d2h(env.ctx, row, d_ring + ((size_t)(e % RING) * 4 + slot) * 8, sizeof row));
    const float loss = row[0] / (int)row[1];                                    // module.cpp:154
    const float l2 = params.weight_decay * row[4] / 2;                          // gcn.cpp:104
    const float acc = (float)row[2] / (int)row[3];                              // gcn.cpp:95
    return {loss + l2, acc};
}

std::pair<float, float> HipGCN::train_epoch() {
    train_epoch_async();
    return read_metrics(epochs_done - 1, 0);
}

std::pair<float, float> HipGCN::eval(int s) {
    eval_async(s);
    return read_metrics(epochs_done - 1, s == 2 ? 1 : 2);
}

// One epoch (train + validation) is a fixed launch sequence whose epoch-dependent inputs all live in device memory, so it
// is captured once into a hipGraph and replayed (single GPU, one stream, device RNG, no per-op timers).  The first epoch
// runs eagerly so every scratch buffer has its final size.
bool HipGCN::enqueue_epoch_replay() {
    const bool graph_ok = env.comm->size() == 1 && !lane && !timers->enabled && !(flags & (HIPGCN_HOST_MASKS | HIPGCN_NO_GRAPH));
    if (!(graph_ok && epochs_done >= 1 && optimizer->can_replay(1))) return false;
    if (!epoch_graph) {
        const long epochs_before = epochs_done;
        const int steps_before = optimizer->steps();
        GCNHIP_CHECK(gcnhip_capture_begin(env.ctx));
        try {
            train_epoch_async();
            eval_async(2);
        } catch (...) {
            // never leave the stream capturing: end the capture, drop whatever it recorded, restore
            // the host-side counters, then let the caller see the failure
            void *broken = nullptr;
            gcnhip_capture_end(env.ctx, &broken);
            if (broken) gcnhip_graph_exec_destroy(broken);
            epochs_done = epochs_before;
            optimizer->note_replayed(steps_before - optimizer->steps());
            throw;
        }
        GCNHIP_CHECK(gcnhip_capture_end(env.ctx, &epoch_graph));
        // the capture only recorded: undo its host-side bookkeeping, then run it for real
        epochs_done = epochs_before;
        optimizer->note_replayed(steps_before - optimizer->steps());
    }
    GCNHIP_CHECK(gcnhip_graph_launch(env.ctx, epoch_graph));
    epochs_done++;
    optimizer->note_replayed(1);
    return true;
}

void HipGCN::run_epochs(int n, float *trace) {
    int done = 0;
    while (done < n) {
        const int chunk = std::min(n - done, RING);
        const long first = epochs_done;
        for (int i = 0; i < chunk; i++) {
            if (enqueue_epoch_replay()) {
            } else if (lane && !(timers->enabled && env.comm->size() == 1)) {
                // (one GPU with per-op timers on: one stream, so that every launch is timed alone — the branch below)
                // validation of epoch i-1 zipped with training of epoch i; the chunk's last validation runs alone
                if (i == 0) train_epoch_async(); else eval_then_train_zipped(2);
                if (i == chunk - 1) eval_on_lane(2);
            } else {
                train_epoch_async();
                eval_async(2);
            }
        }
        sync();
        if (trace) {
            std::vector<float> ring((size_t)RING * 32);
            GCNHIP_CHECK(gcnhip_d2h(env.ctx, ring.data(), d_ring, ring.size() * sizeof(float)));
            for (int i = 0; i < chunk; i++) {
                const uint32_t e = (uint32_t)(first + i);
                for (int slot = 0; slot < 2; slot++) {
                    const float *row = &ring[((size_t)(e % RING) * 4 + slot) * 8];
                    trace[(size_t)(done + i) * 4 + slot * 2] = row[0] / (int)row[1] + params.weight_decay * row[4] / 2;
                    trace[(size_t)(done + i) * 4 + slot * 2 + 1] = (float)row[2] / (int)row[3];
                }
            }
        }
        done += chunk;
    }
}

void HipGCN::readback_create(bool own_stream) {
    if (readback && (readback->ctx != nullptr) != own_stream) readback_destroy();
    if (readback) return;
    readback.reset(new Readback());
    Readback &R = *readback;
    if (own_stream) GCNHIP_CHECK(gcnhip_ctx_create(&R.ctx, device_, nullptr));
    void *q = nullptr;
    GCNHIP_CHECK(gcnhip_host_alloc(&q, (size_t)PIPELINE_DEPTH * READBACK_GROUP_MAX * 32 * sizeof(float)));
    R.host = (float *)q;
    for (int k = 0; k < PIPELINE_DEPTH; k++) {
        if (own_stream) GCNHIP_CHECK(gcnhip_event_create_sync(&R.ev_ready[k]));
        GCNHIP_CHECK(gcnhip_event_create_sync(&R.ev_copied[k]));
    }
}

void HipGCN::readback_destroy() {
    if (!readback) return;
    Readback &R = *readback;
    if (R.ctx) gcnhip_ctx_sync(R.ctx);
    for (int k = 0; k < PIPELINE_DEPTH; k++) {
        if (R.ev_ready[k]) gcnhip_event_destroy(R.ev_ready[k]);
        if (R.ev_copied[k]) gcnhip_event_destroy(R.ev_copied[k]);
    }
    gcnhip_host_free(R.host);
    if (R.ctx) gcnhip_ctx_destroy(R.ctx);
    readback.reset();
}

// everything of epochs e .. e+n-1 (0-based) has been enqueued, the last validation pass on `producer`: behind it, their
// ring rows (32 floats each; slots 0 = train and 1 = validation are read) are copied to slot k of the pinned buffer.  The
// ring wraps at RING rows: at most two copies.  The copy goes on `producer` itself.  With a stream of its own for the
// read-back (round 4's first version) the copy leaves the producer's timeline, but a barrier packet then sits on a second
// hardware queue for as long as the group runs, and the producer's own launches slow down beside it: gcn-hip per epoch,
// own stream -> producer's stream: Cora 102 -> 81 us (captured epoch replayed), Pubmed 124 -> 114 us and Reddit
// 3329 -> 3291 us (validation lane); profiles/r04_cli_run_loop.json, DESIGN.md §4.10.
void HipGCN::readback_enqueue(long e, int n, int k, gcnhip_ctx *producer) {
    Readback &R = *readback;
    gcnhip_ctx *on = producer;
    if (R.ctx) {
        GCNHIP_CHECK(gcnhip_event_record(producer, R.ev_ready[k]));
        GCNHIP_CHECK(gcnhip_stream_wait_event(R.ctx, R.ev_ready[k]));
        on = R.ctx;
    }
    float *dst = R.host + (size_t)k * READBACK_GROUP_MAX * 32;
    const int r0 = (int)((uint32_t)e % RING), n1 = std::min(n, RING - r0);
    GCNHIP_CHECK(gcnhip_d2h_async(on, dst, d_ring + (size_t)r0 * 32, (size_t)n1 * 32 * sizeof(float)));
    if (n > n1) GCNHIP_CHECK(gcnhip_d2h_async(on, dst + (size_t)n1 * 32, d_ring, (size_t)(n - n1) * 32 * sizeof(float)));
    GCNHIP_CHECK(gcnhip_event_record(on, R.ev_copied[k]));
}

void HipGCN::run() {                            // gcn.cpp:130-158
    if (params.early_stopping > 0 || (flags & HIPGCN_SYNC_EPOCHS) || params.epochs < 1) run_synchronous();
    else run_pipelined();
    report_test();
}

// No printed number feeds back into the run (no early stopping): epochs are enqueued ahead of the line being printed, and
// their metrics come back in GROUPS of consecutive epochs — one event wait and one small copy per group, up to
// PIPELINE_DEPTH groups in flight.  The first READBACK_CALIBRATION epochs go one per group; their steady completion rate
// then sets the group size so that a group spans about READBACK_GROUP_SECONDS (1 on Reddit-size graphs, where an epoch is
// milliseconds; 16 on Cora / Pubmed, whose ~100 us epochs would otherwise spend as long on the host's per-epoch event and
// copy calls as on the device).  Lines of a group are printed together, each with time= the group's interval / its size.
// With the validation lane, eval(e) is zipped with train(e+1) exactly as in run_epochs.
void HipGCN::run_pipelined() {
    const bool zipped = lane && !(timers->enabled && env.comm->size() == 1);
    {   // the read-back copies ride on the producer's stream; HIPGCN_READBACK_STREAM=1 gives them their own (measured
        // slower at every size, see readback_enqueue; kept so that the measurement can be repeated)
        readback_create(opt_.readback_stream);
    }
    Readback &R = *readback;
    const bool talk = env.comm->rank() == 0;
    const long E = params.epochs, first = epochs_done;
    long enq = 0, evald = 0, grouped = 0, printed = 0;   // epochs (of this run) with: training pass enqueued / validation pass
                                                         // enqueued / a read-back enqueued / their line out
    struct Group { long e0; int n, slot; };
    std::deque<Group> inflight;
    long n_groups = 0;
    int group = 1;                                       // epochs per read-back group
    bool group_settled = false;                          // pinned by HIPGCN_READBACK_GROUP, or set after the calibration epochs
    if (opt_.readback_group > 0) {
        group = std::max(1, std::min(opt_.readback_group, (int)READBACK_GROUP_MAX));
        group_settled = true;
    }
    double total_train = 0, calib = 0;
    auto t_prev = std::chrono::high_resolution_clock::now();
    const bool verbose = opt_.verbose;
    double host_enqueue_s = 0, host_wait_s = 0;
    while (printed < E) {
        const auto t_enq0 = std::chrono::high_resolution_clock::now();
        while ((int)inflight.size() < PIPELINE_DEPTH && grouped < E) {
            const int n = (int)std::min<long>(group, E - grouped);
            while (evald < grouped + n) {
                if (zipped) {
                    if (enq == 0) { train_epoch_async(); enq = 1; }
                    if (enq < E) { eval_then_train_zipped(2); enq++; }       // eval(evald) beside train(evald + 1)
                    else eval_on_lane(2);
                } else {
                    if (!enqueue_epoch_replay()) { train_epoch_async(); eval_async(2); }
                    enq++;
                }
                evald++;
            }
            const int slot = (int)(n_groups++ % PIPELINE_DEPTH);
            readback_enqueue(first + grouped, n, slot, zipped ? lane->env.ctx : env.ctx);
            inflight.push_back({grouped, n, slot});
            grouped += n;
        }
        const Group g = inflight.front();
        inflight.pop_front();
        const auto t_wait0 = std::chrono::high_resolution_clock::now();
        GCNHIP_CHECK(gcnhip_event_sync(R.ev_copied[g.slot]));
        const auto t_now = std::chrono::high_resolution_clock::now();
        host_enqueue_s += std::chrono::duration<double>(t_wait0 - t_enq0).count();
        host_wait_s += std::chrono::duration<double>(t_now - t_wait0).count();
        const double dt_group = std::chrono::duration_cast<std::chrono::duration<double>>(t_now - t_prev).count();
        t_prev = t_now;
        total_train += dt_group;
        const float dt = (float)(dt_group / g.n);
        for (int i = 0; i < g.n; i++) {
            const float *tr = R.host + ((size_t)g.slot * READBACK_GROUP_MAX + i) * 32, *va = tr + 8;
            const float train_loss = tr[0] / (int)tr[1] + params.weight_decay * tr[4] / 2, train_acc = (float)tr[2] / (int)tr[3];
            const float val_loss = va[0] / (int)va[1] + params.weight_decay * va[4] / 2, val_acc = (float)va[2] / (int)va[3];
            if (talk)
                printf("epoch=%ld train_loss=%.5f train_acc=%.5f val_loss=%.5f val_acc=%.5f time=%.5f\n",
                       g.e0 + i + 1, train_loss, train_acc, val_loss, val_acc, dt);
        }
        printed += g.n;
        if (!group_settled && printed > READBACK_CALIBRATION / 2 && printed <= READBACK_CALIBRATION) calib += dt_group;
        if (!group_settled && printed == READBACK_CALIBRATION) {      // (groups of one until here: printed counts epochs one by one)
            const double per_epoch = calib / (READBACK_CALIBRATION / 2);
            const int want = per_epoch > 0 ? (int)(READBACK_GROUP_SECONDS / per_epoch) : 1;
            group = 1;
            while (group * 2 <= want && group * 2 <= READBACK_GROUP_MAX) group *= 2;
            group_settled = true;
            if (verbose && talk) fprintf(stderr, "[hipgcn] read-back groups of %d epochs (%.1f us per epoch while calibrating)\n", group, 1e6 * per_epoch);
        }
    }
    sync();
    if (talk) printf("total training time=%.5f\n", (float)total_train);
    if (verbose && talk)
        fprintf(stderr, "[hipgcn] run loop: %.1f us per epoch enqueueing, %.1f us per epoch waiting for read-backs\n", 1e6 * host_enqueue_s / E, 1e6 * host_wait_s / E);
}

void HipGCN::run_synchronous() {
    int epoch = 1;
    std::vector<float> loss_history;
    double total_train = 0;
    const bool talk = env.comm->rank() == 0;
    for (; epoch <= params.epochs; epoch++) {
        float train_loss, train_acc, val_loss, val_acc;
        auto t0 = std::chrono::high_resolution_clock::now();
        train_epoch_async();
        if (lane) eval_on_lane(2); else eval_async(2);
        std::tie(train_loss, train_acc) = read_metrics(epochs_done - 1, 0);     // one synchronisation per epoch
        std::tie(val_loss, val_acc) = read_metrics(epochs_done - 1, 1);
        const float dt = std::chrono::duration_cast<std::chrono::duration<float>>(std::chrono::high_resolution_clock::now() - t0).count();
        total_train += dt;
        if (talk)
            printf("epoch=%d train_loss=%.5f train_acc=%.5f val_loss=%.5f val_acc=%.5f time=%.5f\n",
                   epoch, train_loss, train_acc, val_loss, val_acc, dt);
        loss_history.push_back(val_loss);
        if (params.early_stopping > 0 && epoch >= params.early_stopping) {
            float recent_loss = 0.0;
            for (int i = epoch - params.early_stopping; i < epoch; i++) recent_loss += loss_history[i];
            if (val_loss > recent_loss / params.early_stopping) {
                if (talk) printf("Early stopping...\n");
                break;
            }
        }
    }
    if (talk) printf("total training time=%.5f\n", (float)total_train);
}

void HipGCN::report_test() {
    const bool talk = env.comm->rank() == 0;
    float test_loss, test_acc;
    auto t0 = std::chrono::high_resolution_clock::now();
    std::tie(test_loss, test_acc) = eval(3);
    const float dt = std::chrono::duration_cast<std::chrono::duration<float>>(std::chrono::high_resolution_clock::now() - t0).count();
    if (talk) printf("test_loss=%.5f test_acc=%.5f time=%.5f\n", test_loss, test_acc, dt);
}

void HipGCN::get_var(int k, bool grad, std::vector<float> &out, int *rows, int *cols) {
    if (k < 1 || k > 6) throw GcnHipFailure(-1, "get_var: k must be 1..6");
    HipVariable *v = variables[k].get();
    if (k == 3 && !grad && h1_from_fused_eval && !eval_modules.empty()) {
        // the last forward on this stream was an evaluation whose hidden matrix stayed in registers: run it as its own launch
        static_cast<HipSparseMatmul *>(eval_modules[0])->forward_stored();
        h1_from_fused_eval = false;
        sync();
    }
    if (k == 3 && grad && dh1_pack) {           // introspection: rebuild the dense image of the packed gradient
        GCNHIP_CHECK(gcnhip_rowpack_expand(env.ctx, dh1_pack, v->grad, v->ld));
        sync();
    }
    out.resize((size_t)v->rows * v->cols);
    v->download(out.data(), grad);
    if (rows) *rows = v->rows;
    if (cols) *cols = v->cols;
}

void HipGCN::set_weights(const float *w1, const float *w2) {
    variables[2]->upload(w1);
    variables[5]->upload(w2);
    GCNHIP_CHECK(gcnhip_sumsq(env.ctx, variables[2]->data, (int64_t)variables[2]->elems(), optimizer->d_sumsq));
    sync();
}
