// partition.h — contiguous 1-D row partition of the adjacency across `world`
// GPUs, balanced by work (edges + a per-row constant), plus the column remap
// into the padded all-gather layout.  Pure host logic (tested on CPU).
//
// Rank q owns global rows [start[q], start[q+1]).  Gathered matrices are laid
// out as `world` blocks of rows_max rows each, so global row j owned by rank q
// sits at padded row q*rows_max + (j - start[q]) on every rank; ncclAllGather
// needs equal block sizes, hence the padding to the largest block.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

struct RowPartition {
    int world = 1;
    std::vector<int> start;      // [world+1]
    int rows_max = 0;
    int owner(int row) const { return (int)(std::upper_bound(start.begin(), start.end(), row) - start.begin()) - 1; }
    int padded(int row) const { const int q = owner(row); return q * rows_max + (row - start[q]); }
    int rows(int q) const { return start[q + 1] - start[q]; }
};

// cost(row) = degree(row) + row_cost; boundaries at equal shares of the prefix sum
inline RowPartition make_partition(const int *indptr, int n_rows, int world, double row_cost = -1.0) {
    RowPartition p;
    p.world = world;
    p.start.assign(world + 1, 0);
    p.start[world] = n_rows;
    if (row_cost < 0) row_cost = n_rows > 0 ? (double)indptr[n_rows] / n_rows : 0.0;   // = mean degree
    const double total = (double)indptr[n_rows] + row_cost * n_rows;
    int r = 0;
    for (int q = 1; q < world; q++) {
        const double target = total * q / world;
        while (r < n_rows && (double)indptr[r + 1] + row_cost * (r + 1) <= target) r++;
        p.start[q] = std::max(r, p.start[q - 1]);
    }
    for (int q = 0; q < world; q++) p.rows_max = std::max(p.rows_max, p.rows(q));
    if (p.rows_max < 1) p.rows_max = 1;
    return p;
}

// This rank's row block of the adjacency with its column indices rewritten to
// padded all-gather positions, plus the degree of every padded column (the
// edge coefficient needs the GLOBAL degree of the neighbour, module.cpp:92).
struct LocalGraph {
    std::vector<int> indptr, indices, col_deg;
    int n_rows = 0, n_cols = 0;
};

inline LocalGraph build_local_graph(const int *gp, const int *gi, int n_rows, const RowPartition &part, int rank) {
    LocalGraph lg;
    const int r0 = part.start[rank], r1 = part.start[rank + 1];
    lg.n_rows = r1 - r0;
    lg.n_cols = part.world * part.rows_max;
    lg.indptr.resize(lg.n_rows + 1);
    for (int r = 0; r <= lg.n_rows; r++) lg.indptr[r] = gp[r0 + r] - gp[r0];
    std::vector<int> pad(n_rows);
    lg.col_deg.assign((size_t)lg.n_cols, 1);
    for (int q = 0; q < part.world; q++)
        for (int j = part.start[q]; j < part.start[q + 1]; j++) {
            pad[j] = q * part.rows_max + (j - part.start[q]);
            lg.col_deg[pad[j]] = gp[j + 1] - gp[j];
        }
    const long nnz = (long)gp[r1] - gp[r0];
    lg.indices.resize((size_t)nnz);
    for (long e = 0; e < nnz; e++) lg.indices[e] = pad[gi[gp[r0] + e]];
    return lg;
}

// ---------------------------------------------------------------------------------------------------------------
// What a rank holds of a matrix that an aggregation gathers from ("table"), and how the missing rows arrive.
//   ALLGATHER: world blocks of rows_max rows (the padded layout above); every rank receives every block with one
//              in-place ncclAllGather.  Right when a rank's columns cover (nearly) the whole graph — reddit-syn:
//              even whole-community row blocks need 91-100 % of the remote rows.
//   HALO     : [this rank's rows | the rows it needs from peer 0 | from peer 1 | ...] — only rows some local edge
//              points at.  Each rank packs, per peer, the rows that peer needs (send lists) and the pieces move
//              point to point (grouped ncclSend/ncclRecv straight into the table segments).  R-MAT ids carry the
//              generator's quadrant locality: at 8 ranks a row block needs a third of the remote rows.
// Every rank derives the plan of ALL ranks from the whole adjacency (which each holds), so the mode decision and
// the send lists agree everywhere without communication.
struct ExchangePlan {
    int world = 1, rank = 0;
    bool halo = false;
    int n_local = 0;
    int rows_max = 0;               // ALLGATHER block size
    int table_rows = 0;             // rows of a gathered table on this rank
    int own_offset = 0;             // table row of this rank's first row
    // HALO: segment of peer q = table rows [n_local + recv_off[q], n_local + recv_off[q+1]); recv_rows = the
    // peer-local row ids in that segment (ascending); send_rows[send_off[q] .. send_off[q+1]) = local rows peer q needs
    std::vector<int> recv_off, recv_rows, send_off, send_rows;
    std::vector<int> table_global;  // global node id of every table row (-1: padding)
    double halo_share = 1.0;        // max over ranks of (needed remote rows / remote rows): what the mode was decided on
    long recv_total() const { return halo ? (long)recv_rows.size() : (long)(world - 1) * rows_max; }
    long send_total() const { return halo ? (long)send_rows.size() : (long)n_local; }
};

// mode: 0 = decide (HALO when every rank needs at most `halo_below` of the remote rows), 1 = ALLGATHER, 2 = HALO
inline ExchangePlan make_exchange_plan(const int *gp, const int *gi, int n_rows, const RowPartition &part, int rank,
                                       int mode = 0, double halo_below = 0.75) {
    ExchangePlan x;
    const int P = part.world;
    x.world = P; x.rank = rank;
    x.n_local = part.rows(rank);
    x.rows_max = part.rows_max;
    // need[p] = sorted global ids owned by others that rows of p point at.  One pass over the edges per rank block.
    std::vector<std::vector<int>> need(P);
    std::vector<uint8_t> mark((size_t)n_rows);
    double worst = 0.0;
    for (int p = 0; p < P && P > 1; p++) {
        std::fill(mark.begin(), mark.end(), 0);
        const int a = part.start[p], b = part.start[p + 1];
        for (long e = gp[a]; e < gp[b]; e++) mark[gi[e]] = 1;
        long cnt = 0;
        for (int j = 0; j < n_rows; j++)
            if (mark[j] && (j < a || j >= b)) { cnt++; if (p == rank) need[p].push_back(j); else if (j >= part.start[rank] && j < part.start[rank + 1]) need[p].push_back(j); }
        const long remote = (long)n_rows - (b - a);
        if (remote > 0) worst = std::max(worst, (double)cnt / (double)remote);
    }
    x.halo_share = P > 1 ? worst : 0.0;
    x.halo = P > 1 && (mode == 2 || (mode == 0 && worst <= halo_below));
    if (!x.halo) {
        x.table_rows = P * part.rows_max;
        x.own_offset = rank * part.rows_max;
        x.table_global.assign((size_t)x.table_rows, -1);
        for (int q = 0; q < P; q++)
            for (int j = part.start[q]; j < part.start[q + 1]; j++) x.table_global[(size_t)q * part.rows_max + (j - part.start[q])] = j;
        return x;
    }
    // this rank's table: own rows, then per peer the needed rows in ascending id order
    x.own_offset = 0;
    x.recv_off.assign(P + 1, 0);
    x.send_off.assign(P + 1, 0);
    const int r0 = part.start[rank];
    x.table_global.resize((size_t)x.n_local);
    for (int r = 0; r < x.n_local; r++) x.table_global[r] = r0 + r;
    {
        const std::vector<int> &mine = need[rank];       // ascending ids => grouped by owner in rank order
        size_t k = 0;
        for (int q = 0; q < P; q++) {
            x.recv_off[q] = (int)x.recv_rows.size();
            while (k < mine.size() && mine[k] < part.start[q + 1]) {
                if (q != rank) { x.recv_rows.push_back(mine[k] - part.start[q]); x.table_global.push_back(mine[k]); }
                k++;
            }
        }
        x.recv_off[P] = (int)x.recv_rows.size();
    }
    for (int q = 0; q < P; q++) {
        x.send_off[q] = (int)x.send_rows.size();
        if (q != rank)
            for (int j : need[q]) x.send_rows.push_back(j - r0);   // need[q] (q != rank) holds only ids this rank owns
    }
    x.send_off[P] = (int)x.send_rows.size();
    x.table_rows = x.n_local + (int)x.recv_rows.size();
    return x;
}

// This rank's row block with columns rewritten to table rows of `plan` (either mode) + the degree of every table row
inline LocalGraph build_table_graph(const int *gp, const int *gi, int n_rows, const RowPartition &part, const ExchangePlan &plan) {
    LocalGraph lg;
    const int rank = plan.rank, r0 = part.start[rank], r1 = part.start[rank + 1];
    lg.n_rows = r1 - r0;
    lg.n_cols = plan.table_rows;
    lg.indptr.resize(lg.n_rows + 1);
    for (int r = 0; r <= lg.n_rows; r++) lg.indptr[r] = gp[r0 + r] - gp[r0];
    std::vector<int> pos((size_t)n_rows, -1);
    lg.col_deg.assign((size_t)std::max(lg.n_cols, 1), 1);
    for (int t = 0; t < plan.table_rows; t++) {
        const int j = plan.table_global[t];
        if (j < 0) continue;
        pos[j] = t;
        lg.col_deg[t] = gp[j + 1] - gp[j];
    }
    const long nnz = (long)gp[r1] - gp[r0];
    lg.indices.resize((size_t)nnz);
    for (long e = 0; e < nnz; e++) lg.indices[e] = pos[gi[gp[r0] + e]];   // never -1: the plan covers every column of the block
    return lg;
}

// ---------------------------------------------------------------------------------------------------------------
// Rank blocks by STRUCTURE when the node ids carry no locality.  Rank blocks are contiguous ranges of the node order, so
// which rows a rank must fetch depends on that order: R-MAT ids from the generator put the hubs first and give the other
// blocks few distinct neighbours (38 % of the remote rows at 8 ranks), the same graph with shuffled ids needs far more.
// An order is a permutation `order[new] = old`; the model is renumbered with it once, before anything is partitioned
// (the model is permutation-invariant; only which element gets which dropout decision changes with the numbering).
// Candidates, each a function of the graph alone (every rank computes the same):
//   * descending degree — recovers the hub-first layout of a scale-free graph whatever its ids are;
//   * group-major over groups found in the graph (cluster.h: modularity local moving), descending degree inside a group —
//     for graphs made of communities.
// exchange_cost() prices an order at `world` ranks by the rows the neediest rank receives per exchange (what a HALO plan
// moves; an ALLGATHER plan moves (world - 1) * rows_max).
struct OrderCost {
    double halo_share = 1.0;       // max over ranks: needed remote rows / remote rows
    long recv_rows_max = 0;        // ... needed remote rows of the neediest rank
    long rows_max = 0;
};

// the CSR renumbered with order[new] = old (self loop first in every row, the other neighbours in their old order)
inline void permute_csr(const int *gp, const int *gi, int n, const std::vector<int> &order, std::vector<int> &np_, std::vector<int> &ni_) {
    std::vector<int> inv((size_t)n);
    for (int k = 0; k < n; k++) inv[order[k]] = k;
    np_.assign((size_t)n + 1, 0);
    for (int k = 0; k < n; k++) np_[k + 1] = np_[k] + (gp[order[k] + 1] - gp[order[k]]);
    ni_.resize((size_t)np_[n]);
    for (int k = 0; k < n; k++) {
        const int o = order[k];
        int w = np_[k];
        for (int e = gp[o]; e < gp[o + 1]; e++) ni_[w++] = inv[gi[e]];
    }
}

inline OrderCost exchange_cost(const int *gp, const int *gi, int n, int world) {
    OrderCost c;
    const RowPartition part = make_partition(gp, n, world);
    c.rows_max = part.rows_max;
    std::vector<uint8_t> mark((size_t)n);
    c.halo_share = 0.0;
    for (int p = 0; p < world && world > 1; p++) {
        std::fill(mark.begin(), mark.end(), 0);
        const int a = part.start[p], b = part.start[p + 1];
        for (long e = gp[a]; e < gp[b]; e++) mark[gi[e]] = 1;
        long cnt = 0;
        for (int j = 0; j < n; j++) cnt += mark[j] && (j < a || j >= b);
        const long remote = (long)n - (b - a);
        if (remote > 0) c.halo_share = std::max(c.halo_share, (double)cnt / (double)remote);
        c.recv_rows_max = std::max(c.recv_rows_max, cnt);
    }
    return c;
}

inline std::vector<int> degree_order(const int *gp, int n, const int *group = nullptr) {
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (group && group[a] != group[b]) return group[a] < group[b];
        return gp[a + 1] - gp[a] > gp[b + 1] - gp[b];
    });
    return order;
}

// The order to renumber with, or empty: keep the ids.  Renumbering changes which element gets which dropout decision
// (they are keyed by global element index), so a renumbered run is statistically, not numerically, the single-GPU run of
// the given dataset (it IS numerically the single-GPU run of the renumbered dataset: tested).  It is therefore done only
// where it changes the kind of exchange: the id order needs the all-gather (its neediest rank reads more than `plan_at`
// of the remote rows) and the candidate order gets by with halo lists (share <= plan_at) that also move at least `gain`
// fewer rows than the all-gather did.  HIPGCN_STRUCTURE_PARTITION forces the best candidate, HIPGCN_ID_PARTITION none.
struct NodeOrderChoice {
    std::vector<int> order;        // empty: ids kept
    const char *name = "ids";
    OrderCost ids, chosen;
};
inline NodeOrderChoice choose_node_order(const int *gp, const int *gi, int n, int world, const int *group /* cluster.h, or NULL */,
                                         bool force, double plan_at = 0.75, double gain = 0.15) {
    NodeOrderChoice out;
    if (world < 2 || n < 2) return out;
    out.ids = out.chosen = exchange_cost(gp, gi, n, world);
    if (!force && out.ids.halo_share <= plan_at) return out;
    auto price = [&](const OrderCost &c) { return c.halo_share <= plan_at ? c.recv_rows_max : (long)(world - 1) * c.rows_max; };
    long best = price(out.ids);
    std::vector<int> np_, ni_;
    for (int cand = 0; cand < 2; cand++) {
        if (cand == 1 && !group) continue;
        std::vector<int> order = degree_order(gp, n, cand == 1 ? group : nullptr);
        permute_csr(gp, gi, n, order, np_, ni_);
        const OrderCost c = exchange_cost(np_.data(), ni_.data(), n, world);
        const bool accept = force ? (out.order.empty() || price(c) < best)
                                  : (c.halo_share <= plan_at && price(c) < best && (double)price(c) <= (1.0 - gain) * (double)price(out.ids));
        if (accept) {
            best = price(c);
            out.order = std::move(order);
            out.name = cand == 1 ? "group-major (groups found in the graph), descending degree inside" : "descending degree";
            out.chosen = c;
        }
    }
    return out;
}
