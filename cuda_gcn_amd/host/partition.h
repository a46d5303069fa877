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
