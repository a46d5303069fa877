// sparse.h — CSR index, the reference's SparseIndex (src/seq/sparse.h:12-17):
// `indptr` has rows()+1 entries, `indices` holds the column of every stored
// element; values live elsewhere (features) or are implicit (adjacency).
#pragma once
#include <cstddef>
#include <vector>

class SparseIndex {
public:
    std::vector<int> indptr;    // row r owns indices[indptr[r] .. indptr[r+1])
    std::vector<int> indices;

    int rows() const { return indptr.empty() ? 0 : static_cast<int>(indptr.size()) - 1; }
    std::size_t nnz() const { return indices.size(); }
    int row_length(int r) const { return indptr[r + 1] - indptr[r]; }
};
