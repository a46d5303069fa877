// sparse.h — CSR index (src/seq/sparse.h:12-17): indices[nnz], indptr[nrow+1];
// values live elsewhere (features) or are implicit (graph).
#pragma once
#include <vector>

class SparseIndex {
public:
    std::vector<int> indices;
    std::vector<int> indptr;
    int rows() const { return indptr.empty() ? 0 : (int)indptr.size() - 1; }
};
