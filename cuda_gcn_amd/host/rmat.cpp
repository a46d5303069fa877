// rmat.cpp — Graph500 R-MAT generator for the HBM-roofline stress configuration (BASELINE.json configs[4]:
// 2^22 nodes, average degree ~32).  The reference ships no generator and no data; this produces the
// in-memory layout its loader would (src/common/parser.cpp:20-46: CSR, the self loop stored first in
// every row, neighbours in file order = ascending here).
//
// Host-only, multi-threaded, and deterministic in (scale, edge_factor, seed) whatever the thread count:
// every sampled pair draws its quadrant choices from a counter-based hash of (seed, pair index, level).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "gcnhost.h"

namespace {

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int n_threads() {
    unsigned h = std::thread::hardware_concurrency();
    if (const char *e = getenv("GCN_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) h = (unsigned)v; }
    return (int)std::min(16u, std::max(1u, h));
}

template <class F>
void parallel_for(int T, F f) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(f, t);
    f(0);
    for (auto &x : th) x.join();
}

// sort `a` with T chunk sorts and log2(T) rounds of pairwise merges through `b`; the result is in `a`
void parallel_sort(std::vector<uint64_t> &a, std::vector<uint64_t> &b, int T) {
    const size_t n = a.size();
    int P = 1;
    while (P * 2 <= T) P *= 2;                       // power of two chunks
    std::vector<size_t> cut(P + 1);
    for (int i = 0; i <= P; i++) cut[i] = n * (size_t)i / P;
    parallel_for(P, [&](int t) { std::sort(a.begin() + cut[t], a.begin() + cut[t + 1]); });
    b.resize(n);
    uint64_t *src = a.data(), *dst = b.data();
    for (int width = 1; width < P; width *= 2) {
        const int pairs = P / (2 * width);
        parallel_for(pairs, [&](int t) {
            const size_t lo = cut[2 * width * t], mid = cut[2 * width * t + width], hi = cut[2 * width * (t + 1)];
            std::merge(src + lo, src + mid, src + mid, src + hi, dst + lo);
        });
        std::swap(src, dst);
    }
    if (src != a.data()) memcpy(a.data(), src, n * sizeof(uint64_t));
}

}  // namespace

extern "C" {

int gcnhost_rmat_graph(int scale, int edge_factor, uint64_t seed, int **out_indptr, int **out_indices, int64_t *out_nnz) {
    if (scale < 1 || scale > 26 || edge_factor < 1 || !out_indptr || !out_indices || !out_nnz) return -1;
    const uint64_t n = 1ull << scale, m = n * (uint64_t)edge_factor;
    const int T = n_threads();
    // Graph500 parameters a, b, c (d = 1 - a - b - c = 0.05) at 16-bit resolution
    const uint32_t A = (uint32_t)(0.57 * 65536), AB = (uint32_t)(0.76 * 65536), ABC = (uint32_t)(0.95 * 65536);
    std::vector<uint64_t> keys(2 * m), tmp;
    parallel_for(T, [&](int t) {
        for (uint64_t e = m * (uint64_t)t / T; e < m * (uint64_t)(t + 1) / T; e++) {
            uint64_t u = 0, v = 0, h = 0;
            for (int l = 0; l < scale; l++) {
                if ((l & 3) == 0) h = splitmix64(seed ^ splitmix64(e * 8 + (uint64_t)(l >> 2)));
                const uint32_t r = (uint32_t)(h >> (16 * (l & 3))) & 0xFFFFu;
                u = (u << 1) | (r >= AB);
                v = (v << 1) | ((r >= A && r < AB) || r >= ABC);
            }
            // both directions; a self pair becomes the sentinel ~0 (sorted to the end, dropped)
            keys[2 * e] = u == v ? ~0ull : u * n + v;
            keys[2 * e + 1] = u == v ? ~0ull : v * n + u;
        }
    });
    parallel_sort(keys, tmp, T);
    std::vector<uint64_t>().swap(tmp);
    size_t cnt = std::unique(keys.begin(), keys.end()) - keys.begin();
    while (cnt && keys[cnt - 1] == ~0ull) cnt--;
    const uint64_t nnz = cnt + n;                     // + one self loop per node (parser.cpp:30-33)
    if (nnz >= (1ull << 31)) return -1;               // int32 indices, like the reference
    int *indptr = (int *)malloc((n + 1) * sizeof(int));
    int *indices = (int *)malloc(nnz * sizeof(int));
    if (!indptr || !indices) { free(indptr); free(indices); return -1; }
    std::vector<int> deg(n, 1);
    for (size_t i = 0; i < cnt; i++) deg[keys[i] >> scale]++;
    indptr[0] = 0;
    for (uint64_t r = 0; r < n; r++) indptr[r + 1] = indptr[r] + deg[r];
    // keys are sorted by (source, neighbour): key i of source s lands at indptr[s] + 1 + (i - first key of s)
    // = i + s + 1, because s self loops precede it
    parallel_for(T, [&](int t) {
        for (uint64_t r = n * (uint64_t)t / T; r < n * (uint64_t)(t + 1) / T; r++) indices[indptr[r]] = (int)r;
        for (size_t i = cnt * (size_t)t / T; i < cnt * (size_t)(t + 1) / T; i++) {
            const uint64_t s = keys[i] >> scale;
            indices[i + s + 1] = (int)(keys[i] & (n - 1));
        }
    });
    *out_indptr = indptr;
    *out_indices = indices;
    *out_nnz = (int64_t)nnz;
    return 0;
}

void gcnhost_free_array(void *p) { free(p); }

}  // extern "C"
