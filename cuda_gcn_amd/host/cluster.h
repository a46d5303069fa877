// cluster.h — row groups for the aggregation's schedule found in the GRAPH, for datasets whose labels are no
// communities of it (or that bring none).  The GraphSum kernel is bound by how many of the rows it gathers are still in
// an XCD's L2; rows that share neighbours should therefore be in flight together ("row schedule").
// The reference has no counterpart: its GraphSum walks rows in file order (src/seq/module.cpp:85-101).
//
// Method: the local-moving phase of the Louvain method (Blondel, Guillaume, Lambiotte, Lefebvre 2008) from singleton
// groups, asynchronous, in a fixed pseudo-random node order: every node repeatedly joins the neighbouring group with the
// largest modularity gain  k_i,c - tot_c * k_i / 2m  (k_i,c: its neighbours in c; tot_c: degree sum of c; ties and
// non-positive gains keep the current group).  The degree-sum term is what plain label propagation lacks: that floods
// a graph with hubs into ONE group — measured on reddit-syn (56 % of the edges inside 41 planted communities of 5.7 K
// nodes): collapse in three sweeps; with a size bound instead of the term, 397 mixed groups holding 32 % of the edges;
// with the term, the planted communities themselves (38-42 groups, 55-56 % of the edges) in 5 sweeps, 0.8 s.
// A size bound stays (8192 nodes: 2 MiB of 256-byte row slices, half an XCD's L2): a group is a cache working set, so
// merged communities beyond that size are of no use here.  One sweep costs one pass over the edges with a counter array
// (no hashing, no sorting).  Deterministic (fixed order, fixed tie rule), so every rank of a row-partitioned run finds
// the same groups.  The groups only ORDER the task list — any grouping, even a useless one, gives the same bits
// (tested); whether it is used at all is decided by timing it against the other schedules (HipGCN::tune_schedule).
// With the label hint withheld, reddit-syn on one MI355X: plain degree order 197 epochs/s, size-bounded label propagation
// 279, this 297; the labels themselves 297.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

struct StructureGroups {
    std::vector<int> group;      // per node, 0 .. n_groups-1, largest group first
    int n_groups = 0;
    int sweeps = 0;
    double largest_share = 0.0;  // nodes in the largest group / all nodes
    double inside_share = 0.0;   // share of a node's neighbours found in its own group, last sweep
    bool useful = false;         // false: collapsed into one giant group, never left the singletons, or the groups hold few edges
};

inline StructureGroups structure_groups(const int *indptr, const int *indices, int n, int max_sweeps = 10, int max_group = 8192) {
    StructureGroups out;
    if (n <= 0) return out;
    std::vector<int> lab(n), order(n), cnt(n, 0), stamp(n, -1), size(n, 1), seen;
    std::vector<double> tot(n);                       // degree sum of each group (self loops of A + I left out)
    double m2 = 0.0;                                  // 2m: sum of all degrees
    for (int i = 0; i < n; i++) {
        lab[i] = order[i] = i;
        tot[i] = (double)(indptr[i + 1] - indptr[i] - 1);
        m2 += tot[i];
    }
    if (m2 <= 0.0) m2 = 1.0;
    // fixed shuffle (xorshift64*): hubs must not all be visited first, nor in file order
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int i = n - 1; i > 0; i--) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        const uint64_t r = (s * 0x2545F4914F6CDD1Dull) >> 33;
        std::swap(order[i], order[(int)(r % (uint64_t)(i + 1))]);
    }
    int visit = 0;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        long changed = 0, inside = 0, edges = 0;
        for (int k = 0; k < n; k++, visit++) {
            const int i = order[k], cur = lab[i];
            const double ki = (double)(indptr[i + 1] - indptr[i] - 1);
            seen.clear();
            for (int e = indptr[i]; e < indptr[i + 1]; e++) {
                const int j = indices[e];
                if (j == i) continue;                         // the self loop of A + I carries no information
                const int l = lab[j];
                if (stamp[l] != visit) { stamp[l] = visit; cnt[l] = 0; seen.push_back(l); }
                cnt[l]++;
            }
            tot[cur] -= ki; size[cur]--;                      // i leaves; staying is one of the candidates below
            int best = cur, best_n = stamp[cur] == visit ? cnt[cur] : 0;
            double best_gain = (double)best_n - tot[cur] * ki / m2;
            for (int l : seen) {                              // in order of first appearance: deterministic
                if (l == cur || size[l] >= max_group) continue;
                const double gain = (double)cnt[l] - tot[l] * ki / m2;
                if (gain > best_gain) { best_gain = gain; best = l; best_n = cnt[l]; }
            }
            tot[best] += ki; size[best]++;
            if (best != cur) { lab[i] = best; changed++; }
            inside += best_n;
            edges += indptr[i + 1] - indptr[i] - 1;
        }
        out.sweeps = sweep + 1;
        out.inside_share = edges ? (double)inside / (double)edges : 0.0;
        if (getenv("HIPGCN_GROUPS_DEBUG")) fprintf(stderr, "sweep %d: changed %ld inside %.4f\n", sweep, changed, out.inside_share);
        // No structure to find: stop paying for sweeps.  After the first sweep a planted partition at Reddit's mixing has
        // 18-25 % of every node's neighbours in its group, an R-MAT graph 3 % (final values: 49-56 % against 5 %).
        if (sweep == 0 && out.inside_share < 0.06) break;
        if (changed * 200 < n) break;                          // < 0.5 % of the nodes moved
        if (visit > (1 << 30)) break;                          // stamp values stay below 2^31
    }
    // groups by size, largest first; a node that kept a singleton label joins one trailing group
    std::vector<int> ids;
    for (int l = 0; l < n; l++) if (size[l] > 1) ids.push_back(l);
    std::sort(ids.begin(), ids.end(), [&](int a, int b) { return size[a] != size[b] ? size[a] > size[b] : a < b; });
    std::vector<int> remap(n, -1);
    for (size_t g = 0; g < ids.size(); g++) remap[ids[g]] = (int)g;
    const int rest = (int)ids.size();
    bool any_rest = false;
    out.group.resize(n);
    for (int i = 0; i < n; i++) {
        const int g = remap[lab[i]];
        out.group[i] = g >= 0 ? g : rest;
        any_rest |= g < 0;
    }
    out.n_groups = rest + (any_rest ? 1 : 0);
    out.largest_share = ids.empty() ? 0.0 : (double)size[ids[0]] / n;
    // a schedule needs several groups that each fit a cache and together hold most nodes
    long grouped = 0;
    for (int l : ids) grouped += size[l];
    out.useful = ids.size() >= 4 && out.largest_share <= 0.5 && grouped * 2 >= n && out.inside_share >= 0.10;
    return out;
}
