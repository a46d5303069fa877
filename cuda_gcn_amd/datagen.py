"""Seeded synthetic stand-ins for the reference's datasets.

The reference ships no data (its ``data.tgz`` is absent and there is no
network), so every configuration runs on shape-matched synthetic graphs.  The
in-memory layout is exactly what the reference's loader produces
(``src/common/parser.cpp:20-103`` of the reference): graph CSR with the self
loop stored FIRST in every row, feature CSR + values, one label and one split
code (1 train / 2 val / 3 test / 0 unused) per node.  ``write_text`` emits the
three text files (``<name>.graph/.split/.svmlight``) the reference reads.

Definitions follow SURVEY.md §8(d); generator seed 20191210 unless stated.
"""
from __future__ import annotations

import os
import numpy as np

DEFAULT_SEED = 20191210

# name -> (N, F, C, undirected edges, nnz/row (0 = dense), n_train, n_val, n_test)
SHAPES = {
    "cora-syn": (2708, 1433, 7, 5429, 18, 140, 500, 1000),
    "citeseer-syn": (3327, 3703, 6, 4732, 32, 120, 500, 1000),
    "pubmed-syn": (19717, 500, 3, 44338, 50, 60, 500, 1000),
    "reddit-syn": (232965, 602, 41, 11606919, 0, -1, -1, -1),
    "reddit-mini": (23296, 602, 41, 1160692, 0, -1, -1, -1),
    "tiny-syn": (97, 23, 5, 211, 6, 30, 25, 30),
}

# Reddit-shaped variants that differ only in the community structure planted in the graph (round 5: the headline's
# sensitivity to that structure).  `<reddit-syn|reddit-mini>-<variant>`: (share of the edges whose second endpoint is drawn
# inside the first endpoint's class, class sizes).  The plain names keep the round-1 generator: 0.6, equal classes.
#   h0    SURVEY 8(d)'s literal Chung-Lu graph: no planted structure at all (the labels are then independent of the graph)
#   h03   half the mixing of the plain graph
#   zipf  the plain graph's mixing with class sizes ~ 1/rank: the largest class holds 23 % of the nodes (54 K rows = 13.9 MB of
#         256-byte column slices, 3.3 x one XCD's L2), the smallest 0.6 %
REDDIT_VARIANTS = {"h0": (0.0, "equal"), "h03": (0.3, "equal"), "zipf": (0.6, "zipf")}


def reddit_variant(name: str):
    """(base shape name, homophily, class sizes) of a reddit-* dataset name"""
    parts = name.split("-")
    base = "-".join(parts[:2])
    if len(parts) == 2:
        return base, 0.6, "equal"
    h, sizes = REDDIT_VARIANTS[parts[2]]
    return base, h, sizes


def _unique_undirected(u, v, n):
    """drop self loops and duplicate pairs; return (lo, hi) arrays"""
    lo = np.minimum(u, v).astype(np.int64)
    hi = np.maximum(u, v).astype(np.int64)
    keep = lo != hi
    key = np.unique(lo[keep] * n + hi[keep])
    return (key // n).astype(np.int64), (key % n).astype(np.int64)


def _sample_edges(rng, n, m, weights=None, label=None, homophily=0.0):
    """m distinct undirected edges; endpoints uniform or ∝ weights (Chung-Lu).  With `label`
    and homophily h, a fraction h of the edges picks its second endpoint inside the first
    endpoint's class (still ∝ weights), which gives the graph the community structure a GCN
    can learn from."""
    if weights is not None:
        cdf = np.cumsum(weights, dtype=np.float64)
        cdf /= cdf[-1]
    if label is not None and homophily > 0:
        order = np.argsort(label, kind="stable")
        wl = weights[order] if weights is not None else np.ones(n)
        ccdf = np.cumsum(wl, dtype=np.float64)
        n_cls = int(label.max()) + 1
        first = np.searchsorted(label[order], np.arange(n_cls), side="left")
        last = np.searchsorted(label[order], np.arange(n_cls), side="right")
        c_lo = np.where(first > 0, ccdf[np.maximum(first, 1) - 1], 0.0)
        c_hi = ccdf[last - 1]
    lo = np.empty(0, np.int64)
    hi = np.empty(0, np.int64)
    while lo.size < m:
        need = int((m - lo.size) * 1.15) + 64
        if weights is None:
            u = rng.integers(0, n, need)
            v = rng.integers(0, n, need)
        else:
            u = np.searchsorted(cdf, rng.random(need), side="right").clip(0, n - 1)
            v = np.searchsorted(cdf, rng.random(need), side="right").clip(0, n - 1)
            if label is not None and homophily > 0:
                intra = rng.random(need) < homophily
                cu = label[u[intra]]
                r = c_lo[cu] + rng.random(cu.size) * (c_hi[cu] - c_lo[cu])
                v[intra] = order[np.searchsorted(ccdf, r, side="right").clip(0, n - 1)]
        a, b = _unique_undirected(np.concatenate([lo, np.minimum(u, v)]),
                                  np.concatenate([hi, np.maximum(u, v)]), n)
        lo, hi = a, b
    if lo.size > m:
        sel = np.sort(rng.choice(lo.size, m, replace=False))
        lo, hi = lo[sel], hi[sel]
    return lo, hi


def csr_with_self_loops(lo, hi, n):
    """symmetrise and put the self loop first in every row (parser.cpp:30-42)"""
    src = np.concatenate([lo, hi])
    dst = np.concatenate([hi, lo])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    deg = np.bincount(src, minlength=n).astype(np.int64) + 1
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(deg, out=indptr[1:])
    indices = np.empty(indptr[-1], np.int32)
    indices[indptr[:-1]] = np.arange(n, dtype=np.int32)
    # position of each neighbour = row start + 1 + rank inside the row
    rank = np.arange(src.size, dtype=np.int64) - (indptr[src] - src)  # indptr[src]-src = edges before row (no self loops)
    indices[indptr[src] + 1 + rank] = dst.astype(np.int32)
    assert indptr[-1] < 2 ** 31
    return indptr.astype(np.int32), indices


def rmat_graph(scale: int, edge_factor: int = 16, seed: int = DEFAULT_SEED):
    """Graph500 R-MAT adjacency (a, b, c = .57, .19, .19), symmetrised, duplicates and self pairs dropped, in the
    loader's CSR layout (self loop first).  Generated by the multi-threaded C++ helper of libgcnhost
    (host/rmat.cpp: scale 21 in seconds; the numpy version of round 1 took minutes) -> (indptr, indices) int32."""
    import ctypes as C
    from . import _lib
    lib = _lib.gcnhost()
    ip, ix, nnz = C.c_void_p(), C.c_void_p(), C.c_int64()
    if lib.gcnhost_rmat_graph(int(scale), int(edge_factor), int(seed), C.byref(ip), C.byref(ix), C.byref(nnz)) != 0:
        raise RuntimeError(f"gcnhost_rmat_graph(scale={scale}) failed")
    n = 1 << scale
    indptr = np.frombuffer((C.c_int * (n + 1)).from_address(ip.value), dtype=np.int32).copy()
    indices = np.frombuffer((C.c_int * nnz.value).from_address(ix.value), dtype=np.int32).copy()
    lib.gcnhost_free_array(ip)
    lib.gcnhost_free_array(ix)
    return indptr, indices


def make_dataset(name: str, seed: int = DEFAULT_SEED, rows: slice | None = None):
    """Build a dataset dict.  ``name``: a key of SHAPES or ``rmat-<scale>[-<feats>]``."""
    rng = np.random.default_rng(seed)
    if name.startswith("rmat-"):
        parts = name.split("-")
        scale = int(parts[1])
        F = int(parts[2]) if len(parts) > 2 else 256
        C = 41
        g_indptr, g_indices = rmat_graph(scale, 16, seed)
        N = 1 << scale
        nnz_row = 0
        n_train = n_val = n_test = -2
    else:
        homophily, class_sizes = 0.6, "equal"
        if name.startswith("reddit"):
            base, homophily, class_sizes = reddit_variant(name)
            N, F, C, M, nnz_row, n_train, n_val, n_test = SHAPES[base]
        else:
            N, F, C, M, nnz_row, n_train, n_val, n_test = SHAPES[name]
        if name.startswith("reddit"):
            # Chung-Lu, power-law expected degrees (exponent ~2.3), capped; by default 60 % of the edges stay
            # inside a class (Reddit communities are posts of one subreddit = one label)
            w = (np.arange(1, N + 1, dtype=np.float64)) ** (-1.0 / 1.3)
            rng.shuffle(w)
            cap = 2.0e4 * w.sum() / (2.0 * M)
            w = np.minimum(w, cap)
            lrng = np.random.default_rng(seed + 1)
            if class_sizes == "zipf":
                pz = 1.0 / np.arange(1, C + 1, dtype=np.float64)
                pre_label = lrng.choice(C, N, p=pz / pz.sum()).astype(np.int32)
            else:
                pre_label = lrng.integers(0, C, N).astype(np.int32)
            pre_label[:C] = np.arange(C, dtype=np.int32)
            lo, hi = _sample_edges(rng, N, M, w, pre_label, homophily)
        else:
            lo, hi = _sample_edges(rng, N, M)
    if not name.startswith("rmat-"):
        g_indptr, g_indices = csr_with_self_loops(lo, hi, N)

    label = rng.integers(0, C, N).astype(np.int32)
    label[:C] = np.arange(C, dtype=np.int32)       # every class occurs (loader: output_dim = max label + 1)
    if name.startswith("reddit"):
        label = pre_label

    if nnz_row == 0:
        # dense standardised features with a class-dependent shift (learnable signal)
        f_val = rng.standard_normal((N, F), dtype=np.float32)
        f_val[np.arange(N), label % F] += np.float32(0.5 if name.startswith("rmat") else 1.5)
        f_indptr = (np.arange(N + 1, dtype=np.int64) * F).astype(np.int32)
        f_indices = np.tile(np.arange(F, dtype=np.int32), N)
        f_val = f_val.reshape(-1)
    else:
        # nnz_row distinct columns per row; half of them from a class-specific band
        band = max(nnz_row, F // C)
        cols = np.empty((N, nnz_row), np.int64)
        half = nnz_row // 2
        for i in range(N):
            b0 = (int(label[i]) * (F // C)) % max(1, F - band + 1)
            own = b0 + rng.choice(band, half, replace=False)
            rest = rng.choice(F, nnz_row, replace=False)
            rest = rest[~np.isin(rest, own)][: nnz_row - half]
            cols[i] = np.sort(np.concatenate([own, rest]))
        cols[0, -1] = F - 1                          # input_dim = max index + 1
        cols[0] = np.sort(cols[0])
        if np.unique(cols[0]).size != nnz_row:       # keep row 0 duplicate-free
            cols[0] = np.sort(np.concatenate([rng.choice(F - 1, nnz_row - 1, replace=False), [F - 1]]))
        f_indices = cols.reshape(-1).astype(np.int32)
        f_indptr = (np.arange(N + 1, dtype=np.int64) * nnz_row).astype(np.int32)
        if name.startswith("pubmed"):
            f_val = rng.uniform(0.0, 0.2, N * nnz_row).astype(np.float32)   # TF-IDF-like
        else:
            f_val = np.full(N * nnz_row, np.float32(1.0) / np.float32(nnz_row), np.float32)

    split = np.zeros(N, np.int32)
    perm = rng.permutation(N)
    if n_train == -1:                                # reddit: 66 % / 10 % / 24 %
        a, b = int(0.66 * N), int(0.76 * N)
        split[perm[:a]] = 1
        split[perm[a:b]] = 2
        split[perm[b:]] = 3
    elif n_train == -2:                              # rmat: all train except 1 % val, 1 % test
        split[:] = 1
        k = max(1, N // 100)
        split[perm[:k]] = 2
        split[perm[k:2 * k]] = 3
    else:
        split[perm[:n_train]] = 1
        split[perm[n_train:n_train + n_val]] = 2
        split[perm[n_train + n_val:n_train + n_val + n_test]] = 3

    ds = dict(name=name, num_nodes=N, input_dim=F, output_dim=C,
              g_indptr=g_indptr, g_indices=g_indices,
              f_indptr=f_indptr, f_indices=f_indices, f_val=f_val,
              label=label, split=split)
    if name.startswith("reddit"):
        ds["planted_homophily"] = homophily
        ds["class_sizes"] = class_sizes
    return ds


def edge_homophily(ds):
    """share of the stored non-self edges whose two ends carry the same label (what the generator planted, measured)"""
    gp, gi = ds["g_indptr"].astype(np.int64), ds["g_indices"]
    src = np.repeat(np.arange(gp.size - 1), np.diff(gp))
    m = src != gi
    return float((ds["label"][src[m]] == ds["label"][gi[m]]).mean())


def write_text(ds, root: str, name: str | None = None):
    """Write ``root/<name>.graph/.split/.svmlight`` in the reference's formats
    (parser.cpp:20-103).  Values are printed with 9 significant digits so they
    round-trip through strtof bit-exactly."""
    name = name or ds["name"]
    os.makedirs(root, exist_ok=True)
    N = ds["num_nodes"]
    gp, gi = ds["g_indptr"], ds["g_indices"]
    with open(os.path.join(root, name + ".graph"), "w") as f:
        for i in range(N):
            f.write(" ".join(map(str, gi[gp[i] + 1:gp[i + 1]].tolist())) + "\n")   # self loop is implicit
    fp, fi, fv = ds["f_indptr"], ds["f_indices"], ds["f_val"]
    with open(os.path.join(root, name + ".svmlight"), "w") as f:
        for i in range(N):
            toks = ["%d:%.9g" % (k, v) for k, v in zip(fi[fp[i]:fp[i + 1]].tolist(), fv[fp[i]:fp[i + 1]].tolist())]
            f.write(str(int(ds["label"][i])) + (" " if toks else "") + " ".join(toks) + "\n")
    with open(os.path.join(root, name + ".split"), "w") as f:
        for s in ds["split"].tolist():
            f.write("%d\n" % s)


def write_gcnbin(ds, path: str):
    """Binary dataset cache read by the C++ Parser (host/parser.cpp: magic, 3 dims, then seven
    length-prefixed arrays).  `gcn-hip <name>` prefers `<root>/<name>.gcnbin` over the text files."""
    import struct
    with open(path, "wb") as f:
        f.write(b"GCNBIN01")
        f.write(struct.pack("<3i", int(ds["num_nodes"]), int(ds["input_dim"]), int(ds["output_dim"])))
        for key, dt in (("g_indptr", np.int32), ("g_indices", np.int32), ("f_indptr", np.int32), ("f_indices", np.int32),
                        ("f_val", np.float32), ("split", np.int32), ("label", np.int32)):
            a = np.ascontiguousarray(ds[key], dt)
            f.write(struct.pack("<Q", a.size))
            a.tofile(f)


def planted_communities(n_comm=16, size=512, deg=16, p_in=0.95, feats=32, classes=8, seed=DEFAULT_SEED, shuffle=True):
    """A graph of `n_comm` equal communities (a share p_in of every node's edges stays inside its community) whose node ids
    are SHUFFLED, so that contiguous id ranges cut through every community — the case where rank blocks must be formed from
    the graph's structure (host/partition.h, choose_node_order).  Dense features with a class shift, labels = community % classes."""
    rng = np.random.default_rng(seed)
    N = n_comm * size
    comm = np.repeat(np.arange(n_comm), size)
    M = N * deg // 2
    a = rng.integers(0, N, M)
    inside = rng.random(M) < p_in
    b = np.where(inside, comm[a] * size + rng.integers(0, size, M), rng.integers(0, N, M))
    keep = a != b
    a, b = a[keep], b[keep]
    if shuffle:
        perm = rng.permutation(N)
        a, b, comm = perm[a], perm[b], comm[np.argsort(perm)]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    key = np.unique(lo.astype(np.int64) * N + hi)
    g_indptr, g_indices = csr_with_self_loops((key // N).astype(np.int64), (key % N).astype(np.int64), N)
    label = (comm % classes).astype(np.int32)
    f_val = rng.standard_normal((N, feats), dtype=np.float32)
    f_val[np.arange(N), label % feats] += np.float32(1.5)
    split = np.zeros(N, np.int32)
    r = rng.random(N)
    split[r < 0.5] = 1
    split[(r >= 0.5) & (r < 0.7)] = 2
    split[(r >= 0.7) & (r < 0.9)] = 3
    return dict(name="planted", num_nodes=N, input_dim=feats, output_dim=classes, g_indptr=g_indptr, g_indices=g_indices,
                f_indptr=(np.arange(N + 1, dtype=np.int64) * feats).astype(np.int32), f_indices=np.tile(np.arange(feats, dtype=np.int32), N),
                f_val=f_val.reshape(-1), split=split, label=label)
