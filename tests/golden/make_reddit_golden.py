#!/usr/bin/env python3
"""Pins tools/reddit_convert.py to the reference's own offline script.

Run in the BUILD CONTAINER only (it needs /root/reference):

    python3 tests/golden/make_reddit_golden.py

It writes a small GraphSAGE-format dataset (56 nodes with string ids, 9 features, 5 classes; nodes without val/test
annotations, a duplicated link, a self link, a constant feature column, train/val/test nodes) into a
temporary directory, EXECUTES /root/reference/reddit_preprocess.py there unmodified (runpy, cwd = that
directory: the script reads `reddit-*` and writes `reddit.graph/.split/.svmlight` relative to `.`), parses the
three text files it wrote, and stores inputs + parsed outputs in tests/golden/reddit_preprocess.npz.
tests/test_host_cpu.py::test_reddit_convert_matches_reference_script then compares tools/reddit_convert.py with
those arrays on any box.  The reference's source is not copied anywhere: only data goes into the fixture.

The script was written for older libraries; two names it touches no longer exist in this image and are aliased
here, before it runs, without changing what it computes:
  * `from scipy.sparse.linalg.eigen.arpack import eigsh` (reddit_preprocess.py:16) — a dead import (eigsh is never
    called); scipy 1.15 has no such module path.  A `sys.modules` entry provides the name.
  * `G.node[...]` (reddit_preprocess.py:64-66) — networkx removed the `Graph.node` alias of `Graph.nodes` in 2.4;
    the alias is put back on the class (it is networkx's own former definition: `node = nodes`).
"""
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference/reddit_preprocess.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reddit_preprocess.npz")


def make_input(seed=20191210):
    rng = np.random.default_rng(seed)
    n_all, F, C = 56, 9, 5
    # string node ids as in the GraphSAGE Reddit files (their sorted order is lexicographic: "1003" < "97")
    ids = [str(v) for v in range(91, 91 + 19 * n_all, 19)]
    rng.shuffle(ids)                                     # file order != sorted order
    nodes = []
    for k, nid in enumerate(ids):
        n = {"id": nid}
        if k != 7:                                       # node 7 lacks the annotations: the script drops it
            r = rng.random()
            n["test"] = bool(r < 0.25)
            n["val"] = bool(0.25 <= r < 0.45)
        nodes.append(n)
    links = []
    for _ in range(150):
        a, b = rng.choice(n_all, 2, replace=False)
        links.append({"source": ids[a], "target": ids[b]})
    links.append(dict(links[3]))                         # a duplicated link
    links.append({"source": ids[11], "target": ids[11]})   # a self link
    links.append({"source": ids[7], "target": ids[9]})     # a link of the dropped node
    # reddit_preprocess.py:30 evaluates `G.nodes()[0]`: under networkx >= 2 that is a lookup of the node NAMED 0, so such a
    # node must exist for the script to get past that line; it carries no annotations and is removed like node 7
    nodes.append({"id": 0})
    G = {"directed": False, "multigraph": False, "graph": {}, "nodes": nodes, "links": links}
    feats = rng.standard_normal((n_all, F))
    feats[:, 4] = 2.5                                    # constant column: zero variance
    id_map = {nid: k for k, nid in enumerate(ids)}       # node id -> row of feats
    class_map = {nid: int(rng.integers(0, C)) for nid in ids}
    class_map[ids[0]] = C - 1                       # the top class occurs (the loader sizes the output by max label)
    return G, feats, id_map, class_map


def parse_outputs(d):
    graph = [list(map(int, ln.split())) for ln in open(os.path.join(d, "reddit.graph")).read().split("\n")[:-1]]
    split = [int(ln) for ln in open(os.path.join(d, "reddit.split")).read().split()]
    labels, f_indptr, f_idx, f_val = [], [0], [], []
    for ln in open(os.path.join(d, "reddit.svmlight")).read().split("\n")[:-1]:
        tok = ln.split()
        labels.append(int(float(tok[0])))
        for kv in tok[1:]:
            k, v = kv.split(":")
            f_idx.append(int(k)); f_val.append(float(v))
        f_indptr.append(len(f_idx))
    g_indptr = np.cumsum([0] + [len(r) for r in graph])
    return dict(out_g_indptr=np.array(g_indptr, np.int64), out_g_indices=np.array([j for r in graph for j in r], np.int64),
                out_split=np.array(split, np.int64), out_label=np.array(labels, np.int64),
                out_f_indptr=np.array(f_indptr, np.int64), out_f_indices=np.array(f_idx, np.int64), out_f_val=np.array(f_val, np.float64))


def main():
    if not os.path.exists(REF):
        sys.exit("needs /root/reference (build container only)")
    import networkx as nx
    if not hasattr(nx.Graph, "node"):
        nx.Graph.node = nx.Graph.nodes                   # networkx < 2.4's own alias
    for name in ("scipy.sparse.linalg.eigen", "scipy.sparse.linalg.eigen.arpack"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.eigsh = None                               # imported, never called
            sys.modules[name] = m
    G, feats, id_map, class_map = make_input()
    with tempfile.TemporaryDirectory() as d:
        json.dump(G, open(os.path.join(d, "reddit-G.json"), "w"))
        np.save(os.path.join(d, "reddit-feats.npy"), feats)
        json.dump(id_map, open(os.path.join(d, "reddit-id_map.json"), "w"))
        json.dump(class_map, open(os.path.join(d, "reddit-class_map.json"), "w"))
        cwd = os.getcwd()
        os.chdir(d)
        try:
            runpy.run_path(REF, run_name="__main__")
        finally:
            os.chdir(cwd)
        out = parse_outputs(d)
    np.savez_compressed(OUT, in_G=json.dumps(G), in_feats=feats, in_id_map=json.dumps(id_map), in_class_map=json.dumps(class_map), **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
