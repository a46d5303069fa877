#!/usr/bin/env python3
"""GraphSAGE-format dataset -> this repo's loader formats.

Same data semantics as the reference's offline script (reddit_preprocess.py:27-167 of the
reference), without networkx / sklearn:
  * nodes without both 'val' and 'test' annotations are dropped (:53-58);
  * train = neither val nor test; split codes 1 train / 2 val / 3 test (:141-149);
  * nodes are renumbered 0..N-1 in sorted order of their ids (:102-105);
  * features are standardised with mean / std of the TRAIN rows only, std 0 -> 1 (:71-77);
  * adjacency lists keep the graph's neighbour order (first occurrence of a duplicated link); the loader adds the
    self loop in front; a self link in the data stays in the list (the reference's loader then counts it twice);
Pinned: tests/golden/reddit_preprocess.npz holds what the reference script itself wrote for a 56-node fixture
(tests/golden/make_reddit_golden.py ran it unmodified), and tests/test_host_cpu.py compares this converter with it.
Output: <out>/<name>.gcnbin (binary cache, always) and, with --text, the three text files.
Every feature row is written with all its columns (explicit zeros included), so the dense
first-layer path is taken; sklearn's dump_svmlight_file would have dropped exact zeros.

  tools/reddit_convert.py <dir with reddit-G.json, -feats.npy, -id_map.json, -class_map.json> --prefix reddit --out data
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen  # noqa: E402


def convert(src_dir, prefix):
    G = json.load(open(os.path.join(src_dir, prefix + "-G.json")))
    feats = np.load(os.path.join(src_dir, prefix + "-feats.npy")).astype(np.float64)
    id_map = json.load(open(os.path.join(src_dir, prefix + "-id_map.json")))
    class_map = json.load(open(os.path.join(src_dir, prefix + "-class_map.json")))
    nodes = G["nodes"]
    node_ids = [n["id"] for n in nodes]
    ok = [("val" in n and "test" in n) for n in nodes]
    key = (lambda x: x) if all(isinstance(i, int) for i in node_ids) else str
    kept = sorted((i for i, good in zip(node_ids, ok) if good), key=key)
    new_id = {nid: k for k, nid in enumerate(kept)}
    N = len(kept)
    by_id = {n["id"]: n for n in nodes}
    split = np.zeros(N, np.int32)
    for nid, k in new_id.items():
        n = by_id[nid]
        split[k] = 3 if n["test"] else (2 if n["val"] else 1)
    # links: endpoints are node ids, or (older node-link files) positions in the node list
    links = G["links"]
    by_pos = bool(links) and isinstance(links[0]["source"], int) and not all(isinstance(i, int) for i in node_ids)
    nbrs = [[] for _ in range(N)]
    for e in links:
        a, b = (node_ids[e["source"]], node_ids[e["target"]]) if by_pos else (e["source"], e["target"])
        if a in new_id and b in new_id:
            nbrs[new_id[a]].append(new_id[b])
            if a != b:                                # a self link stays ONE entry of the node's own list, as networkx keeps it
                nbrs[new_id[b]].append(new_id[a])     # (reddit_preprocess.py:122-125 then writes the node into its own line)
    indptr = np.zeros(N + 1, np.int64)
    rows = []
    for k in range(N):
        seen, row = set(), [k]                        # the loader's self loop first (parser.cpp:30-33); a self link in the
        for j in nbrs[k]:                             # data follows as a neighbour of its own, exactly as the reference's file would
            if j not in seen:
                seen.add(j); row.append(j)
        rows.append(row)
        indptr[k + 1] = indptr[k] + len(row)
    g_indices = np.fromiter((j for row in rows for j in row), np.int32, count=int(indptr[-1]))
    rows_of = np.array([int(id_map[str(nid)] if str(nid) in id_map else id_map[nid]) for nid in kept])
    X = feats[rows_of]
    tr = X[split == 1]
    mean, std = tr.mean(axis=0), tr.std(axis=0)
    std[std == 0] = 1.0
    X = ((X - mean) / std).astype(np.float32)
    label = np.array([int(class_map[str(nid)] if str(nid) in class_map else class_map[nid]) for nid in kept], np.int32)
    F = X.shape[1]
    return dict(name=prefix, num_nodes=N, input_dim=F, output_dim=int(label.max()) + 1,
                g_indptr=indptr.astype(np.int32), g_indices=g_indices,
                f_indptr=(np.arange(N + 1, dtype=np.int64) * F).astype(np.int32),
                f_indices=np.tile(np.arange(F, dtype=np.int32), N), f_val=X.reshape(-1), split=split, label=label)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("src_dir")
    ap.add_argument("--prefix", default="reddit")
    ap.add_argument("--out", default="data")
    ap.add_argument("--text", action="store_true", help="also write the reference's three text files")
    a = ap.parse_args()
    ds = convert(a.src_dir, a.prefix)
    os.makedirs(a.out, exist_ok=True)
    datagen.write_gcnbin(ds, os.path.join(a.out, a.prefix + ".gcnbin"))
    if a.text:
        datagen.write_text(ds, a.out, a.prefix)
    print(f"{a.prefix}: {ds['num_nodes']} nodes, {(ds['g_indices'].size - ds['num_nodes']) // 2} edges, "
          f"{ds['input_dim']} features, {ds['output_dim']} classes")
