"""Experiment: GraphSum time of ONE rank's row block at P = 8 / 4 when the block is (a) a contiguous id range
(current partition) or (b) whole communities (rows grouped by label, blocks balanced by edges)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck
from tools.bench_ops import timeit

ds = datagen.make_dataset("reddit-syn")
gp, gi, lab = ds["g_indptr"].astype(np.int64), ds["g_indices"], ds["label"]
N = gp.size - 1
deg = np.diff(gp).astype(np.int32)
dev = Device(0); lib = dev.lib
rng = np.random.default_rng(0)


def block_graph(rows):
    """CSR of the given rows (global column ids)"""
    cnt = deg[rows].astype(np.int64)
    ip = np.zeros(rows.size + 1, np.int64); np.cumsum(cnt, out=ip[1:])
    idx = np.concatenate([gi[gp[r]:gp[r + 1]] for r in rows])
    return ip.astype(np.int32), idx


def bench(tag, rows, groups):
    ip, idx = block_graph(rows)
    for sched in ("degree", "label-major"):
        g = dev.graph(ip, idx, n_cols=N, col_deg=deg, row_group=groups if sched == "label-major" else None)
        for dim, ld in ((128, 128), (41, 48)):
            x = dev.buf(rng.standard_normal((N, ld), dtype=np.float32)); o = dev.buf((rows.size, ld))
            ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, x.ptr, ld, o.ptr, ld, dim), "gs"), iters=10)
            print(f"{tag} [{sched}] rows={rows.size} edges={idx.size} d={dim}: {ms:.3f} ms", flush=True)
            x.free(); o.free()
        g.free()


for P in (8, 4):
    # (a) contiguous ids, edge balanced
    tgt = gp[-1] / P
    r1 = int(np.searchsorted(gp, tgt))
    rows = np.arange(0, r1)
    bench(f"P={P} contiguous ids", rows, lab[rows])
    # (b) whole communities: nodes ordered by label, first block up to 1/P of the edges
    order = np.argsort(lab, kind="stable")
    ce = np.cumsum(deg[order].astype(np.int64))
    k = int(np.searchsorted(ce, tgt))
    rows = order[:k]
    bench(f"P={P} community block", rows, lab[rows])
