"""Experiment: GraphSum time on reddit-syn in the generator's node order vs nodes grouped by
community (here: by the true label, the upper bound for a structure-only clustering)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck
from tools.bench_ops import timeit

ds = datagen.make_dataset("reddit-syn")
gp, gi, lab = ds["g_indptr"], ds["g_indices"], ds["label"]
N = gp.size - 1
dev = Device(0); lib = dev.lib
rng = np.random.default_rng(0)

def permuted(order):
    """order[new] = old"""
    inv = np.empty(N, np.int64); inv[order] = np.arange(N)
    deg = np.diff(gp)
    src = np.repeat(np.arange(N), deg)
    keep = gi != src                                    # drop stored self loops, csr_with_self_loops re-adds them
    u, v = inv[src[keep]], inv[gi[keep]]
    m = u < v
    return datagen.csr_with_self_loops(u[m], v[m], N)

def bench(tag, p, i, group=None):
    g = dev.graph(p, i, row_group=group)
    for dim, ld in ((128, 128), (41, 48)):
        x = dev.buf(rng.standard_normal((N, ld), dtype=np.float32)); o = dev.buf((N, ld))
        ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, x.ptr, ld, o.ptr, ld, dim), "gs"), iters=10)
        print(f"{tag}: d={dim}: {ms:.3f} ms", flush=True)
        x.free(); o.free()
    g.free()

bench("generator order", gp, gi)
bench("generator order, tasks grouped by label (no renaming)", gp, gi, lab)
bench("grouped by label", *permuted(np.argsort(lab, kind="stable")))
deg = np.diff(gp)
bench("label, then degree desc", *permuted(np.lexsort((-deg, lab))))
bench("degree desc only", *permuted(np.argsort(-deg, kind="stable")))
o = np.lexsort((-deg, lab))
bench("label+degree renaming, tasks grouped by label", *permuted(o), lab[o])
bench("generator order, 41 RANDOM groups (control: blending without communities)", gp, gi, np.random.default_rng(1).integers(0, 41, N).astype(np.int32))
rank = np.empty(N, np.int64); rank[np.argsort(-deg, kind="stable")] = np.arange(N)
for G in (8, 41, 256):
    bench(f"degree rank dealt round-robin into {G} groups", gp, gi, (rank % G).astype(np.int32))
# communities, each dealt into sub-groups: label-major, then blended inside the label
for G in (4,):
    bench(f"label x {G} dealt sub-groups", gp, gi, (lab.astype(np.int64) * G + rank % G).astype(np.int32))
