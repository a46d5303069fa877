"""Experiment: GraphSum time vs size of the gathered table at a fixed edge count
(how much would a higher L2 hit rate buy?)."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck
from tools.bench_ops import timeit

dev = Device(0); lib = dev.lib
rng = np.random.default_rng(0)
M = 11_600_000
for N in (232965, 116000, 58000, 29000, 14500):
    w = np.arange(1, N + 1, dtype=np.float64) ** (-1.0 / 1.3); rng.shuffle(w)
    w = np.minimum(w, 2.0e4 * w.sum() / (2.0 * M))
    lo, hi = datagen._sample_edges(rng, N, M, w)
    gp, gi = datagen.csr_with_self_loops(lo, hi, N)
    g = dev.graph(gp, gi)
    for dim, ld in ((128, 128), (41, 48)):
        x = dev.buf(rng.standard_normal((N, ld)).astype(np.float32)); o = dev.buf((N, ld))
        ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, x.ptr, ld, o.ptr, ld, dim), "gs"), iters=10)
        print(f"N={N} nnz={gi.size} table={N*ld*4/1e6:.1f}MB d={dim}: {ms:.3f} ms  ({gi.size/ms/1e6:.2f} Gedges/s)", flush=True)
    g.free()
