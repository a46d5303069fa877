"""Experiment (HBM regime): the hidden-width aggregation of an R-MAT graph as TWO launches — one gathering only from the
hottest source rows (the top f of the nodes by degree: 1 % of the rows of rmat-21 are 50 % of all gathered rows, 3 % are 69 %),
one from the rest — against the one-launch aggregation, in which hot and cold rows compete for the same L2 / Infinity Cache
lines.  Uses gcnhip_graph_create_restricted (a source-row mask); partial outputs go to separate buffers (a product form
would add the second launch's sum to the first's: + one read of the output).
    python tools/exp_hot_cold.py [scale=21] [fractions ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck
from tools.exp_source_blocks import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    fracs = [float(a) for a in sys.argv[2:]] or [0.01, 0.03, 0.12]
    gp, gi = datagen.rmat_graph(scale)
    N = gp.size - 1
    deg = np.diff(gp)
    cnt = np.bincount(gi, minlength=N)
    order = np.argsort(-cnt, kind="stable")
    dev = Device(0); lib = dev.lib
    g = dev.graph(gp, gi, row_group=None)
    g.set_schedule(2, None, 256)                      # the schedule HipGCN picks on R-MAT
    rng = np.random.default_rng(0)
    for d, ld in ((128, 128), (41, 48)):
        x = dev.buf(rng.standard_normal((N, ld)).astype(np.float32))
        o = dev.buf((N, ld)); o2 = dev.buf((N, ld))
        base = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, x.ptr, ld, o.ptr, ld, d), "gs"), iters=10)
        print(f"rmat-{scale} d={d}: one launch {base:.3f} ms ({4.0 * d * gi.size / base / 1e6:.0f} GB/s gathered)", flush=True)
        for f in fracs:
            k = int(N * f)
            hot = np.zeros(N, bool); hot[order[:k]] = True
            share = cnt[order[:k]].sum() / cnt.sum()
            gh, gc = g.restricted(hot), g.restricted(~hot)
            gh.set_schedule(2, None, 256); gc.set_schedule(2, None, 256)
            th = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, gh.h, x.ptr, ld, o.ptr, ld, d), "gs"), iters=10)
            tc = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, gc.h, x.ptr, ld, o2.ptr, ld, d), "gs"), iters=10)
            print(f"   hot = top {100 * f:.0f} % of the rows ({k * ld * 4 / 2**20:.0f} MiB, {100 * share:.1f} % of the gathered rows): "
                  f"hot launch {th:.3f} ms ({4.0 * d * gi.size * share / th / 1e6:.0f} GB/s), cold launch {tc:.3f} ms "
                  f"({4.0 * d * gi.size * (1 - share) / tc / 1e6:.0f} GB/s), together {th + tc:.3f} ms = {(th + tc) / base:.2f} x one launch", flush=True)
            gh.free(); gc.free()
        x.free(); o.free(); o2.free()


if __name__ == "__main__":
    main()
