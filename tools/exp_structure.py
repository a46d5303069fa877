"""Experiment (round 5): what moves the hidden-width aggregation on a Reddit-shaped graph WITHOUT planted structure
(reddit-syn-h0: SURVEY 8(d)'s literal Chung-Lu graph; also -h03, -zipf)?  Every launch is the factored operator
(gcnhip_graphsum_ex, scaling 2) at d = 128, timed with HIP events.
  1. row schedules: degree | dealt-256 | label-major
  2. column-slice width: 64 floats (default) | 32 | 16  (context option gs_l) x row loads in flight (gs_u 4 | 2)
  3. hot / cold split: a launch that gathers only from the top-K rows by degree (K rows x 256-byte slices fit an XCD's L2),
     then one for the rest accumulating into the same output (gcnhip_graphsum_ex, accumulate = 1)
    python tools/exp_structure.py [dataset=reddit-syn-h0] [out.json]"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, GsOpts, _ck
from tools.bench_ops import timeit


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit-syn-h0"
    ds = datagen.make_dataset(name)
    gp, gi, N = ds["g_indptr"], ds["g_indices"], ds["num_nodes"]
    nnz = int(gi.size)
    d = 128
    dev = Device(0); lib = dev.lib
    rng = np.random.default_rng(0)
    x = dev.buf(rng.standard_normal((N, d), dtype=np.float32)); o = dev.buf((N, d))
    gso = GsOpts(); gso.scaling = 2
    res = {"dataset": name, "rows": N, "stored_edges": nnz, "d": d, "runs": []}

    def t_gs(g, acc=0):
        gso.accumulate = acc
        return timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum_ex(dev.ctx, g.h, C.byref(gso), x.ptr, d, o.ptr, d, d), "gs_ex"), iters=10)

    def note(what, ms, **kw):
        r = dict(what=what, ms=ms, gathered_GBps=4.0 * d * nnz / ms / 1e6, **kw)
        res["runs"].append(r)
        print(f"[{name}] {what}: {ms:.3f} ms  {r['gathered_GBps']:.0f} GB/s", flush=True)

    g = dev.graph(gp, gi)
    scheds = [("degree", 0, None, 0), ("dealt-256", 2, None, 256), ("label-major", 1, ds["label"], 0)]
    best = None
    for sname, mode, grp, ng in scheds:
        g.set_schedule(mode, grp, ng)
        ms = t_gs(g)
        note(f"schedule {sname}", ms, schedule=sname)
        if best is None or ms < best[1]:
            best = (sname, ms, mode, grp, ng)
    sname, _, mode, grp, ng = best
    g.set_schedule(mode, grp, ng)
    for gl in (0, 8, 4):
        for gu in (0, 2):
            _ck(lib, lib.gcnhip_ctx_set_option(dev.ctx, b"gs_l", gl), "opt")
            _ck(lib, lib.gcnhip_ctx_set_option(dev.ctx, b"gs_u", gu), "opt")
            note(f"{sname}, slices of {(gl or 16) * 4} floats, {gu or 4} row loads in flight", t_gs(g), gs_l=gl, gs_u=gu, schedule=sname)
    _ck(lib, lib.gcnhip_ctx_set_option(dev.ctx, b"gs_l", 0), "opt")
    _ck(lib, lib.gcnhip_ctx_set_option(dev.ctx, b"gs_u", 0), "opt")
    # hot / cold by in-degree (how often a row is gathered)
    cnt = np.bincount(gi, minlength=N)
    order = np.argsort(-cnt, kind="stable")
    for k in (4096, 8192, 16384, 32768):
        hot = np.zeros(N, bool); hot[order[:k]] = True
        share = float(cnt[order[:k]].sum() / cnt.sum())
        gh, gc = g.restricted(hot), g.restricted(~hot)
        th = t_gs(gh)
        tc = t_gs(gc, acc=1)
        note(f"{sname}, hot = top {k} rows ({k * 256 / 2**20:.1f} MiB of slices, {100 * share:.1f} % of the gathered rows): hot {th:.3f} + cold (accumulating) {tc:.3f}",
             th + tc, hot_rows=k, hot_share=share, hot_ms=th, cold_ms=tc, schedule=sname)
        gh.free(); gc.free()
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
