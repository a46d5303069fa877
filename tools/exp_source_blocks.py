"""Experiment: the hidden-width aggregation as Q launches, launch q gathering only from the source rows of block q of the
communities (temporal blocking of the gathered table: each XCD's L2 then sees 1/Q of it at a time), each writing its own
partial output; against the one-launch aggregation.  Uses gcnhip_graph_create_restricted.
    python tools/exp_source_blocks.py [Q ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck


def timeit(dev, fn, iters=20):
    lib = dev.lib
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.gcnhip_event_create(C.byref(e0)); lib.gcnhip_event_create(C.byref(e1))
    for _ in range(3): fn()
    lib.gcnhip_event_record(dev.ctx, e0)
    for _ in range(iters): fn()
    lib.gcnhip_event_record(dev.ctx, e1)
    ms = C.c_float()
    _ck(lib, lib.gcnhip_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
    return ms.value / iters


def main():
    ds = datagen.make_dataset("reddit-syn")
    gp, gi, lab = ds["g_indptr"], ds["g_indices"], ds["label"]
    N = gp.size - 1
    dev = Device(0); lib = dev.lib
    g = dev.graph(gp, gi, row_group=lab)
    d = 128
    rng = np.random.default_rng(0)
    x = dev.buf(rng.standard_normal((N, d)).astype(np.float32))
    o = dev.buf((N, d))
    base = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, x.ptr, d, o.ptr, d, d), "gs"))
    print(f"one launch: {base:.3f} ms", flush=True)
    C_ = int(lab.max()) + 1
    for Q in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        blocks = (lab.astype(np.int64) * Q // C_).astype(np.int32)
        subs = [g.restricted(blocks == q) for q in range(Q)]
        outs = [dev.buf((N, d)) for _ in range(Q)]
        def run():
            for q in range(Q):
                _ck(lib, lib.gcnhip_graphsum(dev.ctx, subs[q].h, x.ptr, d, outs[q].ptr, d, d), "gs")
        t = timeit(dev, run)
        each = [timeit(dev, (lambda q=q: _ck(lib, lib.gcnhip_graphsum(dev.ctx, subs[q].h, x.ptr, d, outs[q].ptr, d, d), "gs")), iters=10) for q in range(Q)]
        print(f"Q={Q}: {t:.3f} ms for the {Q} launches (+ a combine pass of {Q} x {N * d * 4 / 1e6:.0f} MB); each: {[round(e, 3) for e in each]}", flush=True)
        for s in subs: s.free()
        for b in outs: b.free()


if __name__ == "__main__":
    main()
