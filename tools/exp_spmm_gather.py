#!/usr/bin/env python3
"""Would XCD-bound column slices of W help SparseMatmul on a large sparse X (verdict r04, weak 2: 25 x B_sp of fabric traffic at h = 128)?
The aggregation kernel IS that operation (out[i] = sum_j coef_ij W[col_ij]) with slicing built in, so it stands in for the timing:
X's CSR as an adjacency with F columns, W as the gathered table, 64- / 32- / 16-float column slices (context option gs_l).

    python tools/exp_spmm_gather.py [rows] [nnz_row] [cols]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_spmm import synthetic_x, timeit  # noqa: E402
from cuda_gcn_amd.ops import Device, _ck  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
    h = 128
    ip, ix, v = synthetic_x(n, k, F)
    dev = Device(0)
    lib = dev.lib
    rng = np.random.default_rng(1)
    w = dev.buf(rng.standard_normal((F, h)).astype(np.float32))
    out = dev.buf((n, h))
    ep = dev.buf(np.zeros(1, np.uint32))
    res = {}
    f = dev.feat(ip, ix, v, F)
    ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_spmm_fwd(dev.ctx, f.h, f.values_ptr, w.ptr, h, out.ptr, h, h, 0.0, 1, ep.ptr, 0, None), "spf"), 10)
    res["spmm_fwd"] = ms
    print(f"gcnhip_spmm_fwd (csr row kernel): {ms:.3f} ms", flush=True)
    f.free()
    col_deg = np.maximum(np.bincount(ix, minlength=F), 1).astype(np.int32)      # column popularity plays the column degree
    g = dev.graph(ip, ix, n_cols=F, col_deg=col_deg)
    g.set_schedule(0)
    for gl, name in ((0, "64-float slices (2 per row)"), (8, "32-float slices (4)"), (4, "16-float slices (8)")):
        lib.gcnhip_ctx_set_option(dev.ctx, b"gs_l", gl)
        for u in (4, 2):
            lib.gcnhip_ctx_set_option(dev.ctx, b"gs_u", u)
            ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, w.ptr, h, out.ptr, h, h), "gs"), 10)
            res[f"graphsum_l{gl}_u{u}"] = ms
            print(f"aggregation kernel on X's CSR, {name}, {u} row loads in flight: {ms:.3f} ms", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
