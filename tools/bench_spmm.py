#!/usr/bin/env python3
"""SparseMatmul on a SPARSE feature matrix (the CSR forward / CSC backward kernels of csrc/spmm.hip; reference:
src/seq/module.cpp:47-77, src/cuda/cuda_kernel.cu:100-122), through the C-ABI, timed with HIP events.

Legs: a large synthetic X outside launch-bound territory (default N = 2 M rows, 50 stored values per row, F = 50 000
columns, h = 128 and 16) and the BASELINE configs[1] shapes (cora-syn, citeseer-syn, pubmed-syn at h = 16).
Reported against SURVEY §8(d)'s contract figure  B_sp = 4(N+1) + 8 nnzX + 4 F h + 4 N h  (index + value streams once,
W once, the output once) and against the gather model  B_gather = 8 nnzX + 4 nnzX h + 4 N h  (one W row per stored value:
what actually crosses the L2 when W does not fit LDS).

    python tools/bench_spmm.py [--rows 2000000] [--nnz-row 50] [--cols 50000] [--only big|small] [--iters 20]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen  # noqa: E402
from cuda_gcn_amd.ops import Device, _ck  # noqa: E402


def timeit(dev, fn, iters, warmup=3):
    lib = dev.lib
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.gcnhip_event_create(C.byref(e0)); lib.gcnhip_event_create(C.byref(e1))
    for _ in range(warmup):
        fn()
    dev.sync()
    lib.gcnhip_event_record(dev.ctx, e0)
    for _ in range(iters):
        fn()
    lib.gcnhip_event_record(dev.ctx, e1)
    ms = C.c_float()
    _ck(lib, lib.gcnhip_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
    lib.gcnhip_event_destroy(e0); lib.gcnhip_event_destroy(e1)
    return ms.value / iters


def synthetic_x(n, nnz_row, f, seed=0, skew=0.0):
    """n rows of nnz_row distinct sorted columns (uniform, or a share `skew` drawn from the first f/64 columns: a few
    dense columns, as bag-of-words features have) and U(0, 0.2) values"""
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, f, (n, nnz_row), dtype=np.int64)
    if skew > 0:
        hot = rng.random((n, nnz_row)) < skew
        cols = np.where(hot, rng.integers(0, max(1, f // 64), (n, nnz_row)), cols)
    cols.sort(axis=1)
    # make the columns of a row distinct: bump duplicates (keeps them sorted; the last may wrap, which only matters to realism)
    dup = np.zeros_like(cols, dtype=bool)
    dup[:, 1:] = cols[:, 1:] <= cols[:, :-1]
    for _ in range(4):
        if not dup.any():
            break
        cols = cols + dup
        cols.sort(axis=1)
        dup[:, 1:] = cols[:, 1:] <= cols[:, :-1]
    cols = np.minimum(cols, f - 1)
    vals = rng.uniform(0.0, 0.2, n * nnz_row).astype(np.float32)
    indptr = (np.arange(n + 1, dtype=np.int64) * nnz_row).astype(np.int32)
    return indptr, cols.reshape(-1).astype(np.int32), vals


def leg(dev, tag, indptr, indices, vals, F, h, iters, p_drop=0.5):
    lib = dev.lib
    N, nnz = indptr.size - 1, int(indices.size)
    f = dev.feat(indptr, indices, vals, F)
    rng = np.random.default_rng(1)
    ld = (h + 3) // 4 * 4
    w = dev.buf(rng.standard_normal((F, ld)).astype(np.float32))
    out = dev.buf((N, ld))
    dout = dev.buf(rng.standard_normal((N, ld)).astype(np.float32))
    dw = dev.buf((F, ld))
    ep = dev.buf(np.zeros(1, np.uint32))
    b_sp = 4 * (N + 1) + 8 * nnz + 4 * F * h + 4 * N * h
    b_gather_f = 8 * nnz + 4 * nnz * h + 4 * N * h
    b_gather_b = 12 * nnz + 4 * nnz * h + 4 * F * h         # CSC: row + position + value per entry
    res = {"N": N, "nnz": nnz, "F": F, "h": h, "B_sp": b_sp, "W_MB": round(F * ld * 4 / 1e6, 2), "dout_MB": round(N * ld * 4 / 1e6, 2)}
    for pd in (0.0, p_drop):
        ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_spmm_fwd(dev.ctx, f.h, f.values_ptr, w.ptr, ld, out.ptr, ld, h, pd, 1, ep.ptr, 0, None), "spf"), iters)
        res[f"fwd_p{pd}"] = dict(ms=ms, B_sp_GBps=b_sp / ms / 1e6, gather_GBps=b_gather_f / ms / 1e6)
        print(f"[{tag}] h={h} fwd drop={pd}: {1e3 * ms:.1f} us  B_sp/t {b_sp / ms / 1e6:.0f} GB/s  gather model {b_gather_f / ms / 1e6:.0f} GB/s", flush=True)
        ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_spmm_bwd(dev.ctx, f.h, f.values_ptr, dout.ptr, ld, dw.ptr, ld, h, pd, 1, ep.ptr, 0, None), "spb"), iters)
        res[f"bwd_p{pd}"] = dict(ms=ms, B_sp_GBps=b_sp / ms / 1e6, gather_GBps=b_gather_b / ms / 1e6)
        print(f"[{tag}] h={h} bwd drop={pd}: {1e3 * ms:.1f} us  B_sp/t {b_sp / ms / 1e6:.0f} GB/s  gather model {b_gather_b / ms / 1e6:.0f} GB/s", flush=True)
    for b in (w, out, dout, dw, ep):
        b.free()
    f.free()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--nnz-row", type=int, default=50)
    ap.add_argument("--cols", type=int, default=50_000)
    ap.add_argument("--only", choices=["big", "small", "skew", "lds", "rows"], default=None)
    ap.add_argument("--hidden", type=int, nargs="*", default=[128, 16])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = Device(0)
    doc = {}
    if a.only in (None, "small"):
        for name in ("cora-syn", "citeseer-syn", "pubmed-syn"):
            ds = datagen.make_dataset(name)
            doc[name] = leg(dev, name, ds["f_indptr"], ds["f_indices"], ds["f_val"], ds["input_dim"], 16, max(a.iters, 200))
    if a.only in (None, "big"):
        t0 = time.time()
        ip, ix, v = synthetic_x(a.rows, a.nnz_row, a.cols)
        print(f"synthetic X: {a.rows} x {a.cols}, {ix.size} stored values, built in {time.time() - t0:.1f} s", flush=True)
        for h in a.hidden:
            doc[f"big_h{h}"] = leg(dev, "big", ip, ix, v, a.cols, h, a.iters)
    if a.only in (None, "skew"):
        ip, ix, v = synthetic_x(a.rows // 4, a.nnz_row, a.cols, skew=0.3)
        cnt = np.bincount(ix, minlength=a.cols)
        print(f"skewed X: longest column {cnt.max()} entries, median {int(np.median(cnt))}", flush=True)
        for h in a.hidden:
            doc[f"skew_h{h}"] = dict(leg(dev, "skew", ip, ix, v, a.cols, h, a.iters), longest_column=int(cnt.max()))
    if a.only == "lds":
        # W staged in LDS (option spmm_lds = 1) against rows gathered through L1/L2 (0): shapes whose W fits LDS
        for name in ("pubmed-syn", "cora-syn"):
            ds = datagen.make_dataset(name)
            for lds, general in ((0, 1), (1, 1), (0, -1), (1, -1)):
                dev.set_option("spmm_lds", lds)
                dev.set_option("spmm_general", general)
                tag = ("general" if general == 1 else "narrow") + ("+lds" if lds else "")
                doc[f"{name}_{tag}"] = leg(dev, f"{name} {tag}", ds["f_indptr"], ds["f_indices"], ds["f_val"], ds["input_dim"], 16, 300)
        for F in (500, 2000):
            ip, ix, v = synthetic_x(a.rows, a.nnz_row, F)
            for lds, general in ((0, 1), (1, 1), (0, 0), (1, 0)):
                dev.set_option("spmm_lds", lds)
                dev.set_option("spmm_general", general)
                tag = ("general" if general else "narrow") + ("+lds" if lds else "")
                doc[f"F{F}_{tag}"] = leg(dev, f"N={a.rows} F={F} {tag}", ip, ix, v, F, 16, a.iters)
        dev.set_option("spmm_lds", 0)
        dev.set_option("spmm_general", 0)
    if a.only == "rows":
        ip, ix, v = synthetic_x(a.rows, a.nnz_row, a.cols)
        for h in a.hidden:
            for k in (1, 2, 4, 8, 16):
                dev.set_option("spmm_rows", k)
                doc[f"big_h{h}_rows{k}"] = leg(dev, f"rows/wave={k}", ip, ix, v, a.cols, h, a.iters)
        dev.set_option("spmm_rows", 0)
    dev.close()
    txt = json.dumps(doc, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
