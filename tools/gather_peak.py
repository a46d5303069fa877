#!/usr/bin/env python3
"""Measured ceiling of the aggregation's row gather on this chip (driver of tools/gather_peak.hip; run on the GPU box).

For the dataset's own adjacency in the product's order (rows label-major / descending degree, neighbours of a row by
descending degree, rows above 1024 edges cut into segments, 256-byte column slices bound to XCD groups) and for a uniformly
random index stream with the same row lengths, a gather-and-sum kernel with no coefficient stream, no multiply and no
epilogue is swept over row loads in flight (U), resident waves per SIMD and with / without storing the result row.
Reports gathered GB/s (4 bytes x gathered floats per edge x edges / time) per configuration and the best per table;
bench.py prices `roofline.frac` of the hidden-width aggregation against `ceiling_GBps` of the matching table.

    python tools/gather_peak.py [--dataset reddit-syn] [--out profiles/r04_gather_peak.json] [--quick]
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -- python3 tools/gather_peak.py --best-only
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuda_gcn_amd import datagen  # noqa: E402
from cuda_gcn_amd.provenance import source_sha  # noqa: E402

LIB = os.path.join(ROOT, "build", "libgatherpeak.so")


def load_lib():
    if not os.path.exists(LIB):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                        os.path.join(ROOT, "tools", "gather_peak.hip"), "-o", LIB], check=True)
    lib = C.CDLL(LIB)
    lib.gp_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_long]
    lib.gp_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.gp_run2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.gp_destroy.argtypes = [C.c_void_p]
    return lib


def schedule_key(ds, schedule):
    """the row-group key csrc/ctx.hip sorts the task list by (gcnhip_graph_set_schedule): labels, nothing, or degree rank dealt
    into G groups"""
    n = ds["num_nodes"]
    deg = np.diff(ds["g_indptr"].astype(np.int64))
    if schedule == "label-major":
        return ds["label"].astype(np.int64)
    if schedule.startswith("dealt-"):
        G = int(schedule.split("-")[1])
        order = np.argsort(-deg, kind="stable")
        key = np.empty(n, np.int64)
        key[order] = np.arange(n) % G
        return key
    return np.zeros(n, np.int64)


def product_order(ds, group_major=True, key=None):
    """the task list and index array as csrc/ctx.hip builds them: neighbours of a row by descending degree (stable),
    rows by (group key, descending degree) — key = label when group_major, else none; rows above 1024 edges in 1024-edge segments"""
    gp, gi = ds["g_indptr"].astype(np.int64), ds["g_indices"]
    n = gp.size - 1
    deg = np.diff(gp)
    row_of = np.repeat(np.arange(n), deg)
    # per row: neighbours sorted by (-degree, id) — std::sort of pairs (-deg, id) in graph_create_impl
    order = np.lexsort((gi, -deg[gi], row_of))
    idx = gi[order].astype(np.int32)
    if key is None:
        key = ds["label"].astype(np.int64) if group_major else np.zeros(n, np.int64)
    rows = np.lexsort((np.arange(n), -deg, key))                 # stable: (key asc, degree desc)
    e0, e1, tr = [], [], []
    for r in rows.tolist():
        a, b = int(gp[r]), int(gp[r + 1])
        if b - a <= 1024:
            e0.append(a); e1.append(b); tr.append(r)
        else:
            for s in range(a, b, 1024):
                e0.append(s); e1.append(min(b, s + 1024)); tr.append(r)
    return np.array(e0, np.int32), np.array(e1, np.int32), np.array(tr, np.int32), idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="reddit-syn")
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--best-only", action="store_true", help="only the best configuration of each stream (for a PMC pass)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--schedule", default="label-major", help="row schedule of the product's own stream: label-major | degree | dealt-<G> (what HipGCN picked for this graph)")
    ap.add_argument("--coef", action="store_true", help="also time the gather with a coefficient stream (separate array / interleaved pairs)")
    a = ap.parse_args()
    lib = load_lib()
    t0 = time.time()
    ds = datagen.make_dataset(a.dataset)
    n = ds["num_nodes"]
    nnz = int(ds["g_indices"].size)
    e0, e1, tr, idx = product_order(ds, key=schedule_key(ds, a.schedule))
    print(f"[gather_peak] {a.dataset}: {n} rows, {nnz} stored edges, {e0.size} tasks; host preparation {time.time() - t0:.1f} s", flush=True)
    rng = np.random.default_rng(1)
    own = f"own index stream, {a.schedule} row order (the product's schedule)"
    streams = {own: (e0, e1, tr, idx)}
    if a.schedule != "degree":
        streams["own index stream, descending-degree row order"] = product_order(ds, False)
    streams["uniformly random rows, same row lengths"] = (e0, e1, tr, rng.integers(0, n, nnz).astype(np.int32))
    tables = [("d=128 (ld 128, 256-byte slices on XCD pairs)", 128, 128), ("d=41 (ld 48)", 48, 41)]
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or None
    except Exception:
        commit = None
    doc = {"dataset": a.dataset, "rows": n, "stored_edges": nnz, "tasks": int(e0.size), "schedule": a.schedule,
           "_meta": {"commit": commit or (open(os.path.join(ROOT, ".commit_for_profiles")).read().strip() if os.path.exists(os.path.join(ROOT, ".commit_for_profiles")) else None),
                     "sources": source_sha(["graphsum.hip", "tools/gather_peak.hip"])},
           "kernel": "tools/gather_peak.hip: gather + sum only (no coefficient stream, no multiply, no epilogue); one wave per task, XCD-sliced like graphsum_vec_kernel",
           "unit": "GB/s of gathered row bytes = 4 * dim * edges / avg launch time", "results": []}
    sweeps_u = (4,) if a.best_only else ((2, 4) if a.quick else (1, 2, 4, 8))
    sweeps_w = (8,) if a.best_only else ((8,) if a.quick else (4, 6, 8))
    for sname, (se0, se1, str_, sidx) in streams.items():
        h = C.c_void_p()
        se0, se1, str_, sidx = (np.ascontiguousarray(x, np.int32) for x in (se0, se1, str_, sidx))
        rc = lib.gp_create(C.byref(h), se0.ctypes.data, se1.ctypes.data, str_.ctypes.data, int(se0.size), sidx.ctypes.data, nnz, n * 128)
        if rc != 0:
            sys.exit("gp_create failed")
        for tname, ld, dim in tables:
            best = None
            for store in ((0,) if a.best_only else (0, 1)):
                for U in sweeps_u:
                    for w in sweeps_w:
                        ms = C.c_float()
                        if lib.gp_run(h, ld, dim, U, w, store, a.iters, C.byref(ms)) != 0:
                            sys.exit("gp_run failed")
                        gbps = 4.0 * dim * nnz / (ms.value * 1e-3) / 1e9
                        r = {"stream": sname, "table": tname, "table_MB": round(n * ld * 4 / 1e6, 1), "loads_in_flight": U, "waves_per_simd": w,
                             "store_result": bool(store), "ms": ms.value, "GBps": gbps}
                        doc["results"].append(r)
                        print(f"[gather_peak] {sname[:28]:28s} {tname[:6]:6s} U={U} waves/SIMD={w} store={store}: {ms.value:.3f} ms  {gbps:.0f} GB/s", flush=True)
                        if best is None or gbps > best["GBps"]:
                            best = r
            doc.setdefault("best", []).append(best)
            if a.coef and sname == own:
                # what the product's coefficient stream costs on top of the pure gather, and whether interleaving it with the
                # indices (one 8-byte load per lane instead of two 4-byte loads) gets that back
                for mode, label in ((0, "no coefficients"), (1, "coefficients from their own array"), (2, "(index, coefficient) pairs, one array")):
                    ms = C.c_float()
                    lib.gp_run2(h, ld, dim, 4, 8, 1, a.iters, mode, C.byref(ms))
                    doc.setdefault("coefficient_stream", []).append({"table": tname, "mode": label, "ms": ms.value, "GBps": 4.0 * dim * nnz / (ms.value * 1e-3) / 1e9})
                    print(f"[gather_peak] coefficient stream, {tname[:6]}: {label}: {ms.value:.3f} ms", flush=True)
        lib.gp_destroy(h)
    # the ceilings bench.py uses: the product's own stream and schedule, best configuration, per table
    doc["ceiling_GBps"] = {b["table"].split(" ")[0]: b["GBps"] for b in doc["best"] if b["stream"] == own}
    doc["ceiling_uniform_random_GBps"] = {b["table"].split(" ")[0]: b["GBps"] for b in doc["best"] if b["stream"].startswith("uniformly")}
    txt = json.dumps(doc, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(json.dumps({k: doc[k] for k in ("ceiling_GBps", "ceiling_uniform_random_GBps")}))


if __name__ == "__main__":
    main()
