#!/usr/bin/env python3
"""Round 6, verdict r05 item 1: hub rows of the aggregation served from LDS — priced before it goes anywhere near the product.

Persistent workgroups (tools/gather_peak.hip: gather_lds_kernel) own contiguous chunks of the product's task list; each fills
its LDS once with the H rows its chunk's edges reference most and reads those edges from LDS, the others from the table as
today.  A task's edges are stably partitioned hot-first and the hot count is rounded down to whole rounds of the lane groups,
so the sum keeps the plain kernel's order on that edge order: the check compares the two bit for bit.

    python tools/exp_lds_hubs.py [--dataset reddit-syn] [--schedule label-major] [--out profiles/r06_lds_hubs.json]
    (PMC: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -- python3 tools/exp_lds_hubs.py --one lds|plain ...)
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
sys.path.insert(0, os.path.join(ROOT, "tools"))
from cuda_gcn_amd import datagen  # noqa: E402
import gather_peak as gp          # noqa: E402


def equal_work_bounds(pre, lo, hi, parts, align=1):
    b = [lo]
    for k in range(1, parts):
        tgt = pre[lo] + (pre[hi] - pre[lo]) * k // parts
        t = int(np.searchsorted(pre, tgt))
        t = min(hi, (t + align - 1) // align * align)
        b.append(max(t, b[-1]))
    b.append(hi)
    return b


def lds_plan(e0, e1, tr, idx, n, groups, W, H, G=4, min_refs=2):
    """-> tasks4 [T,4] {e0, e1, nl, row}, indices (slot | row), cols (rows, same order), chunk_bounds, hot [groups*W, H], share"""
    T = e0.size
    nnz = idx.size
    work = (e1 - e0).astype(np.int64) + 8
    pre = np.concatenate([[0], np.cumsum(work)])
    gb = equal_work_bounds(pre, 0, T, groups, 4)
    cb = []
    for g in range(groups):
        cb += equal_work_bounds(pre, gb[g], gb[g + 1], W)[:-1]
    cb.append(T)
    cb = np.array(cb, np.int32)
    n_chunks = groups * W
    task_chunk = np.repeat(np.arange(n_chunks), np.diff(cb))
    by_e0 = np.argsort(e0, kind="stable")
    assert e0[by_e0[0]] == 0 and np.all(e1[by_e0][:-1] == e0[by_e0][1:]) and e1[by_e0[-1]] == nnz, "task edge ranges must partition the edge array"
    edge_task = np.repeat(by_e0, (e1 - e0)[by_e0])
    edge_chunk = task_chunk[edge_task]
    slot = np.full(nnz, -1, np.int32)
    hot = np.full((n_chunks, H), -1, np.int32)
    by_chunk = np.argsort(edge_chunk, kind="stable")
    cstart = np.searchsorted(edge_chunk[by_chunk], np.arange(n_chunks + 1))
    slotmap = np.full(n, -1, np.int32)
    for c in range(n_chunks):
        ed = by_chunk[cstart[c]:cstart[c + 1]]
        if ed.size == 0:
            continue
        cols = idx[ed]
        cnt = np.bincount(cols, minlength=n)
        cand = np.flatnonzero(cnt >= min_refs)
        if cand.size > H:
            o = np.lexsort((cand, -cnt[cand]))[:H]          # most referenced first, ties by row id: deterministic
            cand = cand[o]
        else:
            cand = cand[np.lexsort((cand, -cnt[cand]))]
        hot[c, :cand.size] = cand
        slotmap[cand] = np.arange(cand.size, dtype=np.int32)
        slot[ed] = slotmap[cols]
        slotmap[cand] = -1
    cold = (slot < 0).astype(np.int8)
    pos = np.arange(nnz)
    perm = np.lexsort((pos, cold, e0[edge_task]))           # inside every task: hot edges first, each part in the old order
    new_cols = idx[perm]
    new_slot = slot[perm]
    n_hot = np.zeros(T, np.int64)
    np.add.at(n_hot, edge_task, 1 - cold)
    nl = (n_hot // G * G).astype(np.int32)                # whole rounds of the G lane groups
    # entries past nl of a task keep the table row even if hot
    off = pos - e0[edge_task[perm]]                         # position inside its task (perm keeps tasks' ranges)
    use_slot = off < nl[edge_task[perm]]
    indices = np.where(use_slot, new_slot, new_cols).astype(np.int32)
    assert np.all(indices[use_slot] >= 0)
    tasks4 = np.stack([e0, e1, nl, tr], axis=1).astype(np.int32)
    return tasks4, indices, new_cols.astype(np.int32), cb, hot, float(use_slot.mean()), float((1 - cold).mean())


def load_lib():
    lib = gp.load_lib()
    lib.gl_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p,
                              C.c_int, C.c_int, C.c_int, C.c_long]
    lib.gl_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.POINTER(C.c_float), C.POINTER(C.c_long)]
    lib.gl_run2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.POINTER(C.c_float), C.POINTER(C.c_long)]
    lib.gl_destroy.argtypes = [C.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="reddit-syn")
    ap.add_argument("--schedule", default="label-major")
    ap.add_argument("--out", default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--one", default=None, help="plain | lds: only that kernel of the first configuration (PMC passes)")
    ap.add_argument("--configs", default=None, help="comma list of table:lanes:NW:H:U, table = h (ld 128, d 128) | c (ld 48, d 41) | n (ld 128, 32-float slices)")
    a = ap.parse_args()
    lib = load_lib()
    t0 = time.time()
    ds = datagen.make_dataset(a.dataset)
    n = ds["num_nodes"]
    e0, e1, tr, idx = gp.product_order(ds, key=gp.schedule_key(ds, a.schedule))
    nnz = int(idx.size)
    print(f"[lds_hubs] {a.dataset}/{a.schedule}: {n} rows, {nnz} edges, {e0.size} tasks; {time.time() - t0:.1f} s", flush=True)
    configs = a.configs.split(",") if a.configs else ["h:16:16:319:4", "h:16:16:639:8", "h:16:8:159:4", "n:8:16:639:4", "c:16:16:319:4"]
    doc = {"dataset": a.dataset, "schedule": a.schedule, "rows": n, "edges": nnz, "tasks": int(e0.size), "results": [],
           "kernel": "tools/gather_peak.hip: gather_lds_kernel (persistent, hub rows in LDS) vs gather_plain2_kernel (one wave per task) on the same tasks and edge order"}
    try:
        doc["_meta"] = {"commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or
                        (open(os.path.join(ROOT, ".commit_for_profiles")).read().strip() if os.path.exists(os.path.join(ROOT, ".commit_for_profiles")) else None)}
    except Exception:
        doc["_meta"] = {"commit": None}
    for cfg in configs:
        table, lanes, NW, H, U = cfg.split(":")[:5]
        ordered = int(cfg.split(":")[5]) if cfg.count(":") >= 5 else 0     # > 0: tasks drawn in order, per XCD, in batches of this many
        lanes, NW, H, U = int(lanes), int(NW), int(H), int(U)
        ld, dim = (48, 41) if table == "c" else (128, 128)
        n_slices = 1 if table == "c" else 128 // (lanes * 4)
        groups = 8 // n_slices
        wg_per_cu = max(1, min(32 // NW, (160 * 1024) // (H * lanes * 16 + 16)))   # 32 waves per CU = 8 per SIMD when the LDS allows
        W = 32 * wg_per_cu
        wgs = W
        if ordered:
            W = 1                                          # one hot list per XCD group
        t1 = time.time()
        tasks4, indices, cols, cb, hot, share, share_hot = lds_plan(e0, e1, tr, idx, n, groups, W, max(H, 1), 64 // lanes, min_refs=2 if H else 1 << 30)
        print(f"[lds_hubs] {cfg}: groups {groups} W {W} ({wg_per_cu} workgroups per CU, {H * lanes * 16 // 1024} KB each): edges served from LDS {share:.3f} "
              f"(hot before rounding {share_hot:.3f}); plan {time.time() - t1:.1f} s", flush=True)
        h = C.c_void_p()
        rc = lib.gl_create(C.byref(h), tasks4.ctypes.data, int(tasks4.shape[0]), indices.ctypes.data, cols.ctypes.data, nnz,
                           cb.ctypes.data, np.ascontiguousarray(hot).ctypes.data, groups, W, max(H, 1), max(n, int(tasks4.shape[0])) * 128)
        if rc != 0:
            sys.exit("gl_create failed")
        r = {"config": cfg, "table": {"h": "d=128, 64-float slices", "n": "d=128, 32-float slices", "c": "d=41 (ld 48)"}[table], "lanes_per_row": lanes,
             "waves_per_workgroup": NW, "workgroups_per_cu": wg_per_cu, "hub_rows": H, "lds_KB": H * lanes * 16 // 1024, "loads_in_flight": U,
             "share_of_edges_from_lds": share, "ordered_draw_batch": ordered}
        modes = [("plain", 0), ("lds", 1)] if not a.one else [(a.one, 0 if a.one == "plain" else 1)]
        for rep in range(1 if a.one else 2):               # A/B/A/B: the clock drifts with what ran before
            for name, mode in modes:
                ms = C.c_float(); nd = C.c_long(-2)
                check = 1 if (mode == 1 and rep == 0 and not a.one) else 0
                if lib.gl_run2(h, ld, dim, lanes, U, NW, mode, 1, a.iters, check, ordered, wgs, C.byref(ms), C.byref(nd)) != 0:
                    sys.exit("gl_run failed")
                r.setdefault(name + "_ms", []).append(round(ms.value, 4))
                if check:
                    r["floats_differing_from_plain"] = int(nd.value)
                print(f"[lds_hubs]   {name:5s} {ms.value:.4f} ms  {4.0 * dim * nnz / (ms.value * 1e-3) / 1e9:.0f} GB/s" + (f"  differing floats: {nd.value}" if check else ""), flush=True)
        doc["results"].append(r)
        lib.gl_destroy(h)
        if a.one:
            break
    if a.out:
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
