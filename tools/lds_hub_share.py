#!/usr/bin/env python3
"""CPU pricing of "hub rows served from LDS" (verdict r05 item 1; no GPU needed).

For the product's own task list (tools/gather_peak.py: product_order) the share of gathered rows an LDS-resident hot set of H
rows would serve, per way of choosing the set:
  global   : the H highest-degree columns of the whole graph (one set for every workgroup)
  group    : per XCD task group (4 contiguous task ranges at two column slices), the H most referenced columns of its edges
  chunk    : per persistent workgroup (each XCD group's range cut into W contiguous equal-edge chunks), the H most
             referenced columns of the chunk's own edges
    python tools/lds_hub_share.py reddit-syn [label-major] [--groups 4] [--wgs 64]
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from cuda_gcn_amd import datagen  # noqa: E402
import gather_peak as gp          # noqa: E402


def top_share(cols, H, n):
    cnt = np.bincount(cols, minlength=n)
    if H >= n:
        return 1.0
    top = np.partition(cnt, n - H)[n - H:]
    return float(top.sum()) / max(1, cols.size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset")
    ap.add_argument("schedule", nargs="?", default="label-major")
    ap.add_argument("--groups", type=int, default=4)
    ap.add_argument("--wgs", type=int, nargs="*", default=[32, 64, 128])
    ap.add_argument("--H", type=int, nargs="*", default=[160, 320, 640, 1280])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    t0 = time.time()
    ds = datagen.make_dataset(a.dataset)
    n = ds["num_nodes"]
    e0, e1, tr, idx = gp.product_order(ds, key=gp.schedule_key(ds, a.schedule))
    nnz = idx.size
    print(f"{a.dataset}/{a.schedule}: {n} rows {nnz} edges {e0.size} tasks ({time.time()-t0:.1f}s)", flush=True)
    deg = np.diff(ds["g_indptr"].astype(np.int64))
    work = (e1 - e0).astype(np.int64) + 8
    pre = np.concatenate([[0], np.cumsum(work)])
    doc = {"dataset": a.dataset, "schedule": a.schedule, "rows": n, "edges": int(nnz), "rows_of_result": []}
    def bounds(lo, hi, parts):
        b = [lo]
        for k in range(1, parts):
            tgt = pre[lo] + (pre[hi] - pre[lo]) * k // parts
            b.append(int(np.searchsorted(pre, tgt)))
        b.append(hi)
        return b
    gb = bounds(0, e0.size, a.groups)
    # edges of a task range as one flat array (tasks are contiguous edge ranges, but not contiguous to each other)
    def edges_of(t0_, t1_):
        if t1_ <= t0_:
            return np.zeros(0, np.int32)
        return np.concatenate([idx[e0[t]:e1[t]] for t in range(t0_, t1_)]) if t1_ - t0_ < 64 else idx[np.concatenate([np.arange(e0[t], e1[t]) for t in range(t0_, t1_)])]
    for H in a.H:
        order = np.argsort(-deg, kind="stable")[:H]
        hot = np.zeros(n, bool); hot[order] = True
        r = {"H": H, "global": float(hot[idx].mean())}
        grp = 0.0
        for g in range(a.groups):
            ce = edges_of(gb[g], gb[g + 1])
            grp += top_share(ce, H, n) * ce.size
        r["group"] = grp / nnz
        for W in a.wgs:
            tot = 0.0
            for g in range(a.groups):
                cb = bounds(gb[g], gb[g + 1], W)
                for w in range(W):
                    ce = edges_of(cb[w], cb[w + 1])
                    if ce.size:
                        tot += top_share(ce, H, n) * ce.size
            r[f"chunk_W{W}"] = tot / nnz
        doc["rows_of_result"].append(r)
        print(json.dumps(r), flush=True)
    if a.out:
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
