#!/usr/bin/env python3
"""The reference's headline experiment through the shipped command line: `gcn-hip reddit-syn` from its binary cache,
100 epochs, at the report's shape (hidden 16: the reference's default, report.pdf p.8: 106.2 s CUDA / 595.4 s sequential
for 100 Reddit epochs) and at BASELINE.json's (hidden 128).  Writes the cache once, runs every configuration as its own
process, prints one JSON document (load time, model build time, per-epoch `time=` statistics, `total training time=`).

    python tools/run_cli_reddit.py [--out profiles/r04_cli_reddit.json] [--dataset reddit-syn] [--epochs 100]
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuda_gcn_amd import clirun, datagen  # noqa: E402


def summary(r):
    t = [e["time"] for e in r["epochs"]]
    keep = {k: r.get(k) for k in ("command", "env", "load_s", "model_build_s", "total_training_time_s", "epochs_per_s", "ms_per_epoch",
                                  "epochs_per_s_after_warmup", "process_wall_s", "test")}
    keep["n_epochs"] = len(t)
    keep["epoch_time_s"] = {"first": t[0], "median": statistics.median(t), "min": min(t), "max": max(t)}
    keep["last_epoch"] = r["epochs"][-1]
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="reddit-syn")
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--out", default=None)
    ap.add_argument("--hidden", type=int, nargs="*", default=[128, 16])
    a = ap.parse_args()
    t0 = time.perf_counter()
    ds = datagen.make_dataset(a.dataset)
    t_gen = time.perf_counter() - t0
    doc = {"dataset": a.dataset, "nodes": int(ds["num_nodes"]), "stored_edges": int(ds["g_indices"].size),
           "generate_s": round(t_gen, 2),
           "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None,
           "reference_published": "report.pdf p.8 (hidden 16, 100 epochs, Reddit): 106.2 s cuda_gcn on a V100-class GPU, 595.4 s seq_gcn",
           "runs": []}
    td = tempfile.mkdtemp(prefix="gcn_cli_")
    try:
        path, t_write = clirun.write_cache(ds, os.path.join(td, "data"))
        doc["cache_MB"] = round(os.path.getsize(path) / 1e6, 1)
        doc["cache_write_s"] = round(t_write, 2)
        del ds
        for hidden in a.hidden:
            for label, env in (("default (pipelined epochs, validation lane, aggregate-first evaluation)", {}),
                               ("GCN_SYNC_EPOCHS=1 (the reference's loop: wait for every epoch)", {"GCN_SYNC_EPOCHS": "1"}),
                               ("reference operation order, one stream, synchronous (GCN_REFERENCE_ORDER=1 GCN_EVAL_LANE=0 GCN_SYNC_EPOCHS=1)",
                                {"GCN_REFERENCE_ORDER": "1", "GCN_EVAL_LANE": "0", "GCN_SYNC_EPOCHS": "1"})):
                r = clirun.run(a.dataset, td, hidden=hidden, epochs=a.epochs, env=dict(env, GCN_SEED="1"))
                s = summary(r)
                s["hidden"] = hidden
                s["schedule"] = label
                doc["runs"].append(s)
                print(f"[cli] hidden {hidden}, {label}: load {s['load_s']} s, build {s['model_build_s']} s, "
                      f"total training time {s['total_training_time_s']} s = {s['epochs_per_s']:.1f} epochs/s", file=sys.stderr, flush=True)
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)
    txt = json.dumps(doc, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
