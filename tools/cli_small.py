#!/usr/bin/env python3
"""The command line's run loop on a dataset (default: the three small BASELINE configs[1] shapes), us per epoch from
`total training time=`: validation lane (two streams of eager launches) against one stream replaying the captured epoch,
the read-back copy on its own stream against on the producer's, and the reference's wait-per-epoch loop.  Every
configuration runs as its own process, `--repeat` times; the best and the median are printed.
    python tools/cli_small.py [--epochs 4000] [--datasets cora-syn citeseer-syn pubmed-syn reddit-syn] [--repeat 3] [--out X.json]"""
import argparse
import json
import os
import statistics
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import clirun, datagen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=4000, help="per run; reddit-syn always runs 100 at hidden 128")
ap.add_argument("--datasets", nargs="*", default=["cora-syn", "citeseer-syn", "pubmed-syn"])
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--hidden", default="-")
ap.add_argument("--out", default=None)
a = ap.parse_args()
CONFIGS = (("default", {}),
           ("GCN_EVAL_LANE=0 (one stream, captured epoch replayed; read-back copies on it)", {"GCN_EVAL_LANE": "0"}),
           ("GCN_EVAL_LANE=0 HIPGCN_READBACK_STREAM=1", {"GCN_EVAL_LANE": "0", "HIPGCN_READBACK_STREAM": "1"}),
           ("GCN_EVAL_LANE=0 HIPGCN_READBACK_GROUP=1", {"GCN_EVAL_LANE": "0", "HIPGCN_READBACK_GROUP": "1"}),
           ("GCN_EVAL_LANE=1 (two streams of eager launches; read-back copies on the lane's)", {"GCN_EVAL_LANE": "1"}),
           ("GCN_EVAL_LANE=1 HIPGCN_READBACK_STREAM=1", {"GCN_EVAL_LANE": "1", "HIPGCN_READBACK_STREAM": "1"}),
           ("GCN_EVAL_LANE=0 GCN_SYNC_EPOCHS=1 (the reference's loop)", {"GCN_EVAL_LANE": "0", "GCN_SYNC_EPOCHS": "1"}))
doc = {"epochs": a.epochs, "repeat": a.repeat, "unit": "us per epoch = total training time / epochs", "results": []}
for name in a.datasets:
    epochs, hidden = (100, "128") if name.startswith("reddit") else (a.epochs, a.hidden)
    ds = datagen.make_dataset(name)
    td = tempfile.mkdtemp(prefix="gcn_cli_")
    clirun.write_cache(ds, os.path.join(td, "data"))
    del ds
    for label, env in CONFIGS:
        us = []
        for _ in range(a.repeat):
            r = clirun.run(name, td, hidden=hidden, epochs=epochs, env=dict(env, GCN_SEED="1"))
            us.append(1e6 * r["total_training_time_s"] / epochs)
        doc["results"].append({"dataset": name, "epochs": epochs, "hidden": hidden, "config": label, "env": env, "us_per_epoch_best": min(us), "us_per_epoch_median": statistics.median(us)})
        print(f"{name:13s} {label:78s}: best {min(us):7.1f}  median {statistics.median(us):7.1f} us per epoch", flush=True)
if a.out:
    open(a.out, "w").write(json.dumps(doc, indent=1) + "\n")
