#!/usr/bin/env python3
"""One-rank RCCL device time per collective at the epoch's real message sizes (gcnhost_rccl_collective_us): the launch +
kernel floor of an in-place all-gather / all-reduce on the stream — a lower bound on what a peer adds, taken by
tools/comm_model.py instead of an assumed latency.  Run on the GPU box.

    python tools/rccl_latency.py [--dataset reddit-syn] [--hidden 128]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import _lib, datagen  # noqa: E402
from cuda_gcn_amd.model import nccl_unique_id  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dataset", default="reddit-syn")
ap.add_argument("--hidden", type=int, default=128)
a = ap.parse_args()
N, F, Cc = (datagen.SHAPES[a.dataset][i] for i in (0, 1, 2)) if a.dataset in datagen.SHAPES else (1 << int(a.dataset.split("-")[1]), 256, 41)
lib = _lib.gcnhost()
out = {"dataset": a.dataset, "world": 1, "what": "HIP-event time per collective, 50 back-to-back calls on the stream, one rank (no peer): a floor"}
for P in (2, 4, 8):
    rows = (N + P - 1) // P
    for name, ld in (("class width (48 floats per row)", 48), ("hidden width", (a.hidden + 15) // 16 * 16), ("mask bits", (a.hidden + 31) // 32)):
        ag, ar = C.c_double(), C.c_double()
        rc = lib.gcnhost_rccl_collective_us(0, 0, 1, nccl_unique_id(), rows * ld, F * a.hidden + a.hidden * Cc + 4, 50, C.byref(ag), C.byref(ar))
        if rc != 0:
            sys.exit(lib.gcnhost_last_error().decode())
        out[f"P={P} block of {rows} rows, {name}"] = {"allgather_us": round(ag.value, 2), "block_MB": round(rows * ld * 4 / 1e6, 2)}
        out["allreduce_us (weight gradients + 4 scalars)"] = round(ar.value, 2)
print(json.dumps(out, indent=1))
