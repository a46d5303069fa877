"""Run the shipped program — `gcn-hip <dataset> [...]` (host/main.cpp; the reference's entry is src/main.cpp:15-48) — on a
dataset dict: write its binary cache (`<name>.gcnbin`, host/parser.cpp), start the binary as a child process, parse the
reference's output lines (src/seq/gcn.cpp:133-158).  Used by bench.py's `cli` block and tools/run_cli_reddit.py, so that
what is reported is what a user of the command line gets, not what the library can do under another driver.
"""
from __future__ import annotations

import os
import re
import subprocess
import tempfile
import time

from . import datagen

HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(HERE, "bin", "gcn-hip")


def write_cache(ds, root, name=None):
    """<root>/<name>.gcnbin; a dense X goes without its index array (the loader accepts that: every row = 0..F-1)"""
    name = name or ds["name"]
    os.makedirs(root, exist_ok=True)
    d = ds
    if ds["f_val"].size == ds["num_nodes"] * ds["input_dim"]:
        d = dict(ds, f_indices=ds["f_indices"][:0])
    path = os.path.join(root, name + ".gcnbin")
    t0 = time.perf_counter()
    datagen.write_gcnbin(d, path)
    return path, time.perf_counter() - t0


def parse_output(stdout, stderr=""):
    ep = []
    out = {"epochs": ep}
    for line in stdout.splitlines():
        if line.startswith("epoch="):
            ep.append({k: float(v) for k, v in (t.split("=") for t in line.split())})
        elif line.startswith("total training time="):
            out["total_training_time_s"] = float(line.split("=")[1])
        elif line.startswith("test_loss="):
            out["test"] = {k: float(v) for k, v in (t.split("=") for t in line.split())}
        elif line.startswith("Early stopping"):
            out["early_stopped"] = True
    m = re.search(r"dataset loaded in ([0-9.]+) s, model built in ([0-9.]+) s", stderr)
    if m:
        out["load_s"], out["model_build_s"] = float(m.group(1)), float(m.group(2))
    return out


def run(name, cwd, hidden="-", epochs="-", dropout="-", early_stopping="-", env=None, timeout=600):
    """`gcn-hip <name> - - hidden - dropout - - epochs early_stopping` in cwd (which holds data/<name>.gcnbin or the text
    files); returns the parsed lines + wall time of the whole process"""
    if not os.path.exists(BINARY):
        raise RuntimeError(f"{BINARY} not built (make host)")
    args = [BINARY, name, "-", "-", str(hidden), "-", str(dropout), "-", "-", str(epochs), str(early_stopping)]
    t0 = time.perf_counter()
    r = subprocess.run(args, cwd=cwd, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(f"gcn-hip failed ({r.returncode}): {r.stderr[-2000:]}")
    out = parse_output(r.stdout, r.stderr)
    out["command"] = " ".join(["gcn-hip"] + args[1:])
    out["env"] = dict(env or {})
    out["process_wall_s"] = wall
    out["stderr_tail"] = r.stderr[-6000:]
    n = len(out["epochs"])
    if n and out.get("total_training_time_s"):
        out["epochs_per_s"] = n / out["total_training_time_s"]
        out["ms_per_epoch"] = 1e3 * out["total_training_time_s"] / n
        # the first epochs carry one-time costs (scratch sizing, graph capture): the steady rate beside the whole-run one
        if n >= 20:
            tail = out["epochs"][n // 5:]
            t = sum(e["time"] for e in tail)
            out["epochs_per_s_after_warmup"] = len(tail) / t if t > 0 else None
    return out


def run_on_dataset(ds, hidden="-", epochs="-", env=None, workdir=None, keep=False):
    """write the cache into a scratch directory, run the program there, drop the directory"""
    td = workdir or tempfile.mkdtemp(prefix="gcn_cli_")
    try:
        path, t_write = write_cache(ds, os.path.join(td, "data"))
        out = run(ds["name"], td, hidden=hidden, epochs=epochs, env=env)
        out["cache_MB"] = round(os.path.getsize(path) / 1e6, 1)
        out["cache_write_s"] = round(t_write, 2)
        return out
    finally:
        if not keep and not workdir:
            import shutil
            shutil.rmtree(td, ignore_errors=True)
