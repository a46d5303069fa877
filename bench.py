#!/usr/bin/env python3
"""bench.py — Reddit-shape full-batch GCN training on N MI355X of one node.

A "step" is one EPOCH exactly as the reference times it: train_epoch() +
eval(validation) (src/seq/gcn.cpp:136-140 of the reference) = 2 forwards, 1
backward, Adam, 2 x (loss, L2, accuracy).  Workload: reddit-syn (BASELINE.json
configs[2]: 232 965 nodes, 11 606 919 undirected edges, 602 dense features ->
128 hidden -> 41 classes; synthetic, seeded — the real dataset is not
available offline).  All inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
RANK / WORLD_SIZE are then set), or this file does: a parent that has not
touched the GPU (no torch import, no native library) starts the N ranks with
torch.distributed.run as CHILD processes under ONE overall deadline
(GCN_BENCH_DEADLINE, default 420 s).  It first runs the RCCL self-test
(all-gather, all-reduce, halo send/receive, split communicator) as a throw-away
group of ranks, then the plain one-stream schedule, whose JSON line it prints at
once; then, while enough of the deadline is left, the two overlapped schedules
(exchanges on their own stream beside the locally owned columns' aggregation;
validation forward on a second stream) as fresh children, and prints the fastest
valid line last.

Prints ONE JSON line on rank 0 (contract fields + "roofline" + "cpu_baseline").
  * `value` / `ms_per_step`: exactly K epochs, barrier + synchronize on both sides, max over ranks,
    on the product's default path (per-op timers OFF: one GPU replays the captured hipGraph epoch);
  * `bursts`: the same timed region repeated, median / min / max epochs/s;
  * `roofline`: the dominant kernel (GraphSum at the hidden width) timed with HIP events on its own
    stream in a separate pass of the same process (timers ON), against the bound that binds it;
  * `roofline.hbm_regime`: the same kernel on an R-MAT graph whose table (1 GiB) is far beyond the
    Infinity Cache — the fraction of the HBM roofline proper;
  * `value_no_label_hint`: the headline with the dataset's labels withheld from the aggregation's row schedule (row
    groups are then found in the graph itself by modularity local moving); `value_no_row_groups`: no row groups at all.
"""
import argparse
import json
import os
import signal
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only supports dmabuf IPC (RCCL needs it)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (~6.3 TB/s achievable)
L2_AGG_GBPS = 34500.0         # "L2 (per XCD)": 4 MiB per XCD, ~34.5 TB/s aggregate
MALL_GATHER_GBPS = 8600.0     # "Indexed rows": 38 MB table, uniformly random rows served by the Infinity Cache: 33.5 GB/s per CU = 8.6 TB/s
F32_MFMA_PEAK_TF = 157.3      # "Peak FP32 (matrix)": v_mfma_f32_32x32x2_f32 / 16x16x4_f32, 64 FLOP/clk/SIMD at 2.4 GHz
BF16_MFMA_PEAK_TF = 2500.0    # "Peak BF16/FP16 MFMA": ~2.5 PF dense
PMC_FILES = ["r06_graphsum_pmc.json", "r05_graphsum_pmc.json", "r04_graphsum_pmc.json", "r03_graphsum_pmc.json", "r02_graphsum_pmc.json"]           # newest first (profiles/)
PMC_RMAT_FILES = ["r06_graphsum_pmc_rmat.json", "r05_graphsum_pmc_rmat.json", "r04_graphsum_pmc_rmat.json", "r03_graphsum_pmc_rmat.json", "r02_graphsum_pmc_rmat.json"]
PMC_RMAT22_FILES = ["r04_graphsum_pmc_rmat22.json"]                      # the in-model launch of BASELINE configs[4] (scale 22, 2 GiB table)
GATHER_PEAK_FILES = ["r06_gather_peak.json", "r05_gather_peak.json", "r04_gather_peak.json"]     # measured ceiling of the cache-regime gather (tools/gather_peak.py)
GEMM_PMC_FILES = ["r06_gemm_bf16x3_pmc.json", "r05_gemm_bf16x3_pmc.json", "r05_gemm_pmc.json"]                                   # SQ_VALU_MFMA_BUSY_CYCLES etc. of the dense first-layer kernels (tools/pmc_gemm.sh)
STRUCTURE_LEGS = [("value_structure_free", "reddit-syn-h0"), ("value_h03", "reddit-syn-h03"), ("value_zipf", "reddit-syn-zipf")]
GS_KERNEL = "graphsum_vec_kernel<16, 4, true, false>"       # the hidden-width launch on a cache-resident table (graphsum.hip, launch_vec)
GS_KERNEL_HBM = "graphsum_vec_kernel<16, 2, true, false>"   # ... past the Infinity Cache (two row loads in flight)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def graph_structure(ds):
    """what the generator planted and what the graph shows: edge homophily, class sizes, degrees"""
    import numpy as np
    from cuda_gcn_amd import datagen
    if "planted_homophily" not in ds:
        return None
    deg = np.diff(ds["g_indptr"])
    cls = np.bincount(ds["label"], minlength=ds["output_dim"])
    return {"generator": "Chung-Lu, power-law expected degrees (exponent ~2.3, capped at 2e4), seeded (cuda_gcn_amd/datagen.py)",
            "planted_homophily": ds["planted_homophily"], "class_sizes": ds["class_sizes"],
            "measured_edge_homophily": round(datagen.edge_homophily(ds), 4),
            "largest_class_rows": int(cls.max()), "smallest_class_rows": int(cls.min()),
            "largest_class_slice_MB": round(int(cls.max()) * 256 / 1e6, 2),       # its rows as 256-byte column slices (one XCD's L2: 4 MiB)
            "max_degree": int(deg.max()), "mean_degree": round(float(deg.mean()), 2)}


def structure_note(ds):
    if "planted_homophily" not in ds:
        return ""
    return (f"; synthetic graph with PLANTED community structure: {100 * ds['planted_homophily']:.0f} % of the edges drawn inside the "
            f"endpoint's class ({ds['output_dim']} {ds['class_sizes']}-size classes = the labels); structure-free and other variants: "
            "value_structure_free / value_h03 / value_zipf")


def b_gs(n_rows, nnz, d, in_bytes=4):
    """algorithmic bytes of one GraphSum call: indptr + indices + one d-value
    neighbour row per edge (f32, or 2-byte bf16 with --bf16-tables) + one f32 output row per node"""
    return 4 * (n_rows + 1) + 4 * nnz + in_bytes * nnz * d + 4 * n_rows * d


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--dataset", default="reddit-syn")
    ap.add_argument("--bursts", type=int, default=4, help="extra repetitions of the K-step timed region (median/min/max)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cli", action="store_true", help="skip the `cli` block (the shipped gcn-hip binary on the same workload from its .gcnbin cache)")
    ap.add_argument("--cli-epochs", type=int, default=100, help="epochs of the `cli` block's run (the reference's default epoch count)")
    ap.add_argument("--no-extras", action="store_true", help="skip the HBM-regime leg, the structure legs, the structure-blind reruns and the `cli` block")
    ap.add_argument("--no-structure-legs", action="store_true",
                    help="skip the Reddit-shaped graphs with other planted structure (homophily 0 / 0.3, Zipf class sizes)")
    ap.add_argument("--pin-schedule", default="label", help="--profile-run: the row schedule to pin (label | degree | dealt-256 | structure)")
    ap.add_argument("--profile-run", action="store_true",
                    help="the run to put under rocprofv3 --kernel-trace --stats: one stream (no validation lane), the aggregation's row schedule "
                         "pinned (HIPGCN_SCHEDULE, no tuning launches), no extras / cli / CPU baseline, so that the summary's average per kernel "
                         "is the average of the timed launches")
    ap.add_argument("--sustain", type=int, default=1500, help="epochs of the long back-to-back region of the extras (0: skip)")
    ap.add_argument("--hbm-scale", type=int, default=21, help="R-MAT scale of the HBM-regime leg (21: 1 GiB table at d=128)")
    ap.add_argument("--no-row-groups", action="store_true", help="plain descending-degree aggregation schedule (no label hint)")
    ap.add_argument("--bf16-tables", action="store_true",
                    help="opt-in, NOT the headline: GraphSum gathers bfloat16 copies of its inputs (f32 sums); reported as dtype f32+bf16-tables")
    ap.add_argument("--eval-lane", choices=["auto", "on", "off"], default="auto",
                    help="validation forward on a second stream, overlapped with the next training epoch (auto: on with one GPU, off with several)")
    ap.add_argument("--overlap", choices=["on", "off"], default="off",
                    help="several GPUs: exchanges on their own stream beside the aggregation of the locally owned columns (HIPGCN_OVERLAP_EXCHANGE)")
    ap.add_argument("--selftest", action="store_true", help="several GPUs: run gcnhost_rccl_selftest_world on every rank and exit")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher (N > 1)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _descendants(pid):
    """every live process below `pid` (children, their children, ...), from /proc: torch.distributed.run starts each rank in a
    session of its own, so killing the launcher's process group alone leaves hung ranks behind — holding the GPU and our pipe"""
    kids = {}
    for d in os.listdir("/proc"):
        if not d.isdigit():
            continue
        try:
            with open(f"/proc/{d}/stat") as f:
                st = f.read()
            ppid = int(st[st.rindex(")") + 2:].split()[1])
        except (OSError, ValueError, IndexError):
            continue
        kids.setdefault(ppid, []).append(int(d))
    out, todo = [], [pid]
    while todo:
        for c in kids.get(todo.pop(), []):
            out.append(c)
            todo.append(c)
    return out


def _kill_tree(p):
    """SIGKILL exactly the processes this parent started through `p`: the launcher's group and every descendant found under it"""
    tree = _descendants(p.pid)
    for target in [p.pid] + tree:
        try:
            os.killpg(os.getpgid(target), signal.SIGKILL) if target == p.pid else os.kill(target, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    for _ in range(20):                          # ranks started between the listing and the kill
        more = [c for c in _descendants(p.pid) if c not in tree]
        if not more:
            break
        for c in more:
            try:
                os.kill(c, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
        tree += more
        time.sleep(0.1)
    return len(tree) + 1


def _run_ranks(n_gpus, extra, timeout, env=None):
    """one group of N rank processes (children of this GPU-free parent); returns (rc, JSON line or None)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + extra
    log("starting", n_gpus, "ranks:", " ".join(cmd[1:]), "" if not env else f"env {env}")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True, env=dict(os.environ, **(env or {})))
    try:
        out, _ = p.communicate(timeout=timeout)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        log(f"ranks still running after {timeout:.0f} s: killing the {n_gpus}-rank group under pid {p.pid}")
        n = _kill_tree(p)                        # exactly what this parent started: the launcher's group and every rank below it
        try:
            out, _ = p.communicate(timeout=15)
        except subprocess.TimeoutExpired:        # something still holds the pipe: do not wait for it
            log(f"the pipe of the killed group did not close ({n} processes signalled): moving on")
            out = ""
            try:
                p.stdout.close()
            except Exception:
                pass
        rc = 124
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith("{") and ('"metric"' in ln or '"selftest"' in ln):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if rc == 0 and line is None:
        rc = 1
    return rc, line


def launch_ranks(n_gpus, argv):
    """Parent of an N-rank run.  Runs before torch or any native library is imported: this process never
    initialises the GPU, it only starts children, so nothing here is an exec from a GPU process.

    ONE overall deadline (GCN_BENCH_DEADLINE seconds from now, default 420) covers everything below; every step takes its
    time limit from what is left and an OPTIONAL step is skipped when less than its own estimate remains, so the whole
    call ends inside the deadline whatever hangs.  Order (each step a fresh group of children, so that a hang or failure
    of a later, less proven step cannot take an earlier result with it):
      0. the RCCL self-test (gcnhost_rccl_selftest_world: all-gather, all-reduce, the halo exchange by grouped
         ncclSend/ncclRecv, the split communicator) with a short limit.  Passed: the model runs may pick the exchange
         per graph (HIPGCN_EXCHANGE=auto).  Failed, hung or skipped: they are pinned to the in-place all-gather.
      1. the plain schedule: everything on one stream.  Its JSON line is the safe result and is PRINTED THE MOMENT IT
         EXISTS (flushed), before anything optional is started.  If it fails with a per-graph exchange it is retried once
         pinned to the all-gather — if the time left allows.
      2. unless the caller chose a schedule: `--overlap on` (exchanges on their own stream beside the aggregation of the
         locally owned columns), then `--eval-lane on` (validation forward on a second stream and communicator).
    When an optional schedule beat the plain one, the best line is printed LAST (a reader that keeps the last JSON line
    gets the best valid result, one that keeps the first gets the safe one); `config.schedule` says which and
    `other_schedules` lists the rest, with what was skipped or killed and why."""
    t_start = time.monotonic()
    deadline = t_start + float(os.environ.get("GCN_BENCH_DEADLINE", "420"))
    margin = 5.0                                     # to kill a group, collect its output and print

    def left():
        return deadline - time.monotonic() - margin
    timeout = float(os.environ.get("GCN_BENCH_TIMEOUT", "600"))
    t_try = float(os.environ.get("GCN_BENCH_LANE_TIMEOUT", "300"))
    argv = list(argv)
    env = {}
    # what a plain run needs at the very least (import torch on a fresh box, the dataset, the model build, the epochs): never
    # spend on the self-test what would leave the safe result less than this
    plain_estimate = float(os.environ.get("GCN_BENCH_PLAIN_ESTIMATE", "150"))
    if "HIPGCN_EXCHANGE" not in os.environ and os.environ.get("GCN_BENCH_TRANSPORT") != "host":
        t_self = min(float(os.environ.get("GCN_BENCH_SELFTEST_TIMEOUT", "180")), left() - plain_estimate)
        if t_self >= 30:
            rc0, line0 = _run_ranks(n_gpus, ["--gpus", str(n_gpus), "--selftest"], t_self)
            selftest_ok = rc0 == 0
            log("RCCL self-test", "passed: exchange decided per graph" if selftest_ok else f"FAILED (rc {rc0}): every exchange pinned to the all-gather")
        else:
            selftest_ok = False
            log(f"RCCL self-test skipped ({left():.0f} s left of the deadline): every exchange pinned to the all-gather")
        env["HIPGCN_EXCHANGE"] = "auto" if selftest_ok else "allgather"
    chosen = "--eval-lane" in argv or "--overlap" in argv
    base = argv if chosen else argv + ["--eval-lane", "off", "--overlap", "off"]
    t0 = time.monotonic()
    rc, line = _run_ranks(n_gpus, base, max(10.0, min(timeout, left())), env)
    t_plain = time.monotonic() - t0
    if rc != 0 and env.get("HIPGCN_EXCHANGE") == "auto" and left() >= min(plain_estimate, 1.2 * t_plain):
        log(f"run failed (rc {rc}) with a per-graph exchange: retrying pinned to the all-gather")
        env["HIPGCN_EXCHANGE"] = "allgather"
        t0 = time.monotonic()
        rc, line = _run_ranks(n_gpus, base, max(10.0, min(timeout, left())), env)
        t_plain = time.monotonic() - t0
    if rc != 0:
        log(f"plain schedule failed (rc {rc})")
        return rc
    print(line, flush=True)                          # the safe result: out before any optional step starts
    if chosen:
        return 0
    lines = [json.loads(line)]
    others = []
    # an optional schedule runs what the plain one ran without the CPU baseline: its own duration + slack is the estimate
    estimate = 1.2 * t_plain
    for extra in (["--eval-lane", "off", "--overlap", "on"], ["--eval-lane", "on", "--overlap", "off"]):
        if left() < estimate:
            log(f"schedule {extra} skipped: {left():.0f} s left of the deadline, a run needs ~{estimate:.0f} s")
            others.append({"schedule": " ".join(extra), "value": None, "skipped": f"{left():.0f} s left, ~{estimate:.0f} s needed"})
            continue
        rc2, line2 = _run_ranks(n_gpus, argv + extra + ["--no-cpu-baseline"], min(t_try, left()), env)
        if rc2 == 0:
            lines.append(json.loads(line2))
        else:
            log(f"schedule {extra} failed or timed out (rc {rc2}); keeping what has been measured")
            others.append({"schedule": " ".join(extra), "value": None, "failed_rc": rc2})
    best = max(lines, key=lambda d: d["value"])
    for d in lines:
        if d is not best:
            others.append({"schedule": d["config"].get("schedule"), "value": d["value"], "ms_per_step": d["ms_per_step"]})
    if best.get("cpu_baseline") is None:
        best["cpu_baseline"] = lines[0].get("cpu_baseline")      # timed once, in the first group (same box, same workload)
    best["other_schedules"] = others
    best["launcher"] = {"deadline_s": deadline - t_start, "used_s": round(time.monotonic() - t_start, 1), "plain_run_s": round(t_plain, 1)}
    if best is not lines[0] or others:
        print(json.dumps(best), flush=True)          # last line = best valid line (the plain one again, now with `other_schedules`)
    return 0


# ----------------------------------------------------------------------------------------------- CPU baseline
def host_description():
    """the box the CPU baseline ran on: logical cores (nproc), CPU model, compiler and flags of the timed objects"""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    try:
        gxx = subprocess.run(["g++", "--version"], capture_output=True, text=True, timeout=10).stdout.splitlines()[0]
    except Exception:
        gxx = None
    return {"host_cores": os.cpu_count(), "host_cores_usable": usable, "cpu_model": model, "compiler_here": gxx}


def cpu_baseline(ds_full, hidden, budget_s=30.0):
    """gcn-seq timed on this box's host cores, rank 0 only: the reference's own objects
    (oracle/_ref/libref.so, built in the build container from the reference's sources where they lie;
    kind "reference") when that library travelled with the repo, else the oracle, our single-threaded
    C restatement pinned bit-for-bit to it (kind "port").  Epochs (train + validation) of the full
    workload, one model, timed one by one: the median of at least two when they fit the budget, else the
    1/10-scale graph scaled by 10 (cost is linear in nodes and edges).  The other one of the two is timed
    on the 1/10-scale graph as a cross-check."""
    from cuda_gcn_amd import datagen
    from oracle.pyoracle import Oracle, Ref

    def epochs(factory, ds, n):
        m = factory.model(ds, seed_time=1, hidden_dim=hidden, dropout=0.5)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            m.train_epoch(); m.eval(2)
            ts.append(time.perf_counter() - t0)
        m.close()
        return ts

    port = Oracle()
    ref = Ref() if Ref.available() else None
    # flags: oracle/Makefile builds the reference's sources with the reference's own flags (its Makefile:6) and the
    # restatement with the same optimisation level, no -march, no fast-math
    main_impl, kind, what, flags = (ref, "reference", "the reference's src/seq objects (oracle/_ref/libref.so)", "g++ -O3 -std=c++11 (the reference's Makefile flags), 1 thread") if ref else \
                                   (port, "port", "oracle/gcn_oracle.c", "gcc -O3 -std=gnu11, no -march, no fast-math, 1 thread")
    mini = datagen.make_dataset("reddit-mini")
    t_mini = statistics.median(epochs(main_impl, mini, 2))
    scale = ds_full["num_nodes"] / mini["num_nodes"]
    host = host_description()
    if 2 * t_mini * scale <= budget_s:
        n = max(2, min(3, int(budget_s / (t_mini * scale))))
        ts = epochs(main_impl, ds_full, n)
        t_full = statistics.median(ts)
        out = dict(value=1.0 / t_full, unit="epochs/s", cores=1, kind=kind,
                   sample=f"median of {n} epochs (train+val) of the full workload on one model: {', '.join('%.2f' % t for t in ts)} s wall; {what}, {flags}",
                   epochs_s=[round(t, 3) for t in ts])
    else:
        out = dict(value=1.0 / (t_mini * scale), unit="epochs/s", cores=1, kind=kind,
                   sample=f"median of 2 epochs of reddit-mini (1/10 nodes and edges, same widths) = {t_mini:.2f} s, scaled x{scale:.1f} "
                          f"to the full graph; {what}, {flags}")
    out.update(host)
    out["build"] = flags
    if ref:
        out["port_cross_check"] = dict(mini_epoch_s_reference=round(t_mini, 3), mini_epoch_s_port=round(statistics.median(epochs(port, mini, 2)), 3))
    return out


# ----------------------------------------------------------------------------------------------- HBM-regime leg
def _pmc(files, kernel, avg_launch_ms=None, tol=0.10):
    """PMC summary of exactly `kernel` (full name with template arguments) from the newest committed profile that has it,
    accepted only when it describes the launch that was just timed: the profile records the kernel's median duration under
    the profiler, and a file whose duration differs from the HIP-event average of this run by more than `tol` is refused
    (a kernel change would otherwise leave the roofline computed from stale fabric bytes).  A file taken on another version of
    the kernel's source (`_meta.sources`, cuda_gcn_amd/provenance.py) is still quoted, with `stale: true`.
    Returns (entry, source, why_not)."""
    why = "no committed PMC profile names this kernel"
    for f in files:
        p = os.path.join(ROOT, "profiles", f)
        if not os.path.exists(p):
            continue
        doc = json.load(open(p))
        k = doc.get(kernel)
        if k is None:
            continue
        ref_us = k.get("median_duration_us_under_pmc")
        if avg_launch_ms is not None and ref_us:
            dev = abs(1e3 * avg_launch_ms - ref_us) / ref_us
            if dev > tol:
                why = f"profiles/{f}: {kernel} took {ref_us:.0f} us under the profiler, {1e3 * avg_launch_ms:.0f} us here ({100 * dev:.0f} % apart): refused"
                continue
        from cuda_gcn_amd.provenance import stale_reason
        why_stale = stale_reason(doc.get("_meta"))      # the kernel source the counters were taken on vs this tree's
        k = dict(k, commit=doc.get("_meta", {}).get("commit"), duration_checked=bool(avg_launch_ms is not None and ref_us),
                 stale=why_stale is not None, stale_why=why_stale)
        return k, "profiles/" + f, None
    return None, None, why


def measured_gather_ceiling(dataset, dim, rows, stored_edges, schedule):
    """GB/s of the best gather+sum+store microbenchmark on this dataset's own index stream, task order and XCD slicing
    (tools/gather_peak.hip, swept over loads in flight and resident waves; profiles/r0N_gather_peak.json).  A file is used only
    if it describes THIS graph and schedule: dataset name, row and stored-edge counts, and the row schedule it was taken on.
    -> (GB/s, source, entry, why_not)"""
    why = "no committed gather ceiling for this configuration (tools/gather_peak.py)"
    for f in GATHER_PEAK_FILES:
        p = os.path.join(ROOT, "profiles", f)
        if not os.path.exists(p):
            continue
        doc = json.load(open(p))
        key = f"d={dim}"
        if doc.get("dataset") != dataset or key not in doc.get("ceiling_GBps", {}):
            continue
        if doc.get("rows") != rows or doc.get("stored_edges") != stored_edges:
            why = f"profiles/{f}: taken on {doc.get('rows')} rows / {doc.get('stored_edges')} stored edges, this run has {rows} / {stored_edges}: refused"
            continue
        if doc.get("schedule", "label-major") != schedule:
            why = f"profiles/{f}: taken on the {doc.get('schedule', 'label-major')} schedule, this run uses {schedule}: refused"
            continue
        return doc["ceiling_GBps"][key], "profiles/" + f, doc, None
    return None, None, None, why


def hbm_regime_leg(scale, dim, device, launches=10):
    """GraphSum at the hidden width on an R-MAT graph (BASELINE configs[4] family) whose gathered table is far
    larger than the 256 MiB Infinity Cache, through the same C-ABI entry point the model calls, with the
    `dealt-256` row schedule HipGCN picks for such graphs.  B_gs(d) / HIP-event time against the 8 TB/s HBM peak."""
    import ctypes as C
    import numpy as np
    from cuda_gcn_amd import datagen
    from cuda_gcn_amd.ops import Device, _ck
    t0 = time.perf_counter()
    gp, gi = datagen.rmat_graph(scale)
    N, nnz = gp.size - 1, int(gi.size)
    t_gen = time.perf_counter() - t0
    dev = Device(device)
    lib = dev.lib
    t0 = time.perf_counter()
    g = dev.graph(gp, gi)
    _ck(lib, lib.gcnhip_graph_set_schedule(dev.ctx, g.h, 2, None, 256), "set_schedule")
    t_prep = time.perf_counter() - t0
    ld = (dim + 15) // 16 * 16 if dim > 32 else (dim + 3) // 4 * 4
    x = dev.buf(np.random.default_rng(0).standard_normal((N, ld), dtype=np.float32))
    o = dev.buf((N, ld))
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.gcnhip_event_create(C.byref(e0)); lib.gcnhip_event_create(C.byref(e1))

    from cuda_gcn_amd.ops import GsOpts
    gso = GsOpts()
    gso.scaling = 1                     # the factored operator, as HipGCN launches the aggregation by default (no coefficient stream)

    def run():
        _ck(lib, lib.gcnhip_graphsum_ex(dev.ctx, g.h, C.byref(gso), x.ptr, ld, o.ptr, ld, dim), "gcnhip_graphsum_ex")
    for _ in range(2):
        run()
    dev.sync()
    lib.gcnhip_event_record(dev.ctx, e0)
    for _ in range(launches):
        run()
    lib.gcnhip_event_record(dev.ctx, e1)
    ms = C.c_float()
    _ck(lib, lib.gcnhip_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
    lib.gcnhip_event_destroy(e0); lib.gcnhip_event_destroy(e1)
    avg_s = ms.value * 1e-3 / launches
    x.free(); o.free(); g.free(); dev.close()
    bytes_per_launch = b_gs(N, nnz, dim)
    achieved = bytes_per_launch / avg_s / 1e9
    k, src, why_not = _pmc(PMC_RMAT_FILES, GS_KERNEL_HBM if N * ld * 4 > 256 * 2**20 else GS_KERNEL, 1e3 * avg_s, tol=0.15)
    traffic = k.get("traffic_bytes_per_launch") if k else None
    return {"workload": f"GraphSum d={dim} (factored operator, gcnhip_graphsum_ex) on rmat-{scale} (N={N}, {nnz} stored edges, max degree {int(np.diff(gp).max())}); "
                        f"gathered table {N * ld * 4 / 2**20:.0f} MiB >> 256 MiB Infinity Cache; schedule dealt-256",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "bytes_per_launch": bytes_per_launch, "avg_launch_ms": 1e3 * avg_s, "launches": launches,
            "traffic": traffic, "traffic_source": src if k else why_not,
            "traffic_frac_of_peak": None if traffic is None else traffic / avg_s / 1e9 / HBM_PEAK_GBPS,
            "setup_s": {"generate": round(t_gen, 2), "graph_create_and_schedule": round(t_prep, 2)}}


def dense_leg(ds, hidden, device, iters=10):
    """The three dense first-layer products of an epoch (SparseMatmul on a dense X: module.cpp:47-77) on this workload's own X
    through the C-ABI entry points the model calls, each timed with HIP events on the stream it runs on: training forward
    (input dropout fused), evaluation forward, weight gradient.  TFLOP/s of 2*nnzX*h against the MFMA peak of the pipe the
    kernel computes on; the busy share of the MFMA pipes comes from the committed counter passes (tools/pmc_gemm.sh)."""
    import ctypes as C
    import numpy as np
    from cuda_gcn_amd.ops import Device, _ck
    N, F = ds["num_nodes"], ds["input_dim"]
    dev = Device(device)
    lib = dev.lib
    f = dev.feat(ds["f_indptr"], None, ds["f_val"], F)
    rng = np.random.default_rng(0)
    w1 = dev.buf((rng.standard_normal((F, hidden)) * 0.09).astype(np.float32))
    h0 = dev.buf((N, hidden))
    g0 = dev.buf(rng.standard_normal((N, hidden), dtype=np.float32))
    dw = dev.buf((F, hidden))
    ep = dev.buf(np.zeros(1, np.uint32))
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.gcnhip_event_create(C.byref(e0)); lib.gcnhip_event_create(C.byref(e1))

    def timeit(fn):
        for _ in range(2):
            fn()
        dev.sync()
        lib.gcnhip_event_record(dev.ctx, e0)
        for _ in range(iters):
            fn()
        lib.gcnhip_event_record(dev.ctx, e1)
        ms = C.c_float()
        _ck(lib, lib.gcnhip_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
        return ms.value / iters
    v = C.c_int(0)
    method = "bf16x3" if (lib.gcnhip_ctx_get_option(dev.ctx, b"gemm_bf16x3", C.byref(v)) == 0 and v.value >= 1) else "f32"
    peak = BF16_MFMA_PEAK_TF if method == "bf16x3" else F32_MFMA_PEAK_TF
    flop = 2.0 * N * F * hidden
    # MFMA work per launch: the exact-f32 kernels issue one f32 MFMA flop per algorithmic flop; the bf16x3 kernels issue 6 bf16
    # products per f32 product (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi)
    mfma_flop = flop * (6 if method == "bf16x3" else 1)
    legs = {}
    for key, fn in (("train_forward_dropout", lambda: _ck(lib, lib.gcnhip_spmm_fwd(dev.ctx, f.h, f.values_ptr, w1.ptr, hidden, h0.ptr, hidden, hidden, 0.5, 1, ep.ptr, 0, None), "fwd")),
                    ("eval_forward", lambda: _ck(lib, lib.gcnhip_spmm_fwd(dev.ctx, f.h, f.values_ptr, w1.ptr, hidden, h0.ptr, hidden, hidden, 0.0, 0, ep.ptr, 0, None), "fwd0")),
                    ("weight_gradient_dropout", lambda: _ck(lib, lib.gcnhip_spmm_bwd(dev.ctx, f.h, f.values_ptr, g0.ptr, hidden, dw.ptr, hidden, hidden, 0.5, 1, ep.ptr, 0, None), "bwd"))):
        ms = timeit(fn)
        legs[key] = {"ms": ms, "algorithmic_TFLOPs": flop / ms / 1e9, "frac_of_f32_mfma_peak": flop / ms / 1e9 / F32_MFMA_PEAK_TF,
                     "mfma_TFLOPs_issued": mfma_flop / ms / 1e9, "frac_of_pipe_peak": mfma_flop / ms / 1e9 / peak}
    lib.gcnhip_event_destroy(e0); lib.gcnhip_event_destroy(e1)
    for b in (w1, h0, g0, dw, ep):
        b.free()
    f.free(); dev.close()
    pmc = None
    for fpmc in GEMM_PMC_FILES:
        pth = os.path.join(ROOT, "profiles", fpmc)
        if os.path.exists(pth):
            from cuda_gcn_amd.provenance import stale_reason
            doc = json.load(open(pth))
            why = stale_reason(doc.get("_meta"))
            pmc = {"source": "profiles/" + fpmc, "stale": why is not None, "stale_why": why, "kernels": doc}
            break
    return {"bound": "mfma", "method": method, "pipe": "bf16 MFMA (three-plane split of f32 operands, f32 accumulate)" if method == "bf16x3" else "f32 MFMA (exact)",
            "peak": peak, "peak_f32_mfma": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "flop_per_launch": flop,
            "workload": f"X [{N} x {F}] (this run's feature matrix) x W1 [{F} x {hidden}]; gcnhip_spmm_fwd / gcnhip_spmm_bwd, HIP events, {iters} launches each, back to back",
            "kernels": legs,
            "frac": min(v["frac_of_pipe_peak"] for v in legs.values()),
            "mfma_busy_pmc": pmc}


def first_layer_method(ds, hidden):
    """how X.W1 and its weight gradient are summed in this run; the switches are HipGCNOptions::gemm
    (HIPGCN_GEMM=f32|bf16x3) and the context option gemm_bf16x3 (GCNHIP_GEMM_BF16X3=0|1|2)"""
    dense = ds["f_indptr"].size - 1 == ds["num_nodes"] and ds["f_val"].size == ds["num_nodes"] * ds["input_dim"]
    if not dense:
        return "sparse X: CSR / CSC row kernels, f32 FMA"
    off = os.environ.get("HIPGCN_GEMM", "") == "f32" or os.environ.get("GCNHIP_GEMM_BF16X3", "") == "0"
    if hidden == 128 and not off:
        return ("bf16x3: f32 operands split exactly into three bf16 planes, six plane products on the bf16 MFMA pipe, f32 accumulate "
                "(error within the f32 summation bound; GCNHIP_GEMM_BF16X3=0 selects the f32-MFMA kernels)")
    return "f32 MFMA"


def class_layer_method(ds, hidden):
    """how H1.W2, dH1 and dW2 are summed in this run (csrc/class_bf16x3.h)"""
    off = os.environ.get("HIPGCN_GEMM", "") == "f32" or os.environ.get("GCNHIP_GEMM_BF16X3", "") == "0"
    if hidden == 128 and ds["output_dim"] <= 64 and ds["num_nodes"] >= 2048 and not off:
        return ("bf16x3 (as the first layer): H1.W2 in one launch, dH1 + dW2 fused in one launch + a slab sum; "
                "loss, accuracy and gradient rows in the epilogue of the class-width aggregation")
    return "f32 MFMA row-stream kernels; loss in the epilogue of the class-width aggregation"


# ----------------------------------------------------------------------------------------------- one rank
def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    if args.profile_run:
        # what `rocprofv3 --kernel-trace --stats -- python3 bench.py --profile-run` should see: the timed epochs and nothing that
        # shares their kernel names (no schedule-tuning launches, no second stream beside the timed launches, no extra models)
        args.eval_lane, args.no_extras, args.no_cli, args.no_cpu_baseline = "off", True, True, True
        os.environ.setdefault("HIPGCN_SCHEDULE", args.pin_schedule)
    if args.no_extras:
        args.no_cli = True
    hang = os.environ.get("GCN_BENCH_TEST_HANG")      # tests only: "overlap" / "lane" make the children of that optional schedule sleep
    if hang and "WORLD_SIZE" in os.environ and ((hang == "overlap" and args.overlap == "on") or (hang == "lane" and args.eval_lane == "on")):
        time.sleep(100000)
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import torch                       # device plumbing + rendezvous only; imported BEFORE the native libs so
    import torch.distributed as dist   # that one HIP runtime / one RCCL is loaded in the process
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    device = int(os.environ.get("GCN_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)    # control plane; data plane is RCCL in libgcnhost

    from cuda_gcn_amd import datagen
    from cuda_gcn_amd.model import (HipGCNModel, EVAL_LANE, NO_EVAL_LANE, BF16_TABLES, NO_ROW_GROUPS, NO_LABEL_HINT, NO_AGG_FIRST_EVAL,
                                    ALL_ROWS, OVERLAP_EXCHANGE, nccl_unique_id)

    if args.selftest:
        # every RCCL primitive the row-partitioned epoch uses, against real peers, before a model depends on them
        from cuda_gcn_amd import _lib
        if world < 2:
            sys.exit("bench.py --selftest needs more than one rank")
        box = [nccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        lib = _lib.gcnhost()
        rc = lib.gcnhost_rccl_selftest_world(device, rank, world, box[0])
        if rc != 0:
            sys.exit(f"rank {rank}: gcnhost_rccl_selftest_world failed: {rc}: {lib.gcnhost_last_error().decode()}")
        dist.barrier()
        if rank == 0:
            print(json.dumps({"selftest": "ok", "n_gpus": world}), flush=True)
        dist.destroy_process_group()
        return

    def barrier():
        if world > 1:
            dist.barrier()

    t0 = time.perf_counter()
    ds = datagen.make_dataset(args.dataset)          # same seed on every rank -> identical graph
    t_data = time.perf_counter() - t0
    if rank == 0:
        log(f"dataset {args.dataset} ready in {t_data:.1f} s")
    nccl_id, host_ag, host_ar = None, None, None
    if world > 1 and os.environ.get("GCN_BENCH_TRANSPORT") == "host":
        # rehearsal without RCCL (e.g. several ranks sharing one GPU): collectives staged through the host
        # over gloo.  Never a performance number.
        def host_ag(user, ptr, block):
            t = torch.from_numpy(np.ctypeslib.as_array(ptr, (block * world,)))
            dist.all_gather(list(t.view(world, block).unbind(0)), t[rank * block:(rank + 1) * block].clone())

        def host_ar(user, ptr, n):
            dist.all_reduce(torch.from_numpy(np.ctypeslib.as_array(ptr, (n,))))
    elif world > 1:
        box = [nccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        nccl_id = box[0]
    # auto: one GPU — the validation forward runs on a second stream beside the next epoch's hidden-width aggregation
    # (no communication involved; 277 -> 289 epochs/s).  Several GPUs — the plain one-stream schedule: the overlapped one
    # (second stream + split communicator) is opt-in there until it has been measured on a multi-GPU node; the
    # self-launching parent above tries both.
    lane_on = args.eval_lane == "on" or (args.eval_lane == "auto" and world == 1)
    lane_flag = EVAL_LANE if lane_on else NO_EVAL_LANE
    overlap_on = args.overlap == "on" and world > 1
    base_flags = lane_flag | (BF16_TABLES if args.bf16_tables else 0) | (OVERLAP_EXCHANGE if overlap_on else 0)
    n_epochs_total = args.warmup + args.steps * (2 + args.bursts) + 64 + args.sustain

    def build(flags, data=None):
        t0 = time.perf_counter()
        m = HipGCNModel(data if data is not None else ds, seed=1, device=device, flags=flags, rank=rank, world=world, nccl_id=nccl_id,
                        host_allgather=host_ag, host_allreduce=host_ar,
                        hidden_dim=args.hidden, dropout=0.5, epochs=n_epochs_total)
        return m, time.perf_counter() - t0     # host preprocessing (edge order, schedules) + every H2D copy + schedule timing

    def timed_region(m, k):
        """EXACTLY k epochs, bracketed by barrier + synchronize on both sides; max over ranks; returns (seconds, trace)"""
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr = m.run_epochs(k)                         # k epochs enqueued back to back, one sync at the end
        torch.cuda.synchronize(); barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        return dt, tr

    model, t_build = build(base_flags | (NO_ROW_GROUPS if args.no_row_groups else 0))
    info = model.info()
    exchange = model.exchange() if world > 1 else None
    transport = model.transport() if world > 1 else None
    model.run_epochs(args.warmup, want_trace=False)
    dt, trace = timed_region(model, args.steps)      # <- the headline: default product path, per-op timers off
    burst_eps = [args.steps / dt]
    for _ in range(args.bursts):
        d, _tr = timed_region(model, args.steps)
        burst_eps.append(args.steps / d)
    if rank == 0:
        log(f"timed: {args.steps / dt:.2f} epochs/s; bursts {['%.1f' % b for b in burst_eps]}")
    # one long region on the same model (one GPU, extras on): seconds of back-to-back epochs, so that a sustained rate stands
    # beside the K-step figure (clocks and temperatures settle; an external busy-sampler sees the card at work)
    sustained = None
    if world == 1 and not args.no_extras and args.sustain > 0:
        # socket power and shader clock of the visible card while the region runs (rocm-smi from a side thread, ~0.2 s a
        # sample, read-only): DESIGN.md §4.4 — every hot kernel of this epoch alone draws 1.25-1.36 kW of the 1.4 kW cap
        import threading
        smi, stop = [], threading.Event()

        def sample():
            while not stop.is_set():
                try:
                    o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                    d = json.loads(o).get("card0", {})
                    smi.append((float(d["Current Socket Graphics Package Power (W)"]), float(d["sclk clock speed:"].strip("()Mhz"))))
                except Exception:
                    stop.wait(0.5)
        th = threading.Thread(target=sample, daemon=True)
        th.start()
        d_s, _tr = timed_region(model, args.sustain)
        stop.set()
        th.join(timeout=10)
        sustained = {"epochs": args.sustain, "seconds": round(d_s, 3), "epochs_per_s": args.sustain / d_s}
        tail = smi[len(smi) // 3:]                              # the first third: ramp
        if tail:
            sustained.update({"socket_power_W": round(sum(x[0] for x in tail) / len(tail), 1), "sclk_MHz": round(sum(x[1] for x in tail) / len(tail), 1),
                              "smi_samples": len(tail), "smi": "rocm-smi --showpower --showclocks, sampled during the region"})
        log(f"sustained: {args.sustain / d_s:.2f} epochs/s over {args.sustain} epochs ({d_s:.1f} s)" +
            (f"; socket {sustained['socket_power_W']} W, sclk {sustained['sclk_MHz']} MHz" if tail else ""))

    # ---- separate pass, same process and model: per-op HIP-event timers on (the epoch then runs eagerly, one
    # event pair per op on the stream the op runs on; on one GPU everything runs on ONE stream in this pass, so a launch
    # is timed alone and not beside the validation lane's kernels).  The dominant kernel's launches are timed here.
    n_tm = min(args.steps, 20)
    model.set_timers(True)
    model.run_epochs(2, want_trace=False)
    model.timers_reset()
    dt_tm, _ = timed_region(model, n_tm)
    s_wide, n_wide = model.timer("graphsum_wide")
    if n_wide == 0:                                   # hidden <= 64: the wide timer never fires; use all GraphSum launches
        s_f, n_f = model.timer("graphsum_fw")
        s_b, n_b = model.timer("graphsum_bw")
        s_wide, n_wide = s_f + s_b, n_f + n_b
    breakdown = {}
    comm_calls = 0
    for name in ("spmatmul_fw", "spmatmul_bw", "graphsum_fw", "graphsum_bw", "matmul_fw", "matmul_bw", "loss_fw", "adam", "comm"):
        s, n = model.timer(name)
        if n:
            breakdown[name] = round(1e3 * s / n_tm, 4)      # ms per epoch
        if name == "comm":
            comm_calls = n                                  # exchanges + all-reduces timed with device events on the stream they run on
    model.set_timers(False)
    # train-only epochs (no validation forward; one host read-back per epoch), outside the timed region
    n_tr = min(args.steps, 20)
    barrier(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_tr):
        model.train_epoch()
    torch.cuda.synchronize(); barrier()
    train_only_ms = 1e3 * (time.perf_counter() - t1) / n_tr
    schedule = model.schedule()
    slice_floats = model.slice_floats()
    model.close()

    out = None
    if rank == 0:
        ib = 2 if args.bf16_tables else 4
        if args.hidden > 64:
            d_eff = float(args.hidden)
            bytes_per_launch = b_gs(info["local_rows"], info["local_edges"], args.hidden, ib)
        else:                                         # average over the hidden- and class-width launches (3 + 3 per epoch)
            d_eff = (args.hidden + ds["output_dim"]) / 2
            bytes_per_launch = (b_gs(info["local_rows"], info["local_edges"], args.hidden, ib) +
                                b_gs(info["local_rows"], info["local_edges"], ds["output_dim"], ib)) / 2
        table_mb = ds["num_nodes"] * args.hidden * 4 / 1e6
        avg_s = s_wide / max(n_wide, 1)
        algorithmic = bytes_per_launch / avg_s / 1e9
        gathered = ib * info["local_edges"] * d_eff / avg_s / 1e9
        kernel = ("graphsum_bf16_kernel<16> (bf16 table; timer includes the f32->bf16 conversion)" if args.bf16_tables else
                  f"{GS_KERNEL if ds['num_nodes'] * args.hidden * 4 <= 256 * 2**20 else GS_KERNEL_HBM}, 64-float column slices per XCD group (GraphSum d={args.hidden})" if args.hidden > 64 else
                  f"graphsum_vec_kernel (GraphSum, d={args.hidden} and d={ds['output_dim']} launches averaged)")
        if overlap_on and args.hidden > 64:
            kernel += "; --overlap on: a launch pair per aggregation (own-column edges, then the rest), timed together incl. any exposed wait for the exchange"
        # fabric traffic and L2 hit rate per launch of the same kernel from the committed rocprofv3 PMC passes
        # (FETCH_SIZE, WRITE_SIZE and the TCC hit counters need separate runs, so they cannot be collected live here);
        # a profile is accepted only for exactly this kernel at (within 10 %) the duration timed in this run
        pmc, pmc_src, pmc_why = (None, None, "no PMC profile is committed for this configuration")
        if args.dataset == "reddit-syn" and args.hidden == 128 and not args.bf16_tables and not args.no_row_groups and world == 1:
            pmc, pmc_src, pmc_why = _pmc(PMC_FILES, GS_KERNEL, 1e3 * avg_s)
        elif args.dataset.startswith("rmat-22") and args.hidden == 128 and not args.bf16_tables and world == 1:
            pmc, pmc_src, pmc_why = _pmc(PMC_RMAT22_FILES, GS_KERNEL_HBM, 1e3 * avg_s, tol=0.15)      # BASELINE configs[4]'s own launch
        traffic = pmc.get("traffic_bytes_per_launch") if pmc else None
        cache_resident = table_mb * 1e6 <= 256 * 2**20
        ceil_gbps, ceil_src, ceil_doc, ceil_why = (None, None, None, "not a single-GPU f32 cache-resident hidden-width run")
        if cache_resident and world == 1 and not args.bf16_tables and not args.no_row_groups and args.hidden > 64:
            ceil_gbps, ceil_src, ceil_doc, ceil_why = measured_gather_ceiling(args.dataset, args.hidden, ds["num_nodes"], int(ds["g_indices"].size), schedule)
        guide = None
        if cache_resident and pmc and "traffic_bytes_per_launch" in pmc:
            # The ceiling that follows from MI355X_MICROARCH.md alone, for a table that sits in the Infinity Cache (B_gs / t exceeds the
            # HBM peak, so HBM is not the bound): every gathered byte crosses an XCD's L2 (34.5 TB/s aggregate, "L2") and the bytes that
            # miss it (PMC: 2*FETCH_SIZE + WRITE_SIZE per launch) come from the Infinity Cache, whose best row-gather rate on record
            # there is 8.6 TB/s ("Indexed rows").  A launch cannot be shorter than the longer of the two transfers.
            gathered_bytes = ib * info["local_edges"] * d_eff
            fabric = pmc["traffic_bytes_per_launch"]
            t_l2, t_fabric = gathered_bytes / (L2_AGG_GBPS * 1e9), fabric / (MALL_GATHER_GBPS * 1e9)
            guide = {"peak": gathered_bytes / max(t_l2, t_fabric) / 1e9, "floor_ms_l2": 1e3 * t_l2, "floor_ms_infinity_cache": 1e3 * t_fabric}
        if guide:
            roof = {"bound": "cache-gather", "kernel": kernel, "achieved": gathered, "peak": guide["peak"], "unit": "GB/s", "frac": gathered / guide["peak"],
                    "traffic": traffic, "l2_hit_rate": pmc.get("l2_hit_rate"),
                    "floor_ms_l2": guide["floor_ms_l2"], "floor_ms_infinity_cache": guide["floor_ms_infinity_cache"],
                    "peak_source": "MI355X_MICROARCH.md: L2 34.5 TB/s aggregate; Infinity-Cache row gather 8.6 TB/s ('Indexed rows'); fabric bytes from " + str(pmc_src),
                    "what": "achieved = gathered neighbour-row bytes (4*nnz*d) / HIP-event launch time; peak = the same bytes / max(bytes / 34.5 TB/s "
                            "L2 aggregate, PMC fabric bytes / 8.6 TB/s) - the two-resource floor that follows from the guide's figures for a table "
                            "resident in the Infinity Cache (8.6 TB/s is the best Infinity-Cache row-gather rate the guide records)"}
            # beside it, NOT as `frac`: the best rate of a kernel that only gathers, sums and stores on this dataset's own index stream,
            # task order and XCD slicing (tools/gather_peak.hip) — the kernel against a stripped copy of itself
            roof["frac_vs_measured_gather"] = gathered / ceil_gbps if ceil_gbps else None
            roof["peak_measured_gather"] = ceil_gbps
            roof["peak_measured_gather_source"] = ceil_src if ceil_gbps else ceil_why
            if ceil_gbps:
                roof["peak_measured_gather_commit"] = ceil_doc.get("_meta", {}).get("commit")
                roof["peak_without_store"] = ceil_doc.get("ceiling_without_store_GBps", {}).get(f"d={args.hidden}")
                roof["peak_uniform_random_rows"] = ceil_doc.get("ceiling_uniform_random_GBps", {}).get(f"d={args.hidden}")
                roof["measured_gather_exceeded"] = bool(gathered > ceil_gbps)     # a stale or mismatched ceiling shows here
                # the launch's L2-MISS traffic (PMC fabric bytes / launch time) against the rate the same microbenchmark measures for
                # uniformly random rows of the same table (~3 % L2 hits: practically all of it Infinity-Cache traffic) — the resource
                # round 6 found the launch waits for (docs/NOTEBOOK_r6.md 1)
                rnd = roof["peak_uniform_random_rows"]
                if rnd and traffic:
                    roof["fabric_rate_GBps"] = traffic / avg_s / 1e9
                    roof["frac_vs_measured_fabric_rate"] = traffic / avg_s / 1e9 / rnd
        else:
            roof = {"bound": "hbm", "kernel": kernel, "achieved": algorithmic, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": algorithmic / HBM_PEAK_GBPS, "traffic": traffic,
                    "what": "B_gs(d) / HIP-event launch time against the HBM peak" +
                            ("; the table is Infinity-Cache resident, so this may exceed 1 (cache-served); no PMC profile accepted for the "
                             "two-resource cache bound: " + str(pmc_why) if cache_resident else "")}
        # stored numbers quoted beside this run's live timing (fabric bytes, the measured gather ceiling): were they taken on THIS
        # tree's kernel source?  (`_meta.sources` of the file against the sha256 of the sources here, cuda_gcn_amd/provenance.py)
        from cuda_gcn_amd.provenance import stale_reason
        pmc_stale = pmc.get("stale_why") if pmc else None
        ceil_stale = stale_reason(ceil_doc.get("_meta")) if ceil_doc else None
        roof.update({"stale": bool(pmc_stale or ceil_stale), "traffic_stale": bool(pmc_stale), "traffic_stale_why": pmc_stale,
                     "peak_measured_gather_stale": bool(ceil_stale), "peak_measured_gather_stale_why": ceil_stale,
                     "traffic_source": pmc_src if pmc else pmc_why, "traffic_commit": pmc.get("commit") if pmc else None,
                     "traffic_profile_duration_us": pmc.get("median_duration_us_under_pmc") if pmc else None,
                     "bytes_per_launch": bytes_per_launch, "avg_launch_ms": 1e3 * avg_s, "launches": n_wide,
                     "table_MB": round(table_mb, 1),
                     "breakdown_schedule": "one-stream (per-op timers on: every launch timed alone; the headline runs the validation forward "
                                           "on a second stream, so the breakdown does not add up to ms_per_step)" if (world == 1 and lane_on) else "as timed",
                     # SURVEY 8(d) contract figure B_gs(d)/t against the HBM peak; > 1 means cache-served
                     "hbm_algorithmic_achieved": algorithmic, "hbm_algorithmic_frac": algorithmic / HBM_PEAK_GBPS,
                     # PMC fabric bytes (2*FETCH_SIZE + WRITE_SIZE) per launch / launch time
                     "hbm_traffic_achieved": None if traffic is None else traffic / avg_s / 1e9,
                     "hbm_traffic_frac": None if traffic is None else traffic / avg_s / 1e9 / HBM_PEAK_GBPS})
        if world > 1:
            # the exchange's share: device time of the collectives per epoch (HIP events on the stream they run on) and the
            # bytes this rank receives per epoch; with --overlap on the exchanges run beside the aggregations, so their
            # device time is not all exposed
            comm_ms = breakdown.get("comm", 0.0)
            roof.update({"comm_ms_per_epoch": comm_ms, "comm_share_of_timers_pass": comm_ms / max(1e3 * dt_tm / n_tm, 1e-9),
                         "collectives_per_epoch": comm_calls / max(n_tm, 1),
                         "comm_us_per_collective": 1e3 * comm_ms / max(comm_calls / max(n_tm, 1), 1e-9),
                         "per_rank": "rank 0's launches on its row block"})
        n_lab = int((ds["split"] == 1).sum())
        out = {
            "metric": "epochs_per_sec", "value": args.steps / dt, "unit": "epochs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32+bf16-tables" if args.bf16_tables else "f32", "data": "synthetic",
            "config": {"workload": f"{args.dataset} full-batch 2-layer GCN, N={ds['num_nodes']}, "
                                   f"{(ds['g_indices'].size - ds['num_nodes']) // 2} undirected edges, "
                                   f"{ds['input_dim']}->{args.hidden}->{ds['output_dim']}, dropout 0.5, Adam; "
                                   "step = train_epoch + eval(val)" + structure_note(ds),
                       "graph_structure": graph_structure(ds),
                       "parallelism": f"row-partition x{world}" if world > 1 else "single GPU",
                       "train_nodes": n_lab, "aggregation_schedule": schedule, "aggregation_slice_floats": slice_floats,
                       "first_layer_products": first_layer_method(ds, args.hidden),
                       "class_layer_products": class_layer_method(ds, args.hidden),
                       "eval_lane": "on" if lane_on else "off",
                       "overlap_exchange": "on" if overlap_on else "off",
                       "schedule": ("plain one-stream" if not (lane_on or overlap_on) else
                                    " + ".join(x for x in ["exchange overlap" if overlap_on else "", "validation lane" if lane_on else ""] if x)),
                       "exchange": exchange,       # rank 0's view: all-gather of row blocks or halo lists, rows moved per exchange
                       "transport": transport[0] if transport else None,
                       "rccl_ranks": transport[1] if (transport and transport[0] == "rccl") else None,     # ncclCommCount of the model's communicator
                       "eval_forward": "reference order A^.(X.W1)" if os.environ.get("HIPGCN_NO_AGG_FIRST_EVAL") else
                                       ("aggregate-first ReLU((A^.X).W1), A^.X built once at load (dense X)" +
                                        ("" if os.environ.get("HIPGCN_NO_EVAL_FUSION") or args.hidden != 128 else "; .W2 in the same launch, H1 not stored")),
                       "logit_rows": "all" if os.environ.get("HIPGCN_ALL_ROWS") else "rows of the scored split only"},
            "bursts": {"epochs_per_s": [round(b, 2) for b in burst_eps], "median": statistics.median(burst_eps), "min": min(burst_eps),
                       "max": max(burst_eps), "steps_each": args.steps},
            "sustained": sustained,
            "roofline": roof,
            "breakdown_ms_per_epoch": breakdown, "timers_pass_ms_per_epoch": round(1e3 * dt_tm / n_tm, 4),
            "train_only_ms_per_epoch": round(train_only_ms, 4),
            "final": {"train_loss": float(trace[-1, 0]), "train_acc": float(trace[-1, 1]),
                      "val_loss": float(trace[-1, 2]), "val_acc": float(trace[-1, 3])},
            "setup_s": {"dataset": round(t_data, 2), "model_build_incl_h2d": round(t_build, 2)},
        }

    extras = world == 1 and not args.no_extras
    if extras and not args.no_row_groups and schedule == "label-major":
        # the same headline with the labels withheld from the schedule: row groups found in the graph (host/cluster.h)
        m2, _ = build(base_flags | NO_LABEL_HINT)
        m2.run_epochs(args.warmup, want_trace=False)
        d2, _tr = timed_region(m2, args.steps)
        out["value_no_label_hint"] = args.steps / d2
        out["config"]["aggregation_schedule_no_label_hint"] = m2.schedule()
        m2.close()
        log(f"no label hint: {args.steps / d2:.2f} epochs/s")
    if extras and not args.no_row_groups and schedule != "degree":
        # ... and with no row groups of any kind (plain descending degree)
        m2, _ = build(base_flags | NO_ROW_GROUPS)
        m2.run_epochs(args.warmup, want_trace=False)
        d2, _tr = timed_region(m2, args.steps)
        out["value_no_row_groups"] = args.steps / d2
        m2.close()
        log(f"no row groups: {args.steps / d2:.2f} epochs/s")
    if extras and not args.no_row_groups:
        # ... and with every forward in the reference's own operation order and every row of the logits computed
        # (no A^.X built once for the evaluation forwards, no skipping of rows the loss never reads)
        m3, _ = build(base_flags | NO_AGG_FIRST_EVAL | ALL_ROWS)
        m3.run_epochs(args.warmup, want_trace=False)
        d3, _tr = timed_region(m3, args.steps)
        out["value_reference_op_order_all_rows"] = args.steps / d3
        m3.close()
        log(f"reference op order, all rows: {args.steps / d3:.2f} epochs/s")
    dense_x = ds["f_indices"] is not None and ds["f_indptr"][1] == ds["input_dim"]
    if extras and args.hidden == 128 and dense_x and not args.no_row_groups and "bf16x3" in first_layer_method(ds, args.hidden):
        # ... and with every dense product on the exact-f32 MFMA kernels (HIPGCN_GEMM=f32: the f32 pipe, 1/16 of the bf16 rate;
        # sequential f32 multiply-add like module.cpp:11-77) — the same model, schedule and timed region as the headline
        os.environ["HIPGCN_GEMM"] = "f32"
        try:
            m4, _ = build(base_flags)
            m4.run_epochs(args.warmup, want_trace=False)
            d4, _tr = timed_region(m4, args.steps)
            out["value_f32_products"] = args.steps / d4
            m4.close()
            log(f"f32-MFMA products (HIPGCN_GEMM=f32): {args.steps / d4:.2f} epochs/s")
        except Exception as e:
            out["value_f32_products"] = None
            out["value_f32_products_error"] = repr(e)
        finally:
            del os.environ["HIPGCN_GEMM"]
    if extras and not args.no_structure_legs and args.dataset == "reddit-syn" and not args.no_row_groups:
        # The headline graph has planted communities (the labels).  The same shape with other structure, each on the product's default
        # path with the row schedule it times fastest at load: no planted structure at all (SURVEY 8(d)'s literal Chung-Lu graph),
        # half the mixing, and the headline's mixing with Zipf class sizes (largest class 3.3 x one XCD's L2 of column slices).
        legs = {}
        for key, name in STRUCTURE_LEGS:
            try:
                t0 = time.perf_counter()
                d2 = datagen.make_dataset(name)
                t_gen = time.perf_counter() - t0
                m2, t_b = build(base_flags, d2)
                m2.run_epochs(args.warmup, want_trace=False)
                dd, tr2 = timed_region(m2, args.steps)
                n2 = min(args.steps, 10)
                m2.set_timers(True)
                m2.run_epochs(2, want_trace=False)
                m2.timers_reset()
                timed_region(m2, n2)
                sw, nw = m2.timer("graphsum_wide")
                m2.set_timers(False)
                inf2 = m2.info()
                legs[name] = {"epochs_per_s": args.steps / dd, "ms_per_epoch": 1e3 * dd / args.steps, "aggregation_schedule": m2.schedule(),
                              "aggregation_slice_floats": m2.slice_floats(),
                              "hidden_width_launch_ms": 1e3 * sw / max(nw, 1),
                              "hidden_width_gathered_GBps": 4.0 * inf2["local_edges"] * args.hidden / (sw / max(nw, 1)) / 1e9,
                              "graph_structure": graph_structure(d2), "final_val_acc": float(tr2[-1, 3]),
                              "setup_s": {"dataset": round(t_gen, 2), "model_build_incl_h2d": round(t_b, 2)}}
                out[key] = args.steps / dd
                m2.close()
                del d2
                log(f"{name}: {args.steps / dd:.2f} epochs/s, schedule {legs[name]['aggregation_schedule']} / {legs[name]['aggregation_slice_floats']}-float slices, hidden-width launch {legs[name]['hidden_width_launch_ms']:.3f} ms")
            except Exception as e:      # an extra: report its failure, keep the headline
                legs[name] = {"error": repr(e)}
                out[key] = None
        out["structure_legs"] = legs
    if extras and args.hidden > 64 and dense_x:
        try:
            rd = dense_leg(ds, args.hidden, device)
            # The same products inside the epoch, from the per-op HIP-event timers of this run (forward: the training and the
            # validation product together, with their plane-packing / keep-bit launches; backward: the product and its slab sum).
            # The stand-alone leg above repeats one MFMA-bound launch back to back and runs into the socket power cap (lower
            # shader clock); in the epoch the products alternate with memory-bound kernels and run at the higher clock.
            bd = out.get("breakdown_ms_per_epoch") or {}
            if bd.get("spmatmul_fw") and bd.get("spmatmul_bw"):
                mult = 6 if rd["method"] == "bf16x3" else 1
                fw_tf, bw_tf = 2 * rd["flop_per_launch"] / bd["spmatmul_fw"] / 1e9, rd["flop_per_launch"] / bd["spmatmul_bw"] / 1e9
                rd["in_epoch"] = {"forward_pair_ms": bd["spmatmul_fw"], "forward_algorithmic_TFLOPs": fw_tf, "forward_frac_of_pipe_peak": mult * fw_tf / rd["peak"],
                                  "weight_gradient_ms": bd["spmatmul_bw"], "weight_gradient_algorithmic_TFLOPs": bw_tf,
                                  "weight_gradient_frac_of_pipe_peak": mult * bw_tf / rd["peak"],
                                  "source": "breakdown_ms_per_epoch of this run (one-stream timers pass; every launch of the op included)"}
                rd["frac_in_epoch"] = min(rd["in_epoch"]["forward_frac_of_pipe_peak"], rd["in_epoch"]["weight_gradient_frac_of_pipe_peak"])
                # `frac` = the products as the epoch runs them; the back-to-back legs (one MFMA-bound launch repeated: the socket sits
                # at its power cap and the shader clock drops) under their own name
                rd["frac_power_capped"] = rd["frac"]
                rd["frac"] = rd["frac_in_epoch"]
                rd["frac_source"] = "in_epoch (per-op HIP-event timers of this run); frac_power_capped = the slowest back-to-back leg"
            # both floors of one product: the MFMA pipe it issues on, and its compulsory bytes (X once, W1, the output) at the HBM peak
            n_, f_ = ds["num_nodes"], ds["input_dim"]
            bytes_ = 4.0 * (n_ * f_ + f_ * args.hidden + n_ * args.hidden)
            rd["floors_ms"] = {"mfma_pipe": 1e3 * rd["flop_per_launch"] * (6 if rd["method"] == "bf16x3" else 1) / (rd["peak"] * 1e12),
                               "hbm": 1e3 * bytes_ / (HBM_PEAK_GBPS * 1e9), "hbm_bytes": bytes_,
                               "what": "per product launch: MFMA flops issued / pipe peak; (X + W1 + output) bytes / 8 TB/s"}
            out["roofline_dense"] = rd
        except Exception as e:
            out["roofline_dense"] = {"error": repr(e)}
    if extras:
        try:
            leg = hbm_regime_leg(args.hbm_scale, args.hidden if args.hidden > 64 else 128, device)
            out["roofline"]["hbm_regime"] = leg
            # the same as scalars, so that a reader that keeps only the first level of `roofline` keeps the HBM-roofline numbers
            out["roofline"].update({"hbm_regime_achieved": leg["achieved"], "hbm_regime_frac": leg["frac"],
                                    "hbm_regime_avg_launch_ms": leg["avg_launch_ms"],
                                    "hbm_regime_traffic_frac_of_peak": leg["traffic_frac_of_peak"]})
            log("hbm regime leg:", "%.0f GB/s" % leg["achieved"])
        except Exception as e:          # the leg is an extra: report its failure, keep the headline
            out["roofline"]["hbm_regime"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_cli:
        # The shipped program on the same workload: `gcn-hip <dataset> - - <hidden> - - - - <epochs>` as a child process,
        # reading the dataset from its binary cache, with the command line's own defaults — epochs/s from ITS
        # `total training time=` line (src/seq/gcn.cpp:152), so the driver's record states what the drop-in binary
        # delivers next to what the library delivers under this file's driver loop.
        try:
            from cuda_gcn_amd import clirun
            r = clirun.run_on_dataset(ds, hidden=args.hidden, epochs=args.cli_epochs, env={"GCN_SEED": "1"})
            rs = clirun.run_on_dataset(ds, hidden=args.hidden, epochs=max(10, args.cli_epochs // 5),
                                       env={"GCN_SEED": "1", "GCN_SYNC_EPOCHS": "1", "GCN_REFERENCE_ORDER": "1", "GCN_EVAL_LANE": "0"})
            out["cli"] = {"command": r["command"], "epochs": len(r["epochs"]), "total_training_time_s": r["total_training_time_s"],
                          "epochs_per_s": r["epochs_per_s"], "ms_per_epoch": r["ms_per_epoch"],
                          "epochs_per_s_after_warmup": r.get("epochs_per_s_after_warmup"),
                          "load_s": r.get("load_s"), "model_build_s": r.get("model_build_s"), "process_wall_s": round(r["process_wall_s"], 2),
                          "cache_MB": r["cache_MB"], "cache_write_s": r["cache_write_s"],
                          "final": r["epochs"][-1], "test": r.get("test"),
                          "schedule": "the command line's defaults at early_stopping == 0: epochs enqueued ahead of the printed line, validation "
                                      "lane, aggregate-first evaluation (host/main.cpp)",
                          "reference_loop": {"command": rs["command"], "env": rs["env"], "epochs": len(rs["epochs"]),
                                             "epochs_per_s": rs["epochs_per_s"], "ms_per_epoch": rs["ms_per_epoch"],
                                             "what": "one epoch, wait, print (gcn.cpp:133-151), the reference's operation order, one stream"}}
            log(f"cli: {r['epochs_per_s']:.1f} epochs/s over {len(r['epochs'])} epochs (load {r.get('load_s')} s, build {r.get('model_build_s')} s); "
                f"reference loop {rs['epochs_per_s']:.1f}")
        except Exception as e:          # an extra: report its failure, keep the headline
            out["cli"] = {"error": repr(e)}
    if rank == 0:
        # rank 0, any N: the CPU path on this box's host cores (the other ranks wait at the barrier below)
        # (SURVEY 8d / the bench contract: at N = 1; with several ranks only on request, GCN_BENCH_CPU_BASELINE_ANY_N=1)
        if args.no_cpu_baseline or (world > 1 and not os.environ.get("GCN_BENCH_CPU_BASELINE_ANY_N")):
            out["cpu_baseline"] = None
        else:
            try:
                out["cpu_baseline"] = cpu_baseline(ds, args.hidden)
            except Exception as e:      # never lose the finished line to the baseline
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
