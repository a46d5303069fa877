"""bench.py as the driver invokes it.

CPU: `python bench.py --gpus 2` with no WORLD_SIZE must start its own ranks as child processes (never exec
from a process that touched the GPU) and hand back their failure as a non-zero exit — here the children stop at
"no GPU visible".
GPU (-m gpu): the 1-GPU line carries the contract fields with a roofline fraction <= 1 against the bound that
binds; `--gpus 2` self-launches two ranks (both on GPU 0, host-staged transport: the one-GPU box has no second
device for RCCL) and prints ONE JSON line with n_gpus = 2.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, env=None, timeout=900):
    e = dict(os.environ, OMP_NUM_THREADS="1", **(env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, BENCH] + [str(a) for a in args], env=e, capture_output=True, text=True, timeout=timeout)
    return p


def json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_launcher_parent_returns_child_failure_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = run_bench(["--gpus", 2, "--steps", 1, "--warmup", 0, "--eval-lane", "off"], env={"GCN_BENCH_TIMEOUT": "300"})
    assert p.returncode != 0
    assert "starting 2 ranks" in p.stderr and "no GPU visible" in p.stderr, p.stderr[-2000:]
    assert json_lines(p.stdout) == []


def test_launcher_does_not_touch_gpu_before_spawning():
    """the parent path of bench.py must not import torch or load a native library before it starts the ranks"""
    src = open(BENCH).read()
    main_body = src[src.index("def main():"):]
    spawn_at = main_body.index("launch_ranks(")
    assert "import torch" not in main_body[:spawn_at]
    assert "cuda_gcn_amd" not in main_body[:spawn_at]
    head = src[:src.index("def main():")]
    top_level_imports = [ln for ln in head.splitlines() if ln.startswith("import ") or ln.startswith("from ")]
    assert not any("torch" in ln or "cuda_gcn_amd" in ln or "numpy" in ln for ln in top_level_imports), top_level_imports


CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


@pytest.mark.gpu
def test_bench_one_gpu_line():
    p = run_bench(["--dataset", "reddit-mini", "--steps", 4, "--warmup", 1, "--bursts", 1, "--no-cpu-baseline", "--no-extras"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1
    out = lines[0]
    for k in CONTRACT:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["value"] > 0
    assert abs(out["ms_per_step"] * out["value"] - 1e3) < 1e-6 * 1e3
    assert len(out["bursts"]["epochs_per_s"]) == 2
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["launches"] > 0 and r["avg_launch_ms"] > 0


@pytest.mark.gpu
def test_bench_extras_on_a_small_graph():
    """the extra legs of the one-GPU line (label hint withheld, no row groups, reference operation order, the R-MAT leg of
    the roofline — here at a small scale, so the code path and not the HBM regime is what is exercised)"""
    p = run_bench(["--dataset", "reddit-mini", "--steps", 4, "--warmup", 1, "--bursts", 0, "--no-cpu-baseline", "--hbm-scale", 15])
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_lines(p.stdout)[0]
    assert out["config"]["eval_lane"] == "on"                  # one GPU: validation forward on the second stream
    assert out["value_reference_op_order_all_rows"] > 0
    if out["config"]["aggregation_schedule"] == "label-major":
        assert out["value_no_label_hint"] > 0 and "aggregation_schedule_no_label_hint" in out["config"]
    if out["config"]["aggregation_schedule"] != "degree":
        assert out["value_no_row_groups"] > 0
    leg = out["roofline"]["hbm_regime"]
    assert "error" not in leg, leg
    assert leg["bound"] == "hbm" and leg["achieved"] > 0 and leg["launches"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_self_launched():
    p = run_bench(["--gpus", 2, "--dataset", "reddit-mini", "--steps", 3, "--warmup", 1, "--bursts", 0, "--no-cpu-baseline"],
                  env={"GCN_BENCH_TRANSPORT": "host", "GCN_BENCH_DEVICE": "0", "GCN_BENCH_TIMEOUT": "600"})
    assert p.returncode == 0, p.stderr[-3000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    assert out["config"]["parallelism"] == "row-partition x2"
    assert "comm" in out["breakdown_ms_per_epoch"]
    assert 0 < out["final"]["train_loss"] < 10
