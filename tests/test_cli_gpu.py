"""The PRODUCT, not just the library: `gcn-hip <dataset>` (cuda_gcn_amd/host/main.cpp; the reference's entry is
src/main.cpp:15-48 -> Parser -> GCN::run, src/seq/gcn.cpp:130-158) fed from the binary dataset cache
(`<name>.gcnbin`; replaces src/common/parser.cpp:20-103 for graphs whose text form is gigabytes),
beside `gcn-seq` reading the TEXT form of the same data.

Tolerances are test_model_gpu.py's: identical Glorot weights (same GCN_SEED) and, with GCN_HOST_MASKS=1, identical
dropout decisions, so per-epoch losses agree to 2e-4 in the first ten epochs / 2e-3 after (f32 summation order), accuracies
to two rows of the split.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from cuda_gcn_amd import datagen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "cuda_gcn_amd", "bin", "gcn-hip")
SEQ = os.path.join(ROOT, "oracle", "gcn-seq")


def run_cli(binary, cwd, args, **env):
    r = subprocess.run([binary] + args, cwd=cwd, env=dict(os.environ, **env), capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = r.stdout.strip().splitlines()
    ep = [dict((k, float(v)) for k, v in (t.split("=") for t in l.split())) for l in lines if l.startswith("epoch=")]
    return lines, ep, r.stderr


def total_and_test(lines):
    tot = [l for l in lines if l.startswith("total training time=")]
    tst = [l for l in lines if l.startswith("test_loss=")]
    assert len(tot) == 1 and len(tst) == 1
    return float(tot[0].split("=")[1]), dict((k, float(v)) for k, v in (t.split("=") for t in tst[0].split()))


def compare(ea, eb, ds, early=10):
    assert len(ea) == len(eb)
    n_tr, n_va = int((ds["split"] == 1).sum()), int((ds["split"] == 2).sum())
    for i, (x, y) in enumerate(zip(ea, eb)):
        tol = 2e-4 if i < early else 2e-3
        assert x["epoch"] == y["epoch"] == i + 1
        assert abs(x["train_loss"] - y["train_loss"]) <= tol + 1e-5, (i, x, y)      # (+ the %.5f of the printed line)
        assert abs(x["val_loss"] - y["val_loss"]) <= tol + 1e-5, (i, x, y)
        assert abs(x["train_acc"] - y["train_acc"]) <= 2.0 / n_tr + 1e-5 and abs(x["val_acc"] - y["val_acc"]) <= 2.0 / n_va + 1e-5, (i, x, y)


@pytest.fixture(scope="module")
def seq_binary():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "gcn-seq"], check=True)
    return SEQ


@pytest.mark.parametrize("sync", [False, True])
def test_reddit_mini_from_gcnbin_beside_gcn_seq_on_text(seq_binary, sync):
    """reddit-mini (23 296 nodes, dense 602 features -> 128 -> 41: the MFMA first layer, split rows, label-major schedule,
    validation lane, aggregate-first evaluation — the CLI's default schedule; sync: GCN_SYNC_EPOCHS=1, the reference's loop):
    gcn-hip loads the .gcnbin, gcn-seq parses 224 MB of text; dropout 0.5 with the CPU path's decisions replayed"""
    ds = datagen.make_dataset("reddit-mini")
    args = ["reddit-mini", "-", "-", "128", "-", "0.5", "-", "-", "4"]
    with tempfile.TemporaryDirectory() as td_bin, tempfile.TemporaryDirectory() as td_txt:
        os.makedirs(os.path.join(td_bin, "data"))
        datagen.write_gcnbin(ds, os.path.join(td_bin, "data", "reddit-mini.gcnbin"))
        la, ea, err = run_cli(HIP, td_bin, args, GCN_SEED="3", GCN_HOST_MASKS="1", GCN_SYNC_EPOCHS="1" if sync else "0")
        assert "Loaded binary cache." in la and "RUNNING ON GPU" in la
        assert "dataset loaded in" in err
        datagen.write_text(ds, os.path.join(td_txt, "data"), "reddit-mini")
        lb, eb, _ = run_cli(seq_binary, td_txt, args, GCN_SEED="3")
        assert "RUNNING ON CPU" in lb
    assert len(ea) == 4
    compare(ea, eb, ds)
    ta, xa = total_and_test(la)
    tb, xb = total_and_test(lb)
    assert abs(xa["test_loss"] - xb["test_loss"]) <= 2e-4 + 1e-5 and abs(xa["test_acc"] - xb["test_acc"]) <= 2.0 / int((ds["split"] == 3).sum()) + 1e-5
    assert abs(sum(e["time"] for e in ea) - ta) <= 1e-3 * max(1.0, ta)            # the total is the sum of the printed times (gcn.cpp:139-140,152)


def test_pubmed_from_gcnbin_100_epochs_default_schedule(seq_binary):
    """BASELINE configs[1] through the CLI with the reference's defaults (hidden 16, 100 epochs): sparse X (CSR/CSC kernels),
    pipelined epochs; the cache written by the C++ Parser itself from the text files (gcnhost_dataset_save_binary)"""
    from cuda_gcn_amd import model
    ds = datagen.make_dataset("pubmed-syn")
    with tempfile.TemporaryDirectory() as td:
        root = os.path.join(td, "data")
        datagen.write_text(ds, root, "pubmed-syn")
        lb, eb, _ = run_cli(seq_binary, td, ["pubmed-syn"], GCN_SEED="5", GCN_HOST_MASKS="1")
        parsed = model.load_dataset(root, "pubmed-syn")                              # text -> GCNData (C++ Parser)
        model.save_binary(parsed, os.path.join(root, "pubmed-syn.gcnbin"))
        for ext in (".graph", ".svmlight", ".split"):
            os.remove(os.path.join(root, "pubmed-syn" + ext))                        # only the cache is left
        la, ea, _ = run_cli(HIP, td, ["pubmed-syn"], GCN_SEED="5", GCN_HOST_MASKS="1")
        assert "Loaded binary cache." in la
    assert len(ea) == len(eb) == 100
    compare(ea, eb, ds)
    # epochs were enqueued ahead of the printed line: the lines still arrive one per epoch, in order, each with its own time
    assert all(e["time"] >= 0 for e in ea)


def test_reddit_convert_fixture_to_gcnbin_to_gcn_hip(seq_binary):
    """SURVEY §8f rank 3's last hop: the GraphSAGE-format fixture the reference's own reddit_preprocess.py was pinned on
    (tests/golden/reddit_preprocess.npz) -> tools/reddit_convert.py -> .gcnbin -> gcn-hip, beside gcn-seq on the text form
    the converter writes (a self link in the data: the loader's self loop + the link, i.e. the node twice in its own row)"""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "reddit_preprocess.npz"))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "src")
        os.makedirs(src)
        open(os.path.join(src, "reddit-G.json"), "w").write(str(gold["in_G"]))
        np.save(os.path.join(src, "reddit-feats.npy"), gold["in_feats"])
        open(os.path.join(src, "reddit-id_map.json"), "w").write(str(gold["in_id_map"]))
        open(os.path.join(src, "reddit-class_map.json"), "w").write(str(gold["in_class_map"]))
        bin_root, txt_root = os.path.join(td, "b", "data"), os.path.join(td, "t", "data")
        conv = os.path.join(ROOT, "tools", "reddit_convert.py")
        subprocess.run([sys.executable, conv, src, "--prefix", "reddit", "--out", bin_root], check=True, capture_output=True)
        subprocess.run([sys.executable, conv, src, "--prefix", "reddit", "--out", txt_root, "--text"], check=True, capture_output=True)
        os.remove(os.path.join(txt_root, "reddit.gcnbin"))
        args = ["reddit", "-", "-", "16", "-", "0.5", "-", "-", "30"]
        la, ea, _ = run_cli(HIP, os.path.join(td, "b"), args, GCN_SEED="9", GCN_HOST_MASKS="1")
        lb, eb, _ = run_cli(seq_binary, os.path.join(td, "t"), args, GCN_SEED="9")
    assert "Loaded binary cache." in la
    N = int(gold["out_split"].size)
    ds = {"split": gold["out_split"]}
    assert N == 55 and len(ea) == 30
    compare(ea, eb, ds)


def test_worker_thread_path_on_one_gpu(seq_binary):
    """GCN_THREADS=1: the host-thread-per-GPU path of GCN_GPUS > 1 (main.cpp: worker threads, join, failure policy) with
    one GPU — same lines as the direct call"""
    ds = datagen.make_dataset("cora-syn")
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "data"))
        datagen.write_gcnbin(ds, os.path.join(td, "data", "cora-syn.gcnbin"))
        args = ["cora-syn", "-", "-", "-", "-", "-", "-", "-", "20"]
        la, ea, _ = run_cli(HIP, td, args, GCN_SEED="4")
        lb, eb, _ = run_cli(HIP, td, args, GCN_SEED="4", GCN_THREADS="1", GCN_GPUS="1")
        # a GPU count the box does not have is refused before anything is built
        r = subprocess.run([HIP] + args, cwd=td, env=dict(os.environ, GCN_GPUS="64"), capture_output=True, text=True)
    assert r.returncode != 0 and "GCN_GPUS=64" in r.stderr
    strip = lambda e: {k: v for k, v in e.items() if k != "time"}
    assert [strip(e) for e in ea] == [strip(e) for e in eb] and len(ea) == 20       # device RNG dropout: same seed, same bits


def test_pipelined_and_synchronous_loops_print_the_same_numbers():
    """HipGCN::run_pipelined (epochs enqueued ahead, metrics copied back behind them) against the reference's loop
    (GCN_SYNC_EPOCHS=1): every printed number identical, with and without the validation lane, on the dense path too"""
    for name, hidden in (("cora-syn", "16"), ("reddit-mini", "128")):
        ds = datagen.make_dataset(name)
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, "data"))
            datagen.write_gcnbin(ds, os.path.join(td, "data", name + ".gcnbin"))
            args = [name, "-", "-", hidden, "-", "-", "-", "-", "12"]
            runs = []
            for env in ({}, {"GCN_SYNC_EPOCHS": "1"}, {"GCN_EVAL_LANE": "0"}, {"GCN_EVAL_LANE": "0", "GCN_SYNC_EPOCHS": "1"}):
                la, ea, _ = run_cli(HIP, td, args, GCN_SEED="2", **env)
                _, tst = total_and_test(la)
                runs.append(([{k: v for k, v in e.items() if k != "time"} for e in ea], {k: v for k, v in tst.items() if k != "time"}))
        assert len(runs[0][0]) == 12
        for r in runs[1:]:
            assert r == runs[0], name


def test_grouped_read_back_prints_every_epoch_once_in_order():
    """run_pipelined brings the metrics back in groups of consecutive epochs once the first 16 have set the group size
    (HIPGCN_READBACK_GROUP pins it; 7 does not divide the 1024-row metrics ring, so a group straddles the wrap): 1100 Cora
    epochs print the same 1100 lines as the wait-per-epoch loop, whatever the grouping, the lane or the stream the copies
    ride on, and the time= fields add up to `total training time=`"""
    ds = datagen.make_dataset("cora-syn")
    n = 1100
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "data"))
        datagen.write_gcnbin(ds, os.path.join(td, "data", "cora-syn.gcnbin"))
        args = ["cora-syn", "-", "-", "-", "-", "-", "-", "-", str(n)]
        runs = []
        for env in ({"GCN_SYNC_EPOCHS": "1", "GCN_EVAL_LANE": "0"}, {}, {"HIPGCN_READBACK_GROUP": "1"}, {"HIPGCN_READBACK_GROUP": "7"},
                    {"HIPGCN_READBACK_GROUP": "64"}, {"HIPGCN_READBACK_GROUP": "7", "GCN_EVAL_LANE": "1"},
                    {"HIPGCN_READBACK_GROUP": "7", "HIPGCN_READBACK_STREAM": "1"}, {"GCN_EVAL_LANE": "1", "HIPGCN_READBACK_STREAM": "1"}):
            la, ea, _ = run_cli(HIP, td, args, GCN_SEED="5", **env)
            tot, tst = total_and_test(la)
            assert [e["epoch"] for e in ea] == list(range(1, n + 1)), env
            assert abs(sum(e["time"] for e in ea) - tot) <= n * 1e-5 + 1e-3, env        # each line rounds to %.5f
            runs.append(([{k: v for k, v in e.items() if k != "time"} for e in ea], {k: v for k, v in tst.items() if k != "time"}))
    for r in runs[1:]:
        assert r == runs[0]
