"""CPU-only tests of the C++ host logic in libgcnhost.so (no GPU calls):
the text loader and binary cache, the Glorot/RNG replay, the row partition."""
import os
import re
import tempfile

import numpy as np
import pytest

from cuda_gcn_amd import datagen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
KEYS = ("g_indptr", "g_indices", "f_indptr", "f_indices", "f_val", "split", "label")


def test_gcnhost_exports_every_declared_symbol():
    from cuda_gcn_amd import _lib
    lib = _lib.gcnhost()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gcnhost.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(gcnhost_\w+)\s*\(", txt)) - {"gcnhost_allgather_fn", "gcnhost_allreduce_fn"})
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.GCNHOST_SYMBOLS) == names


@pytest.mark.parametrize("case", ["noeol", "ragged", "plain"])
def test_parser_matches_reference_fixtures(case):
    """the C++ loader against the arrays the reference's Parser produced (tests/golden/parser.npz)"""
    from cuda_gcn_amd import model
    g = np.load(os.path.join(GOLD, "parser.npz"))
    with tempfile.TemporaryDirectory() as td:
        for ext, key in ((".graph", "graph_txt"), (".svmlight", "svm_txt"), (".split", "split_txt")):
            with open(os.path.join(td, case + ext), "w") as f:
                f.write(str(g[f"parse_{case}_{key}"]))
        ds = model.load_dataset(td, case)
    for k in KEYS:
        want = g[f"parse_{case}_{k}"]
        assert np.array_equal(ds[k].view(np.uint32) if k == "f_val" else ds[k], want.view(np.uint32) if k == "f_val" else want), k
    assert [ds["num_nodes"], ds["input_dim"], ds["output_dim"]] == g[f"parse_{case}_dims"].tolist()


def test_parser_missing_file_fails():
    from cuda_gcn_amd import model
    with tempfile.TemporaryDirectory() as td:
        with pytest.raises(model.GcnHostError, match="Cannot read input"):
            model.load_dataset(td, "nothing")


def test_parser_vs_oracle_and_binary_cache(oracle):
    from cuda_gcn_amd import model
    ds = datagen.make_dataset("tiny-syn", seed=11)
    with tempfile.TemporaryDirectory() as td:
        datagen.write_text(ds, td, "t")
        a = model.load_dataset(td, "t")
        b = oracle.parse(td, "t")
        for k in KEYS:
            assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], ds[k]), k
        model.save_binary(a, os.path.join(td, "t.gcnbin"))
        for ext in (".graph", ".split", ".svmlight"):
            os.remove(os.path.join(td, "t" + ext))
        c = model.load_dataset(td, "t")             # only the cache is left
        for k in KEYS:
            assert np.array_equal(a[k], c[k]), k
        assert (c["num_nodes"], c["input_dim"], c["output_dim"]) == (ds["num_nodes"], ds["input_dim"], ds["output_dim"])


def test_seeding_is_private_to_the_model():
    """Several models are built at the same time in one process (gcn-hip: a host thread per GPU; the tests: logical ranks as
    threads).  The reference seeds through libc's process-wide srand / rand; with that, two threads seeding at once drew each
    other's numbers and a rank started from other weights than its peers.  Eight threads seed and draw 3 000 times each,
    half of them with another seed, while the main thread hammers the process-wide generator: every draw must be the
    reference's weights for ITS seed (the golden fixture for seed 7)."""
    import ctypes, threading
    from cuda_gcn_amd import model
    mods = np.load(os.path.join(GOLD, "modules.npz"))
    want7 = mods["glorot_seed7_30x20"]
    want9 = model.glorot(600, 30, 20, seed=9)
    assert not np.array_equal(want7, want9)
    libc = ctypes.CDLL(None)
    stop, bad = threading.Event(), []

    def body(k):
        seed, want = (7, want7) if k % 2 == 0 else (9, want9)
        for _ in range(3000):
            if not np.array_equal(model.glorot(600, 30, 20, seed=seed), want):
                bad.append(k)
                return

    def noise():
        while not stop.is_set():
            libc.srand(123)
            libc.rand()

    th = [threading.Thread(target=body, args=(k,)) for k in range(8)]
    nz = threading.Thread(target=noise)
    nz.start()
    for t in th:
        t.start()
    for t in th:
        t.join()
    stop.set()
    nz.join()
    assert not bad, bad


def test_glorot_and_masks_replay_reference_rng(oracle):
    """the host RNG in the product must reproduce the reference's stream: same seed ->
    same initial weights as gcn-seq, same dropout decisions in HOST_MASKS mode"""
    from cuda_gcn_amd import model
    mods = np.load(os.path.join(GOLD, "modules.npz"))
    assert np.array_equal(model.glorot(600, 30, 20, seed=7), mods["glorot_seed7_30x20"])
    for seed in (1, 42, 20191210):
        oracle.rand_seed_time(seed)
        w1 = oracle.glorot(23 * 16, 23, 16)
        w2 = oracle.glorot(16 * 5, 16, 5)
        assert np.array_equal(model.glorot(23 * 16, 23, 16, seed), w1)
        assert np.array_equal(model.glorot(16 * 5, 16, 5, seed, skip_draws=23 * 16), w2)
        # dropout decisions that follow in the stream
        x = np.ones(1000, np.float32)
        _, mask = oracle.dropout_fwd(x, 0.5)
        assert np.array_equal(model.host_masks(1000, 0.5, seed, skip_draws=23 * 16 + 16 * 5), mask.astype(np.uint8))


@pytest.mark.parametrize("name,world", [("cora-syn", 1), ("cora-syn", 2), ("cora-syn", 8), ("tiny-syn", 3), ("tiny-syn", 4)])
def test_partition_covers_and_balances(name, world):
    from cuda_gcn_amd import model
    ds = datagen.make_dataset(name)
    gp = ds["g_indptr"]
    start, rows_max = model.partition(gp, world)
    n = gp.size - 1
    assert start[0] == 0 and start[-1] == n and np.all(np.diff(start) >= 0)
    assert rows_max == max(1, int(np.diff(start).max()))
    # work = edges + mean-degree per row, within 25 % of the ideal share (+ one heavy row)
    cost = np.diff(gp) + gp[-1] / n
    share = [cost[start[q]:start[q + 1]].sum() for q in range(world)]
    assert max(share) <= 1.25 * cost.sum() / world + cost.max()


def test_gcnbin_written_from_python_loads_in_cpp():
    from cuda_gcn_amd import model
    ds = datagen.make_dataset("tiny-syn", seed=5)
    with tempfile.TemporaryDirectory() as td:
        datagen.write_gcnbin(ds, os.path.join(td, "t.gcnbin"))
        back = model.load_dataset(td, "t")
    for k in KEYS:
        assert np.array_equal(back[k], ds[k]), k
    assert (back["num_nodes"], back["input_dim"], back["output_dim"]) == (ds["num_nodes"], ds["input_dim"], ds["output_dim"])


def test_graphsage_converter_on_a_small_fixture():
    """tools/reddit_convert.py on a 6-node GraphSAGE-format dataset: dropped node, sorted renumbering,
    split codes, train-only standardisation, self loop first"""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reddit_convert
    nodes = [{"id": "n3", "val": False, "test": False}, {"id": "n1", "val": True, "test": False},
             {"id": "n2", "val": False, "test": True}, {"id": "n0", "val": False, "test": False},
             {"id": "broken"}, {"id": "n4", "val": False, "test": False}]
    links = [{"source": 0, "target": 1}, {"source": 1, "target": 2}, {"source": 3, "target": 0}, {"source": 4, "target": 0},
             {"source": 5, "target": 3}]
    feats = np.arange(30, dtype=np.float64).reshape(6, 5) ** 1.5
    id_map = {n["id"]: i for i, n in enumerate(nodes)}
    class_map = {"n0": 2, "n1": 0, "n2": 1, "n3": 1, "n4": 0, "broken": 0}
    with tempfile.TemporaryDirectory() as td:
        json.dump({"nodes": nodes, "links": links}, open(os.path.join(td, "x-G.json"), "w"))
        np.save(os.path.join(td, "x-feats.npy"), feats)
        json.dump(id_map, open(os.path.join(td, "x-id_map.json"), "w"))
        json.dump(class_map, open(os.path.join(td, "x-class_map.json"), "w"))
        ds = reddit_convert.convert(td, "x")
    assert ds["num_nodes"] == 5 and ds["output_dim"] == 3 and ds["input_dim"] == 5     # "broken" dropped
    # sorted ids n0..n4 -> 0..4; edges n3-n1, n1-n2, n0-n3, n4-n0
    want = {0: [0, 3, 4], 1: [1, 3, 2], 2: [2, 1], 3: [3, 1, 0], 4: [4, 0]}
    for r, row in want.items():
        assert ds["g_indices"][ds["g_indptr"][r]:ds["g_indptr"][r + 1]].tolist() == row
    assert ds["split"].tolist() == [1, 2, 3, 1, 1] and ds["label"].tolist() == [2, 0, 1, 1, 0]
    X = ds["f_val"].reshape(5, 5)
    tr = X[[0, 3, 4]]
    assert np.allclose(tr.mean(axis=0), 0, atol=1e-6) and np.allclose(tr.std(axis=0), 1, atol=1e-5)


def test_truncated_or_inconsistent_gcnbin_is_rejected_and_text_is_parsed(tmp_path):
    """a cache whose magic matches but whose arrays are cut off, oversized or inconsistent with the header must
    not leave a half-filled GCNData behind: the loader falls back to the text files and gives their arrays"""
    import struct
    from cuda_gcn_amd import model
    ds = datagen.make_dataset("tiny-syn")
    root = str(tmp_path)
    datagen.write_text(ds, root, "t")
    want = model.load_dataset(root, "t")
    good = os.path.join(root, "good.gcnbin")
    datagen.write_gcnbin(ds, good)
    blob = open(good, "rb").read()
    bad = {
        "cut": blob[:len(blob) // 2],
        "huge_len": blob[:20] + struct.pack("<Q", 1 << 60) + blob[28:],
        "short_split": None,
        "bad_column": None,
    }
    d2 = dict(ds); d2["split"] = ds["split"][:-3]
    datagen.write_gcnbin(d2, os.path.join(root, "tmp.gcnbin")); bad["short_split"] = open(os.path.join(root, "tmp.gcnbin"), "rb").read()
    d3 = dict(ds); gi = ds["g_indices"].copy(); gi[5] = ds["num_nodes"] + 7; d3["g_indices"] = gi
    datagen.write_gcnbin(d3, os.path.join(root, "tmp.gcnbin")); bad["bad_column"] = open(os.path.join(root, "tmp.gcnbin"), "rb").read()
    os.remove(os.path.join(root, "tmp.gcnbin"))
    for tag, data in bad.items():
        with open(os.path.join(root, "t.gcnbin"), "wb") as f:
            f.write(data)
        got = model.load_dataset(root, "t")
        for k in ("g_indptr", "g_indices", "f_indptr", "f_indices", "f_val", "split", "label"):
            assert np.array_equal(got[k], want[k]), (tag, k)
        assert got["num_nodes"] == want["num_nodes"]
    with open(os.path.join(root, "t.gcnbin"), "wb") as f:       # and the intact cache is accepted
        f.write(blob)
    got = model.load_dataset(root, "t")
    assert np.array_equal(got["g_indices"], ds["g_indices"]) and np.array_equal(got["f_val"], ds["f_val"])


@pytest.mark.parametrize("name,world,mode", [("tiny-syn", 3, 2), ("cora-syn", 4, 2), ("rmat-12", 8, 0), ("rmat-12", 2, 0), ("cora-syn", 2, 1)])
def test_exchange_plans_are_consistent_across_ranks(name, world, mode, oracle):
    """every rank derives its plan from the whole adjacency: the plans must agree on the mode, rank q's send list for
    p must be exactly what p expects from q, and aggregating through the table (own rows + received rows) must give
    the rows the single-process oracle computes"""
    from cuda_gcn_amd import model
    if name.startswith("rmat"):
        gp, gi = datagen.rmat_graph(int(name.split("-")[1]))
    else:
        ds = datagen.make_dataset(name)
        gp, gi = ds["g_indptr"], ds["g_indices"]
    N = gp.size - 1
    start, _ = model.partition(gp, world)
    plans = [model.exchange_plan(gp, gi, world, r, mode) for r in range(world)]
    assert len({p["halo"] for p in plans}) == 1 and len({round(p["halo_share"], 12) for p in plans}) == 1
    halo = plans[0]["halo"]
    if name == "rmat-12" and world == 8:
        assert halo and plans[0]["halo_share"] < 0.75            # R-MAT ids carry locality: the automatic choice is HALO
    if mode == 1:
        assert not halo
    x = np.random.default_rng(1).standard_normal((N, 5)).astype(np.float32)
    want = oracle.graphsum(gp, gi, x, 5)
    deg = np.diff(gp).astype(np.int64)
    for p_rank, p in enumerate(plans):
        r0, r1 = int(start[p_rank]), int(start[p_rank + 1])
        assert p["n_local"] == r1 - r0
        tg = p["table_global"]
        assert np.array_equal(tg[p["own_offset"]:p["own_offset"] + p["n_local"]], np.arange(r0, r1))
        if halo:
            for q in range(world):
                seg = p["recv_rows"][p["recv_off"][q]:p["recv_off"][q + 1]]
                sent = plans[q]["send_rows"][plans[q]["send_off"][p_rank]:plans[q]["send_off"][p_rank + 1]]
                assert np.array_equal(seg, sent), (p_rank, q)
                assert np.array_equal(tg[p["n_local"] + p["recv_off"][q]:p["n_local"] + p["recv_off"][q + 1]], seg + int(start[q]))
            assert p["recv_off"][p_rank] == p["recv_off"][p_rank + 1]
            assert p["table_rows"] == p["n_local"] + p["recv_rows"].size < N + 1
        # the table a completed exchange leaves on this rank, then the aggregation in numpy
        table = np.where(tg[:, None] >= 0, x[np.maximum(tg, 0)], np.nan).astype(np.float32)
        ip, ix, cd = p["indptr"], p["indices"], p["col_deg"]
        assert np.array_equal(tg[ix], gi[gp[r0]:gp[r1]])                      # columns decode to the same nodes
        assert np.array_equal(cd[ix], deg[gi[gp[r0]:gp[r1]]])
        src = np.repeat(np.arange(r1 - r0), np.diff(ip))
        coef = (1.0 / np.sqrt((deg[r0:r1][src] * cd[ix].astype(np.int64)).astype(np.float32)).astype(np.float64)).astype(np.float32)
        got = np.zeros((r1 - r0, 5), np.float64)
        np.add.at(got, src, coef[:, None].astype(np.float64) * table[ix].astype(np.float64))
        assert np.allclose(got, want[r0:r1], rtol=1e-5, atol=1e-5)


def _structure_groups(gp, gi):
    import ctypes as C
    from cuda_gcn_amd import _lib
    lib = _lib.gcnhost()
    n = gp.size - 1
    gp, gi = np.ascontiguousarray(gp, np.int32), np.ascontiguousarray(gi, np.int32)
    grp = np.full(max(n, 1), -7, np.int32)
    ng, sw, us, ls = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    assert lib.gcnhost_structure_groups(gp.ctypes.data, gi.ctypes.data, n, grp.ctypes.data, C.byref(ng), C.byref(sw), C.byref(ls), C.byref(us)) == 0
    return grp[:n], ng.value, sw.value, ls.value, bool(us.value)


def test_structure_groups_find_planted_communities():
    """row groups for the aggregation's schedule from the graph alone (host/cluster.h, modularity local moving with a
    size bound): on a planted partition the groups hold as many of the edges as the planted labels do, none exceeds the
    size bound, nothing collapses into one group, and the result is deterministic"""
    ds = datagen.make_dataset("reddit-mini")
    gp, gi, lab = ds["g_indptr"], ds["g_indices"], ds["label"]
    n = gp.size - 1
    grp, ng, sweeps, largest, useful = _structure_groups(gp, gi)
    assert useful and 4 <= ng <= n and 1 <= sweeps <= 10
    assert grp.min() == 0 and grp.max() == ng - 1
    sizes = np.bincount(grp, minlength=ng)
    assert sizes[:-1].max() <= 8192 and np.all(np.diff(sizes[:-1]) <= 0)          # bounded, largest first (the last may be the rest)
    assert abs(largest - sizes[0] / n) < 1e-12
    src = np.repeat(np.arange(n), np.diff(gp))
    nl = src != gi
    inside = float((grp[src[nl]] == grp[gi[nl]]).mean())
    chance = float(((sizes / n) ** 2).sum())
    label_inside = float((lab[src[nl]] == lab[gi[nl]]).mean())
    assert inside >= 4 * chance and inside >= 0.95 * label_inside, (inside, chance, label_inside)
    again = _structure_groups(gp, gi)
    assert np.array_equal(again[0], grp) and again[1:] == (ng, sweeps, largest, useful)


def test_structure_groups_degenerate_inputs():
    # no edges but the self loops: nothing merges -> not useful, every node in the one trailing group
    n = 100
    gp = np.arange(n + 1, dtype=np.int32); gi = np.arange(n, dtype=np.int32)
    grp, ng, _, _, useful = _structure_groups(gp, gi)
    assert not useful and ng == 1 and not grp.any()
    # a clique below the size bound: one group -> not useful
    m = 50
    gp = (np.arange(m + 1) * m).astype(np.int32); gi = np.tile(np.arange(m, dtype=np.int32), m)
    grp, ng, _, largest, useful = _structure_groups(gp, gi)
    assert not useful and ng == 1 and largest == 1.0
    # the empty graph
    grp, ng, _, _, useful = _structure_groups(np.zeros(1, np.int32), np.zeros(0, np.int32))
    assert grp.size == 0 and ng == 0 and not useful
    # an R-MAT graph (no communities to find): valid groups, whatever their use
    gp, gi = datagen.rmat_graph(12)
    grp, ng, _, _, _ = _structure_groups(gp, gi)
    assert grp.min() >= 0 and grp.max() == ng - 1


def test_reddit_convert_matches_reference_script():
    """tools/reddit_convert.py against the output of the reference's own reddit_preprocess.py on a 56-node GraphSAGE-format
    fixture (tests/golden/reddit_preprocess.npz, written by tests/golden/make_reddit_golden.py, which executed the script
    unmodified in the build container): node order, adjacency lists, split codes, labels and train-standardised features."""
    import importlib.util
    import json
    import tempfile
    gold = np.load(os.path.join(ROOT, "tests", "golden", "reddit_preprocess.npz"))
    spec = importlib.util.spec_from_file_location("reddit_convert", os.path.join(ROOT, "tools", "reddit_convert.py"))
    rc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rc)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "reddit-G.json"), "w").write(str(gold["in_G"]))
        np.save(os.path.join(d, "reddit-feats.npy"), gold["in_feats"])
        open(os.path.join(d, "reddit-id_map.json"), "w").write(str(gold["in_id_map"]))
        open(os.path.join(d, "reddit-class_map.json"), "w").write(str(gold["in_class_map"]))
        ds = rc.convert(d, "reddit")
    N = gold["out_split"].size
    assert ds["num_nodes"] == N == 55                         # 57 nodes in the file, two without val/test annotations
    assert np.array_equal(ds["split"], gold["out_split"])
    assert np.array_equal(ds["label"], gold["out_label"])
    assert ds["output_dim"] == int(gold["out_label"].max()) + 1
    # adjacency: the loader puts the self loop in front of each line of the .graph file (parser.cpp:30-33)
    gp, gi = gold["out_g_indptr"], gold["out_g_indices"]
    assert ds["g_indptr"][-1] == gi.size + N
    n_self_links = 0
    for k in range(N):
        ours = ds["g_indices"][ds["g_indptr"][k]:ds["g_indptr"][k + 1]]
        theirs = gi[gp[k]:gp[k + 1]]
        assert ours[0] == k and np.array_equal(ours[1:], theirs), (k, ours, theirs)
        n_self_links += int((theirs == k).sum())
    assert n_self_links == 1                                  # the fixture's self link survives as an entry of its own line
    # features: sklearn's dump_svmlight_file drops exact zeros; ours keeps every column (dense first-layer path)
    F = ds["input_dim"]
    assert F == int(gold["out_f_indices"].max()) + 1 == 9
    dense = np.zeros((N, F), np.float64)
    fp = gold["out_f_indptr"]
    for k in range(N):
        dense[k, gold["out_f_indices"][fp[k]:fp[k + 1]]] = gold["out_f_val"][fp[k]:fp[k + 1]]
    ours = ds["f_val"].reshape(N, F)
    assert np.all(ours[:, 4] == 0) and np.all(dense[:, 4] == 0)   # the constant column: variance 0, scaled by 1 -> exactly 0
    assert np.allclose(ours, dense.astype(np.float32), rtol=2e-6, atol=1e-7), np.abs(ours - dense).max()


def test_node_order_by_structure_when_ids_carry_no_locality():
    """rank blocks are contiguous in the node order; with shuffled ids a graph of communities makes every rank read nearly
    all remote rows — the order found in the graph (host/partition.h: choose_node_order over cluster.h's groups) must bring
    the neediest rank's rows per exchange down far enough for a halo plan; graphs whose ids are already local are left alone"""
    from cuda_gcn_amd import datagen, model
    ds = datagen.planted_communities()
    N = ds["num_nodes"]
    for world in (2, 4, 8):
        c = model.choose_node_order(ds["g_indptr"], ds["g_indices"], world)
        assert c["renumbered"] and np.array_equal(np.sort(c["order"]), np.arange(N))
        assert c["ids_share"] > 0.75 and c["new_share"] <= 0.5                  # all-gather under the ids, halo lists after
        assert c["new_recv_rows"] * 2 <= min(c["ids_recv_rows"], c["allgather_rows"])
    local = datagen.planted_communities(shuffle=False)                            # communities contiguous in the ids
    c = model.choose_node_order(local["g_indptr"], local["g_indices"], 4)
    assert not c["renumbered"] and np.array_equal(c["order"], np.arange(N)) and c["ids_share"] <= 0.5
    # R-MAT with shuffled ids: no communities to find, and the id order is no worse than the generator's — it stays
    gp, gi = datagen.rmat_graph(13)
    n = gp.size - 1
    perm = np.random.default_rng(3).permutation(n).astype(np.int32)
    inv = np.argsort(perm).astype(np.int32)
    deg = np.diff(gp)
    sgp = np.zeros(n + 1, np.int64)
    sgp[1:] = np.cumsum(deg[inv])
    sgi = np.concatenate([perm[gi[gp[o]:gp[o + 1]]] for o in inv]).astype(np.int32)
    c = model.choose_node_order(sgp.astype(np.int32), sgi, 8)
    assert c["ids_share"] <= 0.75                                                 # the halo plan is still what make_exchange_plan picks
    assert np.array_equal(np.sort(c["order"]), np.arange(n))


def test_gcn_hip_writes_the_cache_and_refuses_to_run_without_a_gpu(tmp_path):
    """`GCN_WRITE_CACHE=1 gcn-hip <name>`: the text files are parsed, data/<name>.gcnbin is written (identical arrays when
    loaded back), and on a box without a GPU the program then exits with an error — it has no CPU path"""
    import subprocess
    from cuda_gcn_amd import model, _lib
    import ctypes
    n = ctypes.c_int(-1)
    if _lib.gcnhip().gcnhip_device_count(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    binary = os.path.join(ROOT, "cuda_gcn_amd", "bin", "gcn-hip")
    if not os.path.exists(binary):
        pytest.skip("gcn-hip not built")
    ds = datagen.make_dataset("tiny-syn")
    root = str(tmp_path / "data")
    datagen.write_text(ds, root, "tiny-syn")
    r = subprocess.run([binary, "tiny-syn"], cwd=str(tmp_path), env=dict(os.environ, GCN_WRITE_CACHE="1"), capture_output=True, text=True)
    assert r.returncode != 0 and "no GPU available" in r.stderr and "Parse Split Succeeded." in r.stdout
    assert os.path.exists(os.path.join(root, "tiny-syn.gcnbin")) and "wrote" in r.stderr
    for ext in (".graph", ".svmlight", ".split"):
        os.remove(os.path.join(root, "tiny-syn" + ext))
    back = model.load_dataset(root, "tiny-syn")
    for k in ("g_indptr", "g_indices", "f_indptr", "f_indices", "f_val", "split", "label"):
        assert np.array_equal(back[k], ds[k]), k


# ---- provenance of stored measurements (round 6): bench.py marks a quoted profile stale when the kernel source it was taken on changed
def test_profile_provenance_hashes_and_stale_reason(tmp_path, monkeypatch):
    from cuda_gcn_amd import provenance
    h = provenance.source_sha(["graphsum.hip", "tools/gather_peak.hip", "no_such_file.h"])
    assert len(h["graphsum.hip"]) == 16 and len(h["tools/gather_peak.hip"]) == 16 and h["no_such_file.h"] is None
    assert provenance.stale_reason({"sources": {"graphsum.hip": h["graphsum.hip"]}}) is None
    assert "graphsum.hip" in provenance.stale_reason({"sources": {"graphsum.hip": "0" * 16}})
    assert "no source hashes" in provenance.stale_reason({"commit": "abc"}) and "no source hashes" in provenance.stale_reason(None)


def test_bench_quotes_a_profile_of_another_tree_as_stale(tmp_path, monkeypatch):
    """bench._pmc: a PMC file whose recorded source hash differs from the tree's is still returned (the numbers are what there is),
    with stale = True and the reason"""
    import importlib, json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    k, src, why = bench._pmc(bench.PMC_FILES, bench.GS_KERNEL, None)
    assert k is not None and src.endswith("r06_graphsum_pmc.json") and isinstance(k["stale"], bool)
    assert k["stale"] == (k["stale_why"] is not None)         # (whether it IS stale depends on the tree: a kernel edit makes it so until the file is re-taken)
    fake = tmp_path / "profiles"
    fake.mkdir()
    doc = json.load(open(os.path.join(root, "profiles", "r06_graphsum_pmc.json")))
    doc["_meta"]["sources"] = {"graphsum.hip": "0" * 16}
    json.dump(doc, open(fake / "r06_graphsum_pmc.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    k, src, why = bench._pmc(["r06_graphsum_pmc.json"], bench.GS_KERNEL, None)
    assert k["stale"] is True and "graphsum.hip" in k["stale_why"]
