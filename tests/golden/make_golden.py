"""Generate the committed golden fixtures from the REFERENCE's own objects.

Run in the build container only (needs oracle/_ref/libref.so, i.e.
/root/reference):   python tests/golden/make_golden.py
Every expected value below is produced by the reference's classes through
oracle/ref_shim.cpp; inputs come from seeded numpy / cuda_gcn_amd.datagen.
The fixtures are data (inputs + expected outputs), not reference source.
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Ref  # noqa: E402
from cuda_gcn_amd import datagen  # noqa: E402


def ds_digest(ds):
    h = hashlib.sha256()
    for k in ("g_indptr", "g_indices", "f_indptr", "f_indices", "f_val", "label", "split"):
        h.update(np.ascontiguousarray(ds[k]).tobytes())
    return h.hexdigest()


def karate():
    import networkx as nx
    g = nx.karate_club_graph()
    n = g.number_of_nodes()
    e = np.array(sorted((min(a, b), max(a, b)) for a, b in g.edges()), np.int64)
    indptr, indices = datagen.csr_with_self_loops(e[:, 0], e[:, 1], n)
    label = np.array([0 if g.nodes[i]["club"] == "Mr. Hi" else 1 for i in range(n)], np.int32)
    return n, indptr, indices, label


def module_cases(r, out):
    rng = np.random.default_rng(1234)
    graphs = {}
    n, gp, gi, _ = karate()
    graphs["karate"] = (n, gp, gi)
    ds = datagen.make_dataset("tiny-syn")
    graphs["tiny"] = (ds["num_nodes"], ds["g_indptr"], ds["g_indices"])
    # a ragged graph: isolated nodes (self loop only) and one hub
    n3 = 41
    lo = np.concatenate([np.zeros(30, np.int64), np.array([5, 7, 9])])
    hi = np.concatenate([np.arange(1, 31), np.array([6, 8, 33])])
    gp3, gi3 = datagen.csr_with_self_loops(lo, hi, n3)
    graphs["ragged"] = (n3, gp3, gi3)
    for gname, (n, gp, gi) in graphs.items():
        out[f"gs_{gname}_indptr"] = gp
        out[f"gs_{gname}_indices"] = gi
        for dim in (1, 7, 16, 41):
            x = rng.standard_normal((n, dim)).astype(np.float32)
            out[f"gs_{gname}_d{dim}_in"] = x
            out[f"gs_{gname}_d{dim}_out"] = r.graphsum(gp, gi, x, dim)
            out[f"gs_{gname}_d{dim}_bwd"] = r.graphsum(gp, gi, x, dim, backward=True)
    # matmul
    for (m, n, p) in ((34, 16, 7), (97, 128, 41), (5, 3, 2), (1, 1, 1)):
        a = rng.standard_normal((m, n)).astype(np.float32)
        b = rng.standard_normal((n, p)).astype(np.float32)
        cg = rng.standard_normal((m, p)).astype(np.float32)
        k = f"mm_{m}x{n}x{p}"
        out[k + "_a"], out[k + "_b"], out[k + "_cg"] = a, b, cg
        out[k + "_c"] = r.matmul_fwd(a, b, m, n, p)
        ag, bg = r.matmul_bwd(a, b, cg, m, n, p)
        out[k + "_ag"], out[k + "_bg"] = ag, bg
    # sparse matmul on tiny-syn features (+ an empty row)
    fp, fi, fv = ds["f_indptr"].copy(), ds["f_indices"], ds["f_val"]
    F, N = ds["input_dim"], ds["num_nodes"]
    for p in (16, 8, 3):
        w = rng.standard_normal((F, p)).astype(np.float32)
        cg = rng.standard_normal((N, p)).astype(np.float32)
        vals = rng.standard_normal(fv.size).astype(np.float32)
        k = f"sp_tiny_p{p}"
        out[k + "_val"], out[k + "_w"], out[k + "_cg"] = vals, w, cg
        out[k + "_c"] = r.spmm_fwd(fp, fi, vals, w, F, p)
        out[k + "_wg"] = r.spmm_bwd(fp, fi, vals, cg, F, p)
    out["sp_tiny_indptr"], out["sp_tiny_indices"] = fp, fi
    out["sp_tiny_F"] = np.int32(F)
    # cross entropy: unlabelled rows, tie rows, large logits
    n, c = 50, 7
    lg = (rng.standard_normal((n, c)) * 4).astype(np.float32)
    lg[3] = 2.0                       # all-tie row
    lg[4, :] = [50, -50, 0, 1, 2, 3, 88]
    tr = rng.integers(-1, c, n).astype(np.int32)
    tr[3] = 2
    tr[4] = 6
    tr[10:20] = -1
    out["ce_logits"], out["ce_truth"] = lg, tr
    for training in (0, 1):
        loss, shifted, grad = r.xent_fwd(lg, tr, c, bool(training))
        out[f"ce_loss_t{training}"] = np.float32(loss)
        out[f"ce_shifted_t{training}"] = shifted
        if training:
            out["ce_grad"] = grad
    # relu / dropout (dropout consumes the RNG from a set state)
    x = rng.standard_normal(777).astype(np.float32)
    x[5] = 0.0
    x[6] = -0.0
    g = rng.standard_normal(777).astype(np.float32)
    y, gb = r.relu(x, g)
    out["relu_x"], out["relu_g"], out["relu_y"], out["relu_gb"] = x, g, y, gb
    for p in (0.5, 0.0, 0.9):
        r.rand_set_state(0x1234567, 0x89abcdef)
        y, gb = r.dropout(x, p, g)
        out[f"drop_p{p}_y"], out[f"drop_p{p}_gb"] = y, gb
    out["drop_x"], out["drop_g"] = x, g
    out["drop_state"] = np.array([0x1234567, 0x89abcdef], np.uint64)
    # rng stream + seeding + glorot
    r.rand_seed_time(42)
    out["rng_seed42_state"] = np.array(r.rand_get_state(), np.uint64)
    out["rng_seed42_first64"] = r.rand_stream(64)
    r.rand_seed_time(7)
    out["glorot_seed7_30x20"] = r.glorot(600, 30, 20)
    # adam: 10 steps, 32 elements, with and without decay
    w0 = rng.standard_normal(32).astype(np.float32)
    gs = rng.standard_normal((10, 32)).astype(np.float32)
    out["adam_w0"], out["adam_grads"] = w0, gs
    out["adam_w_decay"] = r.adam_steps(w0, gs, 1, 0.01, 5e-4)
    out["adam_w_nodecay"] = r.adam_steps(w0, gs, 0, 0.01, 5e-4)


def trace_cases(r, out):
    for name, hidden, epochs in (("cora-syn", 16, 100), ("citeseer-syn", 16, 30), ("pubmed-syn", 16, 20), ("tiny-syn", 16, 100)):
        ds = datagen.make_dataset(name)
        out[f"{name}_digest"] = np.array(ds_digest(ds))
        seeds = (1, 2, 3) if name in ("cora-syn", "tiny-syn") else (1,)
        for seed in seeds:
            for dropout in (0.5, 0.0):
                m = r.model(ds, seed_time=seed, hidden_dim=hidden, dropout=dropout, epochs=epochs)
                tr = np.zeros((epochs, 4), np.float32)
                for e in range(epochs):
                    tr[e, 0], tr[e, 1] = m.train_epoch()
                    tr[e, 2], tr[e, 3] = m.eval(2)
                test = np.array(m.eval(3), np.float32)
                k = f"{name}_s{seed}_d{dropout}"
                out[k + "_trace"] = tr
                out[k + "_test"] = test
                out[k + "_w1sum"] = np.float64(m.var(2).astype(np.float64).sum())
                out[k + "_w2sum"] = np.float64(m.var(5).astype(np.float64).sum())
                if name == "tiny-syn" and seed == 1:
                    out[k + "_w1"] = m.var(2)
                    out[k + "_w2"] = m.var(5)
                m.close()
                print(k, tr[-1], test)


def parser_cases(r, out):
    """text fixtures exercising the loader's edge cases"""
    cases = {
        # trailing line without '\n' is dropped in every file
        "noeol": ("1 2\n0\n0", "0 0:1.5 3:2\n1 1:0.25\n2 2:1", "1\n2\n3"),
        # isolated node (empty line), junk ends a row, blank feature row
        "ragged": ("1\n0 2 x 3\n\n1\n", "1 0:1 4:-2.5e-1\n0\n\n2 1:3 2:4\n", "1\n0\n2\n3\n"),
        "plain": ("1 2\n0 2\n0 1\n", "0 0:0.1 1:0.2\n1 2:0.3\n2 0:1e-3 2:7\n", "1\n2\n3\n"),
    }
    for cname, (g, s, sp) in cases.items():
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, "data"))
            for ext, txt in ((".graph", g), (".svmlight", s), (".split", sp)):
                with open(os.path.join(td, "data", cname + ext), "w") as f:
                    f.write(txt)
            ds = r.parse(td, cname)
        out[f"parse_{cname}_graph_txt"] = np.array(g)
        out[f"parse_{cname}_svm_txt"] = np.array(s)
        out[f"parse_{cname}_split_txt"] = np.array(sp)
        for k in ("g_indptr", "g_indices", "f_indptr", "f_indices", "f_val", "split", "label"):
            out[f"parse_{cname}_{k}"] = ds[k]
        out[f"parse_{cname}_dims"] = np.array([ds["num_nodes"], ds["input_dim"], ds["output_dim"]], np.int32)
    # text round trip of a generated dataset
    ds = datagen.make_dataset("tiny-syn")
    with tempfile.TemporaryDirectory() as td:
        datagen.write_text(ds, os.path.join(td, "data"), "tiny-syn")
        back = r.parse(td, "tiny-syn")
    for k in ("g_indptr", "g_indices", "f_indptr", "f_indices", "f_val", "split", "label"):
        assert np.array_equal(back[k], ds[k]), k
    assert (back["num_nodes"], back["input_dim"], back["output_dim"]) == (ds["num_nodes"], ds["input_dim"], ds["output_dim"])


if __name__ == "__main__":
    assert Ref.available(), "build oracle/_ref first (make -C oracle)"
    r = Ref()
    mods, traces, parse = {}, {}, {}
    module_cases(r, mods)
    trace_cases(r, traces)
    parser_cases(r, parse)
    np.savez_compressed(os.path.join(HERE, "modules.npz"), **mods)
    np.savez_compressed(os.path.join(HERE, "traces.npz"), **traces)
    np.savez_compressed(os.path.join(HERE, "parser.npz"), **parse)
    for f in ("modules.npz", "traces.npz", "parser.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
