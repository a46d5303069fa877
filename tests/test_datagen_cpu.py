"""CPU tests of the synthetic-data helpers that need no GPU: the threaded R-MAT generator of libgcnhost
(host/rmat.cpp) — layout the reference's loader would produce (src/common/parser.cpp:20-46), symmetry,
determinism whatever the thread count."""
import os
import subprocess
import sys

import numpy as np

from cuda_gcn_amd import datagen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rmat_graph_layout_and_symmetry():
    scale = 12
    n = 1 << scale
    gp, gi = datagen.rmat_graph(scale)
    assert gp.dtype == np.int32 and gi.dtype == np.int32 and gp.size == n + 1 and gp[0] == 0 and gp[-1] == gi.size
    assert np.array_equal(gi[gp[:-1]], np.arange(n))                      # self loop first (parser.cpp:30-33)
    src = np.repeat(np.arange(n), np.diff(gp))
    first = np.zeros(gi.size, bool); first[gp[:-1]] = True
    s, d = src[~first], gi[~first].astype(np.int64)
    assert np.all(s != d)                                                 # no second self edge
    key = s * n + d
    assert np.all(np.diff(key) > 0)                                       # rows ascending, no duplicates
    assert np.array_equal(np.sort(d * n + s), key)                        # symmetric
    deg = np.diff(gp)
    assert 20 < deg.mean() < 33 and deg.max() > 20 * deg.mean()           # edge factor 16, skewed (R-MAT hubs)


def test_rmat_graph_is_deterministic_across_thread_counts():
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from cuda_gcn_amd import datagen; "
            "gp, gi = datagen.rmat_graph(13, 8, 5); print(int(gp.astype(np.int64).sum()), int((gi.astype(np.int64) * np.arange(gi.size)).sum() %% (1 << 61)))" % ROOT)
    outs = set()
    for t in ("1", "3", "8"):
        outs.add(subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GCN_HOST_THREADS=t), capture_output=True, text=True, check=True).stdout)
    assert len(outs) == 1, outs
    a, b = datagen.rmat_graph(10, 16, 1), datagen.rmat_graph(10, 16, 2)
    assert not np.array_equal(a[1][:2000], b[1][:2000])                    # the seed matters


def test_make_dataset_rmat_small():
    ds = datagen.make_dataset("rmat-10-32")
    assert ds["num_nodes"] == 1024 and ds["input_dim"] == 32 and ds["f_val"].size == 1024 * 32
    assert set(np.unique(ds["split"])) == {1, 2, 3}
