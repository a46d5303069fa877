"""Edge cases of the training path on the GPU against the CPU oracle: odd widths
(scalar-load kernels), many classes, wide hidden layers, isolated nodes, a single
labelled node per split, heavy dropout."""
import numpy as np
import pytest

from cuda_gcn_amd import datagen
from tests.test_model_gpu import check_trace, oracle_trace

pytestmark = pytest.mark.gpu


def synth(N, F, C, M, nnz_row, seed, dense=False, isolated=0):
    rng = np.random.default_rng(seed)
    lo, hi = datagen._sample_edges(rng, N - isolated, M)
    gp, gi = datagen.csr_with_self_loops(lo, hi, N)          # the last `isolated` nodes keep only the self loop
    label = rng.integers(0, C, N).astype(np.int32)
    label[:C] = np.arange(C)
    if dense:
        x = rng.standard_normal((N, F)).astype(np.float32)
        fp = (np.arange(N + 1) * F).astype(np.int32)
        fi = np.tile(np.arange(F, dtype=np.int32), N)
        fv = x.reshape(-1)
    else:
        cols = np.sort(np.stack([rng.choice(F, nnz_row, replace=False) for _ in range(N)]), axis=1)
        cols[0, -1] = F - 1
        fp = (np.arange(N + 1) * nnz_row).astype(np.int32)
        fi = cols.reshape(-1).astype(np.int32)
        fv = rng.standard_normal(N * nnz_row).astype(np.float32)
    split = rng.integers(1, 4, N).astype(np.int32)
    return dict(num_nodes=N, input_dim=F, output_dim=C, g_indptr=gp, g_indices=gi, f_indptr=fp, f_indices=fi,
                f_val=fv, split=split, label=label)


@pytest.mark.parametrize("N,F,C,hidden,dense,isolated", [
    (300, 40, 3, 10, False, 0),       # hidden % 4 != 0: every matrix takes the padded-ld path
    (257, 33, 5, 6, True, 7),         # dense X at odd sizes (16x16x4 MFMA path), isolated nodes
    (500, 64, 100, 64, False, 0),     # 100 classes: two register chunks in the loss kernel
    (400, 50, 41, 256, False, 3),     # wide hidden layer: 8 XCD slices, LDS-resident 256-row operand
    (130, 70, 7, 128, True, 0),       # dense X with hidden 128 (128x128 MFMA tiles) on a tiny graph
])
def test_odd_shapes_vs_oracle(oracle, N, F, C, hidden, dense, isolated):
    from cuda_gcn_amd.model import HipGCNModel, HOST_MASKS
    ds = synth(N, F, C, 4 * N, 5, seed=N + hidden, dense=dense, isolated=isolated)
    for dropout, flags in ((0.0, 0), (0.5, HOST_MASKS)):
        want, _, om = oracle_trace(oracle, ds, 2, 12, hidden_dim=hidden, dropout=dropout)
        m = HipGCNModel(ds, seed=2, flags=flags, hidden_dim=hidden, dropout=dropout, epochs=12)
        got = np.array([m.train_epoch() + m.eval(2) for _ in range(12)], np.float32)
        check_trace(got, want, ds)
        m.close(); om.close()


def test_single_labelled_node_and_heavy_dropout(oracle):
    from cuda_gcn_amd.model import HipGCNModel, HOST_MASKS
    ds = synth(120, 30, 4, 300, 4, seed=9)
    ds["split"][:] = 0
    ds["split"][5] = 1
    ds["split"][6] = 2
    ds["split"][7] = 3
    want, wtest, om = oracle_trace(oracle, ds, 3, 10, hidden_dim=16, dropout=0.9)
    m = HipGCNModel(ds, seed=3, flags=HOST_MASKS, hidden_dim=16, dropout=0.9, epochs=10)
    got = np.array([m.train_epoch() + m.eval(2) for _ in range(10)], np.float32)
    assert np.abs(got[:, [0, 2]] - want[:, [0, 2]]).max() <= 2e-3
    assert np.array_equal(got[:, [1, 3]], want[:, [1, 3]])          # 0 or 1 with a single row
    m.close(); om.close()


def test_empty_split_gives_nan_like_reference(oracle):
    """count == 0 -> 0/0 = NaN loss, as the reference (module.cpp:154)"""
    from cuda_gcn_amd.model import HipGCNModel
    ds = synth(60, 20, 3, 100, 3, seed=1)
    ds["split"][ds["split"] == 2] = 0
    m = HipGCNModel(ds, seed=1, hidden_dim=8, dropout=0.0, epochs=2)
    om = oracle.model(ds, seed_time=1, hidden_dim=8, dropout=0.0)
    a, b = m.train_epoch(), om.train_epoch()
    assert abs(a[0] - b[0]) <= 2e-4
    va, vb = m.eval(2), om.eval(2)
    assert np.isnan(va[0]) and np.isnan(vb[0])
    m.close(); om.close()


def test_c_abi_rejects_bad_arguments():
    """the C-ABI never faults on bad operands: it returns -1 ("gcnhip: invalid argument") and the
    Python front end raises; nothing is leaked into the context (the next call works)"""
    from cuda_gcn_amd.ops import Device, GcnHipError
    dev = Device(0)
    gp = np.array([0, 2, 3], np.int32)
    with pytest.raises(GcnHipError):
        dev.graph(gp, np.array([0, 5, 1], np.int32))                  # column 5 of a 2-node graph
    with pytest.raises(GcnHipError):
        dev.graph(gp, np.array([0, -1, 1], np.int32))
    with pytest.raises(GcnHipError):
        dev.graph(np.array([0, 3, 2], np.int32), np.array([0, 1, 1], np.int32))   # indptr not monotone
    with pytest.raises(GcnHipError):
        dev.feat(np.array([0, 1, 2], np.int32), np.array([0, 9], np.int32), np.ones(2, np.float32), 4)
    g = dev.graph(gp, np.array([0, 1, 1], np.int32))
    x = np.ones((2, 300), np.float32)
    with pytest.raises(GcnHipError):
        dev.matmul_fwd(np.ones((3, 800), np.float32), np.ones((800, 40), np.float32))   # inner dim above the LDS panel limit
    out = dev.graphsum(g, x)
    assert np.isfinite(out).all()
    # round-2 entry points: NULL operands, split ranges outside the plan, a finish without parts
    import ctypes as C
    lib = dev.lib
    h = C.c_void_p()
    assert lib.gcnhip_graph_create_restricted(dev.ctx, C.byref(h), None, None) == -1
    assert lib.gcnhip_graph_create_restricted(dev.ctx, C.byref(h), g.h, None) == -1
    n, F, p = 300, 128, 128
    xs = np.ones((n, F), np.float32)
    f = dev.feat((np.arange(n + 1) * F).astype(np.int32), np.tile(np.arange(F, dtype=np.int32), n), xs.reshape(-1), F)
    rps, ns = dev.spmm_bwd_plan(f, p)
    assert ns >= 1 and rps >= 32
    d = dev.buf(np.ones((n, p), np.float32)); dw = dev.buf(np.zeros((F, p), np.float32))
    part = lambda s0, s1: lib.gcnhip_spmm_bwd_part(dev.ctx, f.h, f.values_ptr, d.ptr, p, p, 0.0, 0, None, 0, None, s0, s1, 0)
    assert part(0, ns + 1) == -1 and part(-1, 1) == -1 and part(2, 1) == -1
    assert lib.gcnhip_spmm_bwd_part(dev.ctx, f.h, f.values_ptr, d.ptr, p, 41, 0.0, 0, None, 0, None, 0, 1, 0) == -1   # no split-K plan at 41 columns
    assert part(0, ns) == 0 and lib.gcnhip_spmm_bwd_finish(dev.ctx, f.h, dw.ptr, p, p) == 0
    assert np.allclose(dw.download(), n)                      # X^T . 1 with X = 1: every entry is the row count
    f.free(); g.free()
    dev.close()


def test_wide_aggregation_on_split_rows_needs_reserve_width(oracle):
    """an adjacency object with split (hub) rows sizes its segment scratch for 256 columns; a wider call is refused with a
    message that names gcnhip_graph_reserve_width, works after the reservation, and a restricted child made
    AFTER the reservation inherits the width"""
    import ctypes as C
    from cuda_gcn_amd.ops import Device, _ck
    from tests.test_ops_gpu import hub_graph, close_mag
    gp, gi = hub_graph(3000, 2600, 5)
    n = gp.size - 1
    dim = 320
    dev = Device(0)
    lib = dev.lib
    g = dev.graph(gp, gi)
    x = np.random.default_rng(1).standard_normal((n, dim)).astype(np.float32)
    xin, out = dev.buf(x), dev.buf(np.zeros((n, dim), np.float32))
    rc = lib.gcnhip_graphsum(dev.ctx, g.h, xin.ptr, dim, out.ptr, dim, dim)
    assert rc == -1 and b"gcnhip_graph_reserve_width" in lib.gcnhip_last_error()
    _ck(lib, lib.gcnhip_graph_reserve_width(dev.ctx, g.h, dim), "reserve_width")
    _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, xin.ptr, dim, out.ptr, dim, dim), "graphsum")
    want = oracle.graphsum(gp, gi, x, dim)
    mag = oracle.graphsum(gp, gi, np.abs(x), dim)
    close_mag(out.download(), want, mag)
    # restricted child created after the reservation: same width
    bits = np.full(n // 32 + 2, 0xFFFFFFFF, np.uint32)
    child = C.c_void_p()
    _ck(lib, lib.gcnhip_graph_create_restricted(dev.ctx, C.byref(child), g.h, bits.ctypes.data), "create_restricted")
    out2 = dev.buf(np.zeros((n, dim), np.float32))
    _ck(lib, lib.gcnhip_graphsum(dev.ctx, child, xin.ptr, dim, out2.ptr, dim, dim), "graphsum on the restricted object")
    close_mag(out2.download(), want, mag)
    # a subset registered on one object is refused on another
    rs = C.c_void_p()
    _ck(lib, lib.gcnhip_graph_add_rowset(dev.ctx, g.h, bits.ctypes.data, C.byref(rs)), "add_rowset")
    rc = lib.gcnhip_graphsum_rowset(dev.ctx, child, rs, xin.ptr, dim, out2.ptr, dim, dim, None)
    assert rc == -1 and b"another adjacency object" in lib.gcnhip_last_error()
    lib.gcnhip_graph_destroy(dev.ctx, child)
    g.free()
    dev.close()


@pytest.mark.parametrize("dim", [128, 41, 256])
def test_in_kernel_segment_sum_equals_the_two_launch_form(oracle, dim, experiments):
    """split (hub) rows: the wave that finishes a row's last segment adds the partials itself (graphsum.hip, Guideline 16
    hand-off; opt-in, context option gs_fold).  Same bits as the default graphsum_finalize launch, launch after launch on CHANGING inputs —
    a partial served from a stale cache line of the previous launch would show here — and with the ReLU+dropout+bits
    epilogue on the same rows"""
    import os
    from cuda_gcn_amd.ops import Device, _ck
    from tests.test_ops_gpu import close_mag
    rng = np.random.default_rng(dim)
    n = 6000
    lo = np.concatenate([np.repeat(np.arange(4), [5200, 3100, 2049, 1025]), rng.integers(4, n, 9000)])
    hi = np.concatenate([rng.choice(np.arange(4, n), 5200, False), rng.choice(np.arange(4, n), 3100, False),
                         rng.choice(np.arange(4, n), 2049, False), rng.choice(np.arange(4, n), 1025, False), rng.integers(4, n, 9000)])
    a, b = datagen._unique_undirected(lo, hi, n)
    gp, gi = datagen.csr_with_self_loops(a, b, n)
    dev = Device(0)
    lib = dev.lib
    g = dev.graph(gp, gi)
    ld = (dim + 15) // 16 * 16
    xin, out = dev.buf(np.zeros((n, ld), np.float32)), dev.buf(np.zeros((n, ld), np.float32))
    try:
        for it in range(6):
            x = np.zeros((n, ld), np.float32)
            x[:, :dim] = rng.standard_normal((n, dim)).astype(np.float32)
            xin.upload(x)
            got = []
            for two_launch in (False, True, False):
                dev.set_option("gs_fold", 0 if two_launch else 1)
                out.upload(np.full((n, ld), 7.0, np.float32))
                _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, xin.ptr, ld, out.ptr, ld, dim), "graphsum")
                got.append(out.download()[:, :dim].copy())
            assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
            if dim % 32 == 0:                                  # the ReLU + dropout + mask-bits epilogue on the same rows
                wpr = dim // 32
                bits = dev.buf(np.zeros((n, wpr), np.uint32))
                ep = dev.buf(np.array([it + 1], np.uint32))
                eb = []
                for two_launch in (False, True):
                    dev.set_option("gs_fold", 0 if two_launch else 1)
                    out.upload(np.full((n, ld), 7.0, np.float32)); bits.upload(np.zeros((n, wpr), np.uint32))
                    _ck(lib, lib.gcnhip_graphsum_relu_dropout_bits(dev.ctx, g.h, xin.ptr, ld, out.ptr, ld, dim, 1, 0.5, 99, ep.ptr, 0, None,
                                                                   bits.ptr, wpr), "graphsum_relu_dropout_bits")
                    eb.append((out.download().copy(), bits.download().copy()))
                assert np.array_equal(eb[0][0], eb[1][0]) and np.array_equal(eb[0][1], eb[1][1])
                unpacked = ((eb[0][1][:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, dim).astype(bool)
                assert np.array_equal(unpacked, eb[0][0][:, :dim] > 0)
            if it == 0:
                close_mag(got[0], oracle.graphsum(gp, gi, x[:, :dim].copy(), dim), oracle.graphsum(gp, gi, np.abs(x[:, :dim]).copy(), dim))
    finally:
        dev.set_option("gs_fold", 0)
    g.free()
    dev.close()
