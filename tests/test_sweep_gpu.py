"""Seeded random sweep over shapes the fixed cases do not name: every op of the aggregation /
transform path against the CPU oracle with the sum-order bound of test_ops_gpu.close_mag.
Widths, strides, row counts and degree profiles are drawn per case (primes, ragged tails, padded
strides, isolated rows, hub rows crossing the split length)."""
import numpy as np
import pytest

from cuda_gcn_amd import datagen
from tests.test_ops_gpu import bf16_round, close_mag

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from cuda_gcn_amd.ops import Device
    d = Device(0)
    yield d
    d.close()


def random_graph(rng):
    n = int(rng.integers(2, 1500))
    m = int(rng.integers(0, 6 * n))
    if m:
        u, v = rng.integers(0, n, m), rng.integers(0, n, m)
        if rng.random() < 0.5:                                  # a hub: one node wired to many
            hub = int(rng.integers(0, n))
            k = int(rng.integers(1, min(n, 3000)))
            u = np.concatenate([u, np.full(k, hub)]); v = np.concatenate([v, rng.integers(0, n, k)])
        lo, hi = datagen._unique_undirected(u, v, n)
    else:
        lo = hi = np.zeros(0, np.int64)
    return datagen.csr_with_self_loops(lo, hi, n)


@pytest.mark.parametrize("case", range(24))
def test_graphsum_random(dev, oracle, case):
    rng = np.random.default_rng(1000 + case)
    gp, gi = random_graph(rng)
    n = gp.size - 1
    dim = int(rng.choice([1, 2, 3, 5, 7, 8, 13, 16, 31, 32, 41, 47, 64, 65, 96, 127, 128, 130, 200, 256, 300]))
    pad = int(rng.choice([0, 0, 1, 3, 4, 7]))
    ld = dim + pad
    x = (rng.standard_normal((n, dim)) * np.exp(rng.uniform(-3, 3, (n, 1)))).astype(np.float32)
    g = dev.graph(gp, gi, row_group=rng.integers(0, 5, n).astype(np.int32) if rng.random() < 0.5 else None)
    want, mag = oracle.graphsum(gp, gi, x, dim), oracle.graphsum(gp, gi, np.abs(x), dim)
    close_mag(dev.graphsum(g, x, ld_in=ld, ld_out=ld), want, mag)
    keep = rng.random(n) < rng.uniform(0.05, 0.95)
    xm = x * keep[:, None]
    close_mag(dev.graphsum(g, xm, ld_in=ld, ld_out=ld, row_nonzero=keep), oracle.graphsum(gp, gi, xm, dim), mag)
    # fused ReLU + dropout epilogue with injected decisions
    km = (rng.random((n, dim)) < 0.7).astype(np.uint8)
    got = dev.graphsum_relu_dropout(g, x, True, 0.3, keep_mask=km, ld=ld)
    base = np.asarray(dev.graphsum(g, x, ld_in=ld, ld_out=ld))
    assert np.allclose(got, np.maximum(base, 0) * km * np.float32(1 / (1 - np.float32(0.3))), rtol=1e-6, atol=0)
    # bf16 table of the same rows
    codes, xr = bf16_round(x)
    ldb = (dim + 7) // 8 * 8 + 8 * int(rng.integers(0, 3))
    tab = dev.to_bf16(x, ld_dst=ldb)
    assert np.array_equal(tab[:, :dim], codes)
    close_mag(dev.graphsum_bf16(g, tab, dim), oracle.graphsum(gp, gi, xr, dim), oracle.graphsum(gp, gi, np.abs(xr), dim))
    g.free()


@pytest.mark.parametrize("case", range(16))
def test_matmul_random(dev, oracle, case):
    rng = np.random.default_rng(2000 + case)
    m = int(rng.integers(1, 3000)); n = int(rng.choice([1, 3, 16, 17, 64, 100, 128, 200, 256])); p = int(rng.choice([1, 2, 7, 16, 41, 47, 48, 64, 100]))
    pad = int(rng.choice([0, 0, 1, 3]))
    a = rng.standard_normal((m, n)).astype(np.float32)
    a[rng.random((m, n)) < 0.4] = 0
    b = rng.standard_normal((n, p)).astype(np.float32)
    dc = rng.standard_normal((m, p)).astype(np.float32)
    lda, ldb = n + pad, p + pad
    close_mag(dev.matmul_fwd(a, b, lda=lda, ldb=ldb, ldc=ldb), oracle.matmul_fwd(a, b, m, n, p), oracle.matmul_fwd(np.abs(a), np.abs(b), m, n, p))
    da, db = dev.matmul_bwd(a, b, dc, lda=lda, ldb=ldb, lddc=ldb)
    oa, ob = oracle.matmul_bwd(a, b, dc, m, n, p)
    ma, mb = oracle.matmul_bwd(np.abs(a), np.abs(b), np.abs(dc), m, n, p)
    close_mag(da, oa, ma); close_mag(db, ob, mb)
    da2, _ = dev.matmul_bwd(a, b, dc, lda=lda, ldb=ldb, lddc=ldb, fused_scale=4.0)
    close_mag(da2, np.where(a > 0, oa * np.float32(4), 0), 4 * ma)
    bits = dev.pack_positive(a, ld=lda)
    assert np.array_equal(dev.matmul_bwd_da_bits(b, dc, bits, 4.0, ldb=ldb, lddc=ldb, ldda=lda), da2)


@pytest.mark.parametrize("case", range(12))
def test_spmm_random(dev, oracle, case):
    rng = np.random.default_rng(3000 + case)
    N = int(rng.integers(1, 900)); F = int(rng.choice([5, 33, 64, 100, 129, 256, 602])); p = int(rng.choice([3, 16, 41, 64, 128, 130]))
    dense = case % 2 == 0
    if dense:
        fp = (np.arange(N + 1) * F).astype(np.int32); fi = np.tile(np.arange(F, dtype=np.int32), N)
    else:
        cnt = rng.integers(0, min(F, 40) + 1, N)
        fp = np.zeros(N + 1, np.int32); np.cumsum(cnt, out=fp[1:])
        fi = np.concatenate([np.sort(rng.choice(F, c, replace=False)) for c in cnt] + [np.zeros(0, np.int64)]).astype(np.int32)
    vals = rng.standard_normal(fi.size).astype(np.float32)
    w = rng.standard_normal((F, p)).astype(np.float32)
    dout = rng.standard_normal((N, p)).astype(np.float32)
    f = dev.feat(fp, fi, vals, F)
    keep = (rng.random(fi.size) < 0.5).astype(np.uint8)
    for vd, kw in ((vals, {}), ((vals * np.where(keep != 0, np.float32(2), np.float32(0))).astype(np.float32), dict(p_drop=0.5, keep_mask=keep))):
        close_mag(dev.spmm_fwd(f, w, **kw), oracle.spmm_fwd(fp, fi, vd, w, p), oracle.spmm_fwd(fp, fi, np.abs(vd), np.abs(w), p))
        close_mag(dev.spmm_bwd(f, dout, **kw), oracle.spmm_bwd(fp, fi, vd, dout, F, p), oracle.spmm_bwd(fp, fi, np.abs(vd), np.abs(dout), F, p))
    f.free()
