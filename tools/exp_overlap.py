"""Experiment: does the MFMA-bound first-layer GEMM overlap with the gather-bound GraphSum when they are
enqueued on two streams?  (one GPU; the validation lane relies on this)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck

ds = datagen.make_dataset("reddit-syn")
N, F, h = ds["num_nodes"], ds["input_dim"], 128
A, B = Device(0), Device(0)                      # two contexts = two streams on the same GPU
lib = A.lib
rng = np.random.default_rng(0)
g = A.graph(ds["g_indptr"], ds["g_indices"], row_group=ds["label"])
x = A.buf(rng.standard_normal((N, h), dtype=np.float32)); o = A.buf((N, h))
f = B.feat(ds["f_indptr"], None, ds["f_val"], F)
w1 = B.buf(rng.standard_normal((F, h)).astype(np.float32)); h0 = B.buf((N, h)); ep = B.buf(np.zeros(1, np.uint32))

def gs(n):
    for _ in range(n):
        _ck(lib, lib.gcnhip_graphsum(A.ctx, g.h, x.ptr, h, o.ptr, h, h), "gs")
def gemm(n):
    for _ in range(n):
        _ck(lib, lib.gcnhip_spmm_fwd(B.ctx, f.h, f.values_ptr, w1.ptr, h, h0.ptr, h, h, 0.5, 1, ep.ptr, 0, None), "f")
def wall(fn):
    A.sync(); B.sync(); t0 = time.perf_counter(); fn(); A.sync(); B.sync(); return 1e3 * (time.perf_counter() - t0)

gs(3); gemm(3)
K = 20
t_gs = wall(lambda: gs(K)); t_ge = wall(lambda: gemm(K))
t_both = wall(lambda: (gs(K), gemm(K)))
def inter():
    for _ in range(K):
        gs(1); gemm(1)
t_inter = wall(inter)
print(f"{K} x GraphSum d=128: {t_gs:.2f} ms; {K} x GEMM: {t_ge:.2f} ms; sum {t_gs + t_ge:.2f} ms")
print(f"two streams, queued back to back: {t_both:.2f} ms; interleaved enqueue: {t_inter:.2f} ms")
