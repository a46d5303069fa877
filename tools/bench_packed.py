"""Micro-benchmark: the hidden layer's backward gather on reddit-syn shapes, dense dH1 (gcnhip_matmul_bwd_fused +
gcnhip_graphsum) against packed rows (gcnhip_matmul_bwd_packed + gcnhip_graphsum_packed).  HIP-event times."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck
from tools.bench_ops import timeit

name = sys.argv[1] if len(sys.argv) > 1 else "reddit-syn"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
density = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
ds = datagen.make_dataset(name)
gp, gi, lab = ds["g_indptr"], ds["g_indices"], ds["label"]
N = gp.size - 1
dev = Device(0); lib = dev.lib
g = dev.graph(gp, gi, row_group=lab)
rng = np.random.default_rng(0)
p, ldp = 41, 48
h = np.where(rng.random((N, n)) < density, 1.0, 0.0).astype(np.float32)
hb = dev.buf(h); bb = dev.buf(rng.standard_normal((n, ldp)).astype(np.float32)); dcb = dev.buf(rng.standard_normal((N, ldp)).astype(np.float32))
da = dev.buf((N, n)); db = dev.buf((n, ldp)); out = dev.buf((N, n))
pk = C.c_void_p(); _ck(lib, lib.gcnhip_rowpack_create(dev.ctx, C.byref(pk), N, n), "pack")
t_pd = timeit(dev, lambda: _ck(lib, lib.gcnhip_matmul_bwd_fused(dev.ctx, hb.ptr, n, bb.ptr, ldp, dcb.ptr, ldp, da.ptr, n, None, ldp, N, n, p, 2.0), "f"))
t_gd = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, da.ptr, n, out.ptr, n, n), "g"))
ref = out.download()
t_pp = timeit(dev, lambda: _ck(lib, lib.gcnhip_matmul_bwd_packed(dev.ctx, hb.ptr, n, bb.ptr, ldp, dcb.ptr, ldp, da.ptr, n, pk, None, ldp, N, n, p, 2.0), "p"))
t_gp = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum_packed(dev.ctx, g.h, pk, da.ptr, n, out.ptr, n), "gp"))
same = np.array_equal(out.download().view(np.uint32), ref.view(np.uint32))
print(f"{name} n={n} density={density}: producer dense {t_pd:.3f} ms, packed {t_pp:.3f} ms; gather dense {t_gd:.3f} ms, packed {t_gp:.3f} ms; identical={same}")
