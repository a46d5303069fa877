"""dense first-layer GEMMs only (for counter profiling)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd.ops import Device, _ck
from tools.bench_ops import timeit
N, F, h = 232965, (int(sys.argv[1]) if len(sys.argv) > 1 else 602), 128
dev = Device(0); lib = dev.lib
rng = np.random.default_rng(0)
x = rng.standard_normal(N * F, dtype=np.float32)
f = dev.feat((np.arange(N + 1, dtype=np.int64) * F).astype(np.int32), None, x, F)
w1 = dev.buf(rng.standard_normal((F, h)).astype(np.float32)); h0 = dev.buf(rng.standard_normal((N, h), dtype=np.float32))
dw = dev.buf((F, h)); ep = dev.buf(np.zeros(1, np.uint32))
for pd in (0.0, 0.5):
    ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_spmm_fwd(dev.ctx, f.h, f.values_ptr, w1.ptr, h, h0.ptr, h, h, pd, 1, ep.ptr, 0, None), "f"), iters=10)
    print(f"fwd p={pd}: {ms:.3f} ms {2.0*N*F*h/ms/1e9:.1f} TF", flush=True)
    ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_spmm_bwd(dev.ctx, f.h, f.values_ptr, h0.ptr, h, dw.ptr, h, h, pd, 1, ep.ptr, 0, None), "b"), iters=10)
    print(f"bwd p={pd}: {ms:.3f} ms {2.0*N*F*h/ms/1e9:.1f} TF", flush=True)
