#!/usr/bin/env python3
"""The class layer's products at Reddit size (N = 232 965, hidden 128, 41 classes in rows of 48 floats), HIP events, back to back:
gcnhip_matmul_fwd and gcnhip_matmul_bwd_ex (mask bits + row factors, as HipGCN calls them), with the bf16x3 kernels of
csrc/class_bf16x3.h (option gemm_bf16x3 = 2) and with the f32-MFMA kernels of csrc/dense_kernels.h (0).

    python tools/bench_class.py [N] [iters] [noabl]   (run on the GPU box; counter passes: PMC_PROG="tools/bench_class.py 232965 10 noabl"
                                                       PMC_MATCH="class_,slab_reduce,rowstream,atb" tools/pmc_gemm.sh <out>)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd.ops import Device, _ck  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 232965
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    noabl = "noabl" in sys.argv[3:]                      # counter passes: only the product's kernels
    h, Cc, ldc = 128, 41, 48
    dev = Device(0)
    lib = dev.lib
    rng = np.random.default_rng(0)
    h1 = dev.buf((rng.standard_normal((N, h)) * (rng.random((N, h)) < 0.25)).astype(np.float32))
    w2 = dev.buf(rng.standard_normal((h, ldc)).astype(np.float32))
    z0 = dev.buf((N, ldc))
    dz = dev.buf((rng.standard_normal((N, ldc)) * 1e-3).astype(np.float32))
    dh = dev.buf((N, h)); dw2 = dev.buf((h, ldc))
    bits = dev.buf(rng.integers(0, 2**32, (N, 4), dtype=np.uint64).astype(np.uint32))
    rs = dev.buf((1.0 / rng.integers(1, 500, N)).astype(np.float32))
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.gcnhip_event_create(C.byref(e0)); lib.gcnhip_event_create(C.byref(e1))

    def timeit(fn):
        for _ in range(3):
            fn()
        dev.sync()
        lib.gcnhip_event_record(dev.ctx, e0)
        for _ in range(iters):
            fn()
        lib.gcnhip_event_record(dev.ctx, e1)
        ms = C.c_float()
        _ck(lib, lib.gcnhip_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
        return ms.value / iters
    res = {"N": N, "iters": iters}
    for mode, name in ((2, "bf16x3"), (0, "f32")):
        lib.gcnhip_ctx_set_option(dev.ctx, b"gemm_bf16x3", mode)
        f = timeit(lambda: _ck(lib, lib.gcnhip_matmul_fwd(dev.ctx, h1.ptr, h, w2.ptr, ldc, z0.ptr, ldc, N, h, Cc), "fwd"))
        b = timeit(lambda: _ck(lib, lib.gcnhip_matmul_bwd_ex(dev.ctx, h1.ptr, h, w2.ptr, ldc, dz.ptr, ldc, dh.ptr, h, dw2.ptr, ldc, N, h, Cc, 2.0,
                                                             bits.ptr, 4, rs.ptr), "bwd"))
        by_f, by_b = 4.0 * (N * h + N * Cc), 4.0 * (2 * N * h + N * Cc + N * 4)
        res[name] = {"fwd_ms": f, "bwd_ms": b, "fwd_GBps": by_f / f / 1e6, "bwd_GBps": by_b / b / 1e6}
        print(f"[bench_class] {name}: H1.W2 {f * 1e3:.1f} us ({by_f / f / 1e6:.0f} GB/s of {by_f / 1e6:.0f} MB), "
              f"dH1 + dW2 {b * 1e3:.1f} us ({by_b / b / 1e6:.0f} GB/s of {by_b / 1e6:.0f} MB)", flush=True)
    lib.gcnhip_ctx_set_option(dev.ctx, b"gemm_bf16x3", 2)
    if noabl:
        print(json.dumps(res)); return
    for abl, what in ((1, "no matrix work (loads, stores)"), (2, "no stores"), (3, "loads only")):
        lib.gcnhip_ctx_set_option(dev.ctx, b"cls_abl", abl)
        f = timeit(lambda: _ck(lib, lib.gcnhip_matmul_fwd(dev.ctx, h1.ptr, h, w2.ptr, ldc, z0.ptr, ldc, N, h, Cc), "fwd"))
        b = timeit(lambda: _ck(lib, lib.gcnhip_matmul_bwd_ex(dev.ctx, h1.ptr, h, w2.ptr, ldc, dz.ptr, ldc, dh.ptr, h, dw2.ptr, ldc, N, h, Cc, 2.0,
                                                             bits.ptr, 4, rs.ptr), "bwd"))
        res[f"ablation_{abl}"] = {"what": what, "fwd_ms": f, "bwd_ms": b}
        print(f"[bench_class] bf16x3, {what}: H1.W2 {f * 1e3:.1f} us, dH1 + dW2 {b * 1e3:.1f} us", flush=True)
    for rep in range(3):                                  # load order of the backward's row loads (rotated: the clock drifts)
        for abl, what in ((0, "dZ0 row, mask words, row factor"), (4, "mask words and row factor first")):
            lib.gcnhip_ctx_set_option(dev.ctx, b"cls_abl", abl)
            b = timeit(lambda: _ck(lib, lib.gcnhip_matmul_bwd_ex(dev.ctx, h1.ptr, h, w2.ptr, ldc, dz.ptr, ldc, dh.ptr, h, dw2.ptr, ldc, N, h, Cc, 2.0,
                                                                 bits.ptr, 4, rs.ptr), "bwd"))
            res.setdefault(f"bwd_load_order_{abl}", []).append(b)
            print(f"[bench_class] bf16x3 backward, {what}: {b * 1e3:.1f} us", flush=True)
    lib.gcnhip_ctx_set_option(dev.ctx, b"cls_abl", 0)
    for wgs in (1, 2, 3):
        lib.gcnhip_ctx_set_option(dev.ctx, b"cls_wgs", wgs)
        f = timeit(lambda: _ck(lib, lib.gcnhip_matmul_fwd(dev.ctx, h1.ptr, h, w2.ptr, ldc, z0.ptr, ldc, N, h, Cc), "fwd"))
        res[f"fwd_wgs_{wgs}"] = f
        print(f"[bench_class] bf16x3, {wgs} workgroups per CU: H1.W2 {f * 1e3:.1f} us", flush=True)
    lib.gcnhip_ctx_set_option(dev.ctx, b"cls_wgs", 0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
