"""Per-kernel micro-benchmark on reddit-syn shapes (run on the GPU box).
Times each op with HIP events on the context's stream, reports achieved
algorithmic GB/s (SURVEY §8d's B_gs / B_sp / Matmul bytes) or TFLOP/s."""
import ctypes as C
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck


def timeit(dev, fn, iters=20, warmup=3):
    lib = dev.lib
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.gcnhip_event_create(C.byref(e0)); lib.gcnhip_event_create(C.byref(e1))
    for _ in range(warmup):
        fn()
    dev.sync()
    lib.gcnhip_event_record(dev.ctx, e0)
    for _ in range(iters):
        fn()
    lib.gcnhip_event_record(dev.ctx, e1)
    ms = C.c_float()
    lib.gcnhip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit-syn"
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    t0 = time.time()
    if name.startswith("rmat-") and len(sys.argv) > 3 and sys.argv[3] == "graphsum":
        # graph only (the feature matrix of an R-MAT stress graph is synthetic anyway)
        scale = int(name.split("-")[1])
        gp, gi = datagen.rmat_graph(scale)
        N = 1 << scale
        deg = np.diff(gp)
        print(f"rmat scale {scale}: N={N} nnzA={gi.size} max_deg={deg.max()} built in {time.time() - t0:.1f}s", flush=True)
        rng = np.random.default_rng(0)
        variants = [("dealt-256", "dealt")] if not (len(sys.argv) > 4 and sys.argv[4] == "orders") else [("degree order", None)]
        if len(sys.argv) > 4 and sys.argv[4] == "orders":     # which task order suits a graph whose ids carry locality?
            variants += [("random 41 groups", np.random.default_rng(1).integers(0, 41, N).astype(np.int32))]
            variants += [(f"id >> {k}", (np.arange(N, dtype=np.int64) >> k).astype(np.int32)) for k in (16, 12)]
            rank = np.empty(N, np.int64); rank[np.argsort(-deg, kind="stable")] = np.arange(N)
            variants += [(f"degree rank dealt into {G} groups", (rank % G).astype(np.int32)) for G in (8, 41, 256, 2048)]
        for tag, groups in variants:
            dev = Device(0); lib = dev.lib
            g = dev.graph(gp, gi, row_group=None if isinstance(groups, str) else groups)
            if isinstance(groups, str):
                g.set_schedule(2, None, 256)              # the schedule HipGCN picks on R-MAT (and bench.py's HBM-regime leg uses)
            for dim in ((h,) if (len(sys.argv) > 4 and sys.argv[4] == "only_h") else (128, 256, 48)):
                x = dev.buf(rng.standard_normal((N, dim), dtype=np.float32)); o = dev.buf((N, dim))
                d_eff = 41 if dim == 48 else dim
                if int(os.environ.get("GS_SCALING", "0")):
                    from cuda_gcn_amd.ops import GsOpts
                    gso = GsOpts(); gso.scaling = int(os.environ["GS_SCALING"])
                    ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum_ex(dev.ctx, g.h, C.byref(gso), x.ptr, dim, o.ptr, dim, d_eff), "gs_ex"), iters=10)
                else:
                    ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, x.ptr, dim, o.ptr, dim, d_eff), "gs"), iters=10)
                bgs = 4 * (N + 1) + 4 * gi.size + 4 * gi.size * d_eff + 4 * N * d_eff
                print(f"[{tag}] graphsum d={d_eff} ld={dim} table={N * dim * 4 / 2**20:.0f} MiB: {ms:.3f} ms  {bgs / ms / 1e6:.0f} GB/s (B_gs model)", flush=True)
                x.free(); o.free()
            g.free(); dev.close()
        return
    ds = datagen.make_dataset(name)
    print("dataset", name, "built in %.1fs" % (time.time() - t0), flush=True)
    N, F, Cc = ds["num_nodes"], ds["input_dim"], ds["output_dim"]
    nnzA = ds["g_indices"].size
    dev = Device(0)
    lib = dev.lib
    g = dev.graph(ds["g_indptr"], ds["g_indices"], row_group=ds["label"])     # as HipGCN builds it
    f = dev.feat(ds["f_indptr"], ds["f_indices"], ds["f_val"], F)
    print("dense X:", f.dense, "nnzA", nnzA, flush=True)
    rng = np.random.default_rng(0)
    res = {}

    scaling = int(os.environ.get("GS_SCALING", "0"))        # 1-3: the factored operator (gcnhip_graphsum_ex), as HipGCN launches it by default
    from cuda_gcn_amd.ops import GsOpts
    gso = GsOpts()
    gso.scaling = scaling

    def gs(dim, ld):
        x = dev.buf(rng.standard_normal((N, ld)).astype(np.float32))
        o = dev.buf((N, ld))
        if scaling:
            ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum_ex(dev.ctx, g.h, C.byref(gso), x.ptr, ld, o.ptr, ld, dim), "gs_ex"))
        else:
            ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, x.ptr, ld, o.ptr, ld, dim), "gs"))
        bgs = 4 * (N + 1) + 4 * nnzA + 4 * nnzA * dim + 4 * N * dim
        res[f"graphsum_d{dim}_ld{ld}"] = dict(ms=ms, GBps=bgs / ms / 1e6)
        print(f"graphsum d={dim} ld={ld}: {ms:.3f} ms  {bgs / ms / 1e6:.0f} GB/s (B_gs model)", flush=True)
    def gs_bf16(dim, ld):
        x = dev.buf(rng.standard_normal((N, dim)).astype(np.float32))
        t = dev.buf(np.zeros((N, ld), np.uint16))
        o = dev.buf((N, (dim + 3) // 4 * 4))
        ms_c = timeit(dev, lambda: _ck(lib, lib.gcnhip_f32_to_bf16(dev.ctx, x.ptr, dim, t.ptr, ld, N, dim), "cv"))
        ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_graphsum_bf16(dev.ctx, g.h, t.ptr, ld, o.ptr, (dim + 3) // 4 * 4, dim, None, None, 0, 0, 0.0, 0, None, 0, None), "gsb"))
        res[f"graphsum_bf16_d{dim}_ld{ld}"] = dict(ms=ms, convert_ms=ms_c)
        print(f"graphsum bf16 table d={dim} ld={ld}: {ms:.3f} ms (+ f32->bf16 convert {ms_c:.3f} ms)", flush=True)
    only_gs = len(sys.argv) > 3 and sys.argv[3] == 'graphsum'
    if len(sys.argv) > 4 and sys.argv[4] == 'split41':      # would two narrower tables (32 + 9 columns) beat one 48-wide row?
        # round 5 (verdict r04 next 7): 48-float rows (192 B: always two 128-byte lines for 164 useful bytes) against 64-float
        # line-aligned rows and a 32 + 16-float split (one whole line + one half line that never straddles; two launches)
        gs(Cc, 48); gs(Cc, 64); gs(32, 32); gs(9, 12); gs(9, 16); gs(16, 16)
        r48, r64 = res[f"graphsum_d{Cc}_ld48"]["ms"], res[f"graphsum_d{Cc}_ld64"]["ms"]
        split = res["graphsum_d32_ld32"]["ms"] + res["graphsum_d9_ld16"]["ms"]
        print(json.dumps({"class_width_layouts_ms": {"rows_of_48_floats": r48, "rows_of_64_floats": r64, "split_32_plus_16": split,
                                                     "split_parts": [res["graphsum_d32_ld32"]["ms"], res["graphsum_d9_ld16"]["ms"]]}}))
        return
    if len(sys.argv) > 4 and sys.argv[4] == 'only_c64':     # PMC passes: 41 columns in 64-float rows
        gs(Cc, 64)
        return
    if len(sys.argv) > 4 and sys.argv[4] == 'only_split':   # PMC passes: the two launches of the 32 + 16 split (different kernel instantiations)
        gs(32, 32); gs(9, 16)
        return
    if len(sys.argv) > 4 and sys.argv[4] == 'bf16':
        gs(h, h); gs(Cc, 48)
        gs_bf16(h, h); gs_bf16(Cc, 64); gs_bf16(Cc, 48)
        return
    if len(sys.argv) > 4 and sys.argv[4] == 'only_c':       # PMC passes: the class-width launches alone (41 columns in 48-float rows)
        gs(Cc, 48)
        return
    gs(h, h)
    if len(sys.argv) > 4 and sys.argv[4] == 'only_h':       # PMC passes: the hidden-width launches alone under their kernel name
        return
    gs(Cc, (Cc + 3) // 4 * 4)
    gs(Cc, 48)
    gs(Cc, 64)
    gs(Cc, Cc)
    if len(sys.argv) > 4 and sys.argv[4] == 'strides':      # row stride vs L2 channel interleave
        for ld in (160, 224, 288, 192):
            gs(h, ld)
        for ld in (80, 96, 112, 160):
            gs(Cc, ld)
    if only_gs:
        print(json.dumps(res)); return

    w1 = dev.buf(rng.standard_normal((F, h)).astype(np.float32))
    h0 = dev.buf((N, h))
    ep = dev.buf(np.zeros(1, np.uint32))
    small = len(sys.argv) > 3 and sys.argv[3] == 'small'    # only the class-layer products and the loss
    for pd in (() if small else (0.0, 0.5)):
        ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_spmm_fwd(dev.ctx, f.h, f.values_ptr, w1.ptr, h, h0.ptr, h, h, pd, 1, ep.ptr, 0, None), "spf"), iters=10)
        fl = 2.0 * ds["f_indices"].size * h
        res[f"spmm_fwd_p{pd}"] = dict(ms=ms, TFLOPs=fl / ms / 1e9)
        print(f"spmm fwd drop={pd}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s", flush=True)
        dw = dev.buf((F, h))
        ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_spmm_bwd(dev.ctx, f.h, f.values_ptr, h0.ptr, h, dw.ptr, h, h, pd, 1, ep.ptr, 0, None), "spb"), iters=10)
        res[f"spmm_bwd_p{pd}"] = dict(ms=ms, TFLOPs=fl / ms / 1e9)
        print(f"spmm bwd drop={pd}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s", flush=True)

    ldc = (Cc + 15) // 16 * 16          # as HipGCN lays the class-width tables out
    w2 = dev.buf(rng.standard_normal((h, ldc)).astype(np.float32))
    z0 = dev.buf((N, ldc)); dz = dev.buf(rng.standard_normal((N, ldc)).astype(np.float32))
    dh = dev.buf((N, h)); dw2 = dev.buf((h, ldc))
    ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_matmul_fwd(dev.ctx, h0.ptr, h, w2.ptr, ldc, z0.ptr, ldc, N, h, Cc), "mm"))
    by = 4.0 * (N * h + h * Cc + N * Cc)
    res["matmul_fwd"] = dict(ms=ms, GBps=by / ms / 1e6)
    print(f"matmul fwd: {ms:.3f} ms {by / ms / 1e6:.0f} GB/s", flush=True)
    ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_matmul_bwd_fused(dev.ctx, h0.ptr, h, w2.ptr, ldc, dz.ptr, ldc, dh.ptr, h, dw2.ptr, ldc, N, h, Cc, 2.0), "mmb"))
    by = 4.0 * (2 * N * h + N * Cc + N * h)
    res["matmul_bwd_fused"] = dict(ms=ms, GBps=by / ms / 1e6)
    print(f"matmul bwd fused: {ms:.3f} ms {by / ms / 1e6:.0f} GB/s", flush=True)

    if h % 32 == 0:                     # the default path: the mask arrives as one bit per element (written by the aggregation)
        wpr = h // 32
        bits = dev.buf(rng.integers(0, 2**32, (N, wpr), dtype=np.uint64).astype(np.uint32))
        ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_matmul_bwd_fused_bits(dev.ctx, h0.ptr, h, w2.ptr, ldc, dz.ptr, ldc, dh.ptr, h, dw2.ptr, ldc, N, h, Cc, 2.0, bits.ptr, wpr), "mmbb"))
        by = 4.0 * (N * h + 2 * N * Cc + N * h + N * wpr)
        res["matmul_bwd_fused_bits"] = dict(ms=ms, GBps=by / ms / 1e6)
        print(f"matmul bwd fused, mask bits: {ms:.3f} ms {by / ms / 1e6:.0f} GB/s", flush=True)

    tr = dev.buf(np.where(ds["split"] == 1, ds["label"], -1).astype(np.int32))
    r4 = dev.buf(np.zeros(4, np.float32)); r2 = dev.buf(np.zeros(2, np.int32))
    cnt = int((ds["split"] == 1).sum())
    ms = timeit(dev, lambda: _ck(lib, lib.gcnhip_xent_fwd(dev.ctx, z0.ptr, ldc, dz.ptr, ldc, tr.ptr, N, Cc, 1, cnt, 0, r4.ptr, r2.ptr), "xe"))
    res["xent"] = dict(ms=ms, GBps=8.0 * N * Cc / ms / 1e6)
    print(f"xent train: {ms:.3f} ms", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
