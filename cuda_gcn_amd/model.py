"""Python handle on the C++ host driver (libgcnhost.so: HipGCN and its Hip*
modules).  Mirrors the reference's GCN class: construct from params + data,
then train_epoch() / eval(split) / run() (src/seq/gcn.h:24-44).

All compute is in the HIP kernels behind include/gcnhip.h; this file only
marshals numpy arrays.  No GPU or no built library -> an exception.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

MODULAR, HOST_MASKS, TIMERS, NO_GRAPH, EVAL_LANE, NO_EVAL_LANE, NO_REPLICATE_L1, REPLICATE_L1, GATHER_DH1, NO_ROW_GROUPS, NULL_COMM, BF16_TABLES, ALL_ROWS, NO_AGG_FIRST_EVAL, EXCHANGE_ALLGATHER, EXCHANGE_HALO, PACKED_DH1, MASKED_BWD, BWD_PIPELINE, NO_LABEL_HINT, OVERLAP_EXCHANGE, STRUCTURE_PARTITION, ID_PARTITION, SYNC_EPOCHS, EDGE_COEF = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304, 8388608, 16777216
TIMER_NAMES = ["train", "test", "matmul_fw", "matmul_bw", "spmatmul_fw", "spmatmul_bw", "graphsum_fw", "graphsum_bw",
               "loss_fw", "relu_fw", "relu_bw", "dropout_fw", "dropout_bw", "adam", "comm", "graphsum_wide"]


class GcnHostError(RuntimeError):
    pass


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _ck(lib, rc, what):
    if rc != 0:
        raise GcnHostError(f"{what}: error {rc}: {lib.gcnhost_last_error().decode()}")


def default_params(**kw):
    lib = _lib.gcnhost()
    p = lib.gcnhost_params_default()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def nccl_unique_id() -> bytes:
    lib = _lib.gcnhost()
    buf = C.create_string_buffer(128)
    _ck(lib, lib.gcnhost_nccl_unique_id(buf), "gcnhost_nccl_unique_id")
    return buf.raw


class HipGCNModel:
    """ds: dict with num_nodes, input_dim, output_dim, g_indptr, g_indices, f_indptr, f_indices (or None
    for a dense X), f_val, split, label (the reference's GCNData)"""

    def __init__(self, ds, seed=0, device=0, flags=0, rank=0, world=1, nccl_id: bytes | None = None,
                 host_allgather=None, host_allreduce=None, **hyper):
        self.lib = lib = _lib.gcnhost()
        p = default_params(num_nodes=ds["num_nodes"], input_dim=ds["input_dim"], output_dim=ds["output_dim"], **hyper)
        self.params = p
        self._keep = [_i32(ds["g_indptr"]), _i32(ds["g_indices"]), _i32(ds["f_indptr"]),
                      _i32(ds["f_indices"]) if ds.get("f_indices") is not None else None,
                      np.ascontiguousarray(ds["f_val"], np.float32), _i32(ds["split"]), _i32(ds["label"])]
        k = self._keep
        self._ag = _lib.ALLGATHER_FN(host_allgather) if host_allgather else C.cast(None, _lib.ALLGATHER_FN)
        self._ar = _lib.ALLREDUCE_FN(host_allreduce) if host_allreduce else C.cast(None, _lib.ALLREDUCE_FN)
        h = C.c_void_p()
        rc = lib.gcnhost_model_create(C.byref(h), C.byref(p), k[0].ctypes.data, k[1].ctypes.data, k[2].ctypes.data,
                                      k[3].ctypes.data if k[3] is not None else None, k[4].ctypes.data,
                                      k[5].ctypes.data, k[6].ctypes.data, int(seed), int(device), int(flags),
                                      int(rank), int(world), nccl_id, self._ag, self._ar, None)
        _ck(lib, rc, "gcnhost_model_create")
        self.h = h
        self._keep = None           # the C++ side copied everything it needs

    def train_epoch(self):
        a, b = C.c_float(), C.c_float()
        _ck(self.lib, self.lib.gcnhost_model_train_epoch(self.h, C.byref(a), C.byref(b)), "train_epoch")
        return a.value, b.value

    def eval(self, split):
        a, b = C.c_float(), C.c_float()
        _ck(self.lib, self.lib.gcnhost_model_eval(self.h, split, C.byref(a), C.byref(b)), "eval")
        return a.value, b.value

    def run_epochs(self, n, want_trace=True):
        tr = np.zeros((n, 4), np.float32) if want_trace else None
        _ck(self.lib, self.lib.gcnhost_model_run_epochs(self.h, n, tr.ctypes.data if tr is not None else None), "run_epochs")
        return tr

    def run(self):
        _ck(self.lib, self.lib.gcnhost_model_run(self.h), "run")

    def sync(self):
        _ck(self.lib, self.lib.gcnhost_model_sync(self.h), "sync")

    def info(self):
        r, w, s, n = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        e = C.c_int64()
        _ck(self.lib, self.lib.gcnhost_model_info(self.h, C.byref(r), C.byref(w), C.byref(s), C.byref(n), C.byref(e)), "info")
        return dict(rank=r.value, world=w.value, row_start=s.value, local_rows=n.value, local_edges=e.value)

    def row_scale(self):
        """(dinv = 1/sqrt(deg) of this rank's rows, factored?) — factored: var(1), var(3), var(4) are stored pre-multiplied by
        dinv of their row and their gradients accordingly (host/gcn.h); EDGE_COEF restores the reference's values"""
        n = self.info()["local_rows"]
        d = np.ones(n, np.float32)
        f = C.c_int()
        _ck(self.lib, self.lib.gcnhost_model_row_scale(self.h, d.ctypes.data, C.byref(f)), "row_scale")
        return d, bool(f.value)

    def var_reference(self, k, grad=False):
        """variable k as the REFERENCE stores it (gcn.cpp:21-54): the factored model's pre-multiplied rows divided back"""
        v = self.var(k, grad)
        d, factored = self.row_scale()
        if not factored or k in (2, 5):
            return v
        d = d[:, None].astype(np.float64)
        if not grad:
            return (v / d).astype(np.float32) if k in (1, 3, 4) else v            # dinv.H0, dinv.H1, dinv.Z0 ; Z as is
        # gradients: dZ' = dinv.dZ (6), dH1' = dinv.dH1 (3) ; T = dZ0/dinv (4), S = dH0/dinv (1)
        return (v / d).astype(np.float32) if k in (6, 3) else (v * d).astype(np.float32)

    def row_ids(self):
        """(node of the caller's dataset for every local row of this rank, whether the model renumbered the nodes)"""
        n = self.info()["local_rows"]
        ids = np.zeros(n, np.int32)
        ren = C.c_int()
        _ck(self.lib, self.lib.gcnhost_model_row_ids(self.h, ids.ctypes.data, C.byref(ren)), "row_ids")
        return ids, bool(ren.value)

    def exchange(self):
        h, t = C.c_int(), C.c_int()
        r, sn = C.c_int64(), C.c_int64()
        sh = C.c_double()
        _ck(self.lib, self.lib.gcnhost_model_exchange(self.h, C.byref(h), C.byref(r), C.byref(sn), C.byref(t), C.byref(sh)), "exchange")
        return dict(mode="halo" if h.value else "allgather", recv_rows=r.value, send_rows=sn.value, table_rows=t.value, halo_share=sh.value)

    def var(self, k, grad=False):
        r, c = C.c_int(), C.c_int()
        _ck(self.lib, self.lib.gcnhost_model_get_var(self.h, k, int(grad), None, C.byref(r), C.byref(c)), "get_var")
        out = np.zeros((r.value, c.value), np.float32)
        _ck(self.lib, self.lib.gcnhost_model_get_var(self.h, k, int(grad), out.ctypes.data, C.byref(r), C.byref(c)), "get_var")
        return out

    def set_weights(self, w1, w2):
        w1, w2 = np.ascontiguousarray(w1, np.float32), np.ascontiguousarray(w2, np.float32)
        _ck(self.lib, self.lib.gcnhost_model_set_weights(self.h, w1.ctypes.data, w2.ctypes.data), "set_weights")

    def schedule(self):
        """row schedule of the aggregation picked at construction: 'degree', 'label-major', 'dealt-<G>' or
        'structure-major (<G> groups)' — groups found in the graph by modularity local moving"""
        m, g = C.c_int(), C.c_int()
        _ck(self.lib, self.lib.gcnhost_model_schedule(self.h, C.byref(m), C.byref(g)), "schedule")
        return {0: "degree", 1: "label-major", 2: f"dealt-{g.value}", 3: f"structure-major ({g.value} groups)"}[m.value]

    def transport(self):
        """(name of the layer that moves rows between ranks, ranks that layer counts — ncclCommCount under RCCL)"""
        n, buf = C.c_int(), C.create_string_buffer(32)
        _ck(self.lib, self.lib.gcnhost_model_transport(self.h, C.byref(n), buf), "transport")
        return buf.value.decode(), n.value

    def slice_floats(self):
        """column-slice width (floats) of the XCD-sliced hidden-width aggregation, timed at load: 64 or 32"""
        f = C.c_int()
        _ck(self.lib, self.lib.gcnhost_model_slice_floats(self.h, C.byref(f)), "slice_floats")
        return f.value

    def timer(self, name_or_id):
        i = TIMER_NAMES.index(name_or_id) if isinstance(name_or_id, str) else int(name_or_id)
        s, n = C.c_double(), C.c_long()
        _ck(self.lib, self.lib.gcnhost_model_timer(self.h, i, C.byref(s), C.byref(n)), "timer")
        return s.value, n.value

    def timers_reset(self):
        _ck(self.lib, self.lib.gcnhost_model_timers_reset(self.h), "timers_reset")

    def set_timers(self, on):
        _ck(self.lib, self.lib.gcnhost_model_set_timers(self.h, int(bool(on))), "set_timers")

    def close(self):
        if self.h:
            self.lib.gcnhost_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_dataset(root, name):
    """the reference's text formats (or a .gcnbin cache) through the C++ Parser"""
    lib = _lib.gcnhost()
    p = lib.gcnhost_params_default()
    h = C.c_void_p()
    if root and not root.endswith("/"):
        root += "/"
    import time
    t0 = time.perf_counter()
    _ck(lib, lib.gcnhost_dataset_load(C.byref(h), root.encode() if root else None, name.encode(), C.byref(p)), "dataset_load")
    load_s = time.perf_counter() - t0      # the C++ Parser alone (the numpy copies below are the Python front end's)
    ptrs = [C.c_void_p() for _ in range(7)]
    ns = [C.c_int64() for _ in range(4)]
    lib.gcnhost_dataset_arrays(h, C.byref(ptrs[0]), C.byref(ptrs[1]), C.byref(ns[0]), C.byref(ptrs[2]), C.byref(ptrs[3]),
                               C.byref(ptrs[4]), C.byref(ns[1]), C.byref(ptrs[5]), C.byref(ns[2]), C.byref(ptrs[6]), C.byref(ns[3]))

    def arr(ptr, n, t):
        if n == 0:
            return np.zeros(0, t)
        ct = C.c_float if t == np.float32 else C.c_int
        return np.frombuffer((ct * n).from_address(ptr.value), dtype=t).copy()
    N = p.num_nodes
    ds = dict(name=name, num_nodes=N, input_dim=p.input_dim, output_dim=p.output_dim,
              g_indptr=arr(ptrs[0], N + 1, np.int32), g_indices=arr(ptrs[1], ns[0].value, np.int32),
              f_indptr=arr(ptrs[2], ns[3].value + 1, np.int32), f_indices=arr(ptrs[3], ns[1].value, np.int32),
              f_val=arr(ptrs[4], ns[1].value, np.float32), split=arr(ptrs[5], ns[2].value, np.int32),
              label=arr(ptrs[6], ns[3].value, np.int32))
    ds["_handle"] = (lib, h, p)
    ds["_load_s"] = load_s
    return ds


def save_binary(ds, path):
    lib, h, p = ds["_handle"]
    if lib.gcnhost_dataset_save_binary(h, C.byref(p), path.encode()) != 0:
        raise GcnHostError("save_binary failed")


def partition(g_indptr, world):
    lib = _lib.gcnhost()
    gp = _i32(g_indptr)
    start = np.zeros(world + 1, np.int32)
    rm = C.c_int()
    rc = lib.gcnhost_partition(gp.ctypes.data, gp.size - 1, world, start.ctypes.data, C.byref(rm))
    if rc != 0:
        raise GcnHostError("partition failed")
    return start, rm.value


def local_graph(g_indptr, g_indices, world, rank):
    """(indptr, padded indices, col_deg, n_cols) of `rank`'s row block — host-only"""
    lib = _lib.gcnhost()
    gp, gi = _i32(g_indptr), _i32(g_indices)
    nl, nc, nnz = C.c_int(), C.c_int(), C.c_int64()
    rc = lib.gcnhost_local_graph(gp.ctypes.data, gi.ctypes.data, gp.size - 1, world, rank, None, None, None,
                                 C.byref(nl), C.byref(nc), C.byref(nnz))
    if rc != 0:
        raise GcnHostError("local_graph failed")
    ip, ix, cd = np.zeros(nl.value + 1, np.int32), np.zeros(nnz.value, np.int32), np.zeros(nc.value, np.int32)
    lib.gcnhost_local_graph(gp.ctypes.data, gi.ctypes.data, gp.size - 1, world, rank, ip.ctypes.data, ix.ctypes.data,
                            cd.ctypes.data, None, None, None)
    return ip, ix, cd, nc.value


def exchange_plan(g_indptr, g_indices, world, rank, mode=0):
    """the host-side plan of `rank` (host/partition.h) as a dict of numpy arrays — host only"""
    lib = _lib.gcnhost()
    gp, gi = _i32(g_indptr), _i32(g_indices)
    h = C.c_void_p()
    if lib.gcnhost_plan_create(C.byref(h), gp.ctypes.data, gi.ctypes.data, gp.size - 1, world, rank, mode) != 0:
        raise GcnHostError("plan_create failed")
    halo, nl, tr, oo, rm = (C.c_int() for _ in range(5))
    sh = C.c_double()
    nnz, nr, ns = C.c_int64(), C.c_int64(), C.c_int64()
    lib.gcnhost_plan_info(h, C.byref(halo), C.byref(nl), C.byref(tr), C.byref(oo), C.byref(rm), C.byref(sh), C.byref(nnz), C.byref(nr), C.byref(ns))
    ptr = [C.c_void_p() for _ in range(8)]
    lib.gcnhost_plan_arrays(h, *[C.byref(q) for q in ptr])

    def arr(q, n):
        return np.frombuffer((C.c_int * n).from_address(q.value), dtype=np.int32).copy() if n and q.value else np.zeros(0, np.int32)
    is_halo = bool(halo.value)
    out = dict(halo=is_halo, n_local=nl.value, table_rows=tr.value, own_offset=oo.value, rows_max=rm.value, halo_share=sh.value,
               recv_off=arr(ptr[0], world + 1 if is_halo else 0), recv_rows=arr(ptr[1], nr.value),
               send_off=arr(ptr[2], world + 1 if is_halo else 0), send_rows=arr(ptr[3], ns.value),
               table_global=arr(ptr[4], tr.value), indptr=arr(ptr[5], nl.value + 1), indices=arr(ptr[6], nnz.value),
               col_deg=arr(ptr[7], max(tr.value, 1)))
    lib.gcnhost_plan_free(h)
    return out


def choose_node_order(g_indptr, g_indices, world, force=False):
    """what a `world`-rank model does with the node ids of this graph (host only): dict(order, renumbered, ids_share, ids_recv_rows,
    new_share, new_recv_rows, allgather_rows)"""
    lib = _lib.gcnhost()
    gp, gi = _i32(g_indptr), _i32(g_indices)
    n = gp.size - 1
    order = np.zeros(n, np.int32)
    ren = C.c_int()
    s0, s1 = C.c_double(), C.c_double()
    r0, r1, ag = C.c_int64(), C.c_int64(), C.c_int64()
    rc = lib.gcnhost_choose_node_order(gp.ctypes.data, gi.ctypes.data, n, world, int(force), order.ctypes.data, C.byref(ren),
                                       C.byref(s0), C.byref(r0), C.byref(s1), C.byref(r1), C.byref(ag))
    if rc != 0:
        raise GcnHostError("choose_node_order failed")
    return dict(order=order, renumbered=bool(ren.value), ids_share=s0.value, ids_recv_rows=r0.value, new_share=s1.value,
                new_recv_rows=r1.value, allgather_rows=ag.value)


def glorot(size, in_size, out_size, seed, skip_draws=0):
    lib = _lib.gcnhost()
    w = np.zeros(size, np.float32)
    lib.gcnhost_glorot(w.ctypes.data, size, in_size, out_size, int(seed), int(skip_draws))
    return w


def host_masks(n, p, seed, skip_draws=0):
    lib = _lib.gcnhost()
    k = np.zeros(n, np.uint8)
    lib.gcnhost_host_masks(k.ctypes.data, n, p, int(seed), int(skip_draws))
    return k
