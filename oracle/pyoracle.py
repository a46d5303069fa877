"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present,
for the reference's own objects (oracle/_ref/libref.so: built in the build container from the
reference's sources where they lie; the built library travels to the GPU box with the snapshot).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg — never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(ref: bool = True):
    """make liboracle.so / gcn-seq (+ _ref when /root/reference exists)"""
    subprocess.run(["make", "-s", "-C", HERE], check=True)


class _Params(C.Structure):
    _fields_ = [("num_nodes", C.c_int), ("input_dim", C.c_int), ("hidden_dim", C.c_int),
                ("output_dim", C.c_int), ("dropout", C.c_float), ("learning_rate", C.c_float),
                ("weight_decay", C.c_float), ("epochs", C.c_int), ("early_stopping", C.c_int)]


class _Data(C.Structure):
    _fields_ = [("g_indptr", C.c_void_p), ("g_indices", C.c_void_p), ("g_nnz", C.c_int),
                ("f_indptr", C.c_void_p), ("f_indices", C.c_void_p), ("f_val", C.c_void_p), ("f_nnz", C.c_int),
                ("split", C.c_void_p), ("label", C.c_void_p), ("n_split", C.c_int), ("n_label", C.c_int)]


class _AdamParams(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float)]


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def _i(a):
    return np.ascontiguousarray(a, np.int32)


_LIB = None


class Oracle:
    """numpy front end of gcn_oracle.h"""

    def __init__(self, path: str | None = None):
        path = path or os.environ.get("GCN_ORACLE_LIB") or os.path.join(HERE, "liboracle.so")   # GCN_ORACLE_LIB: the sanitizer build
        if not os.path.exists(path):
            build()
        global _LIB
        self.lib = L = _LIB = C.CDLL(path)      # _LIB: tests/conftest.py resets the checker's process-wide switches after every test
        L.or_rand_next.restype = C.c_uint32
        L.or_rand_set_state.argtypes = [C.c_uint64, C.c_uint64]
        L.or_rand_get_state.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.or_rand_seed_time.argtypes = [C.c_uint]
        L.or_params_default.restype = _Params
        L.or_adam_default.restype = _AdamParams
        L.or_gcn_create.restype = C.c_void_p
        L.or_gcn_create.argtypes = [_Params, C.POINTER(_Data), C.c_long]
        L.or_gcn_destroy.argtypes = [C.c_void_p]
        L.or_gcn_train_epoch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.or_gcn_eval.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.or_gcn_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.or_gcn_var_data.restype = C.POINTER(C.c_float)
        L.or_gcn_var_data.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.or_gcn_var_grad.restype = C.POINTER(C.c_float)
        L.or_gcn_var_grad.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.or_timer_total.restype = C.c_double
        L.or_timer_total.argtypes = [C.c_int]
        L.or_parse.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(_Params), C.POINTER(_Data)]

    # --- rng
    def rand_seed_time(self, t):
        self.lib.or_rand_seed_time(C.c_uint(t))

    def rand_set_state(self, s0, s1):
        self.lib.or_rand_set_state(s0, s1)

    def rand_get_state(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.or_rand_get_state(C.byref(a), C.byref(b))
        return a.value, b.value

    def rand_next(self):
        return int(self.lib.or_rand_next())

    def rand_stream(self, n):
        return np.array([self.lib.or_rand_next() for _ in range(n)], np.uint32)

    def glorot(self, size, in_size, out_size):
        w = np.empty(size, np.float32)
        self.lib.or_glorot(w.ctypes.data_as(C.c_void_p), size, in_size, out_size)
        return w

    # --- modules
    def matmul_fwd(self, a, b, m, n, p):
        a, b = _f(a), _f(b)
        c = np.empty(m * p, np.float32)
        self.lib.or_matmul_fwd(a.ctypes, b.ctypes, c.ctypes, m, n, p)
        return c.reshape(m, p)

    def matmul_bwd(self, a, b, cg, m, n, p):
        a, b, cg = _f(a), _f(b), _f(cg)
        ag = np.empty(m * n, np.float32)
        bg = np.empty(n * p, np.float32)
        self.lib.or_matmul_bwd(a.ctypes, b.ctypes, cg.ctypes, ag.ctypes, bg.ctypes, m, n, p)
        return ag.reshape(m, n), bg.reshape(n, p)

    def spmm_fwd(self, indptr, indices, val, b, p):
        indptr, indices, val, b = _i(indptr), _i(indices), _f(val), _f(b)
        n_rows = indptr.size - 1
        c = np.empty(n_rows * p, np.float32)
        self.lib.or_spmm_fwd(indptr.ctypes, indices.ctypes, n_rows, val.ctypes, b.ctypes, c.ctypes, p)
        return c.reshape(n_rows, p)

    def spmm_bwd(self, indptr, indices, val, cg, n, p):
        indptr, indices, val, cg = _i(indptr), _i(indices), _f(val), _f(cg)
        n_rows = indptr.size - 1
        bg = np.empty(n * p, np.float32)
        self.lib.or_spmm_bwd(indptr.ctypes, indices.ctypes, n_rows, val.ctypes, cg.ctypes, bg.ctypes, n, p)
        return bg.reshape(n, p)

    def graphsum(self, indptr, indices, x, dim):
        indptr, indices, x = _i(indptr), _i(indices), _f(x)
        n_rows = indptr.size - 1
        out = np.empty(n_rows * dim, np.float32)
        self.lib.or_graphsum(indptr.ctypes, indices.ctypes, n_rows, x.ctypes, out.ctypes, dim)
        return out.reshape(n_rows, dim)

    def set_wide_degree(self, on: bool):
        """64-bit degree product in GraphSum's coefficient (the documented divergence from module.cpp:92);
        process-wide, off by default.  Bit-identical to the int product wherever that is defined."""
        self.lib.or_set_wide_degree(int(bool(on)))

    def overflowing_edges(self, indptr, indices):
        """stored edges whose int degree product (module.cpp:92) is >= 2^31"""
        indptr, indices = _i(indptr), _i(indices)
        self.lib.or_graphsum_overflowing_edges.restype = C.c_long
        return int(self.lib.or_graphsum_overflowing_edges(indptr.ctypes, indices.ctypes, int(indptr.size - 1)))

    def graphsum_rows(self, indptr, indices, rows, x, dim):
        """rows `rows` of graphsum(indptr, indices, x) -> (out [len(rows), dim], n rows with an overflowing int degree product)"""
        indptr, indices, rows, x = _i(indptr), _i(indices), _i(rows), _f(x)
        out = np.empty(rows.size * dim, np.float32)
        self.lib.or_graphsum_rows.restype = C.c_int
        bad = self.lib.or_graphsum_rows(indptr.ctypes, indices.ctypes, rows.ctypes, int(rows.size), x.ctypes, out.ctypes, dim)
        return out.reshape(rows.size, dim), int(bad)

    def xent_fwd(self, logits, truth, num_classes, training=True):
        """returns (loss, shifted_logits, grad or None)"""
        logits = _f(logits).copy()
        truth = _i(truth)
        n_rows = truth.size
        grad = np.zeros(n_rows * num_classes, np.float32)
        loss = C.c_float()
        self.lib.or_xent_fwd(logits.ctypes, grad.ctypes, truth.ctypes, n_rows, num_classes,
                             int(training), C.byref(loss))
        return loss.value, logits.reshape(n_rows, num_classes), (grad.reshape(n_rows, num_classes) if training else None)

    def relu_fwd(self, x, training=True):
        x = _f(x).copy().reshape(-1)
        mask = np.zeros(x.size, np.uint8)
        self.lib.or_relu_fwd(x.ctypes, mask.ctypes, x.size, int(training))
        return x, mask

    def relu_bwd(self, grad, mask):
        grad = _f(grad).copy().reshape(-1)
        mask = np.ascontiguousarray(mask, np.uint8)
        self.lib.or_relu_bwd(grad.ctypes, mask.ctypes, grad.size)
        return grad

    def dropout_fwd(self, x, p, training=True, want_mask=True):
        x = _f(x).copy().reshape(-1)
        mask = np.zeros(x.size, np.int32)
        self.lib.or_dropout_fwd(x.ctypes, mask.ctypes if want_mask else None, x.size, C.c_float(p), int(training))
        return x, mask

    def dropout_bwd(self, grad, mask, p):
        grad = _f(grad).copy().reshape(-1)
        mask = _i(mask)
        self.lib.or_dropout_bwd(grad.ctypes, mask.ctypes, grad.size, C.c_float(p))
        return grad

    def adam_steps(self, w, grads, decay, lr, weight_decay):
        """grads: [k, n]; returns (w, m, v) after k steps"""
        w = _f(w).copy().reshape(-1)
        grads = _f(grads).reshape(-1, w.size)
        m = np.zeros_like(w)
        v = np.zeros_like(w)
        ap = self.lib.or_adam_default()
        ap.lr, ap.weight_decay = lr, weight_decay
        for s in range(grads.shape[0]):
            g = np.ascontiguousarray(grads[s])
            self.lib.or_adam_step_var(w.ctypes, g.ctypes, m.ctypes, v.ctypes, w.size, int(decay), s + 1, C.byref(ap))
        return w, m, v

    # --- model
    def params(self, ds=None, **kw):
        p = self.lib.or_params_default()
        if ds is not None:
            p.num_nodes, p.input_dim, p.output_dim = ds["num_nodes"], ds["input_dim"], ds["output_dim"]
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def parse(self, root, name):
        if not root.endswith("/"):
            root += "/"
        p = self.lib.or_params_default()
        d = _Data()
        rc = self.lib.or_parse(root.encode(), name.encode(), C.byref(p), C.byref(d))
        if rc != 0:
            return None

        def arr(ptr, n, t):
            if n == 0:
                return np.zeros(0, t)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int if t == np.int32 else C.c_float)), (n,)).astype(t).copy()
        N = p.num_nodes
        nrow_f = d.n_label
        ds = dict(name=name, num_nodes=N, input_dim=p.input_dim, output_dim=p.output_dim,
                  g_indptr=arr(d.g_indptr, N + 1, np.int32), g_indices=arr(d.g_indices, d.g_nnz, np.int32),
                  f_indptr=arr(d.f_indptr, nrow_f + 1, np.int32), f_indices=arr(d.f_indices, d.f_nnz, np.int32),
                  f_val=arr(d.f_val, d.f_nnz, np.float32),
                  split=arr(d.split, d.n_split, np.int32), label=arr(d.label, d.n_label, np.int32))
        self.lib.or_data_free(C.byref(d))
        return ds

    def model(self, ds, seed_time=0, **kw):
        return OracleModel(self, ds, seed_time, **kw)


class OracleModel:
    def __init__(self, orc: Oracle, ds, seed_time, **kw):
        self.o = orc
        self.keep = [_i(ds["g_indptr"]), _i(ds["g_indices"]), _i(ds["f_indptr"]), _i(ds["f_indices"]),
                     _f(ds["f_val"]), _i(ds["split"]), _i(ds["label"])]
        k = self.keep
        self.d = _Data(k[0].ctypes.data, k[1].ctypes.data, k[1].size, k[2].ctypes.data, k[3].ctypes.data,
                       k[4].ctypes.data, k[3].size, k[5].ctypes.data, k[6].ctypes.data, k[5].size, k[6].size)
        self.p = orc.params(ds, **kw)
        self.h = orc.lib.or_gcn_create(self.p, C.byref(self.d), seed_time)

    def train_epoch(self):
        a, b = C.c_float(), C.c_float()
        self.o.lib.or_gcn_train_epoch(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def eval(self, split):
        a, b = C.c_float(), C.c_float()
        self.o.lib.or_gcn_eval(self.h, split, C.byref(a), C.byref(b))
        return a.value, b.value

    def var(self, k, grad=False):
        n = C.c_int()
        fn = self.o.lib.or_gcn_var_grad if grad else self.o.lib.or_gcn_var_data
        ptr = fn(self.h, k, C.byref(n))
        if n.value == 0:
            return np.zeros(0, np.float32)
        return np.ctypeslib.as_array(ptr, (n.value,)).copy()

    def close(self):
        if self.h:
            self.o.lib.or_gcn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ref:
    """numpy front end of oracle/_ref/libref.so (the reference's own objects).
    Built in the build container only; present wherever the built file travelled (tests skip without it)."""

    PATH = os.path.join(HERE, "_ref", "libref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = L = C.CDLL(self.PATH)
        L.ref_rand_next.restype = C.c_uint32
        L.ref_rand_set_state.argtypes = [C.c_uint64, C.c_uint64]
        L.ref_rand_get_state.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ref_rand_seed_time.argtypes = [C.c_long]
        L.ref_gcn_create.restype = C.c_void_p
        L.ref_gcn_create.argtypes = [C.c_int] * 4 + [C.c_float] * 3 + [C.c_int] * 2 + [C.c_void_p] * 7 + [C.c_long]
        L.ref_gcn_destroy.argtypes = [C.c_void_p]
        L.ref_gcn_train_epoch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ref_gcn_eval.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ref_gcn_var_size.argtypes = [C.c_void_p, C.c_int]
        L.ref_gcn_var_data.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_gcn_var_grad.argtypes = [C.c_void_p, C.c_int, C.c_void_p]

    def rand_seed_time(self, t):
        self.lib.ref_rand_seed_time(t)

    def rand_set_state(self, s0, s1):
        self.lib.ref_rand_set_state(s0, s1)

    def rand_get_state(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.ref_rand_get_state(C.byref(a), C.byref(b))
        return a.value, b.value

    def rand_stream(self, n):
        return np.array([self.lib.ref_rand_next() for _ in range(n)], np.uint32)

    def glorot(self, size, in_size, out_size):
        w = np.empty(size, np.float32)
        self.lib.ref_glorot(w.ctypes, size, in_size, out_size)
        return w

    def matmul_fwd(self, a, b, m, n, p):
        a, b = _f(a), _f(b)
        c = np.empty(m * p, np.float32)
        self.lib.ref_matmul_fwd(a.ctypes, b.ctypes, c.ctypes, m, n, p)
        return c.reshape(m, p)

    def matmul_bwd(self, a, b, cg, m, n, p):
        a, b, cg = _f(a), _f(b), _f(cg)
        ag = np.empty(m * n, np.float32)
        bg = np.empty(n * p, np.float32)
        self.lib.ref_matmul_bwd(a.ctypes, b.ctypes, cg.ctypes, ag.ctypes, bg.ctypes, m, n, p)
        return ag.reshape(m, n), bg.reshape(n, p)

    def spmm_fwd(self, indptr, indices, val, b, n, p):
        indptr, indices, val, b = _i(indptr), _i(indices), _f(val), _f(b)
        n_rows = indptr.size - 1
        c = np.empty(n_rows * p, np.float32)
        self.lib.ref_spmm_fwd(indptr.ctypes, indices.ctypes, n_rows, val.ctypes, b.ctypes, c.ctypes, n, p)
        return c.reshape(n_rows, p)

    def spmm_bwd(self, indptr, indices, val, cg, n, p):
        indptr, indices, val, cg = _i(indptr), _i(indices), _f(val), _f(cg)
        n_rows = indptr.size - 1
        bg = np.empty(n * p, np.float32)
        self.lib.ref_spmm_bwd(indptr.ctypes, indices.ctypes, n_rows, val.ctypes, cg.ctypes, bg.ctypes, n, p)
        return bg.reshape(n, p)

    def graphsum(self, indptr, indices, x, dim, backward=False):
        indptr, indices, x = _i(indptr), _i(indices), _f(x)
        n_rows = indptr.size - 1
        out = np.empty(n_rows * dim, np.float32)
        fn = self.lib.ref_graphsum_bwd if backward else self.lib.ref_graphsum_fwd
        fn(indptr.ctypes, indices.ctypes, n_rows, x.ctypes, out.ctypes, dim)
        return out.reshape(n_rows, dim)

    def xent_fwd(self, logits, truth, num_classes, training=True):
        logits = _f(logits).copy()
        truth = _i(truth).copy()
        n_rows = truth.size
        grad = np.zeros(n_rows * num_classes, np.float32)
        loss = C.c_float()
        self.lib.ref_xent_fwd(logits.ctypes, grad.ctypes, truth.ctypes, n_rows, num_classes, int(training), C.byref(loss))
        return loss.value, logits.reshape(n_rows, num_classes), (grad.reshape(n_rows, num_classes) if training else None)

    def relu(self, x, grad=None, training=True):
        x = _f(x).copy().reshape(-1)
        g = _f(grad).copy().reshape(-1) if grad is not None else None
        self.lib.ref_relu(x.ctypes, g.ctypes if g is not None else None, x.size, int(training), int(g is not None))
        return x, g

    def dropout(self, x, p, grad=None, training=True, requires_grad=True):
        x = _f(x).copy().reshape(-1)
        g = _f(grad).copy().reshape(-1) if grad is not None else None
        self.lib.ref_dropout(x.ctypes, g.ctypes if g is not None else None, x.size, C.c_float(p), int(training),
                             int(requires_grad), int(g is not None))
        return x, g

    def adam_steps(self, w, grads, decay, lr, weight_decay):
        w = _f(w).copy().reshape(-1)
        grads = _f(grads).reshape(-1, w.size)
        self.lib.ref_adam_steps(w.ctypes, grads.ctypes, w.size, grads.shape[0], int(decay), C.c_float(lr), C.c_float(weight_decay))
        return w

    def parse(self, cwd, name):
        """parse <cwd>/data/<name>.* with the reference's Parser"""
        old = os.getcwd()
        os.chdir(cwd)
        try:
            out = [C.c_int() for _ in range(7)]
            rc = self.lib.ref_parse(name.encode(), *[C.byref(o) for o in out])
            if rc != 0:
                return None
            N, F, Cc, gnnz, fnnz, ns, nl = [o.value for o in out]
            gp = np.zeros(N + 1, np.int32); gi = np.zeros(gnnz, np.int32)
            fp = np.zeros(nl + 1, np.int32); fi = np.zeros(fnnz, np.int32); fv = np.zeros(fnnz, np.float32)
            sp = np.zeros(ns, np.int32); lb = np.zeros(nl, np.int32)
            self.lib.ref_parse_copy(gp.ctypes, gi.ctypes, fp.ctypes, fi.ctypes, fv.ctypes, sp.ctypes, lb.ctypes)
            return dict(name=name, num_nodes=N, input_dim=F, output_dim=Cc, g_indptr=gp, g_indices=gi,
                        f_indptr=fp, f_indices=fi, f_val=fv, split=sp, label=lb)
        finally:
            os.chdir(old)

    def model(self, ds, seed_time=0, hidden_dim=16, dropout=0.5, learning_rate=0.01, weight_decay=5e-4,
              epochs=100, early_stopping=0):
        return RefModel(self, ds, seed_time, hidden_dim, dropout, learning_rate, weight_decay, epochs, early_stopping)


class RefModel:
    def __init__(self, ref, ds, seed_time, hidden_dim, dropout, lr, wd, epochs, early_stopping):
        self.r = ref
        self.keep = [_i(ds["g_indptr"]), _i(ds["g_indices"]), _i(ds["f_indptr"]), _i(ds["f_indices"]),
                     _f(ds["f_val"]), _i(ds["split"]), _i(ds["label"])]
        k = self.keep
        self.h = ref.lib.ref_gcn_create(ds["num_nodes"], ds["input_dim"], hidden_dim, ds["output_dim"],
                                        dropout, lr, wd, epochs, early_stopping,
                                        *[a.ctypes.data for a in k], seed_time)

    def train_epoch(self):
        a, b = C.c_float(), C.c_float()
        self.r.lib.ref_gcn_train_epoch(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def eval(self, split):
        a, b = C.c_float(), C.c_float()
        self.r.lib.ref_gcn_eval(self.h, split, C.byref(a), C.byref(b))
        return a.value, b.value

    def var(self, k, grad=False):
        n = self.r.lib.ref_gcn_var_size(self.h, k)
        out = np.zeros(n, np.float32)
        (self.r.lib.ref_gcn_var_grad if grad else self.r.lib.ref_gcn_var_data)(self.h, k, out.ctypes.data)
        return out

    def close(self):
        if self.h:
            self.r.lib.ref_gcn_destroy(self.h)
            self.h = None
