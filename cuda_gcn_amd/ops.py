"""numpy front end of the op-level C-ABI (include/gcnhip.h).

Each function mirrors one module of the reference (Matmul, SparseMatmul,
GraphSum, CrossEntropyLoss, ReLU, Dropout, Adam): same operand meaning, numpy
arrays in and out.  Arrays are staged through device buffers owned by a
``Device`` (gcnhip_malloc / h2d / d2h); all compute happens in the HIP kernels.
There is no CPU path here: without libgcnhip.so and a GPU these calls raise.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class GsOpts(C.Structure):
    """gcnhip_gs_opts (include/gcnhip.h)"""
    _fields_ = [("rows", C.c_void_p), ("in_row_bits", C.c_void_p), ("accumulate", C.c_int), ("relu_dropout", C.c_int), ("training", C.c_int),
                ("p", C.c_float), ("seed", C.c_uint64), ("d_epoch", C.c_void_p), ("elem_offset", C.c_uint64), ("keep_mask", C.c_void_p),
                ("pos_bits", C.c_void_p), ("words_per_row", C.c_int), ("scaling", C.c_int), ("loss", C.c_void_p)]


class GsLoss(C.Structure):
    """gcnhip_gs_loss (include/gcnhip.h)"""
    _fields_ = [("truth", C.c_void_p), ("grad", C.c_void_p), ("ld_grad", C.c_int), ("training", C.c_int), ("count", C.c_int),
                ("grad_row_scale", C.c_void_p), ("row_terms", C.c_void_p)]


class GcnHipError(RuntimeError):
    pass


def _ck(lib, code, what):
    if code != 0:
        msg = lib.gcnhip_error_string(code)
        raise GcnHipError(f"{what}: error {code} ({msg.decode() if msg else '?'})")


class Buf:
    """a device allocation with a numpy-ish shape/dtype"""

    def __init__(self, dev: "Device", shape, dtype):
        self.dev = dev
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _ck(dev.lib, dev.lib.gcnhip_malloc(dev.ctx, C.byref(p), max(self.nbytes, 16)), "gcnhip_malloc")
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, self.dtype)
        assert arr.nbytes == self.nbytes, (arr.shape, self.shape)
        _ck(self.dev.lib, self.dev.lib.gcnhip_h2d(self.dev.ctx, self.ptr, arr.ctypes.data, self.nbytes), "gcnhip_h2d")
        return self

    def download(self):
        out = np.empty(self.shape, self.dtype)
        _ck(self.dev.lib, self.dev.lib.gcnhip_d2h(self.dev.ctx, out.ctypes.data, self.ptr, self.nbytes), "gcnhip_d2h")
        return out

    def fill_bytes(self, byte=0):
        _ck(self.dev.lib, self.dev.lib.gcnhip_memset_async(self.dev.ctx, self.ptr, byte, self.nbytes), "memset")
        return self

    def free(self):
        if self.ptr and self.dev.ctx:            # a closed Device has already released the GPU
            self.dev.lib.gcnhip_free(self.dev.ctx, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Device:
    """one GPU context (device + stream + scratch)"""

    def __init__(self, device: int = 0, stream=None):
        self.lib = _lib.gcnhip()
        n = C.c_int()
        _ck(self.lib, self.lib.gcnhip_device_count(C.byref(n)), "gcnhip_device_count")
        if n.value <= device:
            raise GcnHipError(f"no GPU {device} (device count {n.value}); this package has no CPU path")
        ctx = C.c_void_p()
        _ck(self.lib, self.lib.gcnhip_ctx_create(C.byref(ctx), device, stream), "gcnhip_ctx_create")
        self.ctx = ctx

    def close(self):
        if self.ctx:
            self.lib.gcnhip_ctx_destroy(self.ctx)
            self.ctx = None

    def sync(self):
        _ck(self.lib, self.lib.gcnhip_ctx_sync(self.ctx), "sync")

    def set_option(self, name: str, value: int):
        """gcnhip_ctx_set_option: a named option of this context (read by the ops at call time from the context, never from the environment)"""
        _ck(self.lib, self.lib.gcnhip_ctx_set_option(self.ctx, name.encode(), int(value)), f"gcnhip_ctx_set_option({name})")

    def get_option(self, name: str) -> int:
        v = C.c_int()
        _ck(self.lib, self.lib.gcnhip_ctx_get_option(self.ctx, name.encode(), C.byref(v)), f"gcnhip_ctx_get_option({name})")
        return v.value

    def buf(self, arr_or_shape, dtype=np.float32):
        if isinstance(arr_or_shape, np.ndarray):
            return Buf(self, arr_or_shape.shape, arr_or_shape.dtype).upload(arr_or_shape)
        return Buf(self, arr_or_shape, dtype)

    def padded(self, arr, ld):
        """upload a 2-D float array into a buffer with leading dimension ld (pad = NaN to catch misuse)"""
        arr = np.asarray(arr, np.float32)
        out = np.full((arr.shape[0], ld), np.nan, np.float32)
        out[:, :arr.shape[1]] = arr
        return self.buf(out)

    # ---- prepared objects
    def graph(self, indptr, indices, n_cols=None, col_deg=None, row_group=None):
        return Graph(self, indptr, indices, n_cols, col_deg, row_group)

    def feat(self, indptr, indices, values, n_cols):
        return Feat(self, indptr, indices, values, n_cols)

    # ---- ops (numpy in, numpy out)
    def _bits(self, flags):
        bits = np.packbits(np.asarray(flags, bool), bitorder="little")
        return self.buf(np.concatenate([bits, np.zeros((-bits.size) % 4 + 4, np.uint8)]).view(np.uint32))

    def graphsum_masked(self, g: "Graph", x, ld_in=None, ld_out=None, row_nonzero=None, out_rows=None, fill=np.nan):
        """gcnhip_graphsum_masked: rows of x flagged zero are not read, rows of the result not in out_rows are
        not computed (they keep `fill`)"""
        x = np.asarray(x, np.float32)
        dim = x.shape[1]
        ld_in, ld_out = ld_in or dim, ld_out or dim
        xin = self.padded(x, ld_in)
        out = self.buf(np.full((g.n_rows, ld_out), fill, np.float32))
        g.reserve(dim)
        ib = self._bits(row_nonzero) if row_nonzero is not None else None
        ob = self._bits(out_rows) if out_rows is not None else None
        _ck(self.lib, self.lib.gcnhip_graphsum_masked(self.ctx, g.h, xin.ptr, ld_in, out.ptr, ld_out, dim,
                                                       ib.ptr if ib else None, ob.ptr if ob else None), "gcnhip_graphsum_masked")
        return out.download()[:, :dim]

    def graphsum_rowset(self, g: "Graph", rows_handle, x, ld_in=None, ld_out=None, row_nonzero=None, fill=np.nan):
        """gcnhip_graphsum_rowset: only the rows of a subset registered with Graph.add_rowset are computed"""
        x = np.asarray(x, np.float32)
        dim = x.shape[1]
        ld_in, ld_out = ld_in or dim, ld_out or dim
        xin = self.padded(x, ld_in)
        out = self.buf(np.full((g.n_rows, ld_out), fill, np.float32))
        g.reserve(dim)
        ib = self._bits(row_nonzero) if row_nonzero is not None else None
        _ck(self.lib, self.lib.gcnhip_graphsum_rowset(self.ctx, g.h, rows_handle, xin.ptr, ld_in, out.ptr, ld_out, dim,
                                                       ib.ptr if ib else None), "gcnhip_graphsum_rowset")
        return out.download()[:, :dim]

    def graphsum_ex(self, g: "Graph", x, scaling, ld=None, rows=None, row_nonzero=None, prev=None, fill=np.nan):
        """gcnhip_graphsum_ex: the factored operator.  `x` must already hold dinv[col] * (the reference's input) when scaling != 0;
        prev: the rows of `out` before the call (accumulate = 1)"""
        x = np.asarray(x, np.float32)
        dim = x.shape[1]
        ld = ld or (dim + 3) // 4 * 4
        xin = self.padded(x, ld)
        out = self.padded(prev, ld) if prev is not None else self.buf(np.full((g.n_rows, ld), fill, np.float32))
        g.reserve(dim)
        ib = self._bits(row_nonzero) if row_nonzero is not None else None
        o = GsOpts()
        o.rows = rows
        o.in_row_bits = ib.ptr if ib else None
        o.accumulate = 1 if prev is not None else 0
        o.scaling = int(scaling)
        _ck(self.lib, self.lib.gcnhip_graphsum_ex(self.ctx, g.h, C.byref(o), xin.ptr, ld, out.ptr, ld, dim), "gcnhip_graphsum_ex")
        return out.download()[:, :dim]

    def graphsum_loss(self, g: "Graph", x, scaling, truth, rows=None, training=True, grad_row_scale=None, ld=None, epilogue=True, grad_fill=0.0):
        """the logits' aggregation followed by the loss over the labelled rows (truth >= 0), two ways:
        epilogue=True : gcnhip_graphsum_ex with a gcnhip_gs_loss + gcnhip_xent_from_row_terms;
        epilogue=False: gcnhip_graphsum_ex, then gcnhip_xent_fwd_rows_scaled on the stored logits.
        returns dict(logits, grad, loss_sum, count, correct, total)"""
        x = np.asarray(x, np.float32)
        truth = np.ascontiguousarray(truth, np.int32)
        dim = x.shape[1]
        ld = ld or (dim + 3) // 4 * 4
        xin = self.padded(x, ld)
        out = self.buf(np.full((g.n_rows, ld), np.nan, np.float32))
        gb = self.buf(np.full((g.n_rows, ld), grad_fill, np.float32))
        listed = np.flatnonzero(truth >= 0).astype(np.int32)
        tb, rb = self.buf(truth), self.buf(listed if listed.size else np.zeros(1, np.int32))
        sb = self.buf(np.ascontiguousarray(grad_row_scale, np.float32)) if grad_row_scale is not None else None
        res, resi = self.buf(np.zeros(4, np.float32)), self.buf(np.zeros(2, np.int32))
        terms = self.buf(np.full((g.n_rows, 2), np.nan, np.float32))
        count = max(int(listed.size), 1)
        o = GsOpts()
        o.rows = rows
        o.scaling = int(scaling)
        lo = GsLoss()
        if epilogue:
            lo.truth = tb.ptr; lo.grad = gb.ptr; lo.ld_grad = ld; lo.training = int(training); lo.count = count
            lo.grad_row_scale = sb.ptr if sb else None
            lo.row_terms = terms.ptr
            o.loss = C.addressof(lo)
        _ck(self.lib, self.lib.gcnhip_graphsum_ex(self.ctx, g.h, C.byref(o), xin.ptr, ld, out.ptr, ld, dim), "gcnhip_graphsum_ex")
        if epilogue:
            _ck(self.lib, self.lib.gcnhip_xent_from_row_terms(self.ctx, terms.ptr, tb.ptr, rb.ptr, int(listed.size), res.ptr, resi.ptr), "gcnhip_xent_from_row_terms")
        else:
            _ck(self.lib, self.lib.gcnhip_xent_fwd_rows_scaled(self.ctx, out.ptr, ld, gb.ptr, ld, tb.ptr, rb.ptr, int(listed.size), dim, int(training), count, 0,
                                                               res.ptr, resi.ptr, sb.ptr if sb else None), "gcnhip_xent_fwd_rows_scaled")
        r, ri = res.download(), resi.download()
        return dict(logits=out.download()[:, :dim], grad=gb.download(), loss_sum=float(r[0]), count=float(r[1]), correct=int(ri[0]), total=int(ri[1]),
                    res=r, terms=terms.download() if epilogue else None)

    def graphsum(self, g: "Graph", x, ld_in=None, ld_out=None, row_nonzero=None):
        x = np.asarray(x, np.float32)
        dim = x.shape[1]
        ld_in = ld_in or dim
        ld_out = ld_out or dim
        xin = self.padded(x, ld_in)
        out = self.buf(np.full((g.n_rows, ld_out), np.nan, np.float32))
        g.reserve(dim)
        if row_nonzero is None:
            _ck(self.lib, self.lib.gcnhip_graphsum(self.ctx, g.h, xin.ptr, ld_in, out.ptr, ld_out, dim), "gcnhip_graphsum")
        else:
            bits = np.packbits(np.asarray(row_nonzero, bool), bitorder="little")
            bits = np.concatenate([bits, np.zeros((-bits.size) % 4 + 4, np.uint8)]).view(np.uint32)
            bb = self.buf(bits)
            _ck(self.lib, self.lib.gcnhip_graphsum_rowmask(self.ctx, g.h, xin.ptr, ld_in, out.ptr, ld_out, dim, bb.ptr), "gcnhip_graphsum_rowmask")
        return out.download()[:, :dim]

    def gather_rows(self, x, rows):
        """dst[i] = x[rows[i]] through gcnhip_gather_rows (the packing step of the halo exchange)"""
        x = np.ascontiguousarray(x, np.float32)
        rows = np.ascontiguousarray(rows, np.int32)
        xb, rb = self.buf(x), self.buf(rows if rows.size else np.zeros(1, np.int32))
        out = self.buf(np.full((max(rows.size, 1), x.shape[1]), np.nan, np.float32))
        _ck(self.lib, self.lib.gcnhip_gather_rows(self.ctx, xb.ptr, x.shape[1], rb.ptr, int(rows.size), out.ptr), "gcnhip_gather_rows")
        return out.download()[:rows.size]

    def to_bf16(self, x, ld_dst=None):
        """f32 rows -> bf16 table (uint16 [rows, ld_dst]) through gcnhip_f32_to_bf16"""
        x = np.asarray(x, np.float32)
        rows, dim = x.shape
        ld_dst = ld_dst or (dim + 7) // 8 * 8
        xb = self.buf(np.ascontiguousarray(x))
        dst = self.buf(np.full((rows, ld_dst), 0xFFFF, np.uint16))
        _ck(self.lib, self.lib.gcnhip_f32_to_bf16(self.ctx, xb.ptr, dim, dst.ptr, ld_dst, rows, dim), "gcnhip_f32_to_bf16")
        return dst.download()

    def graphsum_bf16(self, g: "Graph", table_u16, dim, ld_out=None, row_nonzero=None, relu_dropout=None, out_rows=None, fill=np.nan):
        """GraphSum over a bf16 table (uint16 [n_cols, ld]); relu_dropout = dict(training, p, seed, epoch, elem_offset, keep_mask);
        out_rows: a handle from Graph.add_rowset (only those rows are computed, the others keep `fill`)"""
        t = np.ascontiguousarray(table_u16, np.uint16)
        ld_in = t.shape[1]
        ld_out = ld_out or dim
        tb = self.buf(t)
        out = self.buf(np.full((g.n_rows, ld_out), fill, np.float32))
        bb = None
        if row_nonzero is not None:
            bits = np.packbits(np.asarray(row_nonzero, bool), bitorder="little")
            bits = np.concatenate([bits, np.zeros((-bits.size) % 4 + 4, np.uint8)]).view(np.uint32)
            bb = self.buf(bits)
        rd = relu_dropout or {}
        g.reserve(dim)
        ep = self.buf(np.array([rd.get("epoch", 0)], np.uint32))
        km = self.buf(np.ascontiguousarray(rd["keep_mask"], np.uint8)) if rd.get("keep_mask") is not None else None
        _ck(self.lib, self.lib.gcnhip_graphsum_bf16(self.ctx, g.h, tb.ptr, ld_in, out.ptr, ld_out, dim, bb.ptr if bb else None,
                                                     out_rows,
                                                     1 if relu_dropout is not None else 0, int(rd.get("training", 0)), float(rd.get("p", 0.0)),
                                                     int(rd.get("seed", 0)), ep.ptr, int(rd.get("elem_offset", 0)), km.ptr if km else None),
            "gcnhip_graphsum_bf16")
        return out.download()[:, :dim]

    def graphsum_relu_dropout(self, g, x, training, p, seed=0, epoch=0, elem_offset=0, keep_mask=None, ld=None):
        x = np.asarray(x, np.float32)
        dim = x.shape[1]
        ld = ld or dim
        xin = self.padded(x, ld)
        out = self.buf(np.full((g.n_rows, ld), np.nan, np.float32))
        g.reserve(dim)
        ep = self.buf(np.array([epoch], np.uint32))
        km = self.buf(np.ascontiguousarray(keep_mask, np.uint8)) if keep_mask is not None else None
        _ck(self.lib, self.lib.gcnhip_graphsum_relu_dropout(self.ctx, g.h, xin.ptr, ld, out.ptr, ld, dim, int(training), p,
                                                             seed, ep.ptr, elem_offset, km.ptr if km else None), "graphsum_relu_dropout")
        return out.download()[:, :dim]

    def graphsum_relu_dropout_bits(self, g, x, training, p, seed=0, epoch=0, elem_offset=0, keep_mask=None, ld=None):
        """-> (out, bits [n_rows x dim/32] uint32): gcnhip_graphsum_relu_dropout_bits"""
        x = np.asarray(x, np.float32)
        dim = x.shape[1]
        ld = ld or dim
        wpr = (dim + 31) // 32
        xin = self.padded(x, ld)
        out = self.buf(np.full((g.n_rows, ld), np.nan, np.float32))
        bits = self.buf(np.full((g.n_rows, wpr), 0xDEADBEEF, np.uint32))
        g.reserve(dim)
        ep = self.buf(np.array([epoch], np.uint32))
        km = self.buf(np.ascontiguousarray(keep_mask, np.uint8)) if keep_mask is not None else None
        _ck(self.lib, self.lib.gcnhip_graphsum_relu_dropout_bits(self.ctx, g.h, xin.ptr, ld, out.ptr, ld, dim, int(training), p, seed, ep.ptr,
                                                                  elem_offset, km.ptr if km else None, bits.ptr, wpr), "graphsum_relu_dropout_bits")
        return out.download()[:, :dim], bits.download()

    def matmul_bwd_fused_bits(self, a, b, dc, scale, bits):
        a, b, dc = (np.asarray(t, np.float32) for t in (a, b, dc))
        m, n = a.shape
        p = b.shape[1]
        ldp = (p + 3) // 4 * 4
        ab, bb, dcb = self.buf(a), self.padded(b, ldp), self.padded(dc, ldp)
        bt = self.buf(np.ascontiguousarray(bits, np.uint32))
        da = self.buf(np.full((m, n), np.nan, np.float32))
        db = self.buf(np.full((n, ldp), np.nan, np.float32))
        _ck(self.lib, self.lib.gcnhip_matmul_bwd_fused_bits(self.ctx, ab.ptr, n, bb.ptr, ldp, dcb.ptr, ldp, da.ptr, n, db.ptr, ldp, m, n, p, scale,
                                                             bt.ptr, bits.shape[1]), "gcnhip_matmul_bwd_fused_bits")
        return da.download(), db.download()[:, :p]

    def matmul_bwd_ex(self, a, b, dc, scale, bits, rowscale=None, ldp=None, p=None):
        """gcnhip_matmul_bwd_ex: da = mask(bits) . (scale * rowscale[r]) . (dc . b^T), db = a^T . dc; dc may come with its
        padding columns (p = the real width)"""
        a, b, dc = (np.asarray(t, np.float32) for t in (a, b, dc))
        m, n = a.shape
        p = p or b.shape[1]
        ldp = ldp or (p + 3) // 4 * 4
        ab, bb, dcb = self.buf(a), self.padded(b, ldp), self.padded(dc, ldp)
        bt = self.buf(np.ascontiguousarray(bits, np.uint32))
        rs = self.buf(np.ascontiguousarray(rowscale, np.float32)) if rowscale is not None else None
        da = self.buf(np.full((m, n), np.nan, np.float32))
        db = self.buf(np.full((n, ldp), np.nan, np.float32))
        _ck(self.lib, self.lib.gcnhip_matmul_bwd_ex(self.ctx, ab.ptr, n, bb.ptr, ldp, dcb.ptr, ldp, da.ptr, n, db.ptr, ldp, m, n, p, scale,
                                                     bt.ptr, bits.shape[1], rs.ptr if rs else None), "gcnhip_matmul_bwd_ex")
        return da.download(), db.download()[:, :p]

    def spmm_fwd(self, f: "Feat", w, p_drop=0.0, seed=0, epoch=0, nnz_offset=0, keep_mask=None, vals=None, ld_w=None, ld_out=None):
        w = np.asarray(w, np.float32)
        p = w.shape[1]
        ld_w = ld_w or p
        ld_out = ld_out or p
        wb = self.padded(w, ld_w)
        out = self.buf(np.full((f.n_rows, ld_out), np.nan, np.float32))
        ep = self.buf(np.array([epoch], np.uint32))
        km = self.buf(np.ascontiguousarray(keep_mask, np.uint8)) if keep_mask is not None else None
        vb = self.buf(np.ascontiguousarray(vals, np.float32)) if vals is not None else None
        vptr = vb.ptr if vb else f.values_ptr
        _ck(self.lib, self.lib.gcnhip_spmm_fwd(self.ctx, f.h, vptr, wb.ptr, ld_w, out.ptr, ld_out, p, p_drop, seed, ep.ptr,
                                                nnz_offset, km.ptr if km else None), "gcnhip_spmm_fwd")
        return out.download()[:, :p]

    def spmm_fwd_relu(self, f: "Feat", w, ld_w=None, ld_out=None):
        """ReLU(X . w) through gcnhip_spmm_fwd_relu (the evaluation form on an aggregated feature object)"""
        w = np.asarray(w, np.float32)
        p = w.shape[1]
        ld_w, ld_out = ld_w or p, ld_out or p
        wb = self.padded(w, ld_w)
        out = self.buf(np.full((f.n_rows, ld_out), np.nan, np.float32))
        _ck(self.lib, self.lib.gcnhip_spmm_fwd_relu(self.ctx, f.h, f.values_ptr, wb.ptr, ld_w, out.ptr, ld_out, p), "gcnhip_spmm_fwd_relu")
        return out.download()[:, :p]

    def spmm_fwd_relu_matmul(self, f: "Feat", w, w2, ld_z=None):
        """gcnhip_spmm_fwd_relu_matmul: ReLU(X . w) . w2 in one launch; returns None when the fused form is not available"""
        w, w2 = np.asarray(w, np.float32), np.asarray(w2, np.float32)
        p, p2 = w.shape[1], w2.shape[1]
        ld_z = ld_z or (p2 + 3) // 4 * 4
        wb, w2b = self.buf(w), self.padded(w2, ld_z)
        z = self.buf(np.full((f.n_rows, ld_z), np.nan, np.float32))
        rc = self.lib.gcnhip_spmm_fwd_relu_matmul(self.ctx, f.h, f.values_ptr, wb.ptr, p, p, w2b.ptr, ld_z, p2, z.ptr, ld_z)
        if rc == -2:
            return None
        _ck(self.lib, rc, "gcnhip_spmm_fwd_relu_matmul")
        return z.download()[:, :p2]

    def spmm_bwd(self, f: "Feat", dout, p_drop=0.0, seed=0, epoch=0, nnz_offset=0, keep_mask=None, vals=None, ld_dout=None, ld_dw=None):
        dout = np.asarray(dout, np.float32)
        p = dout.shape[1]
        ld_dout = ld_dout or p
        ld_dw = ld_dw or p
        db = self.padded(dout, ld_dout)
        dw = self.buf(np.full((f.n_cols, ld_dw), np.nan, np.float32))
        ep = self.buf(np.array([epoch], np.uint32))
        km = self.buf(np.ascontiguousarray(keep_mask, np.uint8)) if keep_mask is not None else None
        vb = self.buf(np.ascontiguousarray(vals, np.float32)) if vals is not None else None
        vptr = vb.ptr if vb else f.values_ptr
        _ck(self.lib, self.lib.gcnhip_spmm_bwd(self.ctx, f.h, vptr, db.ptr, ld_dout, dw.ptr, ld_dw, p, p_drop, seed, ep.ptr,
                                                nnz_offset, km.ptr if km else None), "gcnhip_spmm_bwd")
        return dw.download()[:, :p]

    def spmm_bwd_plan(self, f: "Feat", p):
        rps, ns = C.c_int(), C.c_int()
        _ck(self.lib, self.lib.gcnhip_spmm_bwd_plan(self.ctx, f.h, p, C.byref(rps), C.byref(ns)), "gcnhip_spmm_bwd_plan")
        return rps.value, ns.value

    def spmm_bwd_parts(self, f: "Feat", dout, cuts, p_drop=0.0, seed=0, epoch=0, nnz_offset=0, order=None, make_first=True):
        """the weight gradient as gcnhip_spmm_bwd_part calls on the split ranges between `cuts` (in `order`), then _finish;
        make_first=False: no part makes the keep decisions (a forward with the same arguments left them in the feature object)"""
        dout = np.asarray(dout, np.float32)
        p = dout.shape[1]
        db = self.padded(dout, p)
        dw = self.buf(np.full((f.n_cols, p), np.nan, np.float32))
        ep = self.buf(np.array([epoch], np.uint32))
        ranges = list(zip(cuts[:-1], cuts[1:]))
        for k, i in enumerate(order if order is not None else range(len(ranges))):
            _ck(self.lib, self.lib.gcnhip_spmm_bwd_part(self.ctx, f.h, f.values_ptr, db.ptr, p, p, p_drop, seed, ep.ptr, nnz_offset, None,
                                                         ranges[i][0], ranges[i][1], 1 if (k == 0 and make_first) else 0), "gcnhip_spmm_bwd_part")
        _ck(self.lib, self.lib.gcnhip_spmm_bwd_finish(self.ctx, f.h, dw.ptr, p, p), "gcnhip_spmm_bwd_finish")
        return dw.download()

    def matmul_fwd(self, a, b, lda=None, ldb=None, ldc=None):
        a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
        m, n = a.shape
        p = b.shape[1]
        lda, ldb, ldc = lda or n, ldb or p, ldc or p
        ab, bb = self.padded(a, lda), self.padded(b, ldb)
        cb = self.buf(np.full((m, ldc), np.nan, np.float32))
        _ck(self.lib, self.lib.gcnhip_matmul_fwd(self.ctx, ab.ptr, lda, bb.ptr, ldb, cb.ptr, ldc, m, n, p), "gcnhip_matmul_fwd")
        return cb.download()[:, :p]

    def matmul_bwd(self, a, b, dc, lda=None, ldb=None, lddc=None, fused_scale=None):
        a, b, dc = (np.asarray(t, np.float32) for t in (a, b, dc))
        m, n = a.shape
        p = b.shape[1]
        lda, ldb, lddc = lda or n, ldb or p, lddc or p
        ab, bb, dcb = self.padded(a, lda), self.padded(b, ldb), self.padded(dc, lddc)
        da = self.buf(np.full((m, lda), np.nan, np.float32))
        db = self.buf(np.full((n, ldb), np.nan, np.float32))
        if fused_scale is None:
            _ck(self.lib, self.lib.gcnhip_matmul_bwd(self.ctx, ab.ptr, lda, bb.ptr, ldb, dcb.ptr, lddc, da.ptr, lda, db.ptr, ldb, m, n, p), "gcnhip_matmul_bwd")
        else:
            _ck(self.lib, self.lib.gcnhip_matmul_bwd_fused(self.ctx, ab.ptr, lda, bb.ptr, ldb, dcb.ptr, lddc, da.ptr, lda, db.ptr, ldb,
                                                            m, n, p, fused_scale), "gcnhip_matmul_bwd_fused")
        return da.download()[:, :n], db.download()[:, :p]

    def packed_backward_gather(self, g: "Graph", h, b, dc, scale):
        """The hidden layer's backward through packed rows: da = (h > 0) ? scale * dc . b^T : 0 written as packed rows
        (gcnhip_matmul_bwd_packed), then out = A^ . da gathered from the slots (gcnhip_graphsum_packed).
        Returns (da expanded to dense by gcnhip_rowpack_expand, db, out, number of half rows that did not fit a slot)."""
        h, b, dc = (np.asarray(t, np.float32) for t in (h, b, dc))
        m, n = h.shape
        p = b.shape[1]
        ldp = (p + 3) // 4 * 4
        hb, bb, dcb = self.buf(h), self.padded(b, ldp), self.padded(dc, ldp)
        da = self.buf(np.full((m, n), np.nan, np.float32))
        db = self.buf(np.full((n, ldp), np.nan, np.float32))
        pk = C.c_void_p()
        _ck(self.lib, self.lib.gcnhip_rowpack_create(self.ctx, C.byref(pk), m, n), "gcnhip_rowpack_create")
        try:
            _ck(self.lib, self.lib.gcnhip_matmul_bwd_packed(self.ctx, hb.ptr, n, bb.ptr, ldp, dcb.ptr, ldp, da.ptr, n, pk,
                                                             db.ptr, ldp, m, n, p, scale), "gcnhip_matmul_bwd_packed")
            out = self.buf(np.full((g.n_rows, n), np.nan, np.float32))
            g.reserve(n)
            _ck(self.lib, self.lib.gcnhip_graphsum_packed(self.ctx, g.h, pk, da.ptr, n, out.ptr, n), "gcnhip_graphsum_packed")
            raw = da.download()
            overflow = int(np.isfinite(raw).any(axis=1).sum()) if np.isnan(raw).any() else m
            _ck(self.lib, self.lib.gcnhip_rowpack_expand(self.ctx, pk, da.ptr, n), "gcnhip_rowpack_expand")
            return da.download(), db.download()[:, :p], out.download(), overflow
        finally:
            self.lib.gcnhip_rowpack_destroy(self.ctx, pk)

    def pack_positive(self, h, ld=None):
        """bit (r, c) = h[r, c] > 0, 32 columns per little-endian word -> uint32 [rows, ceil(dim/32)]"""
        h = np.asarray(h, np.float32)
        rows, dim = h.shape
        ld = ld or dim
        wpr = (dim + 31) // 32
        hb = self.padded(h, ld)
        bits = self.buf(np.full((rows, wpr), 0xFFFFFFFF, np.uint32))
        _ck(self.lib, self.lib.gcnhip_pack_positive(self.ctx, hb.ptr, ld, rows, dim, bits.ptr, wpr), "gcnhip_pack_positive")
        return bits.download()

    def matmul_bwd_da_bits(self, b, dc, bits, scale, ldb=None, lddc=None, ldda=None):
        b, dc = np.asarray(b, np.float32), np.asarray(dc, np.float32)
        bits = np.ascontiguousarray(bits, np.uint32)
        n, p = b.shape
        m = dc.shape[0]
        ldb, lddc, ldda = ldb or p, lddc or p, ldda or n
        bb, dcb, bt = self.padded(b, ldb), self.padded(dc, lddc), self.buf(bits)
        da = self.buf(np.full((m, ldda), np.nan, np.float32))
        _ck(self.lib, self.lib.gcnhip_matmul_bwd_da_bits(self.ctx, bb.ptr, ldb, dcb.ptr, lddc, da.ptr, ldda, m, n, p,
                                                          bt.ptr, bits.shape[1], scale), "gcnhip_matmul_bwd_da_bits")
        return da.download()[:, :n]

    def relu_fwd(self, x, training=True):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        xb = self.buf(x)
        mb = self.buf(np.zeros(x.size, np.uint8))
        _ck(self.lib, self.lib.gcnhip_relu_fwd(self.ctx, xb.ptr, mb.ptr, x.size, int(training)), "gcnhip_relu_fwd")
        return xb.download(), mb.download()

    def relu_bwd(self, grad, mask):
        g = self.buf(np.ascontiguousarray(grad, np.float32).reshape(-1))
        mb = self.buf(np.ascontiguousarray(mask, np.uint8))
        _ck(self.lib, self.lib.gcnhip_relu_bwd(self.ctx, g.ptr, mb.ptr, g.shape[0]), "gcnhip_relu_bwd")
        return g.download()

    def dropout_fwd(self, x, p, seed=0, epoch=0, elem_offset=0, keep_in=None, want_mask=True):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        xb = self.buf(x)
        mb = self.buf(np.zeros(x.size, np.int32)) if want_mask else None
        ep = self.buf(np.array([epoch], np.uint32))
        kb = self.buf(np.ascontiguousarray(keep_in, np.uint8)) if keep_in is not None else None
        _ck(self.lib, self.lib.gcnhip_dropout_fwd(self.ctx, xb.ptr, mb.ptr if mb else None, x.size, p, seed, ep.ptr, elem_offset,
                                                   kb.ptr if kb else None), "gcnhip_dropout_fwd")
        return xb.download(), (mb.download() if mb else None)

    def dropout_bwd(self, grad, mask, p):
        g = self.buf(np.ascontiguousarray(grad, np.float32).reshape(-1))
        mb = self.buf(np.ascontiguousarray(mask, np.int32))
        _ck(self.lib, self.lib.gcnhip_dropout_bwd(self.ctx, g.ptr, mb.ptr, g.shape[0], p), "gcnhip_dropout_bwd")
        return g.download()

    def relu_dropout_bwd(self, grad, h, scale):
        grad, h = np.asarray(grad, np.float32), np.asarray(h, np.float32)
        g, hb = self.buf(np.ascontiguousarray(grad)), self.buf(np.ascontiguousarray(h))
        _ck(self.lib, self.lib.gcnhip_relu_dropout_bwd(self.ctx, g.ptr, grad.shape[1], hb.ptr, h.shape[1], grad.shape[0], grad.shape[1], scale), "relu_dropout_bwd")
        return g.download()

    def xent_fwd(self, logits, truth, training=True, count=0, shift_in_place=True, ld=None):
        """returns dict(loss_sum, count, correct, total, logits, grad)"""
        logits = np.asarray(logits, np.float32)
        n, c = logits.shape
        ld = ld or c
        lb = self.padded(logits, ld)
        gb = self.buf(np.full((n, ld), np.nan, np.float32))
        tb = self.buf(np.ascontiguousarray(truth, np.int32))
        res = self.buf(np.zeros(4, np.float32))
        resi = self.buf(np.zeros(2, np.int32))
        _ck(self.lib, self.lib.gcnhip_xent_fwd(self.ctx, lb.ptr, ld, gb.ptr, ld, tb.ptr, n, c, int(training), int(count),
                                                int(shift_in_place), res.ptr, resi.ptr), "gcnhip_xent_fwd")
        r, ri = res.download(), resi.download()
        return dict(loss_sum=float(r[0]), count=float(r[1]), correct=int(ri[0]), total=int(ri[1]),
                    logits=lb.download()[:, :c], grad=gb.download()[:, :c] if training else None)

    def xent_fwd_rows(self, logits, truth, training=True, shift_in_place=False, ld=None, grad_fill=0.0):
        """gcnhip_xent_fwd_rows over the list of labelled rows; grad rows outside the list keep grad_fill"""
        logits = np.asarray(logits, np.float32)
        truth = np.ascontiguousarray(truth, np.int32)
        n, c = logits.shape
        ld = ld or c
        rows = np.flatnonzero(truth >= 0).astype(np.int32)
        lb = self.padded(logits, ld)
        gb = self.buf(np.full((n, ld), grad_fill, np.float32))
        tb, rb = self.buf(truth), self.buf(rows if rows.size else np.zeros(1, np.int32))
        res, resi = self.buf(np.zeros(4, np.float32)), self.buf(np.zeros(2, np.int32))
        _ck(self.lib, self.lib.gcnhip_xent_fwd_rows(self.ctx, lb.ptr, ld, gb.ptr, ld, tb.ptr, rb.ptr, int(rows.size), c, int(training),
                                                     max(int(rows.size), 1), int(shift_in_place), res.ptr, resi.ptr), "gcnhip_xent_fwd_rows")
        r, ri = res.download(), resi.download()
        return dict(loss_sum=float(r[0]), count=float(r[1]), correct=int(ri[0]), total=int(ri[1]),
                    logits=lb.download()[:, :c], grad=gb.download()[:, :c] if training else None)

    def accuracy(self, logits, truth, ld=None):
        logits = np.asarray(logits, np.float32)
        n, c = logits.shape
        ld = ld or c
        lb = self.padded(logits, ld)
        tb = self.buf(np.ascontiguousarray(truth, np.int32))
        resi = self.buf(np.zeros(2, np.int32))
        _ck(self.lib, self.lib.gcnhip_accuracy(self.ctx, lb.ptr, ld, tb.ptr, n, c, resi.ptr), "gcnhip_accuracy")
        ri = resi.download()
        return int(ri[0]), int(ri[1])

    def set_truth(self, split, label, s):
        sb, lb = self.buf(np.ascontiguousarray(split, np.int32)), self.buf(np.ascontiguousarray(label, np.int32))
        tb = self.buf(np.zeros(sb.shape[0], np.int32))
        _ck(self.lib, self.lib.gcnhip_set_truth(self.ctx, tb.ptr, sb.ptr, lb.ptr, sb.shape[0], s), "gcnhip_set_truth")
        return tb.download()

    def sumsq(self, x):
        xb = self.buf(np.ascontiguousarray(x, np.float32).reshape(-1))
        ob = self.buf(np.zeros(1, np.float32))
        _ck(self.lib, self.lib.gcnhip_sumsq(self.ctx, xb.ptr, xb.shape[0], ob.ptr), "gcnhip_sumsq")
        return float(ob.download()[0])

    def adam_steps(self, ws, grads_per_step, decays, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-8, epoch_words=None):
        """ws: list of arrays; grads_per_step: list (steps) of lists (vars). Returns (ws, sumsq of ws[0]).
        epoch_words: a device buffer of two uint32 [counter, done] -> gcnhip_adam_step_advance moves them with every step"""
        bufs = []
        for w in ws:
            w = np.ascontiguousarray(w, np.float32).reshape(-1)
            bufs.append(dict(w=self.buf(w), g=self.buf(np.zeros_like(w)), m=self.buf(np.zeros_like(w)), v=self.buf(np.zeros_like(w)), n=w.size))
        arr = (_lib.AdamVar * len(ws))()
        for k, b in enumerate(bufs):
            arr[k] = _lib.AdamVar(b["w"].ptr, b["g"].ptr, b["m"].ptr, b["v"].ptr, b["n"], int(decays[k]))
        sq = self.buf(np.zeros(1, np.float32))
        b1, b2 = np.float32(beta1), np.float32(beta2)
        for t, grads in enumerate(grads_per_step, start=1):
            for k, g in enumerate(grads):
                bufs[k]["g"].upload(np.ascontiguousarray(g, np.float32).reshape(-1))
            # optim.cpp:26 in float arithmetic
            step = np.float32(lr) * np.sqrt(np.float32(1) - np.power(b2, np.float32(t), dtype=np.float32)) / (np.float32(1) - np.power(b1, np.float32(t), dtype=np.float32))
            if epoch_words is not None:
                _ck(self.lib, self.lib.gcnhip_adam_step_advance(self.ctx, arr, len(ws), float(step), None, None, beta1, beta2, eps, weight_decay, sq.ptr,
                                                                epoch_words.ptr, epoch_words.ptr + 4), "gcnhip_adam_step_advance")
            else:
                _ck(self.lib, self.lib.gcnhip_adam_step(self.ctx, arr, len(ws), float(step), None, None, beta1, beta2, eps, weight_decay, sq.ptr), "gcnhip_adam_step")
        return [b["w"].download() for b in bufs], float(sq.download()[0])


class Graph:
    def __init__(self, dev: Device, indptr, indices, n_cols=None, col_deg=None, row_group=None):
        self.dev = dev
        indptr = np.ascontiguousarray(indptr, np.int32)
        indices = np.ascontiguousarray(indices, np.int32)
        self.n_rows = indptr.size - 1
        self.n_cols = int(n_cols) if n_cols is not None else self.n_rows
        cd = np.ascontiguousarray(col_deg, np.int32) if col_deg is not None else None
        rg = np.ascontiguousarray(row_group, np.int32) if row_group is not None else None
        h = C.c_void_p()
        _ck(dev.lib, dev.lib.gcnhip_graph_create_grouped(dev.ctx, C.byref(h), indptr.ctypes.data, indices.ctypes.data, self.n_rows, self.n_cols,
                                                          cd.ctypes.data if cd is not None else None,
                                                          rg.ctypes.data if rg is not None else None), "gcnhip_graph_create_grouped")
        self.h = h

    def set_schedule(self, mode, row_group=None, n_groups=0):
        rg = np.ascontiguousarray(row_group, np.int32) if row_group is not None else None
        _ck(self.dev.lib, self.dev.lib.gcnhip_graph_set_schedule(self.dev.ctx, self.h, mode, rg.ctypes.data if rg is not None else None,
                                                                  n_groups), "gcnhip_graph_set_schedule")

    def add_rowset(self, wanted):
        """register a subset of the rows (boolean per row); returns the handle gcnhip_graphsum_rowset takes"""
        bits = np.packbits(np.asarray(wanted, bool), bitorder="little")
        bits = np.concatenate([bits, np.zeros((-bits.size) % 4 + 8, np.uint8)]).view(np.uint32)
        h = C.c_void_p()
        _ck(self.dev.lib, self.dev.lib.gcnhip_graph_add_rowset(self.dev.ctx, self.h, bits.ctypes.data, C.byref(h)), "gcnhip_graph_add_rowset")
        return h

    def restricted(self, keep_cols):
        """a second adjacency object without the edges whose source row is outside `keep_cols` (boolean per column)"""
        bits = np.packbits(np.asarray(keep_cols, bool), bitorder="little")
        bits = np.concatenate([bits, np.zeros((-bits.size) % 4 + 8, np.uint8)]).view(np.uint32)
        h = C.c_void_p()
        _ck(self.dev.lib, self.dev.lib.gcnhip_graph_create_restricted(self.dev.ctx, C.byref(h), self.h, bits.ctypes.data), "gcnhip_graph_create_restricted")
        g = Graph.__new__(Graph)
        g.dev, g.n_rows, g.n_cols, g.h = self.dev, self.n_rows, self.n_cols, h
        return g

    def reserve(self, dim):
        """segment scratch for aggregations up to `dim` columns (256 are reserved when the object is built)"""
        if dim > 256:
            _ck(self.dev.lib, self.dev.lib.gcnhip_graph_reserve_width(self.dev.ctx, self.h, int(dim)), "gcnhip_graph_reserve_width")

    def scales(self):
        """(dinv_row, dinv2_row, dinv_col, dinv2_col) of gcnhip_graph_scales, as numpy"""
        ps = [C.c_void_p() for _ in range(4)]
        _ck(self.dev.lib, self.dev.lib.gcnhip_graph_scales(self.h, *[C.byref(q) for q in ps]), "gcnhip_graph_scales")
        out = []
        for q, n in zip(ps, (self.n_rows, self.n_rows, self.n_cols, self.n_cols)):
            a = np.empty(n, np.float32)
            _ck(self.dev.lib, self.dev.lib.gcnhip_d2h(self.dev.ctx, a.ctypes.data, q, a.nbytes), "d2h")
            out.append(a)
        return out

    def coef(self):
        pc = C.c_void_p()
        nr, nnz = C.c_int(), C.c_int()
        _ck(self.dev.lib, self.dev.lib.gcnhip_graph_arrays(self.h, None, None, C.byref(pc), C.byref(nr), C.byref(nnz)), "graph_arrays")
        out = np.empty(nnz.value, np.float32)
        _ck(self.dev.lib, self.dev.lib.gcnhip_d2h(self.dev.ctx, out.ctypes.data, pc, out.nbytes), "d2h")
        return out

    def free(self):
        if self.h and self.dev.ctx:
            self.dev.lib.gcnhip_graph_destroy(self.dev.ctx, self.h)
        self.h = None


class Feat:
    @classmethod
    def aggregated(cls, dev: Device, g: "Graph", x: "Feat"):
        """the feature object of A^.X (gcnhip_feat_create_aggregated)"""
        self = cls.__new__(cls)
        self.dev = dev
        self.n_rows, self.n_cols = g.n_rows, x.n_cols
        h = C.c_void_p()
        _ck(dev.lib, dev.lib.gcnhip_feat_create_aggregated(dev.ctx, C.byref(h), g.h, x.h), "gcnhip_feat_create_aggregated")
        self.h = h
        self.values_ptr = dev.lib.gcnhip_feat_values(h)
        self.dense = True
        return self

    def values(self):
        out = np.empty((self.n_rows, self.n_cols), np.float32)
        _ck(self.dev.lib, self.dev.lib.gcnhip_d2h(self.dev.ctx, out.ctypes.data, self.values_ptr, out.nbytes), "d2h")
        return out

    def __init__(self, dev: Device, indptr, indices, values, n_cols):
        self.dev = dev
        indptr = np.ascontiguousarray(indptr, np.int32)
        indices = np.ascontiguousarray(indices, np.int32) if indices is not None else None
        values = np.ascontiguousarray(values, np.float32)
        self.n_rows = indptr.size - 1
        self.n_cols = int(n_cols)
        h = C.c_void_p()
        _ck(dev.lib, dev.lib.gcnhip_feat_create(dev.ctx, C.byref(h), indptr.ctypes.data, indices.ctypes.data if indices is not None else None,
                                                 values.ctypes.data, self.n_rows, self.n_cols), "gcnhip_feat_create")
        self.h = h
        self.values_ptr = dev.lib.gcnhip_feat_values(h)
        self.dense = bool(dev.lib.gcnhip_feat_is_dense(h))

    def free(self):
        if self.h and self.dev.ctx:
            self.dev.lib.gcnhip_feat_destroy(self.dev.ctx, self.h)
        self.h = None
