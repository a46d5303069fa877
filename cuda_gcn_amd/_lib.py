"""ctypes loader for the in-tree native libraries.

There is no fallback: if libgcnhip.so (hand-written HIP for gfx950 behind the
C-ABI of include/gcnhip.h) is missing or does not load, importing an op fails
loudly.  Build with ``make kernels host`` or ``__graft_entry__.build()``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("GCN_LIBDIR") or os.path.join(HERE, "lib")     # GCN_LIBDIR: `make asan-test` points at build/asan

_cache = {}


class NativeLibraryMissing(RuntimeError):
    pass


def _load(name: str) -> C.CDLL:
    if name in _cache:
        return _cache[name]
    path = os.path.join(LIBDIR, name)
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f"{path} not found: the HIP extension is not built. Run `make kernels host` "
            f"(or __graft_entry__.build()); there is no CPU fallback.")
    try:
        lib = C.CDLL(path)          # RTLD_LOCAL: the C++ host classes must not interpose on anyone else's
    except OSError as e:  # pragma: no cover
        raise NativeLibraryMissing(f"cannot load {path}: {e}") from e
    _cache[name] = lib
    return lib


def gcnhip() -> C.CDLL:
    lib = _load("libgcnhip.so")
    if not getattr(lib, "_typed", False):
        _declare_gcnhip(lib)
        lib._typed = True
    return lib


def gcnhost() -> C.CDLL:
    gcnhip()
    lib = _load("libgcnhost.so")
    if not getattr(lib, "_typed", False):
        _declare_gcnhost(lib)
        lib._typed = True
    return lib


P = C.c_void_p
I = C.c_int
I64 = C.c_int64
U64 = C.c_uint64
F = C.c_float


class AdamVar(C.Structure):
    _fields_ = [("w", P), ("g", P), ("m", P), ("v", P), ("n", I64), ("decay", I)]


# every symbol of include/gcnhip.h: name -> (restype, argtypes)
GCNHIP_SYMBOLS = {
    "gcnhip_device_count": (I, [C.POINTER(I)]),
    "gcnhip_ctx_create": (I, [C.POINTER(P), I, P]),
    "gcnhip_ctx_destroy": (I, [P]),
    "gcnhip_ctx_sync": (I, [P]),
    "gcnhip_ctx_set_corun": (I, [P, I]),
    "gcnhip_ctx_set_option": (I, [P, C.c_char_p, I]),
    "gcnhip_ctx_get_option": (I, [P, C.c_char_p, C.POINTER(I)]),
    "gcnhip_ctx_stream": (P, [P]),
    "gcnhip_error_string": (C.c_char_p, [I]),
    "gcnhip_last_error": (C.c_char_p, []),
    "gcnhip_version": (C.c_char_p, []),
    "gcnhip_experiments": (I, []),
    "gcnhip_malloc": (I, [P, C.POINTER(P), C.c_size_t]),
    "gcnhip_free": (I, [P, P]),
    "gcnhip_memset_async": (I, [P, P, I, C.c_size_t]),
    "gcnhip_h2d": (I, [P, P, P, C.c_size_t]),
    "gcnhip_d2h": (I, [P, P, P, C.c_size_t]),
    "gcnhip_d2d_async": (I, [P, P, P, C.c_size_t]),
    "gcnhip_host_alloc": (I, [C.POINTER(P), C.c_size_t]),
    "gcnhip_host_free": (I, [P]),
    "gcnhip_d2h_async": (I, [P, P, P, C.c_size_t]),
    "gcnhip_graph_create": (I, [P, C.POINTER(P), P, P, I, I, P]),
    "gcnhip_graph_create_grouped": (I, [P, C.POINTER(P), P, P, I, I, P, P]),
    "gcnhip_graph_set_schedule": (I, [P, P, I, P, I]),
    "gcnhip_graph_reserve_width": (I, [P, P, I]),
    "gcnhip_graph_destroy": (I, [P, P]),
    "gcnhip_graph_arrays": (I, [P, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(I), C.POINTER(I)]),
    "gcnhip_graphsum": (I, [P, P, P, I, P, I, I]),
    "gcnhip_graphsum_rowmask": (I, [P, P, P, I, P, I, I, P]),
    "gcnhip_graphsum_masked": (I, [P, P, P, I, P, I, I, P, P]),
    "gcnhip_graph_add_rowset": (I, [P, P, P, C.POINTER(P)]),
    "gcnhip_graph_create_restricted": (I, [P, C.POINTER(P), P, P]),
    "gcnhip_graph_clone": (I, [P, C.POINTER(P), P]),
    "gcnhip_rowset_size": (I, [P, C.POINTER(I)]),
    "gcnhip_graphsum_rowset": (I, [P, P, P, P, I, P, I, I, P]),
    "gcnhip_graphsum_relu_dropout": (I, [P, P, P, I, P, I, I, I, F, U64, P, U64, P]),
    "gcnhip_graphsum_relu_dropout_bits": (I, [P, P, P, I, P, I, I, I, F, U64, P, U64, P, P, I]),
    "gcnhip_graphsum_ex": (I, [P, P, P, P, I, P, I, I]),
    "gcnhip_graph_scales": (I, [P, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P)]),
    "gcnhip_feat_scale_rows": (I, [P, P, P]),
    "gcnhip_xent_fwd_rows_scaled": (I, [P, P, I, P, I, P, P, I, I, I, I, I, P, P, P]),
    "gcnhip_xent_from_row_terms": (I, [P, P, P, P, I, P, P]),
    "gcnhip_matmul_bwd_ex": (I, [P, P, I, P, I, P, I, P, I, P, I, I, I, I, F, P, I, P]),
    "gcnhip_graphsum_part": (I, [P, P, P, P, I, P, I, I, P, I, I, I, F, U64, P, U64, P]),
    "gcnhip_feat_create": (I, [P, C.POINTER(P), P, P, P, I, I]),
    "gcnhip_feat_create_aggregated": (I, [P, C.POINTER(P), P, P]),
    "gcnhip_feat_destroy": (I, [P, P]),
    "gcnhip_feat_is_dense": (I, [P]),
    "gcnhip_feat_values": (P, [P]),
    "gcnhip_feat_nnz": (I64, [P]),
    "gcnhip_spmm_fwd": (I, [P, P, P, P, I, P, I, I, F, U64, P, U64, P]),
    "gcnhip_spmm_fwd_relu": (I, [P, P, P, P, I, P, I, I]),
    "gcnhip_spmm_fwd_relu_matmul": (I, [P, P, P, P, I, I, P, I, I, P, I]),
    "gcnhip_spmm_bwd": (I, [P, P, P, P, I, P, I, I, F, U64, P, U64, P]),
    "gcnhip_spmm_bwd_plan": (I, [P, P, I, C.POINTER(I), C.POINTER(I)]),
    "gcnhip_spmm_bwd_part": (I, [P, P, P, P, I, I, F, U64, P, U64, P, I, I, I]),
    "gcnhip_spmm_bwd_finish": (I, [P, P, P, I, I]),
    "gcnhip_matmul_fwd": (I, [P, P, I, P, I, P, I, I, I, I]),
    "gcnhip_matmul_bwd": (I, [P, P, I, P, I, P, I, P, I, P, I, I, I, I]),
    "gcnhip_matmul_bwd_fused": (I, [P, P, I, P, I, P, I, P, I, P, I, I, I, I, F]),
    "gcnhip_matmul_bwd_fused_bits": (I, [P, P, I, P, I, P, I, P, I, P, I, I, I, I, F, P, I]),
    "gcnhip_rowpack_create": (I, [P, C.POINTER(P), I, I]),
    "gcnhip_rowpack_destroy": (I, [P, P]),
    "gcnhip_rowpack_expand": (I, [P, P, P, I]),
    "gcnhip_matmul_bwd_packed": (I, [P, P, I, P, I, P, I, P, I, P, P, I, I, I, I, F]),
    "gcnhip_graphsum_packed": (I, [P, P, P, P, I, P, I]),
    "gcnhip_pack_positive": (I, [P, P, I, I, I, P, I]),
    "gcnhip_f32_to_bf16": (I, [P, P, I, P, I, I64, I]),
    "gcnhip_graphsum_bf16": (I, [P, P, P, I, P, I, I, P, P, I, I, F, U64, P, U64, P]),
    "gcnhip_matmul_bwd_da_bits": (I, [P, P, I, P, I, P, I, I, I, I, P, I, F]),
    "gcnhip_gather_rows": (I, [P, P, I, P, I, P]),
    "gcnhip_relu_fwd": (I, [P, P, P, I64, I]),
    "gcnhip_relu_bwd": (I, [P, P, P, I64]),
    "gcnhip_dropout_fwd": (I, [P, P, P, I64, F, U64, P, U64, P]),
    "gcnhip_dropout_bwd": (I, [P, P, P, I64, F]),
    "gcnhip_relu_fwd_2d": (I, [P, P, I, I, I, P, I]),
    "gcnhip_relu_bwd_2d": (I, [P, P, I, I, I, P]),
    "gcnhip_dropout_fwd_2d": (I, [P, P, I, I, I, P, F, U64, P, U64, P]),
    "gcnhip_dropout_bwd_2d": (I, [P, P, I, I, I, P, F]),
    "gcnhip_relu_dropout_bwd": (I, [P, P, I, P, I, I, I, F]),
    "gcnhip_xent_fwd": (I, [P, P, I, P, I, P, I, I, I, I, I, P, P]),
    "gcnhip_xent_fwd_rows": (I, [P, P, I, P, I, P, P, I, I, I, I, I, P, P]),
    "gcnhip_accuracy": (I, [P, P, I, P, I, I, P]),
    "gcnhip_set_truth": (I, [P, P, P, P, I, I]),
    "gcnhip_sumsq": (I, [P, P, I64, P]),
    "gcnhip_adam_step": (I, [P, C.POINTER(AdamVar), I, F, P, P, F, F, F, F, P]),
    "gcnhip_adam_step_advance": (I, [P, C.POINTER(AdamVar), I, F, P, P, F, F, F, F, P, P, P]),
    "gcnhip_counter_add": (I, [P, P, C.c_uint32]),
    "gcnhip_metrics_record": (I, [P, P, I, I, P, P, P, P]),
    "gcnhip_metrics_record_with_next_loss": (I, [P, P, I, I, P, P]),
    "gcnhip_capture_begin": (I, [P]),
    "gcnhip_capture_end": (I, [P, C.POINTER(P)]),
    "gcnhip_graph_launch": (I, [P, P]),
    "gcnhip_graph_exec_destroy": (I, [P]),
    "gcnhip_event_create": (I, [C.POINTER(P)]),
    "gcnhip_event_create_sync": (I, [C.POINTER(P)]),
    "gcnhip_event_destroy": (I, [P]),
    "gcnhip_event_record": (I, [P, P]),
    "gcnhip_stream_wait_event": (I, [P, P]),
    "gcnhip_event_elapsed_ms": (I, [P, P, C.POINTER(F)]),
    "gcnhip_event_sync": (I, [P]),
}


def _declare_gcnhip(lib):
    for name, (res, args) in GCNHIP_SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args


class HostParams(C.Structure):
    _fields_ = [("num_nodes", I), ("input_dim", I), ("hidden_dim", I), ("output_dim", I),
                ("dropout", F), ("learning_rate", F), ("weight_decay", F), ("epochs", I), ("early_stopping", I)]


ALLGATHER_FN = C.CFUNCTYPE(None, P, C.POINTER(C.c_float), C.c_size_t)
ALLREDUCE_FN = C.CFUNCTYPE(None, P, C.POINTER(C.c_double), C.c_size_t)
PP = C.POINTER(P)

GCNHOST_SYMBOLS = {
    "gcnhost_last_error": (C.c_char_p, []),
    "gcnhost_params_default": (HostParams, []),
    "gcnhost_nccl_unique_id": (I, [C.c_char_p]),
    "gcnhost_model_create": (I, [PP, C.POINTER(HostParams), P, P, P, P, P, P, P, C.c_long, I, I, I, I, C.c_char_p,
                                 ALLGATHER_FN, ALLREDUCE_FN, P]),
    "gcnhost_model_destroy": (I, [P]),
    "gcnhost_model_train_epoch": (I, [P, C.POINTER(F), C.POINTER(F)]),
    "gcnhost_model_eval": (I, [P, I, C.POINTER(F), C.POINTER(F)]),
    "gcnhost_model_run_epochs": (I, [P, I, P]),
    "gcnhost_model_run": (I, [P]),
    "gcnhost_model_sync": (I, [P]),
    "gcnhost_model_info": (I, [P, C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(I64)]),
    "gcnhost_model_schedule": (I, [P, C.POINTER(I), C.POINTER(I)]),
    "gcnhost_model_slice_floats": (I, [P, C.POINTER(I)]),
    "gcnhost_model_transport": (I, [P, C.POINTER(I), C.c_char_p]),
    "gcnhost_model_row_ids": (I, [P, P, C.POINTER(I)]),
    "gcnhost_model_row_scale": (I, [P, P, C.POINTER(I)]),
    "gcnhost_model_get_var": (I, [P, I, I, P, C.POINTER(I), C.POINTER(I)]),
    "gcnhost_model_set_weights": (I, [P, P, P]),
    "gcnhost_model_timer": (I, [P, I, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "gcnhost_model_timers_reset": (I, [P]),
    "gcnhost_model_set_timers": (I, [P, I]),
    "gcnhost_dataset_load": (I, [PP, C.c_char_p, C.c_char_p, C.POINTER(HostParams)]),
    "gcnhost_dataset_arrays": (I, [P, PP, PP, C.POINTER(I64), PP, PP, PP, C.POINTER(I64), PP, C.POINTER(I64), PP, C.POINTER(I64)]),
    "gcnhost_dataset_save_binary": (I, [P, C.POINTER(HostParams), C.c_char_p]),
    "gcnhost_dataset_free": (I, [P]),
    "gcnhost_rccl_selftest": (I, [I]),
    "gcnhost_rccl_selftest_world": (I, [I, I, I, C.c_char_p]),
    "gcnhost_rccl_collective_us": (I, [I, I, I, C.c_char_p, C.c_long, C.c_long, I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gcnhost_halo_selftest_host": (I, [I, I, I, ALLGATHER_FN, ALLREDUCE_FN, P]),
    "gcnhost_partition": (I, [P, I, I, P, C.POINTER(I)]),
    "gcnhost_local_graph": (I, [P, P, I, I, I, P, P, P, C.POINTER(I), C.POINTER(I), C.POINTER(I64)]),
    "gcnhost_plan_create": (I, [PP, P, P, I, I, I, I]),
    "gcnhost_plan_info": (I, [P, C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(C.c_double),
                              C.POINTER(I64), C.POINTER(I64), C.POINTER(I64)]),
    "gcnhost_plan_arrays": (I, [P, PP, PP, PP, PP, PP, PP, PP, PP]),
    "gcnhost_plan_free": (I, [P]),
    "gcnhost_model_exchange": (I, [P, C.POINTER(I), C.POINTER(I64), C.POINTER(I64), C.POINTER(I), C.POINTER(C.c_double)]),
    "gcnhost_glorot": (I, [P, I, I, I, C.c_long, I]),
    "gcnhost_host_masks": (I, [P, I64, F, C.c_long, I64]),
    "gcnhost_rmat_graph": (I, [I, I, U64, PP, PP, C.POINTER(I64)]),
    "gcnhost_choose_node_order": (I, [P, P, I, I, I, P, C.POINTER(I), C.POINTER(C.c_double), C.POINTER(I64), C.POINTER(C.c_double), C.POINTER(I64), C.POINTER(I64)]),
    "gcnhost_structure_groups": (I, [P, P, I, P, C.POINTER(I), C.POINTER(I), C.POINTER(C.c_double), C.POINTER(I)]),
    "gcnhost_free_array": (None, [P]),
}


def _declare_gcnhost(lib):
    for name, (res, args) in GCNHOST_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
