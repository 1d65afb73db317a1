"""ctypes binding of libsagnn.so (include/sagnn.h). Fails loudly when the library is absent:
the product path has no CPU or eager-PyTorch fallback."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAGNN_LIB points at an alternative build (diagnostic variants under scratch/); default: in-tree
LIB_PATH = os.environ.get("SAGNN_LIB") or os.path.join(_HERE, "lib", "libsagnn.so")


class SagnnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsagnn error {code}: {msg}")
        self.code = code


class Tuning(ctypes.Structure):
    _fields_ = [("short_thresh", c_int32), ("long_thresh", c_int32), ("chunk_edges", c_int32),
                ("reserved", c_int32)]


class SpmmEpilogue(ctypes.Structure):
    _fields_ = [("leaky", c_float), ("residual", c_void_p), ("ldr", c_int64), ("out", c_void_p),
                ("ldo", c_int64), ("acc_in", c_void_p), ("ld_acc_in", c_int64), ("acc_out", c_void_p),
                ("ld_acc_out", c_int64), ("mask_out", c_void_p), ("mask_in", c_void_p),
                ("out2", c_void_p), ("ldo2", c_int64), ("slope2", c_float), ("acc_in2", c_void_p),
                ("ld_acc_in2", c_int64)]


class PlanInfo(ctypes.Structure):
    _fields_ = [("n_rows", c_int64), ("n_src", c_int64), ("nnz", c_int64),
                ("n_long_rows", c_int64), ("n_chunks", c_int64), ("short_thresh", c_int32),
                ("long_thresh", c_int32), ("chunk_edges", c_int32), ("max_degree", c_int32),
                ("on_device", c_int32), ("reserved", c_int32)]


# name -> (restype, argtypes); must list every symbol include/sagnn.h declares
SIGNATURES = {
    "sagnn_version": (c_int, []),
    "sagnn_set_engine": (c_int, [c_int]),
    "sagnn_get_engine": (c_int, []),
    "sagnn_range_redo_count": (c_int, [POINTER(c_int64), c_int]),
    "sagnn_last_error": (c_size_t, [c_char_p, c_size_t]),
    "sagnn_profile_enable": (c_int, [c_int]),
    "sagnn_profile_read": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(c_int)]),
    "sagnn_csr_check_host": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64]),
    "sagnn_spmm_plan_create": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                       POINTER(Tuning), POINTER(c_void_p)]),
    "sagnn_spmm_plan_destroy": (c_int, [c_void_p]),
    "sagnn_spmm_plan_get_info": (c_int, [c_void_p, POINTER(PlanInfo)]),
    "sagnn_spmm_plan_copy_chunks": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "sagnn_spmm_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "sagnn_spmm_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_float,
                               c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                               c_size_t, c_void_p]),
    "sagnn_spmm_ex_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, POINTER(SpmmEpilogue), c_void_p, c_size_t,
                                  c_void_p]),
    "sagnn_mask_scale_f32": (c_int, [c_void_p, c_int64, c_void_p, c_float, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "sagnn_gnn_interval_ex_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                                          c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                          c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sagnn_gnn_interval_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                                           c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                           c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "sagnn_spmm_batch_create": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_void_p)]),
    "sagnn_spmm_batch_destroy": (c_int, [c_void_p]),
    "sagnn_spmm_batch_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "sagnn_gnn_stack_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_float,
                                    c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "sagnn_gnn_stack_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                        c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                        c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "sagnn_gnn_interval_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                       c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "sagnn_lstm_fwd_train_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_float,
                                         c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sagnn_mhsa_wide_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "sagnn_ln_mhsa_mean_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "sagnn_ln_mhsa_mean_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_void_p, c_size_t, c_void_p]),
    "sagnn_mhsa_mean_wide_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                                         c_void_p]),
    "sagnn_attn_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "sagnn_layernorm_td_bwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p,
                                           c_float, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sagnn_lstm_bwd_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                        c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "sagnn_attn_bwd_front_supported": (c_int, [c_int, c_int, c_int]),
    "sagnn_attn_bwd_front_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sagnn_attn_bwd_tail_supported": (c_int, [c_int]),
    "sagnn_attn_bwd_tail_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sagnn_lstm_bwd_supported": (c_int, [c_int]),
    "sagnn_lstm_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "sagnn_lstm_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "sagnn_lstm_bwd_ws_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sagnn_leaky_add_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "sagnn_pair_score_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                     c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_void_p]),
    "sagnn_pair_score_bwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_int, c_void_p]),
    "sagnn_prod_leaky_sum_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p,
                                         c_int64, c_int, c_void_p]),
    "sagnn_prod_leaky_sum_bwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_float,
                                             c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sagnn_meta_features_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int,
                                        c_void_p]),
    "sagnn_meta_features_bwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int64, c_int, c_void_p]),
    "sagnn_leaky_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_void_p]),
    "sagnn_rowdot_sigmoid_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sagnn_rowdot_sigmoid_bwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                             c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sagnn_hinge_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sagnn_mul_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sagnn_adam_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                                    c_float, c_float, c_int64, c_void_p]),
    "sagnn_adam_multi_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                     c_float, c_float, c_int64, c_void_p]),
    "sagnn_dense_nn_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_int, c_void_p]),
    "sagnn_dense_tn_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p]),
    "sagnn_dense_tn_seg_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    "sagnn_lstm_fwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p,
                                   c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "sagnn_lstm_fwd_state_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_float,
                                         c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "sagnn_layernorm_td_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p,
                                       c_float, c_void_p, c_int64, c_void_p]),
    "sagnn_mhsa_mean_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int64, c_void_p]),
    "sagnn_interval_fusion_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int64, c_void_p, c_size_t, c_void_p]),
    "sagnn_interval_fusion_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C sa-gnn_amd/csrc`). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    buf = ctypes.create_string_buffer(1024)
    load().sagnn_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        raise SagnnError(rc, last_error())
