"""ctypes binding of oracle/c/tf1_path.c — TEST / CPU-BASELINE INFRASTRUCTURE ONLY.

Built by `make -C oracle` (also by __graft_entry__.build()). See the C file for what it restates
(reference model.py:80-92 as TF1 executes it on a CPU)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtf1path.so")
_lib = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.tf1_message_propagate.restype = ctypes.c_int
        _lib.tf1_message_propagate.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _lib.tf1_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(lib().tf1_max_threads())


def message_propagate(indices: np.ndarray, srclats: np.ndarray, n_out: int, leaky: float,
                      threads: int = 0, strict: bool = False, scratch: np.ndarray | None = None):
    """indices [nnz, 2] int32 (row, col), srclats [M, d] float32 -> [n_out, d] float32."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    srclats = np.ascontiguousarray(srclats, dtype=np.float32)
    nnz, d = indices.shape[0], srclats.shape[1]
    out = np.empty((n_out, d), dtype=np.float32)
    if scratch is None:
        scratch = np.empty((max(nnz, 1), d), dtype=np.float32)
    rc = lib().tf1_message_propagate(indices.ctypes.data, nnz, srclats.ctypes.data, d, n_out,
                                     float(leaky), out.ctypes.data, scratch.ctypes.data,
                                     int(threads), int(strict))
    if rc != 0:
        raise IndexError("n_out exceeds max(row)+1+100 (TF-CPU InvalidArgument)")
    return out
