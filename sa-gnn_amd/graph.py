"""Adjacency preparation: scipy CSR (as DataHandler loads it) -> int32 rowptr/colidx on the
device + SpMM plans. Replaces the O(nnz) Python loop of DataHandler.transToLsts
(reference DataHandler.py:47-69, whose normalised values are dead) and the SparseTensor
constants of Recommender.prepareModel (reference model.py:227-237), keeping their edge sets:

  * forward adjacency  = the STORED structure of subMat[k] (duplicates and explicit zeros are
    edges, because edge values are never read: model.py:84-86);
  * transposed adjacency = DataHandler.transpose (DataHandler.py:9-11): scipy's COO->CSR sums
    duplicates, so a duplicated (u, i) counts once there;
  * an empty matrix becomes one phantom edge (0, 0) (DataHandler.py:66-68).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .ops import SpmmPlan


def transpose(mat):
    """Same contract as the reference's DataHandler.transpose (DataHandler.py:9-11)."""
    return sp.csr_matrix(sp.coo_matrix(mat).transpose())


def csr_arrays(mat, phantom_edge: bool = True):
    """(rowptr int32 [n_rows+1], colidx int32 [nnz]) with the edge set transToLsts would emit.

    The reference walks sp.coo_matrix(mat) in stored order and TF's SegmentSum needs the row ids
    sorted; a matrix whose COO rows are not sorted is rejected here the way TF-CPU rejects it."""
    coo = sp.coo_matrix(mat)
    n_rows = int(mat.shape[0])
    row = np.asarray(coo.row, dtype=np.int64)
    col = np.asarray(coo.col, dtype=np.int32)
    if row.size and np.any(np.diff(row) < 0):
        raise ValueError("adjacency rows are not sorted (tf.math.segment_sum would raise)")
    if row.size == 0 and phantom_edge:
        row = np.zeros(1, dtype=np.int64)
        col = np.zeros(1, dtype=np.int32)
    if row.size > np.iinfo(np.int32).max:
        raise ValueError("more than 2^31-1 edges")
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=n_rows), out=rowptr[1:])
    return rowptr.astype(np.int32), np.ascontiguousarray(col)


class IntervalAdj:
    """One direction of one interval graph: what the reference holds as a tf SparseTensor
    (model.py:234 / :236). `.indices`-style access is not offered: the kernels use CSR."""

    def __init__(self, rowptr, colidx, shape, device, tuning=None, validate=True):
        self.dense_shape = (int(shape[0]), int(shape[1]))
        self.plan = SpmmPlan(rowptr, colidx, self.dense_shape[0], self.dense_shape[1], device=device,
                             tuning=tuning, validate=validate)
        self.nnz = self.plan.nnz

    @classmethod
    def from_scipy(cls, mat, device, tuning=None):
        rowptr, colidx = csr_arrays(mat)
        return cls(rowptr, colidx, mat.shape, device, tuning=tuning)


def exact_transpose_arrays(mat):
    """CSR arrays of the transposed STORED pattern of `mat`, multiplicities kept (a duplicated
    (u, i) stays two edges) — the adjoint of csr_arrays(mat), unlike DataHandler.transpose."""
    coo = sp.coo_matrix(mat)
    n_cols = int(mat.shape[1])
    order = np.argsort(np.asarray(coo.col, dtype=np.int64), kind="stable")
    rowptr = np.zeros(n_cols + 1, dtype=np.int64)
    np.cumsum(np.bincount(np.asarray(coo.col, dtype=np.int64), minlength=n_cols), out=rowptr[1:])
    return rowptr.astype(np.int32), np.ascontiguousarray(np.asarray(coo.row, dtype=np.int32)[order])


def interval_pair(sub_mat, device, tuning=None):
    """(subAdj[k], subTpAdj[k]) for one interval matrix (reference model.py:230-237).

    With duplicated stored entries the two patterns are NOT transposes of each other (forward
    counts a duplicate twice, DataHandler.transpose merges it — DataHandler.py:9-11), so the
    backward pass cannot reuse the partner as the adjoint the way it does for canonical matrices.
    The exact adjoints are then built as well and hung on the plans (`partner_adjoint`):
    d/d e_i of the user-side sum gathers through the forward pattern's true transpose (the
    duplicate counts twice, as TF's gather gradient does), d/d e_u of the item-side sum through
    the merged forward pattern."""
    fwd = IntervalAdj.from_scipy(sub_mat, device, tuning)
    tp_mat = transpose(sub_mat)
    tp = IntervalAdj.from_scipy(tp_mat, device, tuning)
    if fwd.nnz != tp.nnz:
        U, I = fwd.dense_shape
        rp, ci = csr_arrays(transpose(tp_mat))                       # (A_tp)^T: rows = users, merged
        fwd.plan.partner_adjoint = SpmmPlan(rp, ci, U, I, device=device, tuning=tuning)
        rp, ci = exact_transpose_arrays(sub_mat)                     # (A_fwd)^T: rows = items, duplicates kept
        tp.plan.partner_adjoint = SpmmPlan(rp, ci, I, U, device=device, tuning=tuning)
    return fwd, tp
