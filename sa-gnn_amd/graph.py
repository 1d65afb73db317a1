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


def interval_pair(sub_mat, device, tuning=None):
    """(subAdj[k], subTpAdj[k]) for one interval matrix (reference model.py:230-237)."""
    return (IntervalAdj.from_scipy(sub_mat, device, tuning),
            IntervalAdj.from_scipy(transpose(sub_mat), device, tuning))
