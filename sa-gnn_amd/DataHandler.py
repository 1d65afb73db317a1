"""Data loader with the reference's contract (reference DataHandler.py:71-167):
`DataHandler().LoadData()` fills `.trnMat`, `.subMat[T]`, `.timeMat`, `.sequence`, `.tstInt`,
`.tstUsrs`, `.test_dict`, `.maxTime` and injects `args.user` / `args.item`.

Input files are the reference's own pickles (`trn_mat_time` = [trnMat, subMat[T], timeMat],
`sequence`, `tst_int`, `test_dict`), supplied by the user under Datasets/<name>/ — none ship with
the reference (`.MISSING_LARGE_BLOBS`). `from_memory` takes the same objects directly (synthetic
runs). The O(nnz) Python normalisation loop of transToLsts is not reproduced: its result is dead
(model.py:84); `transToLsts` below returns the same (indices, data, shape) triple vectorised.
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import scipy.sparse as sp
from scipy.sparse import csr_matrix

from .Params import args
from .graph import transpose  # noqa: F401  (re-exported: reference DataHandler.transpose)


def transToLsts(mat, mask=False, norm=False):
    """reference DataHandler.transToLsts (DataHandler.py:47-69): COO (row, col) int32 pairs in
    stored order, int32 data, shape; one phantom (0, 0) edge for an empty matrix."""
    shape = [mat.shape[0], mat.shape[1]]
    coo = sp.coo_matrix(mat)
    indices = np.stack([coo.row, coo.col], axis=1).astype(np.int32).reshape(-1, 2)
    data = coo.data.astype(np.int32)
    if norm and data.size:
        row_d = np.asarray(1 / (np.sqrt(np.sum(mat, axis=1) + 1e-8) + 1e-8)).reshape(-1)
        col_d = np.asarray(1 / (np.sqrt(np.sum(mat, axis=0) + 1e-8) + 1e-8)).reshape(-1)
        data = np.trunc(data.astype(np.float64) * row_d[indices[:, 0]] * col_d[indices[:, 1]]).astype(np.int32)
    if mask:
        data = (data * ((np.random.uniform(size=data.shape) > 0.5) * 1.0))
    if indices.shape[0] == 0:
        indices = np.array([[0, 0]], dtype=np.int32)
        data = np.array([0], dtype=np.int32)
    return indices, data, shape


def negSamp(temLabel, sampSize, nodeNum, trnPos, item_with_pop=None):
    """reference DataHandler.negSamp (DataHandler.py:28-41): rejection-sample items the user has
    not interacted with."""
    negset = []
    while len(negset) < sampSize:
        cand = np.random.choice(nodeNum)
        if temLabel[cand] == 0 and cand not in trnPos:
            negset.append(cand)
    return negset


def rating_matrix_from_sequence(user_seq, num_users, num_items):
    """reference generate_rating_matrix_test (DataHandler.py:109-125): binary CSR over `sequence`."""
    lens = np.fromiter((len(s) for s in user_seq), dtype=np.int64, count=len(user_seq))
    row = np.repeat(np.arange(len(user_seq)), lens)
    col = np.fromiter((i for s in user_seq for i in s), dtype=np.int64, count=int(lens.sum()))
    return csr_matrix((np.ones(len(col), dtype=np.int64), (row, col)), shape=(num_users, num_items))


class DataHandler:
    def __init__(self, root: str = "./Datasets"):
        name = "Yelp" if args.data == "yelp" else args.data     # reference DataHandler.py:73-80
        self.predir = os.path.join(root, name) + "/"
        self.trnfile = self.predir + "trn_mat_time"
        self.tstfile = self.predir + "tst_int"
        self.sequencefile = self.predir + "sequence"
        self.test_dictfile = self.predir + "test_dict"

    @classmethod
    def from_memory(cls, trn_mat_time, sequence, tst_int=None, test_dict=None):
        self = cls.__new__(cls)
        self.predir = None
        self._finish(trn_mat_time, sequence, tst_int, test_dict)
        return self

    def LoadData(self):
        trn = self.predir + ("noise_%.2f" % args.percent) if args.percent > 1e-8 else self.trnfile
        with open(trn, "rb") as fs:
            trn_mat_time = pickle.load(fs)
        with open(self.tstfile, "rb") as fs:
            tst_int = pickle.load(fs)
        with open(self.sequencefile, "rb") as fs:
            sequence = pickle.load(fs)
        test_dict = None
        if os.path.isfile(self.test_dictfile):
            with open(self.test_dictfile, "rb") as fs:
                test_dict = pickle.load(fs)
        self._finish(trn_mat_time, sequence, tst_int, test_dict)

    def _finish(self, trn_mat_time, sequence, tst_int, test_dict):
        self.sequence = sequence
        if test_dict is not None:
            self.test_dict = test_dict
        args.user, args.item = trn_mat_time[0].shape                      # DataHandler.py:126
        self.trnMat = rating_matrix_from_sequence(sequence, args.user, args.item)
        self.subMat = trn_mat_time[1]
        self.timeMat = trn_mat_time[2]
        if tst_int is None:
            tst_int = [None] * args.user
        self.tstInt = np.array(tst_int, dtype=object)
        self.tstUsrs = np.flatnonzero(np.array([t is not None for t in tst_int]))
        self.prepareGlobalData()

    def prepareGlobalData(self):
        self.maxTime = 1                                                   # DataHandler.py:164
        self.item_with_pop = []
