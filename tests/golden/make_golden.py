"""Generates the committed golden fixtures. Run from the repo root IN THE BUILD CONTAINER:

    python tests/golden/make_golden.py

Part 1 imports the reference's own DataHandler.py from /root/reference (it imports cleanly under
numpy 2.2 / scipy 1.15 once sys.argv is neutralised — Params.py:52 parses argv at import) and
records what transToLsts / transpose return on seeded tiny matrices, including the quirk cases
(empty matrix, trailing empty rows, duplicated entries, explicit zeros, timestamp-sized values).
Only inputs and outputs are stored — no reference source.

Part 2 records the oracle's outputs (oracle/selfgnn_oracle.py) on seeded tiny problems so the GPU
suite can check the HIP path against committed numbers as well as against the live oracle. The
TF-side arithmetic has no reference vectors ("parity unpinned", see the oracle's header).
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def case_matrices():
    rng = np.random.default_rng(20241004)
    cases = {}
    dense = (rng.random((7, 9)) < 0.35) * rng.integers(1, 50, (7, 9))
    cases["random_small"] = sp.csr_matrix(dense.astype(np.intc))
    ts = (rng.random((11, 6)) < 0.4) * rng.integers(1_400_000_000, 1_500_000_000, (11, 6))
    cases["timestamps"] = sp.csr_matrix(ts.astype(np.intc))
    cases["empty"] = sp.csr_matrix((5, 4), dtype=np.intc)
    tr = np.zeros((12, 5), dtype=np.intc)
    tr[0, 1] = 3
    tr[2, 4] = 9
    tr[2, 0] = 1
    cases["trailing_empty_rows"] = sp.csr_matrix(tr)
    # duplicated (row, col) entries and an explicit zero, kept as stored
    indptr = np.array([0, 3, 3, 5, 6], dtype=np.int32)
    indices = np.array([1, 1, 3, 0, 0, 2], dtype=np.int32)
    data = np.array([4, 5, 0, 7, 8, 2], dtype=np.intc)
    cases["duplicates_and_zero"] = sp.csr_matrix((data, indices, indptr), shape=(4, 4))
    one = np.zeros((1, 3), dtype=np.intc)
    one[0, 2] = 6
    cases["single_row"] = sp.csr_matrix(one)
    return cases


def part1_reference_lists():
    argv = sys.argv
    sys.argv = [argv[0]]
    sys.path.insert(0, "/root/reference")
    import DataHandler as RefDH  # the reference's own module
    sys.argv = argv
    out = {}
    for name, m in case_matrices().items():
        out[f"{name}/indptr"] = m.indptr.astype(np.int32)
        out[f"{name}/indices"] = m.indices.astype(np.int32)
        out[f"{name}/data"] = m.data.astype(np.int32)
        out[f"{name}/shape"] = np.array(m.shape, dtype=np.int64)
        for norm in (False, True):
            try:
                idx, dat, shp = RefDH.transToLsts(m, norm=norm)
            except IndexError:
                # a 1-row (or 1-column) matrix makes np.squeeze return a 0-d normaliser and the
                # reference's norm loop raises (DataHandler.py:54-59); recorded, not reproduced
                out[f"{name}/norm{int(norm)}_raises"] = np.array(1)
                continue
            out[f"{name}/fwd_norm{int(norm)}/idx"] = idx
            out[f"{name}/fwd_norm{int(norm)}/data"] = dat
            idx, dat, shp_t = RefDH.transToLsts(RefDH.transpose(m), norm=norm)
            out[f"{name}/tp_norm{int(norm)}/idx"] = idx
            out[f"{name}/tp_norm{int(norm)}/data"] = dat
        out[f"{name}/tp_shape"] = np.array(shp_t, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "reference_translsts.npz"), **out)
    print("wrote reference_translsts.npz with", len(out), "arrays")


def part2_oracle_vectors():
    from oracle import selfgnn_oracle as O
    out = {}
    for d in (32, 64, 128):
        rng = np.random.default_rng(7000 + d)
        U, I, T, L, heads = 37, 53, 3, 2, 16
        adjs, tps = [], []
        for k in range(T):
            dense = (rng.random((U, I)) < 0.08).astype(np.intc)
            if k == 1:
                dense[20:, :] = 0          # trailing empty user rows (< 100)
            m = sp.csr_matrix(dense)
            adjs.append(O.trans_to_lsts(m)[0])
            tps.append(O.trans_to_lsts(O.transpose(m))[0])
            out[f"d{d}/adj{k}"] = adjs[-1]
            out[f"d{d}/tp{k}"] = tps[-1]
        ue = O.xavier_uniform((T, U, d), rng) * 20   # scaled so leaky/LSTM see O(1) values
        ie = O.xavier_uniform((T, I, d), rng) * 20
        uv, iv = O.gnn_stack(ue, ie, adjs, tps, L, 0.5)
        p = O.init_fusion_params(d, rng)
        fu = O.interval_fusion(uv, p, heads)
        fi = O.interval_fusion(iv, p, heads)
        out[f"d{d}/uEmbed"], out[f"d{d}/iEmbed"] = ue, ie
        out[f"d{d}/user_vector"], out[f"d{d}/item_vector"] = uv, iv
        out[f"d{d}/lstm_user"] = O.basic_lstm(uv, p["lstm_W"], p["lstm_b"])
        out[f"d{d}/final_user"], out[f"d{d}/final_item"] = fu, fi
        for k, v in p.items():
            out[f"d{d}/p/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "oracle_tiny.npz"), **out)
    print("wrote oracle_tiny.npz with", len(out), "arrays")


if __name__ == "__main__":
    part1_reference_lists()
    part2_oracle_vectors()
