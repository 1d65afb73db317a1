"""CPU restatement of the arithmetic the fusion GEMMs run on the f16 matrix cores (sa-gnn_amd/csrc/f16_split.h):
two round-to-nearest f16 pieces per fp32 operand, three piece products, fp32 accumulation. numpy's float16 is IEEE
binary16 with round-to-nearest-even and denormals, as the hardware conversions and the matrix core are (checked on
the GPU by tools/microbench/f16_split.hip), so the bounds the kernels rely on can be pinned here without a GPU."""
import numpy as np


def split(v):
    v = np.asarray(v, dtype=np.float32)
    v1 = v.astype(np.float16)
    r = (v - v1.astype(np.float32)).astype(np.float32)          # exact in fp32
    assert np.array_equal(r.astype(np.float64), v.astype(np.float64) - v1.astype(np.float64))
    v2 = (r * np.float32(4096.0)).astype(np.float16)            # scaled residual: a normal f16 number
    return v1, v2


def test_two_pieces_represent_fp32_to_2pow_minus_23():
    rng = np.random.default_rng(0)
    v = (rng.standard_normal(200_000) * np.exp(rng.uniform(-8, 8, 200_000))).astype(np.float32)
    v = v[np.abs(v) < 65504]
    v1, v2 = split(v)
    rec = v1.astype(np.float64) + v2.astype(np.float64) / 4096.0
    big = np.abs(v) >= 2.0 ** -13                                # both pieces normal
    assert (np.abs(rec - v)[big] <= 2.0 ** -23 * np.abs(v)[big]).all()
    # below that the head goes denormal and the scaled residual carries the value: absolute error below 2^-37
    assert (np.abs(rec - v)[~big] <= 2.0 ** -37).all()


def test_values_beyond_the_range_are_what_the_kernels_must_catch():
    with np.errstate(over="ignore"):
        assert np.isinf(np.float32(70000.0).astype(np.float16))  # hence the running max + fp32 redo in the kernels
    assert np.isfinite(np.float32(65504.0).astype(np.float16))


def test_three_piece_products_are_closer_than_an_fp32_chain():
    """K = 128 dot products (the LSTM's reduction length at d = 64): head x head + 2^-12 (head x residual + residual x
    head), accumulated in fp32, against float64; the dropped residual x residual term is <= 2^-22 |a b|."""
    rng = np.random.default_rng(1)
    n, k = 4000, 128
    a = (rng.uniform(-1, 1, (n, k)) * 0.3).astype(np.float32)
    b = (rng.uniform(-1, 1, (n, k)) * rng.choice([1.0, 40.0], (n, 1))).astype(np.float32)
    a1, a2 = split(a)
    b1, b2 = split(b)
    f32 = lambda x: x.astype(np.float32)
    hi = np.zeros(n, np.float32)
    lo = np.zeros(n, np.float32)
    chain = np.zeros(n, np.float32)
    for j in range(k):                                           # piece products are exact in fp32 (11 x 11 bits)
        hi = hi + f32(a1[:, j]) * f32(b1[:, j])
        lo = lo + (f32(a1[:, j]) * f32(b2[:, j]) + f32(a2[:, j]) * f32(b1[:, j]))
        chain = (chain.astype(np.float64) + a[:, j].astype(np.float64) * b[:, j].astype(np.float64)).astype(np.float32)  # fmaf
    got = (lo.astype(np.float64) / 4096.0 + hi.astype(np.float64)).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    sabs = np.abs(a.astype(np.float64) * b.astype(np.float64)).sum(1)
    e_split = (np.abs(got - ref) / sabs).max()
    e_chain = (np.abs(chain - ref) / sabs).max()
    assert e_split <= 3e-7                                        # fp32-grade: a few 2^-24 of the sum of magnitudes
    assert e_split <= 2.0 * e_chain + 1e-8                        # and not worse than an fmaf chain of the same length
