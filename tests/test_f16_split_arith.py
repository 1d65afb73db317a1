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


def test_top_binade_ties_overflow_the_scaled_residual():
    """|v| in [32768, 65504] fits a head, but the residual reaches 16 there and 16 * 4096 = 65536 is beyond f16:
    the kernels therefore send |v| >= 32768 (kF16Lim) to their fp32 pass, not |v| > 65504."""
    with np.errstate(over="ignore"):
        for v in (32784.0, 49168.0):
            v1, v2 = split(np.float32(v))
            assert np.isfinite(v1) and np.isinf(v2), v
        v = np.float32(32767.99)                                  # the largest values still on the fast path
        v1, v2 = split(v)
        assert np.isfinite(v1) and np.isfinite(v2) and abs(float(v1) + float(v2) / 4096.0 - float(v)) <= 2.0 ** -23 * float(v)
        below = np.nextafter(np.float32(32768.0), np.float32(0.0))
        v1, v2 = split(below)
        assert np.isfinite(v1) and np.isfinite(v2)


def test_the_bottom_of_the_window_is_an_absolute_floor():
    """The split's error is max(2^-23 |v|, 2^-37): operands that are small AS A WHOLE lose relative accuracy (a
    gradient row of 1e-9: 4e-3 of its largest entry, 1e-12: everything). This is what the per-segment range check of
    the forward kernels (2^-18) and the power-of-two row scaling of the attention-backward tail exist for."""
    rng = np.random.default_rng(2)
    rel = {}
    for s in (1e-4, 1e-6, 1e-9, 1e-12):
        v = (rng.standard_normal(4096) * s).astype(np.float32)
        v1, v2 = split(v)
        rec = v1.astype(np.float64) + v2.astype(np.float64) / 4096.0
        rel[s] = np.abs(rec - v).max() / np.abs(v).max()
    assert rel[1e-4] <= 2.0 ** -22 and rel[1e-6] <= 1e-5
    assert rel[1e-9] > 1e-4 and rel[1e-12] > 0.5                 # outside the 1e-4 tolerance: must not reach the matrix cores unscaled
    # what the forward kernels leave on the fast path: every aligned 4-element segment's max >= 2^-18
    v = (rng.standard_normal((50_000, 4)) * np.exp(rng.uniform(np.log(2.0 ** -18), 2, (50_000, 1)))).astype(np.float32)
    seg = np.abs(v).max(axis=1, keepdims=True)
    v = v[(seg >= 2.0 ** -18)[:, 0]]
    seg = np.abs(v).max(axis=1, keepdims=True)
    v1, v2 = split(v)
    rec = v1.astype(np.float64) + v2.astype(np.float64) / 4096.0
    assert (np.abs(rec - v) <= 2.0 ** -19 * seg).all()
    assert (np.abs(rec - v)[(seg >= 2.0 ** -14)[:, 0]] <= 2.0 ** -23 * seg[(seg >= 2.0 ** -14)[:, 0]]).all()


def _biased_exponent(m):
    return (np.asarray(m, dtype=np.float32).view(np.uint32) >> 23).astype(np.int64) & 0xFF


def test_power_of_two_row_scaling_of_the_attention_backward_tail():
    """The arithmetic of attn_bwd_tail_f16.hip's GRADIENT RANGE scheme, restated: gradient row r enters the images as
    dQKV[r] 2^(127 - k_r) (k_r = biased exponent of the row's max), y row r as y[r] 2^(k_r - E + 6) with E the largest
    k; dy[r] is descaled per row, dW / db by 2^(E - 133). Rows of 1, 1e-6, 1e-9 and 1e-12 in one product: dy within
    1e-6 of each ROW's own largest entry, dW / db within 1e-6 of the sum of their terms' magnitudes."""
    rng = np.random.default_rng(3)
    rows, d = 512, 64
    sc = rng.choice([1.0, 1e-6, 1e-9, 1e-12], size=rows)
    y = rng.standard_normal((rows, d)).astype(np.float32)
    g = (rng.standard_normal((rows, 3 * d)) * sc[:, None]).astype(np.float32)
    W = (rng.standard_normal((d, 3 * d)) / 8.0).astype(np.float32)
    k = np.clip(_biased_exponent(np.abs(g).max(axis=1)), 1, 253)
    E = int(k.max())
    gs = (g.astype(np.float64) * 2.0 ** (127 - k)[:, None]).astype(np.float32)     # exact: powers of two
    ys = (y.astype(np.float64) * 2.0 ** (k - E + 6)[:, None]).astype(np.float32)
    assert np.abs(gs).max() < 4 and np.abs(ys).max() < 32768

    def rec(v):                                                   # what the two pieces hold
        v1, v2 = split(v)
        return v1.astype(np.float64) + v2.astype(np.float64) / 4096.0
    g_hat, y_hat, W_hat = rec(gs), rec(ys), rec(W)
    dy = (g_hat @ W_hat.T) * 2.0 ** (k - 127)[:, None]
    want_dy = g.astype(np.float64) @ W.astype(np.float64).T
    assert (np.abs(dy - want_dy).max(axis=1) <= 1e-6 * np.abs(want_dy).max(axis=1)).all()
    dW = (y_hat.T @ g_hat) * 2.0 ** (E - 133)
    want_dW = y.astype(np.float64).T @ g.astype(np.float64)
    mag = np.abs(y.astype(np.float64)).T @ np.abs(g.astype(np.float64))
    assert (np.abs(dW - want_dW) <= 1e-6 * mag).all()
    ones = rec((2.0 ** (k - E + 6)).astype(np.float32))          # a power of two: exact in two pieces down to 2^-36
    assert np.array_equal(ones[k - E + 6 >= -36], (2.0 ** (k - E + 6))[k - E + 6 >= -36])
    db = (ones @ g_hat) * 2.0 ** (E - 133)
    assert (np.abs(db - g.astype(np.float64).sum(0)) <= 1e-6 * np.abs(g.astype(np.float64)).sum(0)).all()
    # unscaled, the same operands are outside the tolerance (what round 2 shipped)
    dy_raw = rec(g) @ W_hat.T
    small = sc <= 1e-9
    assert (np.abs(dy_raw - want_dy).max(axis=1)[small] > 1e-4 * np.abs(want_dy).max(axis=1)[small]).all()
