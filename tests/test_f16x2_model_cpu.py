"""The opt-in fp16x2 arithmetic (include/acamd.h AC_GEMM_F16X2; csrc/common.h split2h) restated in numpy, so that what the mode
promises a priori is checked without a GPU: the two-term split's representation error, the three-product form's error per
dot product (operand rounding + the dropped l.l term) against the exact product, the range at which an operand overflows, and
that the error sits inside the K 2^-24 bound of an fp32 dot product for the encoder's K.  The GPU tests
(tests/test_gemm_f16x2_gpu.py) check the kernels against the same formulas bit for bit (planes) and per element (products)."""
import numpy as np

ACT_LOG2, W_LOG2 = 6, 10


def split(x, log2):
    xs = (x.astype(np.float32) * np.float32(2.0 ** log2)).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        h = xs.astype(np.float16)
        l = (xs - h.astype(np.float32)).astype(np.float16)
    return xs.astype(np.float64), h.astype(np.float64), l.astype(np.float64)


def test_two_term_split_error_and_range():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-8, 2.9, 200000)).astype(np.float32)
    x = x[np.abs(x) < 1023.0]
    xs, h, l = split(x, ACT_LOG2)
    err = np.abs(xs - h - l)
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(xs), 2.0 ** -25))      # 22 bits, or fp16's subnormal spacing for l
    assert np.all(err / 2.0 ** ACT_LOG2 <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -31))
    for v, ok in ((1023.0, True), (1023.4, True), (1023.75, False), (1024.0, False), (-5000.0, False)):   # fp16 max 65504, RNE to inf from 65520
        _, h, l = split(np.array([v], np.float32), ACT_LOG2)
        assert bool(np.isfinite(h[0]) and np.isfinite(l[0])) == ok, v
        if not ok:
            assert not np.isfinite(h[0] * 0.5 + l[0] * 0.5)                       # inf - inf: NaN in every product sum it feeds
    for v, ok in ((63.9, True), (63.98, True), (63.99, False), (80.0, False)):
        _, h, l = split(np.array([v], np.float32), W_LOG2)
        assert bool(np.isfinite(h[0])) == ok, v


def test_three_product_form_is_inside_its_bound_and_the_fp32_dot_product_bound():
    rng = np.random.default_rng(1)
    for K in (64, 768, 3072):
        A = rng.standard_normal((64, K)).astype(np.float32)
        A[:, ::97] *= 20.0
        W = (rng.standard_normal((48, K)) / np.sqrt(K)).astype(np.float32)
        _, ha, la = split(A, ACT_LOG2)
        _, hw, lw = split(W, W_LOG2)
        got = (la @ hw.T + ha @ lw.T + ha @ hw.T) * 2.0 ** -(ACT_LOG2 + W_LOG2)     # the three products, exact accumulation
        want = A.astype(np.float64) @ W.astype(np.float64).T
        S = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T
        e = np.abs(got - want)
        assert np.all(e <= 3.0 * 2.0 ** -22 * S + 1e-12), (K, (e / S).max())        # two operand roundings + l.l
        assert np.all(e <= K * 2.0 ** -24 * S)                                      # what an fp32 fma chain may lose (K >= 12)
        assert (e / S).max() < 2.0 ** -22                                           # in practice the roundings average out
