"""CPU: the arithmetic csrc/hvn_conv_x3.hip relies on, restated in numpy (no GPU, no kernel) -- the three claims DESIGN section 4.1 makes.

  1. An fp32 number is the EXACT sum of three bf16 numbers: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even), every
     difference exact in fp32.  (`engine.split_bf16x3` is the host form the weights go through; the kernel's `split3` is the same recipe.)
     DOMAIN: 2^-110 <= |x| < 3.38e38 (and 0).  Below, the low planes underflow bf16's denormal grid (absolute error < 2^-133: harmless);
     within 0.3 % of FLT_MAX the high plane rounds to infinity (such values overflow any fp32 accumulation as well).
  2. A product of two bf16 numbers is exact in fp32 (8 x 8 significand bits), so the nine partial products of a product carry no rounding
     of their own: only the fp32 accumulation rounds, as it does for the fp32 MFMA.
  3. Dropping m*l, l*m and l*l (the 6-term form) perturbs a product by <= ~2^-23 of its magnitude, and over a K-term dot product that is
     the size of what fp32 accumulation itself loses: against float64, the 6-term and 9-term results sit as close as a plain fp32
     accumulation of the unsplit operands does (random data AND cancellation-heavy data)."""
import numpy as np

from hover_net_amd.engine import split_bf16x3


def _f(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def _planes(x):
    return [_f(p) for p in split_bf16x3(x)]


def test_three_bf16_planes_sum_back_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 1, 200000), rng.normal(0, 1e-3, 50000), rng.normal(0, 300, 50000), rng.uniform(-1, 1, 50000) * 2.0 ** rng.integers(-60, 60, 50000),
                        np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -110, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0, 3.37e38, -3.37e38])]).astype(np.float32)
    h, m, l = _planes(x)
    assert np.array_equal((h + m) + l, x)                       # exact in fp32 arithmetic
    assert np.array_equal(h.astype(np.float64) + m + l, x.astype(np.float64))
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)
    # the stated domain limits are where the claim stops holding, and how: tiny values lose < 2^-133, values at FLT_MAX lose everything
    tiny = (rng.uniform(1, 2, 20000) * 2.0 ** -120).astype(np.float32)
    th, tm, tl = _planes(tiny)
    assert not np.array_equal((th + tm) + tl, tiny) and np.abs(((th.astype(np.float64) + tm) + tl) - tiny).max() < 2.0 ** -133
    with np.errstate(invalid="ignore", over="ignore"):
        assert np.isinf(_planes(np.array([3.4e38], np.float32))[0][0])


def test_partial_products_are_exact_in_fp32():
    rng = np.random.default_rng(1)
    a, b = rng.normal(0, 3, 100000).astype(np.float32), rng.normal(0, 0.05, 100000).astype(np.float32)
    for pa in _planes(a):
        for pb in _planes(b):
            assert np.array_equal((pa * pb).astype(np.float64), pa.astype(np.float64) * pb.astype(np.float64))


def _dot_terms(a, b, terms):
    """fp32 accumulation (sequential over k, like an MFMA accumulator chain) of the partial products of a[k] * b[k]."""
    pa, pb = _planes(a), _planes(b)
    pairs = [(i, j) for s in range(4, -1, -1) for i in range(2, -1, -1) for j in [s - i] if 0 <= j <= 2 and (terms == 9 or i + j <= 2)]
    acc = np.zeros(a.shape[:-1], np.float32)
    for k0 in range(0, a.shape[-1], 16):                        # one 32x32x16 MFMA block after the other; inside: the kernel's pair order
        for i, j in pairs:
            acc = acc + np.sum((pa[i][..., k0:k0 + 16] * pb[j][..., k0:k0 + 16]).astype(np.float64), -1).astype(np.float32)
    return acc


def _dot_fp32(a, b):
    acc = np.zeros(a.shape[:-1], np.float32)
    for k in range(a.shape[-1]):
        acc = acc + a[..., k] * b[..., k]
    return acc


def test_six_terms_sit_where_fp32_accumulation_sits():
    rng = np.random.default_rng(2)
    for name, gen in (("random", lambda: (rng.normal(0, 1, (4000, 1024)), rng.normal(0, 0.03, (4000, 1024)))),
                      ("cancelling", lambda: (np.abs(rng.normal(0, 10, (4000, 1024))), rng.normal(0, 0.03, (4000, 1024)) * np.where(np.arange(1024) % 2, 1, -1)))):
        a, b = (v.astype(np.float32) for v in gen())
        ref = np.sum(a.astype(np.float64) * b.astype(np.float64), -1)
        scale = np.sum(np.abs(a.astype(np.float64) * b.astype(np.float64)), -1)
        e32 = np.abs(_dot_fp32(a, b) - ref) / scale
        e9 = np.abs(_dot_terms(a, b, 9) - ref) / scale
        e6 = np.abs(_dot_terms(a, b, 6) - ref) / scale
        # errors relative to sum |a_k b_k|: all three are a few 2^-24 * sqrt(K)-ish; the split forms must not be worse than plain fp32 by more than 2x
        assert np.median(e9) <= 2.0 * np.median(e32) + 1e-9 and np.percentile(e9, 99) <= 2.0 * np.percentile(e32, 99) + 1e-9, (name, np.median(e9), np.median(e32))
        assert np.median(e6) <= 2.0 * np.median(e32) + 1e-9 and np.percentile(e6, 99) <= 2.0 * np.percentile(e32, 99) + 1e-9, (name, np.median(e6), np.median(e32))
        assert np.max(e6) <= 64 * 2.0 ** -24                      # and never more than a few roundings' worth of the products' magnitude
        print("%s: median rel. error vs float64: fp32 %.2e, 9 terms %.2e, 6 terms %.2e" % (name, np.median(e32), np.median(e9), np.median(e6)))
