"""The C oracle (oracle/hvn_oracle.c) against the golden vectors produced by the
reference's own post_proc.py under real scipy + scikit-image
(oracle/make_golden_postproc.py).  Bit-exact, label values included."""
import glob
import os

import numpy as np
import pytest

from oracle import postproc as O

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pp_*.npz")))


def test_golden_present():
    assert len(CASES) >= 8


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[3:-4] for p in CASES])
def test_oracle_matches_reference_golden(path):
    z = np.load(path)
    pred, inst = z["pred"], z["inst"]
    for i in range(pred.shape[0]):
        got = O.proc_np_hv(pred[i][..., -3:])
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, inst[i])


def test_batch_helper_matches_single():
    z = np.load([p for p in CASES if p.endswith("pp_s80t.npz")][0])
    got = O.proc_batch(z["pred"])
    np.testing.assert_array_equal(got, z["inst"])


def test_sobel_kernels_known_answer():
    # OpenCV's getSobelKernels recurrence: binomial row / first difference of it
    import ctypes
    from math import comb

    k = np.zeros(21)
    O.lib().hvn_o_sobel_kernel21(0, k.ctypes.data_as(ctypes.c_void_p))
    assert [int(v) for v in k] == [comb(20, i) for i in range(21)]
    O.lib().hvn_o_sobel_kernel21(1, k.ctypes.data_as(ctypes.c_void_p))
    assert [int(v) for v in k] == [comb(19, i - 1) - comb(19, i) if 0 < i < 20 else (-1 if i == 0 else 1) for i in range(21)]


def test_sobel_small_ksize_structure():
    # derivative of a linear ramp is constant: sum_k k * deriv[k] * 2^20-normalised
    x = np.tile(np.arange(64, dtype=np.float32), (64, 1))
    g = O.sobel21(x, 1)
    inner = g[12:-12, 12:-12]
    assert np.all(inner == inner[0, 0]) and inner[0, 0] > 0
    assert np.all(O.sobel21(x, 0)[12:-12, 12:-12] == 0)


def test_label4_matches_scipy():
    from scipy import ndimage

    rng = np.random.default_rng(0)
    for p in (0.3, 0.5, 0.6, 0.8):
        b = rng.uniform(size=(37, 53)) < p
        want, n = ndimage.label(b)
        got, m = O.label4(b)
        assert n == m
        np.testing.assert_array_equal(got, want)


def test_fill_holes_matches_scipy():
    from scipy import ndimage

    rng = np.random.default_rng(1)
    for p in (0.4, 0.6, 0.7):
        b = rng.uniform(size=(41, 33)) < p
        np.testing.assert_array_equal(O.fill_holes(b.astype(np.int32)), ndimage.binary_fill_holes(b).astype(np.uint8))
