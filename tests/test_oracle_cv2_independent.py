"""CPU: the cv2 leg of the post-processing oracle, held against a SECOND, independently written restatement.

OpenCV is not on the box, so `oracle/hvn_oracle.c` restates normalize / Sobel-21 / GaussianBlur / morphologyEx from the
OpenCV sources' documented behaviour, and the goldens (made by the reference's post_proc.py over oracle/cv2_shim) could only
pin those four to the C code itself.  oracle/cv2_shim_scipy/cv2.py restates them again with scipy.ndimage in float64
(different code, different summation order).  Here: (1) function by function on the golden inputs -- integer results
bit-equal, floating-point within a stated bound that only summation order can explain; (2) end to end -- the reference's
own __proc_np_hv over the second shim reproduces every committed golden instance map exactly (PQ == 1 by the reference's
metrics/stats_utils.py), re-run live when the reference and the secondary interpreter are present, and recorded in
tests/golden/alt_shim_check.json."""
import glob
import importlib.util
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import postproc as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(REPO, "tests", "golden", "pp_*.npz")))


@pytest.fixture(scope="module")
def alt():
    spec = importlib.util.spec_from_file_location("cv2_alt", os.path.join(REPO, "oracle", "cv2_shim_scipy", "cv2.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_kernels_known_answers(alt):
    assert alt.sobel_taps(3, 1).tolist() == [-1, 0, 1] and alt.sobel_taps(3, 0).tolist() == [1, 2, 1]
    assert alt.sobel_taps(5, 1).tolist() == [-1, -2, 0, 2, 1] and alt.sobel_taps(5, 0).tolist() == [1, 4, 6, 4, 1]
    import ctypes

    k = np.zeros(21)
    for order in (0, 1):
        O.lib().hvn_o_sobel_kernel21(order, k.ctypes.data_as(ctypes.c_void_p))
        assert k.tolist() == alt.sobel_taps(21, order).tolist()
    assert alt.getStructuringElement(alt.MORPH_ELLIPSE, (5, 5)).tolist() == [[0, 0, 1, 0, 0], [1] * 5, [1] * 5, [1] * 5, [0, 0, 1, 0, 0]]
    assert alt.getStructuringElement(alt.MORPH_ELLIPSE, (3, 3)).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]   # OpenCV's 3x3 "ellipse" is the cross


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[3:-4] for p in CASES])
def test_filters_agree_on_golden_inputs(alt, path):
    pred = np.load(path)["pred"]
    for i in range(pred.shape[0]):
        for ch, dx in ((-2, 1), (-1, 0)):
            x = pred[i][..., ch]
            n_c, n_s = O.normalize_32f(x), alt.normalize(x)
            # float32 arithmetic `x*scale - min*scale` (C, OpenCV's 32f path) vs float64 `(x-min)/(max-min)` rounded once:
            # both within half an ulp(1.0) of the real value
            assert np.abs(n_c.astype(np.float64) - n_s).max() <= 2.0 ** -23
            s_c, s_s = O.sobel21(n_c, dx), alt.Sobel(n_c, alt.CV_64F, dx, 1 - dx, ksize=21)
            # 21 + 21 taps in float64, different summation order: a few ulp of the largest magnitude
            assert np.abs(s_c - s_s).max() <= 8 * np.spacing(np.abs(s_c).max())
            m_c, m_s = O.normalize_64f32f(s_c), alt.normalize(s_c)
            assert np.abs(m_c.astype(np.float64) - m_s).max() <= 2.0 ** -23
        rng = np.random.default_rng(i)
        d = rng.random(pred.shape[1:3])
        g_c, g_s = O.gauss3_64f(d), alt.GaussianBlur(d, (3, 3), 0)
        assert np.abs(g_c - g_s).max() <= 2 * np.spacing(1.0)        # power-of-two taps: only the two adds can round
        for fill in (0.5, 0.8):
            b = (rng.random(pred.shape[1:3]) < fill).astype(np.uint8)
            np.testing.assert_array_equal(O.morph_open5(b), alt.morphologyEx(b, alt.MORPH_OPEN, alt.getStructuringElement(alt.MORPH_ELLIPSE, (5, 5))))


def test_reflect101_border_is_what_both_use(alt):
    # a ramp: REFLECT_101 (d c b | a b c d | c b a) keeps the derivative sign change at the border that REFLECT would not
    x = np.tile(np.arange(40, dtype=np.float32), (40, 1))
    np.testing.assert_allclose(O.sobel21(x, 1), alt.Sobel(x, alt.CV_64F, 1, 0, ksize=21), rtol=0, atol=1e-6)
    assert O.sobel21(x, 1)[20, 0] == 0.0      # mirror about the first sample: odd kernel on an even extension


def test_recorded_end_to_end_check_says_identical():
    rep = json.load(open(os.path.join(REPO, "tests", "golden", "alt_shim_check.json")))
    have = {os.path.basename(p) for p in glob.glob(os.path.join(REPO, "tests", "golden", "p*_*.npz")) if os.path.basename(p).startswith(("pp_", "proc_"))}
    assert set(rep) == have
    for name, r in rep.items():
        assert r["identical"] == r["maps"], name
        assert r["min_pq"] > 1 - 1e-5, name       # get_fast_pq adds 1e-6 to its denominators: 1.0 is printed as 0.999999...


@pytest.mark.skipif(not (os.path.exists("/root/reference/models/hovernet/post_proc.py") and os.path.exists("/opt/conda/bin/python3.9")),
                    reason="needs the reference tree and the secondary interpreter (build container only)")
def test_live_reference_over_second_shim_reproduces_goldens(tmp_path):
    out = tmp_path / "rep.json"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run(["/opt/conda/bin/python3.9", "-W", "ignore", os.path.join(REPO, "oracle", "check_alt_shim.py"), "--json", str(out)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.load(open(out))
    assert rep == json.load(open(os.path.join(REPO, "tests", "golden", "alt_shim_check.json")))
