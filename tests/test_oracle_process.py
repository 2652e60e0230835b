"""CPU: `process()` (post_proc.py:94-186) -- the oracle's restatement and the product's HOST pieces (contour
tracer csrc/hvn_contour.cpp, records_to_dict) against tests/golden/proc_*.npz, which the reference's own
unmodified `process()` produced (oracle/make_golden_process.py).  Integer work and exact double divisions:
everything compared with ==, contour point order included."""
import glob
import os

import numpy as np
import pytest

from golden_util import assert_same_info, golden_dicts

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "proc_*.npz")))
IDS = [os.path.basename(p)[5:-4] for p in CASES]


def test_golden_present():
    assert len(CASES) >= 5


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_oracle_process_matches_reference_golden(path):
    from oracle import process_np

    z = np.load(path)
    nt = None if int(z["nr_types"]) < 0 else int(z["nr_types"])
    for i, want in enumerate(golden_dicts(z)):
        inst, info = process_np.process(z["pred"][i], nt, True)
        np.testing.assert_array_equal(inst, z["inst"][i])
        assert_same_info(info, want)


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_host_contours_and_dict_match_reference_golden(path):
    """The product's host half of process(): records (computed here with numpy in the layout hvn_instance_table
    writes) -> trace_contours + records_to_dict == the reference's dict."""
    from hover_net_amd import post_proc as PP

    z = np.load(path)
    nt = None if int(z["nr_types"]) < 0 else int(z["nr_types"])
    for i, want in enumerate(golden_dicts(z)):
        inst = z["inst"][i]
        tmap = z["pred"][i][..., 0].astype(np.int32) if nt is not None else None
        recs = []
        for l in np.unique(inst):
            if l <= 0:
                continue
            ys, xs = np.nonzero(inst == l)
            t, tc = -1, 0
            if nt is not None:
                cnt = np.bincount(tmap[ys, xs], minlength=nt)
                t = int(np.argmax(cnt))                      # first maximum = the reference's stable sort on ties
                if t == 0 and (cnt[1:] > 0).any():
                    t = 1 + int(np.argmax(cnt[1:]))
                tc = int(cnt[t])
            recs.append((l, len(ys), ys.min(), ys.max() + 1, xs.min(), xs.max() + 1, float((xs - xs.min()).sum()),
                         float((ys - ys.min()).sum()), t, tc))
        rec = np.array(recs, dtype=PP._REC_DTYPE)
        assert_same_info(PP.records_to_dict(rec, nt, inst), want)


def test_contours0_is_the_last_top_level_component():
    """cv2 returns the RETR_TREE list newest-first per parent: with several 8-connected pieces under one label
    (cannot come out of the watershed, can be handed to the tracer) contours[0] belongs to the piece whose first
    raster pixel comes last; pieces inside a hole of another piece are not top-level."""
    from hover_net_amd import post_proc as PP
    from oracle.process_np import _suzuki

    a = np.zeros((14, 16), np.int32)
    a[1:4, 2:6] = 5
    a[6:13, 1:12] = 5
    a[8:11, 3:10] = 0          # a hole in the second piece ...
    a[9, 5:7] = 5              # ... with a third piece inside it (found last, but not top-level)
    a[2:5, 9:14] = 5           # fourth piece, top-level, starts before the ring
    want = _suzuki.find_contours_tree((a == 5).astype(np.uint8))[0][0].reshape(-1, 2)
    assert want[0].tolist() == [1, 6]
    ys, xs = np.nonzero(a == 5)
    rec = np.array([(5, len(ys), ys.min(), ys.max() + 1, xs.min(), xs.max() + 1, 0., 0., -1, 0)], dtype=PP._REC_DTYPE)
    got = PP.trace_contours(a, rec)[5]
    assert got.tolist() == want.tolist()


def test_threaded_tracer_equals_single_thread(monkeypatch):
    """hvn_trace_contours splits the record table over host threads (>= 256 records): same flat arrays as one thread."""
    from scipy import ndimage

    from hover_net_amd import post_proc as PP

    rng = np.random.default_rng(11)
    a = ndimage.gaussian_filter(rng.normal(size=(1000, 1200)), 2.0) > 0.15
    lab, n = ndimage.label(a)
    assert n > 600
    assert sum((sl[0].stop - sl[0].start) * (sl[1].stop - sl[1].start) for sl in ndimage.find_objects(lab)) > 250000   # threaded path
    inst = lab.astype(np.int32)
    rec = np.zeros(n + 50, PP._REC_DTYPE)                      # trailing empty slots like a real table
    for i, sl in enumerate(ndimage.find_objects(lab)):
        rec[i] = (i + 1, int((lab[sl] == i + 1).sum()), sl[0].start, sl[0].stop, sl[1].start, sl[1].stop, 0., 0., -1, 0)
    monkeypatch.setenv("HVN_HOST_THREADS", "1")
    p1, o1 = PP.trace_contours_flat(inst, rec)
    monkeypatch.setenv("HVN_HOST_THREADS", "7")
    p7, o7 = PP.trace_contours_flat(inst, rec)
    assert np.array_equal(o1, o7) and np.array_equal(p1, p7) and o1[-1] > 5 * n
    assert (np.diff(o1)[n:] == 0).all()
