"""oracle/targets_np.py (training-target restatement) against goldens made by the reference's own
models/hovernet/targets.py (oracle/make_golden_targets.py): bit-exact hv_map / np_map."""
import os

import numpy as np

from oracle import targets_np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "targets.npz")


def cases():
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        yield g["ann%d" % k].astype(np.int32), int(g["crop%d" % k]), g["hv%d" % k], g["np%d" % k]


def test_targets_oracle_matches_reference_golden_bit_exact():
    for ann, crop, hv, npm in cases():
        t = targets_np.gen_targets(ann, (crop, crop))
        assert np.array_equal(t["hv_map"], hv)
        assert np.array_equal(t["np_map"].astype(np.uint8), npm)
        assert np.abs(hv).max() <= 1.0 and (hv != 0).any()


def test_border_instances_are_skipped_like_the_reference():
    ann = np.zeros((100, 100), np.int32)
    ann[0:40, 30:70] = 1        # touches the top border: the unclamped box start is negative -> empty slice -> no HV target
    ann[45:85, 30:70] = 2
    t = targets_np.gen_targets(ann, (100, 100))
    assert not t["hv_map"][0:40].any() and t["hv_map"][45:85].any()
    assert t["np_map"][0:40, 30:70].all()
