"""-m gpu: hvn_gen_targets against the reference-made goldens and the numpy oracle, bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "targets.npz")


def test_targets_match_reference_goldens_bit_exact():
    from hover_net_amd import targets
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        ann, crop = g["ann%d" % k].astype(np.int32), int(g["crop%d" % k])
        t = targets.gen_targets(ann, (crop, crop))
        assert np.array_equal(t["hv_map"], g["hv%d" % k]), k
        assert np.array_equal(t["np_map"].astype(np.uint8), g["np%d" % k]), k


def test_targets_batched_random_maps_equal_the_oracle():
    from hover_net_amd import targets
    from oracle import targets_np
    rng = np.random.default_rng(11)
    anns = np.stack([targets_np.synth_ann(rng, 270, int(rng.integers(10, 120)), bool(i % 2)) for i in range(6)])
    out = targets.gen_targets_device(torch.from_numpy(anns).cuda(), (80, 80))
    hv, npm = out["hv_map"].cpu().numpy(), out["np_map"].cpu().numpy()
    for i in range(anns.shape[0]):
        want = targets_np.gen_targets(anns[i], (80, 80))
        assert np.array_equal(hv[i], want["hv_map"]), i
        assert np.array_equal(npm[i], want["np_map"]), i
    # non-square crop of a non-square map
    a = targets_np.synth_ann(rng, 300, 90, True)[:260]
    got = targets.gen_targets(a, (120, 164))
    want = targets_np.gen_targets(a, (120, 164))
    assert np.array_equal(got["hv_map"], want["hv_map"]) and np.array_equal(got["np_map"], want["np_map"])
