"""GPU: the on-device whole-slide merge (csrc/hvn_wsi_merge.hip, infer_wsi.DeviceMerger) against the host restatement
`infer_wsi.WsiMerger`, which tests/test_infer_wsi.py pins to the reference's own callbacks executed from source
(/root/reference/infer/wsi.py:569-677): same instance map, same dictionary keys in the same order, tile after tile."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tile(rng, h, w, k, dense=False):
    """A tile's local-id map with k blobs (ids 1..k, some absent) + its info dict (some ids without an entry: dropped contours)."""
    m = np.zeros((h, w), np.int32)
    if dense:
        m[:] = 1
    for i in range(1, k + 1):
        if rng.random() < 0.1:
            continue
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        r = int(rng.integers(1, max(2, min(h, w) // 4)))
        m[max(0, y - r):y + r, max(0, x - r):x + r] = i
    ids = [int(i) for i in np.unique(m) if i > 0]
    info = {i: {"bbox": np.zeros((2, 2)), "centroid": np.zeros(2), "contour": np.zeros((3, 2), np.int32), "type": None, "type_prob": None}
            for i in ids if rng.random() < 0.9}
    return m, info


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_device_merger_equals_host_merger(seed):
    from hover_net_amd import infer_wsi as W

    rng = np.random.default_rng(seed)
    H, Wd = 96, 120
    host, dev = W.WsiMerger((H, Wd)), W.DeviceMerger((H, Wd), "cuda", cap=16 if seed % 2 else 1 << 20)   # odd seeds: the id tables must grow
    # phase 1: a grid of normal tiles; phase 2 / 3: overlapping fix-up windows, some without any background inside or on their edge
    grid = [((y, x), (min(y + 32, H), min(x + 40, Wd))) for y in range(0, H, 32) for x in range(0, Wd, 40)]
    for tl, br in grid:
        m, info = _tile(rng, br[0] - tl[0], br[1] - tl[1], int(rng.integers(0, 9)), dense=rng.random() < 0.3)
        host.normal(m, {k: dict(v) for k, v in info.items()}, tl, br, shifted=True)
        dev.normal(torch.from_numpy(m).cuda(), {k: dict(v) for k, v in info.items()}, tl, br)
    fix = [((y, x), (min(y + 24, H), min(x + 30, Wd))) for y in range(4, H - 8, 17) for x in range(3, Wd - 8, 23)]
    for n, (tl, br) in enumerate(fix):
        m, info = _tile(rng, br[0] - tl[0], br[1] - tl[1], int(rng.integers(0, 12)), dense=rng.random() < 0.25)
        host.fixing(m, {k: dict(v) for k, v in info.items()}, tl, br, shifted=True)
        src = torch.from_numpy(m).cuda() if n % 2 == 0 else m          # device-resident tile or a host array from a remote rank
        dev.fixing(src, {k: dict(v) for k, v in info.items()}, tl, br)
        got = dev.inst_map.cpu().numpy()
        assert np.array_equal(got, host.inst_map), "fix-up tile %d" % n
        assert list(dev.inst_info) == list(host.inst_info), "fix-up tile %d" % n
    inst, info = dev.result()
    assert np.array_equal(inst, host.inst_map) and list(info) == list(host.inst_info) and len(info) > 10


def test_stitch_instances_device_merge_equals_host_merge(monkeypatch):
    from hover_net_amd import infer_wsi, net_desc
    from hover_net_amd.synth import synth_pred_maps, synth_state_dict

    net = net_desc.create_model(mode="original", nr_types=5, input_ch=3)
    net.load_state_dict(synth_state_dict("original", 5, seed=1), strict=True)
    net = net.cuda().eval()
    wsi = infer_wsi.WsiInference(net, nr_types=5, batch_size=8, tile_shape=512, ambiguous_size=64)
    maps = torch.from_numpy(synth_pred_maps(1, 1300, 1100, 5, seed=7, k_lo=2, k_hi=8)[0][0]).cuda()
    monkeypatch.setenv("HVN_WSI_HOST_MERGE", "1")
    inst_h, info_h = wsi.stitch_instances(maps)
    monkeypatch.setenv("HVN_WSI_HOST_MERGE", "0")
    inst_d, info_d = wsi.stitch_instances(maps)
    assert np.array_equal(inst_h, inst_d) and list(info_h) == list(info_d) and len(info_d) > 400
    for k in list(info_d)[::37]:
        assert np.array_equal(info_d[k]["bbox"], info_h[k]["bbox"]) and np.array_equal(info_d[k]["contour"], info_h[k]["contour"])
