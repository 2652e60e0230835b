"""-m gpu: on-GPU instance separation (hover_net_amd/csrc/hvn_postproc.hip) through the C ABI
against (a) the golden instance maps made by the reference's own post_proc.py and (b) the C
oracle stage by stage.  Integer work: BIT-EXACT, label values included."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pp_*.npz")))


def _pp():
    from hover_net_amd.post_proc import PostProc

    return PostProc("cuda")


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[3:-4] for p in CASES])
def test_matches_reference_golden(path):
    z = np.load(path)
    pred, inst = z["pred"], z["inst"]
    got = _pp().separate(torch.from_numpy(pred).to("cuda")).cpu().numpy()
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, inst)


def test_stage_taps_match_oracle():
    from hover_net_amd.synth import synth_pred_maps
    from oracle import postproc as O

    pred = synth_pred_maps(6, 80, 80, 5, seed=41)[0]
    inst, blb, dist, marker = [t.cpu().numpy() for t in _pp().separate(torch.from_numpy(pred).to("cuda"), taps=True)]
    for i in range(pred.shape[0]):
        o_inst, o_blb, o_dist, o_marker = O.proc_np_hv(pred[i][..., 1:], taps=True)
        np.testing.assert_array_equal(blb[i], o_blb)
        np.testing.assert_array_equal(dist[i], o_dist)      # float64, bit for bit
        np.testing.assert_array_equal(marker[i], o_marker)
        np.testing.assert_array_equal(inst[i], o_inst)


def test_full_batch_32_tiles_and_properties():
    """BASELINE size (32 maps of 80x80): oracle equality + size-independent properties."""
    from hover_net_amd.synth import synth_pred_maps
    from oracle import postproc as O

    pred = synth_pred_maps(32, 80, 80, 5, seed=42)[0]
    dev = torch.from_numpy(pred).to("cuda")
    pp = _pp()
    inst = pp.separate(dev).cpu().numpy()
    np.testing.assert_array_equal(inst, O.proc_batch(pred))
    # idempotent / deterministic, and independent of a tile's position in the batch
    np.testing.assert_array_equal(pp.separate(dev).cpu().numpy(), inst)
    perm = np.random.default_rng(0).permutation(32)
    np.testing.assert_array_equal(pp.separate(dev[torch.from_numpy(perm)]).cpu().numpy(), inst[perm])
    # every instance lies inside the thresholded blob mask
    assert ((inst > 0) <= (pred[..., 1] >= 0.5)).all()


def _smooth_noise_maps(n, hw, seed, it=6):
    rng = np.random.Generator(np.random.PCG64(seed))

    def smooth(a):
        for _ in range(it):
            a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5.0
        return a

    f = np.stack([smooth(rng.normal(0, 1, (n, hw, hw)).transpose(1, 2, 0)).transpose(2, 0, 1) for _ in range(3)], -1)
    f = f / f.std()
    f[..., 0] = 0.5 + 0.5 * f[..., 0]
    return f.astype(np.float32)


def test_tie_heavy_maps_match_oracle():
    """Marker ties at nearly every pop: dense touching nuclei with h / v quantised to 2..16 grey levels, and quantised smooth
    noise (irregular blobs, markers from noise).  The wave flood decides same-label ties itself and hands only mixed-label
    ties to the exact replay (hvn_postproc.hip ws_window_wave): every map must equal the oracle's exact heap, label for label."""
    from hover_net_amd.synth import synth_pred_maps
    from oracle import postproc as O

    maps = []
    for k, levels in enumerate((2, 3, 4, 8, 16)):
        q = synth_pred_maps(8, 80, 80, None, seed=200 + k, k_lo=5, k_hi=40, noise=0.0)[0]
        q[..., 1:] = np.round(q[..., 1:] * levels) / levels
        maps.append(q)
    for k, levels in enumerate((2, 4, 8)):
        f = _smooth_noise_maps(4, 80, 300 + k)
        f[..., 1:] = np.round(f[..., 1:] * levels) / levels
        maps.append(f)
    pred = np.concatenate(maps, 0)
    got = _pp().separate(torch.from_numpy(pred).to("cuda")).cpu().numpy()
    want = O.proc_batch(pred)
    bad = [i for i in range(pred.shape[0]) if not np.array_equal(got[i], want[i])]
    assert not bad, "maps %s differ from the exact replay" % bad
    assert int(want.max()) > 3


def test_tile_filling_noise_blobs_match_oracle():
    """What a random-init network emits: blobs that fill the tile, hundreds of noise markers inside (windows beyond the LDS)."""
    from oracle import postproc as O

    pred = _smooth_noise_maps(3, 164, 400, it=3)
    pred[..., 0] += 0.25                                   # most of the tile above the 0.5 threshold: one big component
    got = _pp().separate(torch.from_numpy(pred).to("cuda")).cpu().numpy()
    np.testing.assert_array_equal(got, O.proc_batch(pred))


def test_large_tile_uses_hbm_heap():
    from hover_net_amd.synth import synth_pred_maps
    from oracle import postproc as O

    pred = synth_pred_maps(2, 300, 277, None, seed=43)[0]
    got = _pp().separate(torch.from_numpy(pred).to("cuda")).cpu().numpy()
    np.testing.assert_array_equal(got, O.proc_batch(pred))


def test_process_contract_and_instance_table():
    from hover_net_amd import post_proc
    from hover_net_amd.synth import synth_pred_maps
    from oracle import postproc as O

    pred = synth_pred_maps(1, 80, 80, 5, seed=44)[0][0]
    inst, info = post_proc.process(pred, nr_types=5, return_centroids=True)
    want = O.proc_np_hv(pred[..., 1:])
    np.testing.assert_array_equal(inst, want)
    ids = [i for i in np.unique(want) if i > 0]
    assert set(info.keys()) <= set(ids)
    for i in set(ids) - set(info.keys()):   # post_proc.py:140-143: contours with < 3 points are skipped
        ys, xs = np.nonzero(want == i)
        assert min(ys.max() - ys.min(), xs.max() - xs.min()) == 0
    tmap = pred[..., 0].astype(np.int32)
    for i in sorted(info.keys()):
        m = want == i
        ys, xs = np.nonzero(m)
        e = info[i]
        assert e["bbox"].tolist() == [[ys.min(), xs.min()], [ys.max() + 1, xs.max() + 1]]  # misc/utils.py:18-28
        cx = (xs - xs.min()).sum() / float(m.sum()) + xs.min()
        cy = (ys - ys.min()).sum() / float(m.sum()) + ys.min()
        assert e["centroid"].tolist() == [cx, cy]
        c = e["contour"]
        assert c.dtype == np.int32 and c.ndim == 2 and c.shape[1] == 2 and c.shape[0] >= 3
        assert m[c[:, 1], c[:, 0]].all()                          # contour points lie on the instance
        assert tuple(c[0]) == (xs[ys == ys.min()].min(), ys.min())  # findContours starts at the first raster pixel
        tl, tc = np.unique(tmap[m], return_counts=True)
        order = sorted(zip(tl, tc), key=lambda t: t[1], reverse=True)   # post_proc.py:168-177
        t = order[0][0]
        if t == 0 and len(order) > 1:
            t = order[1][0]
        assert e["type"] == int(t)
        assert e["type_prob"] == float(dict(order)[t] / (m.sum() + 1.0e-6))
    inst2, info2 = post_proc.process(pred[..., 1:], nr_types=None, return_centroids=False)
    np.testing.assert_array_equal(inst2, want)
    assert info2 is None


PROC_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "proc_*.npz")))


@pytest.mark.parametrize("path", PROC_CASES, ids=[os.path.basename(p)[5:-4] for p in PROC_CASES])
def test_process_matches_reference_golden(path):
    """`process()` end to end (instance separation + hvn_instance_table on the GPU, contours + dict on the host) against
    the fixtures made by the reference's own unmodified process() (oracle/make_golden_process.py): instance map, dict key
    set and order, bbox, centroid, contour point order, type and type_prob -- all compared with ==."""
    from golden_util import assert_same_info, golden_dicts
    from hover_net_amd import post_proc

    z = np.load(path)
    nt = None if int(z["nr_types"]) < 0 else int(z["nr_types"])
    for i, want in enumerate(golden_dicts(z)):
        inst, info = post_proc.process(z["pred"][i], nr_types=nt, return_centroids=True)
        np.testing.assert_array_equal(inst, z["inst"][i])
        assert inst.dtype == np.int32
        assert_same_info(info, want)
    # the batched device path gives the same records for all maps of the case at once
    pred = torch.from_numpy(z["pred"]).to("cuda")
    inst_b, rec, counts = post_proc.process_batch_device(pred, nr_types=nt, return_centroids=True)
    np.testing.assert_array_equal(inst_b.cpu().numpy(), z["inst"])
    rec_h = rec.cpu().numpy()
    for i, want in enumerate(golden_dicts(z)):
        got = post_proc.records_to_dict(rec_h[i].view(post_proc._REC_DTYPE).reshape(-1), nt, z["inst"][i])
        assert_same_info(got, want)
        assert int(counts[i]) == len(np.unique(z["inst"][i])) - 1


def test_empty_and_full_maps():
    e = np.zeros((3, 40, 40, 3), np.float32)
    e[1, ..., 0] = 1.0
    got = _pp().separate(torch.from_numpy(e).to("cuda")).cpu().numpy()
    assert (got[0] == 0).all()


def test_component_replay_equals_whole_tile_replay():
    """The per-component watershed (default) and the forced whole-tile replay (HVN_WS_GLOBAL=1, a fresh
    process because the switch is read once) agree on structured, noisy and tie-heavy maps."""
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from hover_net_amd.post_proc import PostProc\n"
        "z = np.load(%r); out = PostProc('cuda').separate(torch.from_numpy(z['pred']).to('cuda')).cpu().numpy()\n"
        "assert np.array_equal(out, z['inst']), 'whole-tile replay differs from golden'\n"
    )
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("pp_quant80.npz", "pp_s80.npz", "pp_noise80.npz"):
        path = os.path.join(repo, "tests", "golden", name)
        env = dict(os.environ, HVN_WS_GLOBAL="1")
        r = subprocess.run([sys.executable, "-c", code % (repo, path)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]


def test_oversized_component_uses_hbm_window():
    """One blob far larger than the LDS window limit (2048 px), with several markers inside."""
    from oracle import postproc as O

    H = W = 120
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    pred = np.zeros((1, H, W, 3), np.float32)
    pred[0, ..., 0] = ((yy - 60) ** 2 + (xx - 60) ** 2 < 55 ** 2) * 0.9 + 0.05
    rng = np.random.default_rng(5)
    cx = np.array([30, 60, 90, 45, 75]); cy = np.array([40, 35, 45, 80, 85])
    d = np.stack([np.hypot(yy - cy[i], xx - cx[i]) for i in range(5)])
    k = d.argmin(0)
    pred[0, ..., 1] = np.clip((xx - cx[k]) / 25.0, -1, 1) + rng.normal(0, 0.01, (H, W))
    pred[0, ..., 2] = np.clip((yy - cy[k]) / 25.0, -1, 1) + rng.normal(0, 0.01, (H, W))
    want = O.proc_batch(pred)
    assert len(np.unique(want)) > 2          # several instances inside one connected blob
    got = _pp().separate(torch.from_numpy(pred).to("cuda")).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_large_image_many_components():
    """1000x1000 map with ~800 nuclei: per-component parallel replay (incl. harmless-tie proofs) == oracle."""
    from hover_net_amd.synth import synth_pred_maps
    from oracle import postproc as O

    pred = synth_pred_maps(1, 1000, 1000, None, seed=45, k_lo=2, k_hi=8)[0]
    got = _pp().separate(torch.from_numpy(pred).to("cuda")).cpu().numpy()
    np.testing.assert_array_equal(got, O.proc_batch(pred))
