"""CPU: whole-slide host logic -- tile/chunk/patch geometry and the three-phase merge rules against the
reference's OWN functions (executed from /root/reference/infer/wsi.py source when present), plus
self-contained invariants that also run on the GPU box."""
import copy
import os
import re
import textwrap
import types

import numpy as np
import pytest
from scipy import ndimage

from hover_net_amd import infer_wsi as W

REF = "/root/reference/infer/wsi.py"
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="reference tree only exists in the build container")


def _ref_ns():
    src = open(REF).read()
    ns = {"np": np}
    for name in ("_remove_inst", "_get_patch_top_left_info", "_get_tile_info", "_get_chunk_patch_info"):
        m = re.search(r"def %s\(.*?\n(?=####)" % name, src, re.S)
        exec(compile(m.group(0), REF, "exec"), ns)
    return ns, src


@needs_ref
@pytest.mark.parametrize("shape", [(5000, 7000), (2048, 2048), (4100, 2049), (10000, 12345)])
def test_geometry_matches_reference(shape):
    ns, _ = _ref_ns()
    shp = np.array(shape)
    for a, b in zip(W.get_tile_info(shp, np.array([2048, 2048]), 128), ns["_get_tile_info"](shp, np.array([2048, 2048]), 128)):
        np.testing.assert_array_equal(a, b)
    for pin, pout in ((270, 80), (256, 164)):
        got = W.get_chunk_patch_info(shp, np.array([3000, 3000]), np.array([pin, pin]), np.array([pout, pout]))
        want = ns["_get_chunk_patch_info"](shp, np.array([3000, 3000]), np.array([pin, pin]), np.array([pout, pout]))
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)


def _fake_tile_result(truth, tl, br, rng):
    """What post_proc.process would return for the tile: local ids 1..k + an info dict."""
    crop = truth[tl[0]:br[0], tl[1]:br[1]]
    lab, k = ndimage.label(crop > 0)
    lab = lab.astype(np.int32)
    info = {}
    for i in range(1, k + 1):
        ys, xs = np.nonzero(lab == i)
        if rng.uniform() < 0.05:
            continue  # like an instance skipped for a degenerate contour
        info[i] = {"bbox": np.array([[ys.min(), xs.min()], [ys.max() + 1, xs.max() + 1]]),
                   "centroid": np.array([xs.mean(), ys.mean()]),
                   "contour": np.stack([xs[:3], ys[:3]], -1).astype(np.int32), "type": 1, "type_prob": 0.5}
    return lab, info


@needs_ref
def test_three_phase_merge_matches_reference_callbacks():
    ns, src = _ref_ns()
    body = src[src.index("        def post_proc_normal_tile_callback(args):"):src.index("        #######################\n        pbar_creator")]
    rng = np.random.default_rng(3)
    shape = np.array([1300, 1500])
    truth = (ndimage.gaussian_filter(rng.normal(size=tuple(shape)), 4) > 0.035).astype(np.int32)
    grid, boundary, cross = W.get_tile_info(shape, np.array([512, 512]), 32)
    # reference side
    me = types.SimpleNamespace(wsi_inst_info={}, wsi_inst_map=np.zeros(tuple(shape), np.int32))
    rns = dict(ns, self=me, pbar=types.SimpleNamespace(update=lambda: None), log_info=lambda *a: None)
    exec(compile(textwrap.dedent(body), REF, "exec"), rns)
    mine, pre = W.WsiMerger(shape), W.WsiMerger(shape)
    for phase, tiles in enumerate((grid, boundary, cross)):
        for idx, t in enumerate(tiles):
            res = _fake_tile_result(truth, t[0], t[1], np.random.default_rng(100 * phase + idx))
            ref_cb = rns["post_proc_normal_tile_callback"] if phase == 0 else rns["post_proc_fixing_tile_callback"]
            ref_cb((copy.deepcopy(res), (idx, t[0].copy(), t[1].copy())))
            (mine.normal if phase == 0 else mine.fixing)(res[0].copy(), copy.deepcopy(res[1]), t[0], t[1])
            # the pipelined path hands over entries that already carry the tile origin (added in bulk at dict assembly)
            shifted = copy.deepcopy(res[1])
            for e in shifted.values():
                for f in ("bbox", "centroid", "contour"):
                    e[f] = e[f] + np.asarray(t[0])[::-1]
            (pre.normal if phase == 0 else pre.fixing)(res[0].copy(), shifted, t[0], t[1], shifted=True)
        np.testing.assert_array_equal(mine.inst_map, me.wsi_inst_map)
        np.testing.assert_array_equal(pre.inst_map, me.wsi_inst_map)
        assert list(mine.inst_info) == sorted(me.wsi_inst_info) == list(pre.inst_info)       # insertion order = ascending ids (O(1) running maximum)
    for k, e in me.wsi_inst_info.items():
        for f in ("bbox", "centroid", "contour"):
            np.testing.assert_array_equal(mine.inst_info[k][f], e[f])
            np.testing.assert_array_equal(pre.inst_info[k][f], e[f])
    assert len(mine.inst_info) > 50


def test_records_to_dict_shift_is_the_merge_offset():
    """post_proc.records_to_dict(shift_xy=(x0, y0)) == the three per-instance `+ top_left` of wsi.py:580-584, x added to the
    bbox rows included."""
    from hover_net_amd import post_proc as PP

    inst = np.zeros((30, 40), np.int32)
    inst[3:9, 5:12] = 1
    inst[15:25, 20:33] = 2
    rec = np.zeros(4, PP._REC_DTYPE)
    for l in (1, 2):
        ys, xs = np.nonzero(inst == l)
        rec[l - 1] = (l, len(ys), ys.min(), ys.max() + 1, xs.min(), xs.max() + 1, float((xs - xs.min()).sum()), float((ys - ys.min()).sum()), -1, 0)
    flat = PP.trace_contours_flat(inst, rec)
    plain = PP.records_to_dict(rec, None, contours_flat=flat)
    moved = PP.records_to_dict(rec, None, contours_flat=flat, shift_xy=(700, 90))
    tl = np.array([700, 90])
    for k in plain:
        np.testing.assert_array_equal(moved[k]["bbox"], plain[k]["bbox"] + tl)
        np.testing.assert_array_equal(moved[k]["centroid"], plain[k]["centroid"] + tl)
        np.testing.assert_array_equal(moved[k]["contour"], plain[k]["contour"] + tl)
        assert moved[k]["contour"].dtype == np.int32


def test_geometry_invariants():
    shp = np.array([5000, 7000])
    grid, boundary, cross = W.get_tile_info(shp, np.array([2048, 2048]), 128)
    cover = np.zeros(tuple(shp), np.uint8)
    for tl, br in grid:
        cover[tl[0]:br[0], tl[1]:br[1]] += 1
    assert (cover == 1).all()                       # grid tiles partition the slide
    assert (boundary[:, 1] - boundary[:, 0]).min() == 256 and cross.shape[0] == 2 * 3
    chunk, patch = W.get_chunk_patch_info(shp, np.array([3000, 3000]), np.array([270, 270]), np.array([80, 80]))
    assert ((patch[:, 0, 1] - patch[:, 0, 0]) == 270).all() and ((patch[:, 1, 1] - patch[:, 1, 0]) == 80).all()
    assert (patch[:, 1, 0] - patch[:, 0, 0] == 190).all()  # (sic) wsi.py:176 adds the full in/out difference
    # every patch belongs to exactly one chunk by the reference's selection rule
    owner = np.zeros(patch.shape[0], int)
    for c in chunk:
        s, e = c[0, 0], c[0, 1] - 270
        owner += ((s[0] <= patch[:, 0, 0, 0]) & (patch[:, 0, 0, 0] <= e[0]) & (s[1] <= patch[:, 0, 0, 1]) & (patch[:, 0, 0, 1] <= e[1]))
    assert (owner <= 1).all()
    overhang = (patch[:, 0, 1] > shp).any(axis=1)    # input window sticks out of the slide: the reference never runs these
    assert (owner[~overhang] == 1).all() and (owner[overhang] == 0).all()


def test_select_valid_uses_the_mask():
    shp = np.array([4096, 4096])
    grid, _, _ = W.get_tile_info(shp, np.array([2048, 2048]), 128)
    mask = np.zeros((128, 128), np.uint8)
    mask[:64, 64:] = 1                               # tissue only in the top-right quadrant
    kept = W.select_valid(grid, mask, shp, has_output_info=False)
    assert kept.shape[0] == 1 and kept[0][0].tolist() == [0, 2048]
