"""CPU: the tile manager's host loop (hover_net_amd/infer_manager.py; infer/base.py:21-94 + infer/tile.py:150-387) -- file list,
RAM-bounded caching rounds, and the mat / json / overlay / qupath writers -- with the GPU call replaced by a stub.  The GPU
run of the same loop is tests/test_gpu_dropin.py::test_process_file_list_end_to_end."""
import json
import os

import numpy as np
import pytest
import scipy.io as sio

from hover_net_amd import infer_manager as im
from hover_net_amd import viz


def _fake_result(img, nr_types, raw):
    h, w = img.shape[:2]
    inst = np.zeros((h, w), np.int32)
    inst[2:8, 3:9] = 1
    inst[12:20, 10:15] = 4
    info = {
        1: {"bbox": np.array([[2, 3], [8, 9]]), "centroid": np.array([5.5, 4.5]), "contour": np.array([[3, 2], [3, 7], [8, 7], [8, 2]], np.int32),
            "type_prob": 0.75 if nr_types else None, "type": 2 if nr_types else None},
        4: {"bbox": np.array([[12, 10], [20, 15]]), "centroid": np.array([12.0, 15.5]), "contour": np.array([[10, 12], [10, 19], [14, 19], [14, 12]], np.int32),
            "type_prob": 1.0 if nr_types else None, "type": 1 if nr_types else None},
    }
    res = (inst, info)
    return res + (np.full((h, w, 4 if nr_types else 3), 0.5, np.float32),) if raw else res


def _make_dir(tmp_path, names, shape=(40, 50)):
    d = tmp_path / "in[1]"                     # brackets in the path: glob must not read them as a character class
    d.mkdir()
    rng = np.random.default_rng(0)
    for n in names:
        img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
        if n.endswith(".npy"):
            np.save(d / n, img)
        else:
            from PIL import Image

            Image.fromarray(img).save(d / n)
    return str(d)


def test_hot_colour_table_matches_matplotlib_lut():
    # (plt.get_cmap("hot")(np.arange(6, dtype=np.int32))[..., :3] * 255).astype(np.uint8), matplotlib under /opt/conda python3.9
    assert im.hot_colours(6).tolist() == [[10, 0, 0], [13, 0, 0], [15, 0, 0], [18, 0, 0], [21, 0, 0], [23, 0, 0]]
    assert im.load_type_info(None, None) == {None: ["no label", [0, 0, 0]]}
    assert im.load_type_info(3, None)[2] == ("2", (15, 0, 0))


def test_type_info_json_checked(tmp_path):
    p = tmp_path / "t.json"
    p.write_text(json.dumps({"0": ["nolabe", [0, 0, 0]], "1": ["neopla", [255, 0, 0]]}))
    assert im.load_type_info(2, str(p))[1] == ("neopla", (255, 0, 0))
    with pytest.raises(AssertionError, match="type_id=2"):
        im.load_type_info(3, str(p))


def test_padded_size_is_prepare_patching_size():
    from hover_net_amd import infer_tile

    for shape in [(40, 50, 3), (80, 80, 3), (270, 300, 3), (1000, 1000, 3)]:
        for win, msk in [(270, 80), (256, 164)]:
            padded, _ = infer_tile.prepare_patching(np.zeros(shape, np.uint8), win, msk) if min(shape[:2]) > (win - msk) // 2 else (None, None)
            if padded is not None:
                assert im.padded_nbytes(shape, win, msk) == padded.nbytes


@pytest.mark.parametrize("nr_types", [None, 5])
def test_process_file_list_writes_every_output(tmp_path, nr_types):
    names = ["b.png", "a.png", "c.npy"]
    inp = _make_dir(tmp_path, names)
    out = str(tmp_path / "out")
    seen = []

    def process(images):
        seen.append(len(images))
        return [_fake_result(i, nr_types, True) for i in images]

    type_json = tmp_path / "type.json"
    type_json.write_text(json.dumps({str(k): ["t%d" % k, [40 * k, 10, 200]] for k in range(6)}))
    mgr = im.InferManager({"model_args": {"nr_types": nr_types, "mode": "original"}, "model_path": None},
                          type_info_path=str(type_json) if nr_types else None, process_fn=process)
    done = mgr.process_file_list({"input_dir": inp, "output_dir": out, "batch_size": 8, "mem_usage": 0.2, "draw_dot": True,
                                  "save_qupath": True, "save_raw_map": True, "nr_inference_workers": 8, "nr_post_proc_workers": 16,
                                  "patch_input_shape": 270, "patch_output_shape": 80})
    assert done == ["a", "b", "c"] and seen == [3]          # sorted file order, one caching round
    for name in done:
        mat = sio.loadmat("%s/mat/%s.mat" % (out, name))
        assert mat["inst_map"].shape == (40, 50) and mat["inst_uid"].tolist() == [[1], [4]]
        assert mat["inst_centroid"].tolist() == [[5.5, 4.5], [12.0, 15.5]]
        assert ("inst_type" in mat) == (nr_types is not None)
        if nr_types:
            assert mat["inst_type"].tolist() == [[2], [1]]
        assert mat["raw_map"].shape == (40, 50, 4 if nr_types else 3)
        js = json.load(open("%s/json/%s.json" % (out, name)))
        assert js["mag"] is None and sorted(js["nuc"]) == ["1", "4"]
        assert js["nuc"]["4"]["contour"] == [[10, 12], [10, 19], [14, 19], [14, 12]] and js["nuc"]["1"]["bbox"] == [[2, 3], [8, 9]]
        from PIL import Image

        ov = np.asarray(Image.open("%s/overlay/%s.png" % (out, name)))
        assert ov.shape == (40, 50, 3)
        if nr_types:
            assert ov[7, 8].tolist() == [80, 10, 200] and ov[12, 10].tolist() == [40, 10, 200]    # contour pixels in the type colours
        assert ov[4, 5].tolist() == [255, 0, 0]                                                 # centroid dot (x=5, y=4)
        rows = open("%s/qupath/%s.tsv" % (out, name)).read().splitlines()
        assert rows[0] == "x\ty\tclass\tname\tcolor"
        if nr_types:
            assert rows[1] == "5.5\t4.5\t\tt2\t%d" % ((80 << 16) + (10 << 8) + 200)
        else:
            assert rows[1] == "5.5\t4.5\t\tno label\t0"
    # a second run replaces the output directories (rm_n_mkdir)
    open("%s/json/stale.json" % out, "w").write("{}")
    mgr.process_file_list({"input_dir": inp, "output_dir": out, "save_qupath": False})
    assert not os.path.exists("%s/json/stale.json" % out) and os.path.exists("%s/qupath" % out)   # qupath dir untouched when not requested


def test_cache_rounds_respect_the_budget_and_lose_no_file(tmp_path):
    names = ["%02d.npy" % i for i in range(7)]
    inp = _make_dir(tmp_path, names, shape=(100, 100))
    per_file = 5 * im.padded_nbytes((100, 100, 3), 270, 80)
    seen = []

    def process(images):
        seen.append(len(images))
        return [_fake_result(i, None, False) for i in images]

    mgr = im.InferManager({"model_args": {"nr_types": None, "mode": "original"}, "model_path": None}, process_fn=process)
    done = mgr.process_file_list({"input_dir": inp, "output_dir": str(tmp_path / "o"), "ram_budget_bytes": 3 * per_file + 10})
    assert seen == [3, 3, 1] and mgr.rounds == [3, 3, 1] and done == ["%02d" % i for i in range(7)]
    seen.clear()
    done = mgr.process_file_list({"input_dir": inp, "output_dir": str(tmp_path / "o"), "ram_budget_bytes": per_file // 2})
    assert seen == [1] * 7 and len(done) == 7               # a file larger than the budget is still processed, alone
    with pytest.raises(AssertionError):
        mgr.process_file_list({"input_dir": inp, "output_dir": str(tmp_path / "o"), "mem_usage": 1.5})
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(AssertionError, match="Not Detected"):
        mgr.process_file_list({"input_dir": str(empty), "output_dir": str(tmp_path / "o")})
    with pytest.raises(AssertionError, match="fixed by the model mode"):
        mgr.process_file_list({"input_dir": inp, "output_dir": str(tmp_path / "o"), "patch_input_shape": 256})


def test_overlay_drawing():
    img = np.zeros((20, 20, 3), np.uint8)
    out = viz.visualize_instances_dict(img, {7: {"contour": np.array([[2, 2], [2, 10], [12, 10], [12, 2]]), "centroid": [7.0, 6.0], "type": 1}},
                                       draw_dot=False, type_colour={1: ("a", (1, 2, 3))}, line_thickness=2)
    assert img.sum() == 0                                   # input untouched
    assert out[2, 2].tolist() == [1, 2, 3] and out[10, 12].tolist() == [1, 2, 3] and out[6, 2].tolist() == [1, 2, 3]
    assert out[3, 3].tolist() == [1, 2, 3] and out[4, 4].tolist() == [0, 0, 0]      # 2 px wide
    assert out[6, 7].tolist() == [0, 0, 0]                  # interior untouched without the dot
    # without a type table: random colour, still drawn; out-of-image points clipped
    out = viz.visualize_instances_dict(img, {1: {"contour": np.array([[-3, 5], [25, 5]]), "centroid": [0.0, 0.0]}}, draw_dot=True)
    assert out[5, 10].any() and out[0, 0].tolist() == [255, 0, 0]
    assert viz.visualize_instances_dict(img, {}).sum() == 0


def test_process_wsi_list(tmp_path):
    """infer/wsi.py:698-750: sorted slide list, directories skipped, existing json skipped, mask file or heuristic, empty mask skipped,
    a slide without a backend is logged as a crash and the list goes on; json carries `mag`."""
    from PIL import Image

    inp = tmp_path / "slides"
    inp.mkdir()
    (inp / "subdir").mkdir()
    rng = np.random.default_rng(1)
    tissue = rng.integers(30, 120, (640, 960, 3), dtype=np.uint8)
    np.save(inp / "s1.npy", tissue)
    np.save(inp / "s2.npy", tissue)
    np.save(inp / "s3_blank.npy", np.full((640, 640, 3), 255, np.uint8))
    (inp / "s4.svs").write_bytes(b"not a slide")
    masks = tmp_path / "masks"
    masks.mkdir()
    m = np.zeros((20, 30), np.uint8)
    m[5:10, 5:20] = 7
    Image.fromarray(m).save(masks / "s1.png")
    Image.fromarray(np.zeros((20, 20), np.uint8)).save(masks / "s3_blank.png")
    calls = []

    def wsi_fn(slide, mask):
        calls.append((slide.shape[:2], int(mask.sum()), mask.dtype))
        return None, {3: {"bbox": np.array([[1, 2], [3, 4]]), "centroid": np.array([2.5, 2.0]), "contour": np.array([[2, 1], [3, 3], [2, 3]]), "type_prob": None, "type": None}}

    mgr = im.WsiManager({"model_args": {"nr_types": None, "mode": "fast"}, "model_path": None}, wsi_fn=wsi_fn)
    out = str(tmp_path / "out")
    args = {"input_dir": str(inp), "output_dir": out, "input_mask_dir": str(masks), "proc_mag": 40, "save_thumb": True, "save_mask": True}
    st = mgr.process_wsi_list(args)
    assert st == {"s1": "done", "s2": "done", "s3_blank": "empty mask", "s4": "crash"}
    assert calls[0] == ((640, 960), 75, np.uint8)              # the mask file, binarised
    assert calls[1][0] == (640, 960) and calls[1][1] > 0        # the thresholding heuristic on the 1.25x thumbnail
    js = json.load(open(out + "/json/s1.json"))
    assert js["mag"] == 40 and js["nuc"]["3"]["contour"] == [[2, 1], [3, 3], [2, 3]]
    assert np.asarray(Image.open(out + "/mask/s1.png")).max() == 255 and np.asarray(Image.open(out + "/thumb/s2.png")).shape == (20, 30, 3)
    assert mgr.process_wsi_list(args)["s1"] == "skip" and len(calls) == 2          # finished slides are not redone
    st = mgr.process_wsi_list({"input_dir": str(inp), "output_dir": str(tmp_path / "flat")})
    assert st["s2"] == "done" and os.path.exists(str(tmp_path / "flat") + "/s2.json")   # json at the top level without thumb / mask output
