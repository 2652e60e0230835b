"""-m gpu: the whole HIP network path through the reference-shaped interface
(hover_net_amd.net_desc / run_desc) against the torch fp32 oracle and the golden logits made by
the reference's own code.  Tolerance = BASELINE.json north_star: logits within 1e-3 (fp32)."""
import os

import numpy as np
import pytest
import torch

from test_oracle_net import CASES, crop_to, load_case

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _model(mode, nt, sd):
    from hover_net_amd import net_desc

    net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
    net.load_state_dict(sd, strict=True)
    return net.to("cuda").eval()


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_golden(name):
    mode, nt, sd, tiles, crop, logits, pmap = load_case(name)
    net = _model(mode, nt, sd)
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous().to("cuda")
    out = net(x)
    assert list(out.keys()) == (["np", "hv"] if nt is None else ["tp", "np", "hv"])
    for k, v in logits.items():
        got = crop_to(out[k].cpu().numpy(), crop, (2, 3))
        assert np.isfinite(got).all()
        assert np.abs(got - v).max() <= TOL, (k, np.abs(got - v).max())


@pytest.mark.parametrize("name", CASES)
def test_infer_step_matches_reference_golden(name):
    from hover_net_amd import run_desc

    mode, nt, sd, tiles, crop, logits, pmap = load_case(name)
    net = _model(mode, nt, sd)
    got = run_desc.infer_step(torch.from_numpy(tiles), net)
    assert isinstance(got, np.ndarray) and got.dtype == np.float32
    assert got.shape[-1] == (3 if nt is None else 4)
    got = crop_to(got, crop, (1, 2))
    assert np.abs(got[..., -3:] - pmap[..., -3:]).max() <= TOL
    if nt is not None:
        assert (got[..., 0] != pmap[..., 0]).mean() < 2e-3  # argmax flips only on near-tied logits


def test_batch_matches_oracle_and_is_batch_invariant():
    from hover_net_amd.synth import synth_state_dict, synth_tiles
    from oracle import net_torch

    sd = synth_state_dict("original", 5, seed=21)
    net = _model("original", 5, sd)
    tiles = synth_tiles(3, 270, seed=22)
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
    want = net_torch.forward(sd, x, "original")
    out = net(x.to("cuda"))
    for k in want:
        assert (out[k].cpu() - want[k]).abs().max().item() <= TOL
    one = net(x[1:2].to("cuda"))
    for k in want:  # same kernels, same order of accumulation: a tile's result cannot depend on its batch
        assert torch.equal(one[k][0], out[k][1])


def test_stage_taps_match_interpreter():
    """Localises a failure: compare a few intermediate activations with the torch interpretation."""
    import plan_interp
    from hover_net_amd.synth import synth_state_dict, synth_tiles

    sd = synth_state_dict("original", None, seed=31)
    net = _model("original", None, sd)
    tiles = torch.from_numpy(synth_tiles(1, 270, seed=32))
    eng = net.engine(1)
    eng.run(tiles.to("cuda"))
    torch.cuda.synchronize()
    names = ["conv0", "d0.units.0.conv1", "d0.units.0.conv2", "d0.units.2.conv3", "d1.units.0.conv2", "d3.units.2.conv3", "conv_bot",
             "u3.upadd", "decoder.np.u3.conva", "decoder.np.u3.dense.units.0.conv2", "decoder.np.u3.convf",
             "decoder.np.u2.upadd", "decoder.np.u1.conva"]
    taps = {k: None for k in names}
    plan_interp.run(eng.plan, tiles, taps)
    # buffers are re-used later in the plan, so re-run the HIP plan up to each op
    for i, op in enumerate(eng.plan.ops):
        if op.name not in taps:
            continue
        eng.run(tiles.to("cuda"), upto=i + 1)
        torch.cuda.synchronize()
        got = eng.buffer(op.y, 1).cpu()
        err = (got - taps[op.name]).abs().max().item()
        assert err <= 2e-4, (op.name, err)


def test_process_images_tile_pipeline():
    """infer/tile.py route (mode B of SURVEY 8d): one 270x270 image -> 16 patches -> stitched map ->
    on-GPU instance separation; equals the step-by-step pipeline and the C oracle on the same map."""
    from hover_net_amd import infer_tile, post_proc, run_desc
    from hover_net_amd.synth import synth_state_dict, synth_tiles
    from oracle import postproc as O

    sd = synth_state_dict("original", 5, seed=51)
    net = _model("original", 5, sd)
    img = synth_tiles(1, 300, seed=52)[0][:, :283]      # ragged, non-square source image
    (inst, info), = infer_tile.process_images([img], net, nr_types=5, batch_size=8)
    assert inst.shape == img.shape[:2] and inst.dtype == np.int32
    padded, pinfo = infer_tile.prepare_patching(img, 270, 80)
    patches = torch.from_numpy(infer_tile.extract_patches(padded, pinfo, 270))
    maps = np.concatenate([run_desc.infer_step(patches[i:i + 8], net) for i in range(0, patches.shape[0], 8)])
    full = infer_tile.stitch(maps, pinfo, img.shape)
    np.testing.assert_array_equal(inst, O.proc_np_hv(np.ascontiguousarray(full[..., 1:])))
    inst2, info2 = post_proc.process(np.ascontiguousarray(full), nr_types=5, return_centroids=True)
    np.testing.assert_array_equal(inst, inst2)
    assert sorted(info.keys()) == sorted(info2.keys())


def test_process_file_list_end_to_end(tmp_path):
    """infer/tile.py:150-387 through hover_net_amd.infer_manager: a directory of images (checkpoint loaded from a `.tar` with
    DataParallel-prefixed keys) -> mat / json / overlay / qupath.  The written instance map equals the oracle on the written raw map."""
    import json

    import scipy.io as sio

    from hover_net_amd import infer_manager
    from hover_net_amd.synth import synth_state_dict, synth_tiles
    from oracle import process_np

    sd = synth_state_dict("original", 5, seed=81)
    torch.save({"desc": {"module." + k: v for k, v in sd.items()}}, tmp_path / "net.tar")
    inp = tmp_path / "in"
    inp.mkdir()
    shapes = {"a": (270, 270), "b": (283, 300), "c": (120, 95)}
    for i, (name, (h, w)) in enumerate(shapes.items()):
        np.save(inp / (name + ".npy"), synth_tiles(1, 300, seed=82 + i)[0][:h, :w])
    mgr = infer_manager.InferManager({"model_args": {"nr_types": 5, "mode": "original"}, "model_path": str(tmp_path / "net.tar")})
    out = str(tmp_path / "out")
    done = mgr.process_file_list({"input_dir": str(inp), "output_dir": out, "batch_size": 8, "save_raw_map": True, "save_qupath": True,
                                  "draw_dot": True, "patch_input_shape": 270, "patch_output_shape": 80})
    assert done == ["a", "b", "c"] and mgr.rounds == [3]
    for name, (h, w) in shapes.items():
        mat = sio.loadmat("%s/mat/%s.mat" % (out, name))
        assert mat["inst_map"].shape == (h, w) and mat["raw_map"].shape == (h, w, 4) and mat["raw_map"].dtype == np.float32
        o_inst, o_info = process_np.process(np.ascontiguousarray(mat["raw_map"]), 5, True)
        np.testing.assert_array_equal(mat["inst_map"], o_inst)
        assert mat["inst_uid"].reshape(-1).tolist() == list(o_info)
        js = json.load(open("%s/json/%s.json" % (out, name)))["nuc"]
        assert [int(k) for k in js] == list(o_info)
        for k, v in o_info.items():
            assert js[str(k)]["contour"] == v["contour"].tolist() and js[str(k)]["type"] == v["type"] and js[str(k)]["centroid"] == v["centroid"].tolist()
        assert os.path.exists("%s/overlay/%s.png" % (out, name)) and os.path.exists("%s/qupath/%s.tsv" % (out, name))


def test_two_stream_pipeline_equals_sequential():
    from hover_net_amd import post_proc, run_desc
    from hover_net_amd.pipeline import TilePipeline
    from hover_net_amd.synth import synth_pred_maps, synth_state_dict, synth_tiles

    sd = synth_state_dict("original", 5, seed=61)
    net = _model("original", 5, sd)
    pipe = TilePipeline(net, nr_types=5)
    extra = torch.from_numpy(synth_pred_maps(4, 80, 80, 5, seed=62)[0]).to("cuda")
    want_extra = post_proc.process_batch_device(extra, 5)[0].cpu()
    outs, wants = [], []
    for i in range(3):                                  # back-to-back submits exercise the ping-pong slots
        tiles = torch.from_numpy(synth_tiles(2, 270, seed=70 + i))
        outs.append(pipe.submit(tiles)[0])
        outs.append(pipe.submit(tiles, extra_maps=extra)[0])
    pipe.wait()
    torch.cuda.synchronize()
    for i in range(3):
        tiles = torch.from_numpy(synth_tiles(2, 270, seed=70 + i))
        pred = run_desc.infer_step_device(tiles, net)
        want = post_proc.process_batch_device(pred, 5)[0].cpu()
        assert torch.equal(outs[2 * i].cpu(), want)
        assert torch.equal(outs[2 * i + 1].cpu(), want_extra)


def test_wsi_pipeline_on_synthetic_slide():
    """infer/wsi.py route on a small synthetic slide: chunked, sharded network pass into the HBM-resident
    map (placement checked against single-patch inference), then the three-phase tile stitch on structured
    maps (instance bookkeeping invariants)."""
    from hover_net_amd import infer_wsi, run_desc
    from hover_net_amd.synth import synth_pred_maps, synth_state_dict

    sd = synth_state_dict("original", 5, seed=81)
    net = _model("original", 5, sd)
    rng = np.random.default_rng(82)
    slide = infer_wsi.ArraySlide(rng.integers(0, 256, (900, 1010, 3), dtype=np.uint8))
    wsi = infer_wsi.WsiInference(net, nr_types=5, batch_size=16, chunk_shape=700, tile_shape=512, ambiguous_size=64)
    mask = np.ones((30, 34), np.uint8)
    pred = wsi.raw_prediction(slide, mask)
    assert tuple(pred.shape) == (900, 1010, 4)
    chunk, patch = infer_wsi.get_chunk_patch_info(np.array([900, 1010]), np.array([700, 700]), np.array([270, 270]), np.array([80, 80]))
    inside = [k for k in range(patch.shape[0]) if (patch[k, 0, 1] <= np.array([900, 1010])).all()]
    for k in (inside[0], inside[len(inside) // 2], inside[-1]):
        y, x = patch[k, 0, 0]
        one = run_desc.infer_step(torch.from_numpy(slide.array[y:y + 270, x:x + 270][None]), net)[0]
        got = pred[y + 95:y + 175, x + 95:x + 175].cpu().numpy()
        assert np.abs(got[..., 1:] - one[..., 1:]).max() <= 1e-5     # same kernels; batch position must not matter
    # stage 2 on a structured map with real nuclei
    maps = synth_pred_maps(1, 1100, 1300, 5, seed=83, k_lo=2, k_hi=8)[0][0]
    inst_map, info = wsi.stitch_instances(torch.from_numpy(maps).to("cuda"))
    ids = np.unique(inst_map)
    ids = ids[ids > 0]
    assert len(ids) > 400 and set(info.keys()) <= set(ids.tolist())
    assert len(info) >= 0.95 * len(ids)                                # only degenerate contours are dropped
    for i in list(info)[:50]:
        e = info[i]
        assert e["contour"].shape[1] == 2 and e["type"] is not None
    # every labelled pixel lies in the thresholded blob mask of the map
    assert ((inst_map > 0) <= (maps[..., 1] >= 0.5)).all()
    # deterministic
    inst2, info2 = wsi.stitch_instances(torch.from_numpy(maps).to("cuda"))
    np.testing.assert_array_equal(inst_map, inst2)
    assert sorted(info2) == sorted(info)


def test_valid_step_protocol():
    """run_desc.valid_step (run_desc.py:113-167): same raw-dict protocol, values from the oracle."""
    from hover_net_amd import run_desc
    from hover_net_amd.synth import synth_state_dict, synth_tiles
    from oracle import net_torch

    sd = synth_state_dict("original", 5, seed=91)
    net = _model("original", 5, sd)
    tiles = torch.from_numpy(synth_tiles(2, 270, seed=92))
    batch = {"img": tiles, "np_map": torch.zeros(2, 80, 80, dtype=torch.int32), "hv_map": torch.zeros(2, 80, 80, 2),
             "tp_map": torch.zeros(2, 80, 80, dtype=torch.int32)}
    out = run_desc.valid_step(batch, [{"net": {"desc": net}}, {}])["raw"]
    assert sorted(out) == ["imgs", "pred_hv", "pred_tp", "prob_np", "true_hv", "true_np", "true_tp"]
    want = net_torch.infer_epilogue(net_torch.forward(sd, tiles.permute(0, 3, 1, 2).float(), "original")).numpy()
    assert out["prob_np"].shape == (2, 80, 80) and out["pred_hv"].shape == (2, 80, 80, 2) and out["pred_tp"].shape == (2, 80, 80)
    assert np.abs(out["prob_np"] - want[..., 1]).max() <= TOL and np.abs(out["pred_hv"] - want[..., 2:]).max() <= TOL
    assert (out["pred_tp"] != want[..., 0]).mean() < 2e-3
    assert out["true_np"].dtype == np.int64 and out["true_hv"].dtype == np.float32
