"""CPU: host half of the GPU input pipeline (hover_net_amd/augment.py): the parameter record layout (== hvn_aug_sample of
include/hvn.h), the draw distributions of train_loader.py:123-187, and the epoch / rank slicing of DevicePatchLoader with the
device pipeline stubbed out."""
import re

import numpy as np

from hover_net_amd import augment as G


def test_record_layout_matches_the_c_struct():
    hdr = open("include/hvn.h").read()
    body = hdr[hdr.index("typedef struct hvn_aug_sample {"):hdr.index("} hvn_aug_sample;")]
    fields = re.findall(r"(double|int32_t|float)\s+([^;]+);", body)
    names = []
    for _ty, decl in fields:
        for d in decl.split(","):
            names.append(re.sub(r"\[.*", "", d.strip()))
    assert names == list(G.AUG_DTYPE.names)
    assert G.AUG_DTYPE.itemsize == 128 and G.AUG_DTYPE.fields["order"][1] == 80 and G.AUG_DTYPE.fields["hue"][1] == 96


def test_draws_follow_the_reference_ranges():
    rng = np.random.default_rng(0)
    prm = G.draw_params(rng, np.arange(4000) % 7, 540, 540)
    assert set(np.unique(prm["kind"])) == {0, 1, 2} and set(np.unique(prm["p0"])) == {1, 3, 5} and set(np.unique(prm["p1"])) == {1, 3, 5}
    assert abs(prm["flip_lr"].mean() - 0.5) < 0.03 and abs(prm["flip_ud"].mean() - 0.5) < 0.03 and abs(prm["per_channel"].mean() - 0.5) < 0.03
    assert prm["hue"].min() >= -8 and prm["hue"].max() <= 8 and prm["sat"].min() >= 0.8 and prm["sat"].max() <= 1.2
    assert prm["bright"].min() >= -26 and prm["bright"].max() <= 26 and prm["contrast"].min() >= 0.75 and prm["contrast"].max() <= 1.25
    assert prm["noise_scale"].min() >= 0 and prm["noise_scale"].max() <= 12.75
    assert all(sorted(o) == [0, 1, 2, 3] for o in prm["order"]) and len({tuple(o) for o in prm["order"]}) == 24
    # the inverse matrices invert matrices with scale in [0.8, 1.2], |shear| <= 5 deg: determinant of the forward map = sx sy cos(shear)
    det = 1.0 / (prm["inv"][:, 0] * prm["inv"][:, 4] - prm["inv"][:, 1] * prm["inv"][:, 3])
    assert det.min() >= 0.8 * 0.8 * np.cos(np.deg2rad(5)) - 1e-9 and det.max() <= 1.44 + 1e-9
    # the centre moves by the translation only: +-1 % of the size
    c = np.array([269.5, 269.5, 1.0])
    fwd_c = np.array([np.linalg.inv(np.vstack([p.reshape(2, 3), [0, 0, 1]])) @ c for p in prm["inv"][:200]])
    assert np.abs(fwd_c[:, :2] - 269.5).max() <= 5.4 + 1e-6
    ident = G.identity_params(3)
    assert ident["inv"].tolist() == [[1, 0, 0, 0, 1, 0]] * 3 and (ident["order"] == -1).all() and (ident["kind"] == 3).all()


def test_loader_epochs_and_rank_slices(monkeypatch):
    data = np.zeros((11, 8, 8, 5), np.int32)
    seen = {}
    for rank in range(2):
        ld = G.DevicePatchLoader(data, (4, 4), (2, 2), batch_size=2, mode="train", with_type=True, seed=3, device="cpu", rank=rank, world=2)
        monkeypatch.setattr(ld, "batch", lambda prm, noise=None, generator=None: prm["src"].copy())
        assert len(ld) == 2                                      # 11 // 2 = 5 patches on EVERY rank, ragged batch dropped
        seen[rank] = [np.concatenate(list(ld)) for _epoch in range(2)]
    for epoch in range(2):
        a, b = seen[0][epoch], seen[1][epoch]
        assert len(a) == len(b) == 4 and len(set(a) & set(b)) == 0           # disjoint slices of ONE permutation, same step count
    assert not np.array_equal(seen[0][0], seen[0][1])                            # reshuffled every epoch
    ld = G.DevicePatchLoader(data, (4, 4), (2, 2), batch_size=4, mode="valid", device="cpu")
    monkeypatch.setattr(ld, "batch", lambda prm, noise=None, generator=None: (prm["src"].copy(), prm["kind"].copy()))
    out = list(ld)
    assert [o[0].tolist() for o in out] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]] and all((o[1] == 3).all() for o in out)   # in order, ragged batch kept


def test_loader_reads_patch_files_one_by_one(tmp_path):
    rng = np.random.default_rng(0)
    paths = []
    for i in range(3):
        d = rng.integers(0, 200, (8, 9, 5)).astype(np.int32)
        np.save(tmp_path / ("p%d.npy" % i), d)
        paths.append(str(tmp_path / ("p%d.npy" % i)))
    ld = G.DevicePatchLoader(paths, (4, 4), (2, 2), batch_size=2, mode="valid", with_type=True, device="cpu")
    assert ld.img.shape == (3, 8, 9, 3) and ld.img.dtype.is_floating_point is False and ld.ann.shape == (3, 8, 9, 2)
    assert np.array_equal(ld.ann[1].numpy(), np.load(paths[1])[..., 3:5]) and np.array_equal(ld.img[2].numpy(), np.load(paths[2])[..., :3].astype(np.uint8))
    np.save(tmp_path / "bad.npy", np.zeros((7, 9, 5), np.int32))
    import pytest
    with pytest.raises(ValueError, match="has shape"):
        G.DevicePatchLoader(paths + [str(tmp_path / "bad.npy")], (4, 4), (2, 2), batch_size=2, device="cpu")


def test_device_loaders_callback_for_run_phases():
    from hover_net_amd import train

    data = np.zeros((6, 300, 300, 5), np.int32)
    make = train.device_loaders(data, data[:2], "original", with_type=True, seed=4, device="cpu")
    l0 = make(0, {"train": 2, "valid": 2})
    assert len(l0["train"]) == 3 and len(l0["valid"]) == 1 and l0["train"].input_shape == (270, 270) and l0["train"].mask_shape == (80, 80)
    l1 = make(1, {"train": 4, "valid": 1})
    assert l1["train"] is l0["train"] and len(l1["train"]) == 1 and len(l1["valid"]) == 2      # same resident set, phase-1 batch sizes
    assert train.device_loaders(data, None, "fast", False, device="cpu")(0, {"train": 2, "valid": 2})["valid"] is None
    assert train.device_loaders(data, None, "fast", False, device="cpu")(0, {"train": 2, "valid": 2})["train"].mask_shape == (164, 164)
