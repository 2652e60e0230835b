"""CPU: host logic of the tile path -- patch geometry / stitching against the reference's
own `_prepare_patching` (run from its source when /root/reference exists) and known answers,
and the N>1 rank sharding + all_gather on 2 gloo processes."""
import math
import os
import re

import numpy as np
import pytest
import torch

from hover_net_amd import infer_tile as T

REF_TILE = "/root/reference/infer/tile.py"


def _ref_prepare():
    src = open(REF_TILE).read()
    m = re.search(r"def _prepare_patching\(.*?\n(?=####)", src, re.S)
    class _NP:  # numpy >= 2 dropped the np.lib.pad alias the reference uses (infer/tile.py:71)
        lib = type("lib", (), {"pad": staticmethod(np.pad)})

        def __getattr__(self, k):
            return getattr(np, k)

    ns = {"np": _NP(), "math": math}
    exec(compile(m.group(0), REF_TILE, "exec"), ns)
    return ns["_prepare_patching"]


@pytest.mark.skipif(not os.path.exists(REF_TILE), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("shape,win,msk", [((270, 270), 270, 80), ((256, 256), 256, 164), ((301, 517), 270, 80), ((1000, 731), 256, 164)])
def test_prepare_patching_matches_reference(shape, win, msk):
    img = np.random.default_rng(0).integers(0, 256, shape + (3,), dtype=np.uint8)
    pad_ref, info_ref = _ref_prepare()(img, win, msk)
    pad, info = T.prepare_patching(img, win, msk)
    np.testing.assert_array_equal(pad, pad_ref)
    np.testing.assert_array_equal(info, info_ref)


def test_prepare_patching_known_answer():
    # a 270x270 image in `original` mode -> 4 steps per axis -> 16 patches of 270x270
    img = np.zeros((270, 270, 3), np.uint8)
    pad, info = T.prepare_patching(img, 270, 80)
    assert info.shape == (16, 4) and pad.shape == (95 + 270 + 320, 95 + 270 + 320, 3)  # padt=95, padb=last_h+win-im_h=320
    assert info[:5].tolist() == [[0, 0, 0, 0], [80, 0, 1, 0], [160, 0, 2, 0], [240, 0, 3, 0], [0, 80, 0, 1]]
    assert T.extract_patches(pad, info, 270).shape == (16, 270, 270, 3)


@pytest.mark.parametrize("as_torch", [False, True])
def test_stitch_round_trip(as_torch):
    rng = np.random.default_rng(1)
    src = (301, 517)
    _, info = T.prepare_patching(np.zeros(src + (3,), np.uint8), 270, 80)
    nr, nc = info[:, 2].max() + 1, info[:, 3].max() + 1
    full = rng.normal(size=(nr * 80, nc * 80, 4)).astype(np.float32)
    patches = np.stack([full[y:y + 80, x:x + 80] for y, x, _, _ in info])
    got = T.stitch(torch.from_numpy(patches) if as_torch else patches, info, src)
    np.testing.assert_array_equal(np.asarray(got), full[:src[0], :src[1]])


def test_shard_range_partitions():
    for n in (0, 1, 5, 32, 33, 1000):
        for world in (1, 2, 3, 8):
            spans = [T.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_step(b):
    # stands in for infer_step_device: uint8 [b,H,W,3] -> float32 [b,2,2,3]
    return b.float().reshape(b.shape[0], 2, -1, 3).mean(2, keepdim=True).repeat(1, 1, 2, 1) + torch.arange(3)


def _worker(rank, world, port, n_items, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (n_items, 4, 4, 3), dtype=np.uint8))
        out = T.run_sharded(items, _fake_step, batch_size=3)
        q.put((rank, out.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [1, 7, 16])
def test_two_rank_gloo_sharding_equals_single_process(n_items):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_items
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    items = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (n_items, 4, 4, 3), dtype=np.uint8))
    want = torch.cat([_fake_step(items[i:i + 3]) for i in range(0, n_items, 3)]).numpy() if n_items else None
    # single-process batching differs from the per-rank batching, results must not
    np.testing.assert_allclose(outs[0], want, rtol=0, atol=1e-6)
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("shape,win,msk", [((270, 270, 3), 270, 80), ((1000, 1000, 3), 270, 80), ((333, 517, 3), 256, 164), ((50, 61, 3), 270, 80)])
def test_patch_grid_equals_prepare_patching(shape, win, msk):
    img = np.zeros(shape, np.uint8)
    _padded, info = T.prepare_patching(img, win, msk)
    info2, pad_tl = T.patch_grid(shape, win, msk)
    assert np.array_equal(info, info2) and pad_tl == (win - msk) // 2


@pytest.mark.gpu
@pytest.mark.parametrize("shape,win,msk", [((270, 270, 3), 270, 80), ((401, 333, 3), 270, 80), ((333, 517, 3), 256, 164), ((50, 61, 3), 270, 80)])
def test_device_patch_extraction_equals_host_reflect_padding(shape, win, msk):
    """hvn_extract_patches (reflect padding folded into the gather) against numpy's pad + crop, including an image
    smaller than the padding (repeated reflections)."""
    import torch
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    padded, info = T.prepare_patching(img, win, msk)
    want = T.extract_patches(padded, info, win)
    info2, pad_tl = T.patch_grid(shape, win, msk)
    got = T.extract_patches_device(torch.from_numpy(img).cuda(), info2, win, pad_tl).cpu().numpy()
    assert np.array_equal(got, want)


def test_process_images_bounds_the_patches_in_flight(monkeypatch):
    """A large cache round is worked off in groups of <= max_patches network patches (an oversized image goes alone); the results
    come back in input order."""
    class Net:
        mode = "original"

    calls = []

    def fake_group(images, model, nr_types, batch_size, return_centroids, return_raw):
        calls.append([im.shape[0] for im in images])
        return [("res", im.shape[0]) for im in images]

    monkeypatch.setattr(T, "_process_image_group", fake_group)
    sizes = [80, 160, 240, 1000, 80, 80]            # 1, 4, 9, 169, 1, 1 patches of an 80-pixel output step
    imgs = [np.zeros((s, s, 3), np.uint8) for s in sizes]
    out = T.process_images(imgs, Net(), max_patches=12)
    assert calls == [[80, 160], [240], [1000], [80, 80]]
    assert [o[1] for o in out] == sizes
    calls.clear()
    T.process_images(imgs, Net())
    assert calls == [sizes]
