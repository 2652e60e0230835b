"""CPU (gloo, world_size 2 and 3): the multi-rank branches of the whole-slide and tile paths.

  * `infer_tile.gather_items_to_rank0`: instance maps / record tables / contour arrays of the items each rank owns arrive on
    rank 0 as tensors (uint8 wire format, nothing pickled), ragged shapes and an empty-handed rank included;
  * `infer_tile.gather_to_rank0`: the per-batch fan-in bench.py times at N > 1;
  * `infer_tile.route_to_owners`: per-patch rows go to the one rank that stitches their image (one uneven all_to_all);
  * `WsiInference.run` on 2 / 3 ranks == the same run on 1 rank: every rank OWNS a row slab of the prediction map (even,
    contiguous deal of the patch rows), predicts exactly its slab's patches, holds slab + halo rows only (counted: less than
    the whole map), receives the halo its stage-2 tiles need in one all_to_all, works the tiles whose top row it owns and
    sends the results to rank 0.  The GPU pieces are
    replaced by deterministic CPU stand-ins (`_step`: a fixed function of the patch bytes; `_postproc_tile`: threshold +
    scipy labelling + numpy record table), the orchestration, sharding, exchange and merge code is the product's."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _spawn(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30200 + (os.getpid() * 7 + hash(target.__name__)) % 3000
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _items(rank, world, n=7):
    rng = np.random.default_rng(5)
    out = {}
    for i in range(n):
        h, w, k = int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(0, 5))
        arrs = [rng.integers(0, 99, (h, w)).astype(np.int32), rng.integers(0, 255, (k, 56)).astype(np.uint8),
                rng.integers(0, 50, (3 * k, 2)).astype(np.int32), np.arange(k + 1, dtype=np.int64) * 3]
        if i % world == rank or world == 1:
            out[i] = arrs
    return out


def _gather_worker(rank, world, port, q):
    _init(rank, world, port)
    from hover_net_amd import infer_tile as T

    got = T.gather_items_to_rank0(_items(rank, world))
    empty = T.gather_items_to_rank0({} if rank == 1 else {0: [np.zeros((2, 2), np.int32)]})      # a rank with nothing to send
    t = (torch.full((3, 4), rank, dtype=torch.int32), None, torch.arange(3) + 10 * rank)
    fan = T.gather_to_rank0(t)
    q.put((rank, (None if got is None else {k: [a.copy() for a in v] for k, v in got.items()},
                  None if empty is None else sorted(empty), None if fan is None else [None if x is None else x.numpy() for x in fan])))
    dist.destroy_process_group()


def test_item_gather_and_batch_fan_in():
    out = _spawn(_gather_worker, 2)
    got, empty, fan = out[0]
    assert out[1] == (None, None, None)
    want = _items(0, 1)
    assert sorted(got) == sorted(want)
    for k in want:
        for a, b in zip(got[k], want[k]):
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
    assert empty == [0]
    assert fan[0].tolist() == [[0] * 4] * 3 + [[1] * 4] * 3 and fan[1] is None and fan[2].tolist() == [0, 1, 2, 10, 11, 12]


# ---- WsiInference on fake GPU pieces -------------------------------------------------------------------------------------------
class _FakeNet(torch.nn.Module):
    mode, nr_types = "fast", None

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))

    def engine(self, n):
        return None


def _make_wsi(reads):
    from scipy import ndimage

    from hover_net_amd import infer_wsi, post_proc

    class Slide(infer_wsi.ArraySlide):
        def read_region(self, coords, size):
            reads.append((tuple(int(c) for c in coords), tuple(int(s) for s in size)))
            return super().read_region(coords, size)

    class W(infer_wsi.WsiInference):
        def _step(self, batch):
            # [b, 64, 64, 3] uint8 -> [b, 32, 32, 3]: a nucleus probability from the centre crop's red channel, zero h/v
            c = batch[:, 16:48, 16:48].float()
            return torch.stack([(c[..., 0] > 100).float() * 0.9, c[..., 1] * 0, c[..., 2] * 0], -1)

        def _postproc_tile(self, tile_map):
            p = tile_map[..., 0].cpu().numpy() >= 0.5
            lab, n = ndimage.label(p)
            rec = np.zeros(n, post_proc._REC_DTYPE)
            for i, sl in enumerate(ndimage.find_objects(lab)):
                m = lab[sl] == i + 1
                ys, xs = np.nonzero(m)
                rec[i] = (i + 1, m.sum(), sl[0].start, sl[0].stop, sl[1].start, sl[1].stop, float(xs.sum()), float(ys.sum()), -1, 0)
            return lab.astype(np.int32), rec

    rng = np.random.default_rng(3)
    img = np.zeros((300, 340, 3), np.uint8)
    for _ in range(60):                                            # bright blobs = "nuclei"
        y, x, r = int(rng.integers(8, 292)), int(rng.integers(8, 332)), int(rng.integers(3, 7))
        img[y - r:y + r, x - r:x + r, 0] = 200
    wsi = W(_FakeNet(), nr_types=None, batch_size=5, chunk_shape=160, tile_shape=96, ambiguous_size=8, patch_input_shape=64, patch_output_shape=32)
    wsi.device = torch.device("cpu")
    return wsi, Slide(img)


def _wsi_worker(rank, world, port, q):
    _init(rank, world, port)
    reads = []
    wsi, slide = _make_wsi(reads)
    inst, info = wsi.run(slide, mask=None)
    q.put((rank, (inst, None if info is None else {k: (v["bbox"].tolist(), v["centroid"].tolist(), v["contour"].tolist()) for k, v in info.items()},
                  reads, wsi.stage1_patches, wsi.map_rows_resident)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_wsi_run_on_several_ranks_equals_single_process(world):
    out = _spawn(_wsi_worker, world)
    reads = []
    wsi, slide = _make_wsi(reads)
    inst1, info1 = wsi.run(slide, mask=None)
    inst0, info0 = out[0][0], out[0][1]
    for r in range(1, world):
        assert out[r][0] is None and out[r][1] is None              # results live on rank 0
    np.testing.assert_array_equal(inst0, inst1)
    assert len(info1) > 20
    assert info0 == {k: (v["bbox"].tolist(), v["centroid"].tolist(), v["contour"].tolist()) for k, v in info1.items()}
    # together the ranks predict every patch exactly once, in near-equal shares (the patch rows are dealt evenly) ...
    counts = [out[r][3] for r in range(world)]
    assert sum(counts) == wsi.stage1_patches and min(counts) > 0
    assert max(counts) - min(counts) <= wsi.stage1_patches // 9 + 10       # one patch row of slack (10 patches per row here)
    # ... no rank holds the whole map: slab + the halo its tiles reach into the neighbours
    H = slide.shape[0]
    rows = [out[r][4] for r in range(world)]
    assert all(r_ < H for r_ in rows) and sum(rows) < 2 * H
    # and a rank reads only its rows of the slide
    for r in range(world):
        assert sum(sz[0] * sz[1] for _, sz in out[r][2]) < sum(sz[0] * sz[1] for _, sz in reads)


def _route_worker(rank, world, port, q):
    _init(rank, world, port)
    from hover_net_amd import infer_tile as T

    n = 23
    rng = np.random.default_rng(9)
    full = torch.from_numpy(rng.normal(size=(n, 3, 2)).astype(np.float32))
    owner = rng.integers(0, world, n)
    owner[:4] = world - 1                                           # a source shard with a single destination
    lo, hi = T.shard_range(n, rank, world)
    idx, rows = T.route_to_owners(full[lo:hi].clone(), n, owner)
    none_idx, none_rows = T.route_to_owners(full[lo:hi].clone(), n, np.zeros(n, np.int64))     # everything to rank 0
    q.put((rank, (idx, rows.numpy(), none_idx, none_rows.numpy())))
    dist.destroy_process_group()


def test_route_to_owners_three_ranks():
    world, n = 3, 23
    out = _spawn(_route_worker, world)
    rng = np.random.default_rng(9)
    full = rng.normal(size=(n, 3, 2)).astype(np.float32)
    owner = rng.integers(0, world, n)
    owner[:4] = world - 1
    for r in range(world):
        idx, rows, idx0, rows0 = out[r]
        assert np.array_equal(idx, np.flatnonzero(owner == r)) and np.array_equal(rows, full[idx])
        assert np.array_equal(idx0, np.arange(n) if r == 0 else np.zeros(0, np.int64)) and rows0.shape[0] == (n if r == 0 else 0)
    assert np.array_equal(out[0][3], full)


def test_row_slabs_and_halo_geometry():
    from hover_net_amd import infer_wsi as W

    for H, world, pin, pout in ((40000, 8, 270, 80), (300, 3, 64, 32), (5000, 2, 256, 164), (500, 8, 270, 80)):
        b = W.row_slabs(H, world, pin, pout)
        o = (pin - pout) // 2
        assert b[0] == 0 and b[-1] == H and np.all(np.diff(b) >= 0)
        assert all((x - o) % pout == 0 for x in b[1:-1] if x < H)              # interior boundaries on the patch-output grid
        n_prow = (H - (pin - pout)) // pout + 1
        rows = [(min(b[r + 1], o + n_prow * pout) - max(b[r], o)) // pout for r in range(world)]
        assert sum(rows) >= n_prow - 1 and max(rows) - min(rows) <= 1         # even deal of the patch rows
    # cfg 4: per-rank share of a 40 000^2 map on 8 ranks with 2048 + 2 x 128 tiles: slab (1/8) + halo < 1/4 of the map
    b = W.row_slabs(40000, 8, 270, 80)
    grid, boundary, cross = W.get_tile_info(np.array([40000, 40000]), np.array([2048, 2048]), 128)
    need = W.needed_rows([grid, boundary, cross], b, 40000)
    assert np.all(need[:, 0] <= b[:-1]) and np.all(need[:, 1] >= b[1:])
    assert int((need[:, 1] - need[:, 0]).max()) < 40000 // 4


# ---- tile manager on 2 ranks ------------------------------------------------------------------------------------------------------
def _manager_worker(rank, world, port, q, inp, out, free_ram=None):
    _init(rank, world, port)
    from hover_net_amd import infer_manager as im
    from hover_net_amd import infer_tile as T

    if free_ram is not None:          # ranks that see different amounts of free RAM (they sample it at different moments in real life)
        import types

        import psutil
        psutil.virtual_memory = lambda: types.SimpleNamespace(available=free_ram[rank])

    def process(images):
        """process_images' contract with the GPU pieces replaced: every rank computes the images it owns, rank 0 receives all of them."""
        mine = {}
        for i, img in enumerate(images):
            if i % world == rank:
                inst = (img[..., 0] > 127).astype(np.int32) * (i + 1)
                mine[i] = [inst, np.zeros((0, 56), np.uint8), np.zeros((0, 2), np.int32), np.zeros(1, np.int64)]
        every = T.gather_items_to_rank0(mine)
        every = mine if every is None else every
        return [(np.array(every[i][0]), {}) if i in every else None for i in range(len(images))]

    mgr = im.InferManager({"model_args": {"nr_types": None, "mode": "original"}, "model_path": None}, process_fn=process)
    args = {"input_dir": inp, "output_dir": out}
    if free_ram is None:
        args["ram_budget_bytes"] = 10 ** 12
    else:
        args["mem_usage"] = 0.5
    done = mgr.process_file_list(args)
    dist.barrier()
    q.put((rank, (done, sorted(os.listdir(out + "/mat")) if os.path.isdir(out + "/mat") else None, mgr.rounds)))
    dist.destroy_process_group()


def test_process_file_list_two_ranks_only_rank0_writes(tmp_path):
    import scipy.io as sio

    inp = tmp_path / "in"
    inp.mkdir()
    rng = np.random.default_rng(0)
    imgs = {}
    for k in range(5):
        imgs["im%d" % k] = rng.integers(0, 256, (30 + k, 40, 3), dtype=np.uint8)
        np.save(inp / ("im%d.npy" % k), imgs["im%d" % k])
    out = _spawn(_manager_worker, 2, str(inp), str(tmp_path / "out"))
    done0, mats, _rounds = out[0]
    assert done0 == ["im%d" % k for k in range(5)] and out[1][0] == []          # rank 1 reports nothing written
    assert mats == ["im%d.mat" % k for k in range(5)]
    for k in range(5):                                                             # every image complete on disk, also those rank 1 computed
        m = sio.loadmat(str(tmp_path / "out" / "mat" / ("im%d.mat" % k)))["inst_map"]
        assert np.array_equal(m, (imgs["im%d" % k][..., 0] > 127).astype(np.int32) * (k + 1))


def test_caching_rounds_agree_when_ranks_see_different_free_ram(tmp_path):
    """every rank derived its caching-round budget from its OWN psutil reading; ranks that cut the list into
    different rounds run mismatched collectives.  Rank 0 sees plenty of RAM, rank 1 room for two images: both must form [2, 2, 1]."""
    import scipy.io as sio

    from hover_net_amd import infer_manager as im

    inp = tmp_path / "in"
    inp.mkdir()
    rng = np.random.default_rng(1)
    imgs = {}
    for k in range(5):
        imgs["im%d" % k] = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
        np.save(inp / ("im%d.npy" % k), imgs["im%d" % k])
    per_image = 5 * im.padded_nbytes((30, 40, 3), 270, 80)
    out = _spawn(_manager_worker, 2, str(inp), str(tmp_path / "out"), [10 ** 13, int(2.5 * per_image / 0.5)])
    assert out[0][2] == out[1][2] == [2, 2, 1]
    assert out[0][0] == ["im%d" % k for k in range(5)]
    for k, pos in zip(range(5), (0, 1, 0, 1, 0)):          # the stand-in labels an image by its position inside its round
        m = sio.loadmat(str(tmp_path / "out" / "mat" / ("im%d.mat" % k)))["inst_map"]
        assert np.array_equal(m, (imgs["im%d" % k][..., 0] > 127).astype(np.int32) * (pos + 1))
