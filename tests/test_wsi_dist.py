"""CPU (gloo, world_size 2): the multi-rank branches of the whole-slide and tile paths.

  * `infer_tile.gather_items_to_rank0`: instance maps / record tables / contour arrays of the items each rank owns arrive on
    rank 0 as tensors (uint8 wire format, nothing pickled), ragged shapes and an empty-handed rank included;
  * `infer_tile.gather_to_rank0`: the per-batch fan-in bench.py times at N > 1;
  * `WsiInference.run` on 2 ranks == the same run on 1 rank: stage 1 shards CHUNKS by rank (each chunk read by exactly one
    rank -- counted) + one all-reduce of the map, stage 2 deals tiles round-robin and gathers to rank 0.  The GPU pieces are
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
                  reads, wsi.stage1_patches)))
    dist.destroy_process_group()


def test_wsi_run_two_ranks_equals_single_process():
    out = _spawn(_wsi_worker, 2)
    reads = []
    wsi, slide = _make_wsi(reads)
    inst1, info1 = wsi.run(slide, mask=None)
    inst0, info0, reads0, n0 = out[0]
    inst_r1, info_r1, reads1, n1 = out[1]
    assert inst_r1 is None and info_r1 is None                      # results live on rank 0
    np.testing.assert_array_equal(inst0, inst1)
    assert len(info1) > 20
    assert info0 == {k: (v["bbox"].tolist(), v["centroid"].tolist(), v["contour"].tolist()) for k, v in info1.items()}
    # every chunk is read by exactly one rank, and together the ranks predict every patch exactly once
    assert len(reads) > 2 and sorted(reads0 + reads1) == sorted(reads) and not set(reads0) & set(reads1)
    assert n0 + n1 == wsi.stage1_patches and n0 > 0 and n1 > 0


# ---- tile manager on 2 ranks ------------------------------------------------------------------------------------------------------
def _manager_worker(rank, world, port, q, inp, out):
    _init(rank, world, port)
    from hover_net_amd import infer_manager as im
    from hover_net_amd import infer_tile as T

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
    done = mgr.process_file_list({"input_dir": inp, "output_dir": out, "ram_budget_bytes": 10 ** 12})
    dist.barrier()
    q.put((rank, (done, sorted(os.listdir(out + "/mat")) if os.path.isdir(out + "/mat") else None)))
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
    done0, mats = out[0]
    assert done0 == ["im%d" % k for k in range(5)] and out[1][0] == []          # rank 1 reports nothing written
    assert mats == ["im%d.mat" % k for k in range(5)]
    for k in range(5):                                                             # every image complete on disk, also those rank 1 computed
        m = sio.loadmat(str(tmp_path / "out" / "mat" / ("im%d.mat" % k)))["inst_map"]
        assert np.array_equal(m, (imgs["im%d" % k][..., 0] > 127).astype(np.int32) * (k + 1))
