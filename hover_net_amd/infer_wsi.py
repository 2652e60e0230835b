"""Whole-slide inference -- the data path of `infer/wsi.py:InferManager.process_single_file`
(/root/reference/infer/wsi.py:449-709) with the prediction map resident in HBM.

Kept from the reference (pinned by tests/test_infer_wsi.py against the reference's own functions):
  * geometry: `_get_patch_top_left_info` (wsi.py:64-88), `_get_tile_info` (:92-151),
    `_get_chunk_patch_info` (:155-221), the chunk -> patch selection rule (:341-351) and the
    mask test of `__select_valid_patches` (:300-327);
  * the three post-processing phases over grid / boundary / cross tiles and the sequential merge
    rules of `post_proc_normal_tile_callback` (:569-599) and `post_proc_fixing_tile_callback`
    (:602-677), including the reference's id offsetting by the running maximum id.

Different by design:
  * `pred_map` ([H,W,3|4] float32, 25.6 GB for a 40k x 40k slide) lives in HBM (288 GB per GPU)
    instead of a disk memmap written by a helper process (wsi.py:520-534, 235-258); patch outputs
    are scattered into it with one indexed write per chunk;
  * several GPUs (one process each) OWN the map by rows: the slide's patch rows are dealt evenly into `world` contiguous
    row slabs (`row_slabs`); a rank predicts exactly the patches of its slab (its share of every chunk: it reads only the
    rows of the chunk it needs) into a local tensor of slab + halo rows -- 1 / world of the map, not a copy of it --, and
    the only exchange of stage 1 is the halo: the rows a rank's stage-2 tiles reach into its neighbours' slabs, one
    `all_to_all_single` of row blocks (`exchange_halo`), instead of an all-reduce of `world` full maps;
  * a tile (2048 x 2048 + margins) is post-processed on the GPU (`post_proc.process_batch_device`:
    per-component parallel watershed) instead of a 16-process CPU pool, by the rank whose slab holds the tile's top row;
    instance maps / record tables / contours travel to rank 0 as tensors, and rank 0 applies the merge callbacks in
    tile order;
  * the slide backend is any object with `.shape` and `.read_region((x, y), (w, h))`
    (`ArraySlide` wraps a numpy array / memmap; OpenSlide is not required).
"""
import time

import numpy as np
import torch

from . import infer_tile, post_proc, run_desc


# --------------------------------------------------------------------------------------------
# geometry
def get_patch_top_left_info(img_shape, input_size, output_size):
    img_shape, input_size, output_size = (np.asarray(a) for a in (img_shape, input_size, output_size))
    diff = input_size - output_size
    nr_step = np.floor((img_shape - diff) / output_size) + 1
    last = (diff // 2) + nr_step * output_size
    ys = np.arange(diff[0] // 2, last[0], output_size[0], dtype=np.int32)
    xs = np.arange(diff[1] // 2, last[1], output_size[1], dtype=np.int32)
    # np.meshgrid(ys, xs) flattened: x outer, y inner
    out_tl = np.stack([np.tile(ys, xs.shape[0]), np.repeat(xs, ys.shape[0])], axis=-1)
    return out_tl - diff // 2, out_tl


def get_tile_info(img_shape, tile_shape, ambiguous_size=128):
    img_shape, tile_shape = np.asarray(img_shape), np.asarray(tile_shape)
    tl, _ = get_patch_top_left_info(img_shape, tile_shape, tile_shape)
    br = np.minimum(tl + tile_shape, img_shape)
    grid = np.stack([tl, br], axis=1)
    gx, gy = np.unique(tl[:, 1]), np.unique(tl[:, 0])
    a = ambiguous_size

    def boxes(y_lo, x_lo, y_hi, x_hi):
        # np.meshgrid(first, second) flattened pairs: second outer, first inner
        def mesh(f, s):
            return np.tile(f, s.shape[0]), np.repeat(s, f.shape[0])

        ly, lx = mesh(y_lo, x_lo)
        hy, hx = mesh(y_hi, x_hi)
        return np.stack([np.stack([ly, lx], -1), np.stack([hy, hx], -1)], axis=1)

    bx = boxes(gy, gx[1:] - a, gy + tile_shape[0], gx[1:] + a)
    # the reference builds the horizontal strips with meshgrid(y, x) as well -> y inner, x outer
    by = boxes(gy[1:] - a, gx, gy[1:] + a, gx + tile_shape[1])
    cross = boxes(gy[1:] - 2 * a, gx[1:] - 2 * a, gy[1:] + 2 * a, gx[1:] + 2 * a)
    return grid, np.concatenate([bx, by], axis=0), cross


def get_chunk_patch_info(img_shape, chunk_input_shape, patch_input_shape, patch_output_shape):
    img_shape, chunk_input_shape, pin, pout = (np.asarray(a) for a in (img_shape, chunk_input_shape, patch_input_shape, patch_output_shape))
    rnd = lambda x, y: np.floor(x / y) * y  # noqa: E731
    diff = pin - pout
    chunk_out = rnd(chunk_input_shape - diff, pout).astype(np.int64)
    chunk_in = (chunk_out + diff).astype(np.int64)
    p_in_tl, _ = get_patch_top_left_info(img_shape, pin, pout)
    p_out_tl = p_in_tl + diff
    patch_info = np.stack([np.stack([p_in_tl, p_in_tl + pin], axis=1), np.stack([p_out_tl, p_out_tl + pout], axis=1)], axis=1)
    c_in_tl, _ = get_patch_top_left_info(img_shape, chunk_in, chunk_out)
    c_in_br = c_in_tl + chunk_in
    for ax in (0, 1):  # keep the chunk inside the slide, on the patch grid
        sel = np.nonzero(c_in_br[:, ax] > img_shape[ax])[0]
        c_in_br[sel, ax] = rnd((img_shape[ax] - diff[ax]) - c_in_tl[sel, ax], pout[ax]) + c_in_tl[sel, ax] + diff[ax]
    chunk_info = np.stack([np.stack([c_in_tl, c_in_br], axis=1),
                           np.stack([c_in_tl + diff // 2, c_in_br - diff // 2], axis=1)], axis=1)
    return chunk_info, patch_info


def select_valid(info_list, mask, proc_shape, has_output_info=True):
    """Indices of boxes whose (output) area touches the tissue mask (wsi.py:300-327)."""
    ratio = mask.shape[0] / proc_shape[0]
    keep = []
    for i in range(info_list.shape[0]):
        box = np.squeeze(info_list[i])
        box = np.rint((box[1] if has_output_info else box) * ratio).astype(np.int64)
        if np.sum(mask[box[0][0]:box[1][0], box[0][1]:box[1][1]]) > 0:
            keep.append(i)
    return info_list[keep]


# --------------------------------------------------------------------------------------------
class ArraySlide:
    """Slide backend over an in-memory / memory-mapped uint8 [H,W,3] array (misc/wsi_handler.py API subset)."""

    def __init__(self, array):
        self.array = array
        self.shape = array.shape

    def read_region(self, coords, size):
        x, y = int(coords[0]), int(coords[1])
        w, h = int(size[0]), int(size[1])
        return np.asarray(self.array[y:y + h, x:x + w, :3])

    def thumbnail(self, scale=32):
        """Sub-sampled view (40x -> 1.25x for scale 32), the input of the tissue-mask heuristic."""
        return np.asarray(self.array[::scale, ::scale, :3])


class TiledSlide:
    """Synthetic slide: one uint8 [h,w,3] texture repeated over a virtual H x W extent; `read_region` composes the request
    from the texture by modular indexing, so a 40 000^2 slide (4.8 GB as an array) costs nothing until it is read."""

    def __init__(self, tile, shape):
        self.tile = np.ascontiguousarray(tile[..., :3])
        self.shape = (int(shape[0]), int(shape[1]), 3)

    def read_region(self, coords, size):
        x, y = int(coords[0]), int(coords[1])
        w, h = int(size[0]), int(size[1])
        th, tw = self.tile.shape[:2]
        rows = (np.arange(y, min(y + h, self.shape[0])) % th)
        cols = (np.arange(x, min(x + w, self.shape[1])) % tw)
        return self.tile[rows][:, cols]

    def thumbnail(self, scale=32):
        th, tw = self.tile.shape[:2]
        return self.tile[(np.arange(0, self.shape[0], scale) % th)][:, (np.arange(0, self.shape[1], scale) % tw)]


# --------------------------------------------------------------------------------------------
# row ownership of the prediction map (multi-GPU)
def row_slabs(n_rows, world, pin, pout):
    """[world + 1] row boundaries: rank r owns map rows [b[r], b[r+1]).  The patch rows of the slide (output rows
    o + j * pout, o = (pin - pout) // 2: `get_patch_top_left_info`) are dealt evenly and contiguously; interior boundaries
    lie ON the patch-output grid, so no patch output straddles two owners."""
    o = (int(pin) - int(pout)) // 2
    n_prow = int(np.floor((n_rows - (int(pin) - int(pout))) / int(pout)) + 1)
    cut = [(r * n_prow) // world for r in range(world + 1)]
    b = [0] + [min(n_rows, o + c * int(pout)) for c in cut[1:-1]] + [int(n_rows)]
    return np.asarray(b, np.int64)


class SlabMap:
    """The rows [row0, row0 + t.shape[0]) of the [H, W, C] prediction map, resident on this rank."""

    def __init__(self, t, row0=0):
        self.t, self.row0 = t, int(row0)

    @property
    def rows(self):
        return self.row0, self.row0 + int(self.t.shape[0])

    def window(self, tl, br):
        lo, hi = self.rows
        y0, y1 = int(tl[0]), min(int(br[0]), hi)
        if y0 < lo:
            raise IndexError("rows [%d, %d) asked of a slab holding [%d, %d)" % (y0, y1, lo, hi))
        return self.t[y0 - lo:y1 - lo, int(tl[1]):int(br[1])]


def tile_owner(tiles, bounds):
    """Rank that post-processes each tile: the owner of its top row."""
    if tiles.shape[0] == 0:
        return np.zeros(0, np.int64)
    return np.clip(np.searchsorted(bounds, tiles[:, 0, 0], side="right") - 1, 0, len(bounds) - 2)


def needed_rows(tile_lists, bounds, n_rows):
    """[world, 2]: the row range each rank must hold = its slab + what its tiles (all three phases) reach beyond it."""
    world = len(bounds) - 1
    need = np.stack([bounds[:-1], bounds[1:]], 1).astype(np.int64)
    for tiles in tile_lists:
        own = tile_owner(tiles, bounds)
        for r in range(world):
            t = tiles[own == r]
            if t.shape[0]:
                need[r, 0] = min(need[r, 0], int(t[:, 0, 0].min()))
                need[r, 1] = max(need[r, 1], min(int(n_rows), int(t[:, 1, 0].max())))
    return need


def exchange_halo(local, need, bounds):
    """local: SlabMap over rows need[rank] whose own slab rows are final.  Fills the rest from the owners: rank s sends rank t
    the rows of need[t] that lie in slab[s] -- every rank derives every block size from (need, bounds), so ONE all_to_all_single of
    contiguous row blocks does it (xGMI is point to point: each block crosses one link once)."""
    dist, rank, world = infer_tile._dist()
    if world == 1:
        return local
    row = int(np.prod(local.t.shape[1:], dtype=np.int64))

    def block(holder, asker):      # rows of `holder`'s slab that `asker` needs
        if holder == asker:
            return 0, 0
        lo, hi = max(int(need[asker, 0]), int(bounds[holder])), min(int(need[asker, 1]), int(bounds[holder + 1]))
        return (lo, hi) if hi > lo else (0, 0)

    send_blocks = [block(rank, t) for t in range(world)]
    recv_blocks = [block(s, rank) for s in range(world)]
    parts = [local.t[lo - local.row0:hi - local.row0].reshape(-1) for lo, hi in send_blocks if hi > lo]
    send = torch.cat(parts) if parts else local.t.new_zeros(0)
    recv = local.t.new_empty(sum(hi - lo for lo, hi in recv_blocks) * row)
    infer_tile.a2a_single(recv, send, [(hi - lo) * row for lo, hi in recv_blocks], [(hi - lo) * row for lo, hi in send_blocks])
    off = 0
    for lo, hi in recv_blocks:
        if hi > lo:
            n = (hi - lo) * row
            local.t[lo - local.row0:hi - local.row0] = recv[off:off + n].view((hi - lo,) + tuple(local.t.shape[1:]))
            off += n
    return local


def remove_inst(inst_map, ids):
    if len(ids):
        inst_map[np.isin(inst_map, np.asarray(list(ids)))] = 0
    return inst_map


class WsiMerger:
    """Host state of the three-phase stitch: global instance map + instance dict, updated strictly in tile
    order (the callbacks of wsi.py:569-677 'must be in sequential ordering')."""

    def __init__(self, proc_shape):
        self.inst_map = np.zeros(tuple(proc_shape), np.int32)
        self.inst_info = {}
        self._flag = np.zeros(1 << 16, np.uint8)       # scratch lookup table over instance ids, all zero between calls
        self._ids = []                                 # max-heap (negated) of the ids inserted so far, lazily pruned

    def _max_id(self):
        """max(self.inst_info), the reference's id offset (wsi.py:574/621: `max(wsi_inst_info.keys())` over up to a million
        keys per tile).  Kept exact without assuming anything about the order in which ids arrive: a heap of all inserted ids,
        popped while its top has been removed from the dict -- amortised O(log n) per inserted id."""
        import heapq

        while self._ids and -self._ids[0] not in self.inst_info:
            heapq.heappop(self._ids)
        return -self._ids[0] if self._ids else 0

    def _insert(self, key, entry):
        import heapq

        self.inst_info[key] = entry
        heapq.heappush(self._ids, -key)

    def _table(self, size):
        if self._flag.shape[0] < size:
            self._flag = np.zeros(max(size, 2 * self._flag.shape[0]), np.uint8)
        return self._flag

    def _present(self, a):
        """np.unique(a) for a non-negative id array, by table lookup instead of a sort (a boundary strip is 0.5 Mpix and
        there are ~1100 of them on a 40 000^2 slide)."""
        lo, hi = int(a.min()), int(a.max())
        f = self._table(hi + 1)
        f[a.ravel()] = 1
        ids = np.flatnonzero(f[lo:hi + 1]) + lo
        f[ids] = 0
        return ids

    def _zero_ids(self, a, ids):
        """remove_inst: a[isin(a, ids)] = 0 by table lookup."""
        if len(ids):
            f = self._table(int(a.max()) + 1)
            ids = ids[ids < f.shape[0]]
            f[ids] = 1
            a[f[a] != 0] = 0
            f[ids] = 0
        return a

    def normal(self, pred_inst, info, tile_tl, tile_br, shifted=False):
        """shifted=True: the entries of `info` already carry the tile origin (post_proc.records_to_dict(shift_xy=...))."""
        if len(info) == 0:
            return
        top_left = np.asarray(tile_tl)[::-1]
        off = self._max_id()
        for i, e in info.items():
            if not shifted:
                e["bbox"] = e["bbox"] + top_left        # (sic) the reference adds (x, y) to the (row, col) box
                e["contour"] = e["contour"] + top_left
                e["centroid"] = e["centroid"] + top_left
            self._insert(i + off, e)
        pred_inst = pred_inst.copy()
        pred_inst[pred_inst > 0] += off
        self.inst_map[tile_tl[0]:tile_br[0], tile_tl[1]:tile_br[1]] = pred_inst

    def fixing(self, pred_inst, info, tile_tl, tile_br, shifted=False):
        if len(info) == 0:
            return
        top_left = np.asarray(tile_tl)[::-1]
        off = self._max_id()                                     # before any removal (wsi.py:621-624)
        roi = self.inst_map[tile_tl[0]:tile_br[0], tile_tl[1]:tile_br[1]].copy()
        edge = np.concatenate([roi[[0, -1], :].ravel(), roi[:, [0, -1]].ravel()])
        on_edge = np.unique(edge)[1:]                            # "[1:]  # exclude background" (wsi.py:631, as is: drops the smallest)
        inner = self._present(roi)[1:]
        inner = np.setdiff1d(inner, on_edge, assume_unique=True)
        roi = self._zero_ids(roi, inner)                         # old nuclei fully inside the strip are replaced
        for i in inner.tolist():
            self.inst_info.pop(i, None)
        pred_inst = pred_inst.copy()
        n_local = int(pred_inst.max()) + 1
        touching = np.flatnonzero(np.bincount(pred_inst[roi > 0], minlength=n_local))    # new nuclei overlapping the kept (split) ones
        new_inner = np.setdiff1d(np.flatnonzero(np.bincount(pred_inst.ravel(), minlength=n_local))[1:], touching, assume_unique=True)
        pred_inst = self._zero_ids(pred_inst, touching)
        for i in new_inner.tolist():
            if i not in info:                                    # contour had < 3 points (wsi.py:655-657)
                continue
            e = info[i]
            if not shifted:
                e["bbox"] = e["bbox"] + top_left
                e["contour"] = e["contour"] + top_left
                e["centroid"] = e["centroid"] + top_left
            self._insert(i + off, e)
        pred_inst[pred_inst > 0] += off
        self.inst_map[tile_tl[0]:tile_br[0], tile_tl[1]:tile_br[1]] = roi + pred_inst


class DeviceMerger:
    """`WsiMerger` with the instance map in HBM: the same callbacks in the same tile order, each tile's array work done by
    `hvn_wsi_merge_normal` / `hvn_wsi_merge_fixing` (csrc/hvn_wsi_merge.hip) on a side stream.  What stays on the host is the
    dictionary: per fix-up tile the list of removed ids and the "touching" flags of the new ids come back (a few hundred bytes),
    which also yields the next tile's id offset -- the one true sequential dependency of wsi.py:569-677."""

    def __init__(self, proc_shape, device, cap=1 << 20):
        import ctypes

        from . import lib as L

        self._L, self._ct = L, ctypes
        self.device = torch.device(device)
        self.inst_map = torch.zeros((int(proc_shape[0]), int(proc_shape[1])), dtype=torch.int32, device=self.device)
        self.inst_info = {}
        self._ids = []
        self.stream = torch.cuda.Stream(self.device, priority=-1)
        self.cap = int(cap)                  # id-table size: grows on demand
        self.epoch = 0
        self._max_map_id = 0                 # largest id ever written to the map (ids without a dict entry included)
        self.removed_cap = 1 << 16
        self.stream.wait_stream(torch.cuda.current_stream(self.device))      # inst_map's zero-fill ran on the caller's stream
        with torch.cuda.stream(self.stream):     # every table is (re)allocated and zero-filled ON the merge stream: ordered before the kernels that stamp it
            self.flags = torch.zeros((2, self.cap), dtype=torch.int32, device=self.device)
            self.removed = torch.zeros(self.removed_cap, dtype=torch.int32, device=self.device)
            self.counters = torch.zeros(5, dtype=torch.int32, device=self.device)
            self.touching = torch.empty(1 << 16, dtype=torch.uint8, device=self.device)      # mg_init clears it per tile
        self._h_counters = torch.zeros(5, dtype=torch.int32, pin_memory=True)
        self._h_removed = torch.zeros(self.removed_cap, dtype=torch.int32, pin_memory=True)
        self._h_touching = torch.zeros(1 << 16, dtype=torch.uint8, pin_memory=True)

    _max_id = WsiMerger._max_id
    _insert = WsiMerger._insert

    def _pred(self, pred_inst, ready):
        """The tile's local-id map on the device: the post-processing output itself (`ready` = its event), or an upload of the host
        array a remote rank sent."""
        if torch.is_tensor(pred_inst) and pred_inst.is_cuda:
            if ready is not None:
                self.stream.wait_event(ready)
            return pred_inst.contiguous()
        return torch.from_numpy(np.ascontiguousarray(pred_inst, np.int32)).to(self.device, non_blocking=True)

    def normal(self, pred_inst, info, tile_tl, tile_br, ready=None, n_local=None):
        if len(info) == 0:
            return
        off = self._max_id()
        if n_local is None:                                      # ids without a dict entry (contours under 3 points) can exceed max(info)
            n_local = int(pred_inst.max())
        self._max_map_id = max(self._max_map_id, off + int(max(n_local, max(info))))
        for i, e in info.items():
            self._insert(i + off, e)
        with torch.cuda.stream(self.stream):
            p = self._pred(pred_inst, ready)
            h, w = int(p.shape[0]), int(p.shape[1])
            self._L.check(self._L.lib().hvn_wsi_merge_normal(self.inst_map.data_ptr(), self.inst_map.shape[1], int(tile_tl[0]), int(tile_tl[1]), h, w,
                                                             p.data_ptr(), int(off), self._ct.c_void_p(self.stream.cuda_stream)), "hvn_wsi_merge_normal")
            p.record_stream(self.stream)

    def fixing(self, pred_inst, info, tile_tl, tile_br, ready=None, n_local=None):
        if len(info) == 0:
            return
        off = self._max_id()                                     # before any removal (wsi.py:621-624)
        if n_local is None:                                      # the tile's largest label: new ids WITHOUT a dict entry are written / dropped too
            n_local = int(pred_inst.max())
        n_local = int(max(n_local, max(info)))
        if self._max_map_id + 1 >= self.cap:                     # the id tables index every id in the map (also ids without a dict entry)
            self.cap = max(2 * self.cap, self._max_map_id + 2)
            with torch.cuda.stream(self.stream):                 # zero-fill ordered before mg_scan's stamps (not on the caller's stream)
                self.flags = torch.zeros((2, self.cap), dtype=torch.int32, device=self.device)
        self._max_map_id = max(self._max_map_id, off + n_local)
        if n_local + 1 > self.touching.shape[0]:
            with torch.cuda.stream(self.stream):
                self.touching = torch.empty(2 * (n_local + 1), dtype=torch.uint8, device=self.device)
            self._h_touching = torch.zeros(2 * (n_local + 1), dtype=torch.uint8, pin_memory=True)
        self.epoch += 1
        with torch.cuda.stream(self.stream):
            p = self._pred(pred_inst, ready)
            h, w = int(p.shape[0]), int(p.shape[1])
            self._L.check(self._L.lib().hvn_wsi_merge_fixing(
                self.inst_map.data_ptr(), self.inst_map.shape[1], int(tile_tl[0]), int(tile_tl[1]), h, w, p.data_ptr(), n_local, int(off), self.epoch,
                self.flags.data_ptr(), self.cap, self.removed.data_ptr(), self.removed_cap, self.counters.data_ptr(), self.touching.data_ptr(),
                self._ct.c_void_p(self.stream.cuda_stream)), "hvn_wsi_merge_fixing")
            p.record_stream(self.stream)
            self._h_counters.copy_(self.counters, non_blocking=True)
            self._h_removed[:4096].copy_(self.removed[:4096], non_blocking=True)
            self._h_touching[:n_local + 1].copy_(self.touching[:n_local + 1], non_blocking=True)
        self.stream.synchronize()
        n_rem = int(self._h_counters[0])
        if int(self._h_counters[4]):
            raise RuntimeError("%d window pixels carry ids beyond the id tables (cap %d): the tables were sized from a too small n_local"
                               % (int(self._h_counters[4]), self.cap))
        if n_rem > self.removed_cap:
            raise RuntimeError("%d instances removed by one fix-up tile: beyond the list of %d" % (n_rem, self.removed_cap))
        rem = self._h_removed[:n_rem].numpy() if n_rem <= 4096 else self.removed[:n_rem].cpu().numpy()
        for i in rem.tolist():
            self.inst_info.pop(i, None)
        touching = self._h_touching.numpy()
        pred_min = int(self._h_counters[3])                      # np.unique(pred_inst)[1:] (wsi.py:646): a tile without background loses its smallest id
        for i, e in info.items():                                # ascending local id, like `for inst_id in inner_inst_list`
            if not touching[i] and not (pred_min > 0 and i == pred_min):
                self._insert(i + off, e)

    def result(self):
        self.stream.synchronize()
        return self.inst_map.cpu().numpy(), self.inst_info


# --------------------------------------------------------------------------------------------
class WsiInference:
    def __init__(self, model, nr_types=None, batch_size=32, chunk_shape=10000, tile_shape=2048, ambiguous_size=128,
                 patch_input_shape=None, patch_output_shape=None):
        net = model.module if hasattr(model, "module") and not hasattr(model, "engine") else model
        self.model, self.nr_types, self.batch_size = model, nr_types, batch_size
        pin = patch_input_shape or (270 if net.mode == "original" else 256)     # run_infer.py:145-150
        pout = patch_output_shape or (80 if net.mode == "original" else 164)
        self.pin, self.pout = np.array([pin, pin]), np.array([pout, pout])
        self.chunk_shape = np.array([chunk_shape, chunk_shape])
        self.tile_shape = np.array([tile_shape, tile_shape], np.int64)
        self.ambiguous_size = ambiguous_size
        self.device = next(net.parameters()).device
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)      # one process per GPU: ctypes HIP launches and collectives use the current device
        self.out_ch = 3 if nr_types is None else 4

    # -- stage 1: raw prediction into the HBM-resident map ----------------------------------------
    def tile_lists(self, shape, mask):
        """The three phases' tile boxes after the mask test -- the same on every rank (pure geometry + mask)."""
        grid, boundary, cross = get_tile_info(shape, self.tile_shape, self.ambiguous_size)
        return [select_valid(t, mask, shape, has_output_info=False) for t in (grid, boundary, cross)]

    def raw_prediction(self, slide, mask, as_slab=False):
        """The reference's chunk loop (wsi.py:329-383; every patch belongs to the FIRST chunk that selects it), restricted on each
        rank to the patch rows of its row slab (`row_slabs`): a rank reads, of every chunk that crosses its slab, only the rows
        it needs, predicts those patches and scatters them into its LOCAL tensor (slab + halo rows, 1 / world of the map).  The
        halo rows -- what its stage-2 tiles reach into the neighbours' slabs -- then arrive in one `all_to_all_single`.
        Returns the whole map as a tensor (one rank), or a `SlabMap` (as_slab=True; always on several ranks)."""
        t_start = time.perf_counter()
        shape = np.array(slide.shape[:2])
        H, W = int(shape[0]), int(shape[1])
        dist, rank, world = infer_tile._dist()
        bounds = row_slabs(H, world, self.pin[0], self.pout[0])
        need = needed_rows(self.tile_lists(shape, mask), bounds, H) if world > 1 else np.array([[0, H]])
        lo, hi = int(need[rank, 0]), int(need[rank, 1])
        local = SlabMap(torch.zeros((hi - lo, W, self.out_ch), dtype=torch.float32, device=self.device), lo)
        self.map_rows_resident = hi - lo                       # inspected by the tests / tools: per-rank share of the map
        chunk_info, patch_info = get_chunk_patch_info(shape, self.chunk_shape, self.pin, self.pout)
        between = lambda x, a, b: (a <= x) & (x <= b)  # noqa: E731
        h = int(self.pout[0])
        off = (self.pin - self.pout) // 2
        ar = torch.arange(h, device=self.device)
        out_top = patch_info[:, 0, 0, 0] + off[0]               # output top row of every patch in the slide
        in_slab = (out_top >= bounds[rank]) & (out_top < bounds[rank + 1])
        taken = np.zeros(patch_info.shape[0], bool)
        self.stage1_patches = 0
        for ci in range(chunk_info.shape[0]):
            chunk = chunk_info[ci]
            start, end = chunk[0, 0], chunk[0, 1] - self.pin
            sel = between(patch_info[:, 0, 0, 0], start[0], end[0]) & between(patch_info[:, 0, 0, 1], start[1], end[1]) & ~taken
            taken |= sel
            sel &= in_slab
            if not sel.any():
                continue
            plist = select_valid(np.array(patch_info[sel]), mask, shape)
            if plist.shape[0] == 0:
                continue
            self.stage1_patches += int(plist.shape[0])
            # this rank's rows of the chunk: from its first patch's input top to its last patch's input bottom, full chunk width
            y0 = int(plist[:, 0, 0, 0].min())
            y1 = int(plist[:, 0, 1, 0].max())
            x0, x1 = int(chunk[0][0][1]), int(chunk[0][1][1])
            region = slide.read_region((x0, y0), (x1 - x0, y1 - y0))
            rel = plist[:, 0, 0] - np.array([y0, x0])           # patch input top-left inside the region
            win = int(self.pin[0])
            if self.device.type == "cuda":
                # the region goes up once (300 MB for 10000^2); the overlapping 270^2 crops (11x the bytes) are gathered on
                # the GPU (hvn_extract_patches; every crop is in bounds, so its reflect rule never fires)
                region_dev = torch.from_numpy(np.ascontiguousarray(region[..., :3])).to(self.device)
                patches = infer_tile.extract_patches_device(region_dev, rel.astype(np.int32), win, 0)
                del region_dev
            else:
                patches = torch.from_numpy(np.ascontiguousarray(np.stack([region[y:y + win, x:x + win] for y, x in rel])))
            outs = [self._step(patches[b0:b0 + self.batch_size]).clone() for b0 in range(0, patches.shape[0], self.batch_size)]
            out = torch.cat(outs, 0)
            # output top-left in the slide = input top-left + diff // 2 (the placement rule of _assemble_and_flush,
            # wsi.py:235-258; patch_info[:, 1] is offset by the FULL diff in the reference and only feeds the mask test)
            otl = torch.from_numpy((plist[:, 0, 0] + off).astype(np.int64)).to(self.device)
            rows = (otl[:, 0, None] + ar - lo)[:, :, None].expand(-1, h, h)
            cols = (otl[:, 1, None] + ar)[:, None, :].expand(-1, h, h)
            local.t[rows, cols] = out.to(local.t.device)          # one scatter per chunk
        tm = getattr(self, "timing", None)
        if tm is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
            tm["stage1_own_rows_s"] = time.perf_counter() - t_start
        if world > 1:
            t_h = time.perf_counter()
            exchange_halo(local, need, bounds)
            if tm is not None:
                if self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)
                tm["halo_exchange_s"] = time.perf_counter() - t_h
                tm["halo_rows"] = int((hi - lo) - (bounds[rank + 1] - bounds[rank]))
        return local if (as_slab or world > 1) else local.t

    def _step(self, batch):
        return run_desc.infer_step_device(batch.to(self.device), self.model)

    # -- stage 2: three-phase post-processing ------------------------------------------------------
    def _results_in_order(self, pred_map, tiles):
        """Yields (tile index, pred_inst numpy, inst_info dict with the tile origin already added) in tile order on rank 0
        (nothing on the other ranks).

        `pred_map`: the whole map (tensor) or this rank's `SlabMap`; a tile is worked by the rank owning its top row.
        Three overlapped stages per rank: (a) the GPU instance separation + instance table of tile i+1 and its D2H into pinned
        memory are in flight (`_launch_tile`) while (b) a worker thread traces the contours of tile i on the host cores
        (C++, multi-threaded, no GIL) and (c) the caller merges tile i-1.  One rank: results are yielded as they complete, so
        the sequential merge runs under the GPU work of later tiles.  Several ranks: every rank
        runs the same pipeline over the tiles whose rows it holds, then instance maps / record tables / contour arrays travel to rank 0 as
        tensors (`infer_tile.gather_items_to_rank0`) and rank 0 assembles the dicts."""
        import collections
        from concurrent.futures import ThreadPoolExecutor

        _, rank, world = infer_tile._dist()
        if not isinstance(pred_map, SlabMap):
            pred_map = SlabMap(pred_map, 0)
        bounds = row_slabs(int(self._shape[0]), world, self.pin[0], self.pout[0])
        own = tile_owner(tiles, bounds)
        idxs = [i for i in range(tiles.shape[0]) if own[i] == rank]    # the rank that holds the tile's rows (slab + halo)
        tm = getattr(self, "timing", None)

        def shift_of(i):
            return (int(tiles[i][0][1]), int(tiles[i][0][0]))      # tile origin as (x, y)

        def host_half(i, inst_h, rec_h, release):
            t0 = time.perf_counter()
            arrs = infer_tile.result_to_arrays(inst_h, rec_h, self.nr_types)
            t1 = time.perf_counter()
            out = infer_tile.arrays_to_result(arrs, self.nr_types, shift_xy=shift_of(i)) if world == 1 else [np.array(a) for a in arrs]
            release()
            if tm is not None:
                tm["contours_s"] = tm.get("contours_s", 0.0) + (t1 - t0)
                tm["dict_s"] = tm.get("dict_s", 0.0) + (time.perf_counter() - t1)
            return out

        inflight, futs, mine = collections.deque(), collections.deque(), {}
        depth = 4 if self.device.type == "cuda" else 2          # tiles in flight: one per post-processing lane + one being handed over
        self._dev_results = {}                                     # tile index -> (device local-id map, ready event), one rank only
        with ThreadPoolExecutor(1) as pool:
            def drain(block):
                while futs and (block or futs[0][1].done()):
                    i, f = futs.popleft()
                    if world == 1:
                        yield (i,) + f.result()
                    else:
                        mine[i] = f.result()

            def finish():
                i, wait = inflight.popleft()
                inst_h, rec_h, release = wait()
                if world == 1 and hasattr(wait, "device_result"):
                    nz = np.flatnonzero(rec_h["area"]) if rec_h.size else np.zeros(0, np.int64)
                    self._dev_results[i] = wait.device_result + (int(nz[-1]) + 1 if nz.size else 0,)      # + the tile's largest label
                futs.append((i, pool.submit(host_half, i, inst_h, rec_h, release)))

            for i in idxs:
                tl, br = tiles[i][0], tiles[i][1]
                inflight.append((i, self._launch_tile(pred_map.window(tl, br))))
                if len(inflight) >= depth:
                    finish()
                yield from drain(False)
            while inflight:
                finish()
            yield from drain(True)
        if world > 1:
            every = infer_tile.gather_items_to_rank0(mine, device=self.device)
            if every is not None:
                for i in range(tiles.shape[0]):
                    yield (i,) + infer_tile.arrays_to_result(every[i], self.nr_types, shift_xy=shift_of(i))

    def _launch_tile(self, tile_map):
        """Start the GPU half of one tile; returns wait() -> (int32 instance map, record table, release) on the host.  CUDA:
        kernels + D2H into a pinned slot are enqueued and wait() blocks on the slot's event; `release()` frees the slot once
        the host half is done with the arrays (they alias pinned memory)."""
        if self.device.type != "cuda":
            inst_h, rec_h = self._postproc_tile(tile_map)
            return lambda: (inst_h, rec_h, lambda: None)
        t0 = time.perf_counter()
        # tiles alternate between a few post-processing lanes (own stream + own workspace): one tile's ~30 launches over 4-5
        # Mpixel leave most of the chip idle, two or three tiles in flight fill it
        lanes = self.__dict__.setdefault("_pp_lanes", [])
        if not lanes:
            import os
            for _ in range(max(1, int(os.environ.get("HVN_WSI_LANES", "3")))):
                lanes.append((torch.cuda.Stream(self.device), post_proc.PostProc(self.device)))
            self._pp_next = 0
        stream, pp = lanes[self._pp_next % len(lanes)]
        self._pp_next += 1
        stream.wait_stream(torch.cuda.current_stream(self.device))         # the prediction map is complete on the caller's stream
        with torch.cuda.stream(stream):
            tile = tile_map.contiguous().unsqueeze(0)
            inst = pp.separate(tile)
            rec, _ = pp.table(inst, tile, self.nr_types)
            slot = self._pinned_slot(inst[0].shape, rec[0].shape)
            slot["inst"].copy_(inst[0], non_blocking=True)
            slot["rec"].copy_(rec[0], non_blocking=True)
            slot["event"].record(stream)
        tm = getattr(self, "timing", None)

        def wait():
            slot["event"].synchronize()
            if tm is not None:
                tm["gpu_launch_to_ready_s"] = tm.get("gpu_launch_to_ready_s", 0.0) + (time.perf_counter() - t0)
            return slot["inst"].numpy(), slot["rec"].numpy().view(post_proc._REC_DTYPE).reshape(-1), slot["free"].set

        wait.device_result = (inst[0], slot["event"])              # for the on-device merge: the local-id map stays in HBM
        return wait

    def _pinned_slot(self, inst_shape, rec_shape):
        """A free pinned (instance map, record table) pair for this tile shape; six per shape, waiting for the oldest when
        all are in use (hipHostMalloc is slow, so the slots are kept)."""
        import threading

        pools = self.__dict__.setdefault("_slots", {})
        key = (tuple(inst_shape), tuple(rec_shape))
        ring = pools.setdefault(key, {"slots": [], "next": 0})
        if len(ring["slots"]) < 6:
            slot = {"inst": torch.empty(inst_shape, dtype=torch.int32, pin_memory=True),
                    "rec": torch.empty(rec_shape, dtype=torch.uint8, pin_memory=True),
                    "event": torch.cuda.Event(), "free": threading.Event()}
            ring["slots"].append(slot)
        else:
            slot = ring["slots"][ring["next"] % 6]
            ring["next"] += 1
            slot["free"].wait()
        slot["free"].clear()
        return slot

    def _postproc_tile(self, tile_map):
        """One tile of the prediction map -> (int32 instance map, record table) on the host, synchronously."""
        inst, rec, _ = post_proc.process_batch_device(tile_map.contiguous().unsqueeze(0), self.nr_types, True)
        return inst[0].cpu().numpy(), rec[0].cpu().numpy().view(post_proc._REC_DTYPE).reshape(-1)

    def run(self, slide, mask=None):
        """slide: object with .shape / .read_region; mask: uint8 tissue mask at any scale, None = all tissue, "auto" = the
        reference's 1.25x thresholding heuristic (wsi.py:486-500) on `slide.thumbnail(32)`.
        Returns (inst_map int32 [H,W] numpy, inst_info dict) like `wsi_inst_map` / `wsi_inst_info` on rank 0, (None, None) on
        the other ranks of a multi-GPU run."""
        shape = np.array(slide.shape[:2])
        if isinstance(mask, str) and mask == "auto":
            from . import tissue_mask

            mask = tissue_mask.simple_get_mask(slide.thumbnail(32))
        if mask is None:
            mask = np.ones((max(1, int(shape[0]) // 32), max(1, int(shape[1]) // 32)), np.uint8)
        pred_map = self.raw_prediction(slide, mask, as_slab=True)
        return self.stitch_instances(pred_map, mask, shape=shape)

    def stitch_instances(self, pred_map, mask=None, shape=None):
        """Stage 2 alone: three-phase tile post-processing + merge of an HBM-resident prediction map (a tensor holding the whole
        map, or this rank's `SlabMap` + the slide `shape`)."""
        if shape is None:
            assert not isinstance(pred_map, SlabMap), "a SlabMap does not know the slide's height: pass shape"
            shape = pred_map.shape[:2]
        shape = np.array([int(shape[0]), int(shape[1])])
        self._shape = shape
        if mask is None:
            mask = np.ones((max(1, int(shape[0]) // 32), max(1, int(shape[1]) // 32)), np.uint8)
        import os

        rank = infer_tile._dist()[1]
        on_device = self.device.type == "cuda" and os.environ.get("HVN_WSI_HOST_MERGE", "0") == "0"
        merger = None if rank != 0 else (DeviceMerger(shape, self.device) if on_device else WsiMerger(shape))
        for phase, tiles in enumerate(self.tile_lists(shape, mask)):
            # the merge is sequential by definition (wsi.py:569-677) and runs on rank 0, under the GPU work of later tiles
            for i, inst_h, info in self._results_in_order(pred_map, tiles):
                t0 = time.perf_counter()
                if on_device:
                    dev, ready, n_local = self._dev_results.pop(i, (None, None, None))
                    src = inst_h if dev is None else dev           # remote ranks' tiles arrive as host arrays and are uploaded
                    if phase == 0:
                        merger.normal(src, info, tiles[i][0], tiles[i][1], ready=ready, n_local=int(inst_h.max()) if n_local is None else n_local)
                    else:
                        merger.fixing(src, info, tiles[i][0], tiles[i][1], ready=ready, n_local=int(inst_h.max()) if n_local is None else n_local)
                else:
                    (merger.normal if phase == 0 else merger.fixing)(inst_h, info, tiles[i][0], tiles[i][1], shifted=True)
                if getattr(self, "timing", None) is not None:
                    self.timing["merge_s"] = self.timing.get("merge_s", 0.0) + (time.perf_counter() - t0)
        if rank != 0:
            return None, None
        if on_device:
            t0 = time.perf_counter()
            out = merger.result()
            if getattr(self, "timing", None) is not None:
                self.timing["map_d2h_s"] = time.perf_counter() - t0
            return out
        if rank != 0:
            return None, None
        return merger.inst_map, merger.inst_info
