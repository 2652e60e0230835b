"""Tile inference over one node's GPUs -- the data-parallel core of
`infer/tile.py:InferManager.process_file_list` (/root/reference/infer/tile.py:150-387).

What is kept from the reference: the patch geometry of `_prepare_patching`
(infer/tile.py:46-94: reflect padding, `step = mask_size`, patch order), the stitching of
`_post_process_patches` (infer/tile.py:98-131: sort by (y, x), reshape/transpose, crop to the
source shape) and `post_proc.process` on the stitched map.

What is different by design: the reference wraps the model in `nn.DataParallel`
(infer/base.py:69: one process, weights re-broadcast every forward, outputs gathered on GPU 0)
and post-processes in a CPU process pool.  Here there is ONE PROCESS PER GPU
(torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests):
patches are sharded by contiguous ranges over the ranks, weights stay resident per rank, the
per-patch prediction maps (102 KB each) are exchanged with a single all_gather per call, images
are then dealt round-robin to the ranks for stitching + on-GPU instance separation, and only
int32 instance maps travel back.  Tiles are independent, so this is the only collective.
"""
import math

import numpy as np
import torch


# --------------------------------------------------------------------------------------------
# geometry (host logic, numpy)
def prepare_patching(img, window_size, mask_size):
    """-> (reflect-padded image, patch_info int32 [P,4] = [y, x, row, col]) with the reference's
    padding rule and patch order (infer/tile.py:60-90: x outer, y inner)."""
    step = mask_size
    im_h, im_w = img.shape[0], img.shape[1]

    def last_step(length):
        nr = math.ceil((length - mask_size) / step)
        return int((nr + 1) * step), int(nr + 1)

    last_h, nr_h = last_step(im_h)
    last_w, nr_w = last_step(im_w)
    pad_tl = (window_size - step) // 2
    padded = np.pad(img, ((pad_tl, last_h + window_size - im_h), (pad_tl, last_w + window_size - im_w), (0, 0)), "reflect")
    ys = np.arange(0, last_h, step, dtype=np.int32)
    xs = np.arange(0, last_w, step, dtype=np.int32)
    info = np.empty((nr_w * nr_h, 4), np.int32)
    k = 0
    for ci, x in enumerate(xs):
        for ri, y in enumerate(ys):
            info[k] = (y, x, ri, ci)
            k += 1
    return padded, info


def patch_grid(shape, window_size, mask_size):
    """The patch grid of `prepare_patching` without materialising the padded image: -> (patch_info int32 [P,4],
    pad_tl).  infer/tile.py:60-90."""
    step = mask_size
    im_h, im_w = shape[0], shape[1]
    last_h = int((math.ceil((im_h - mask_size) / step) + 1) * step)
    last_w = int((math.ceil((im_w - mask_size) / step) + 1) * step)
    ys = np.arange(0, last_h, step, dtype=np.int32)
    xs = np.arange(0, last_w, step, dtype=np.int32)
    info = np.empty((len(xs) * len(ys), 4), np.int32)
    k = 0
    for ci, x in enumerate(xs):
        for ri, y in enumerate(ys):
            info[k] = (y, x, ri, ci)
            k += 1
    return info, (window_size - step) // 2


def extract_patches_device(img_dev, info, window_size, pad_tl):
    """img_dev: uint8 device tensor [H,W,3] (unpadded) -> uint8 device tensor [P,win,win,3]: reflect padding and the
    overlapping crops are produced on the GPU (hvn_extract_patches), so each source pixel crosses PCIe once."""
    import ctypes

    from . import lib as L

    L.require_gpu()
    assert img_dev.dtype == torch.uint8 and img_dev.is_cuda and img_dev.dim() == 3 and img_dev.shape[2] == 3
    img_dev = img_dev.contiguous()
    coords = torch.from_numpy(np.ascontiguousarray(info[:, :2], np.int32)).to(img_dev.device)
    out = torch.empty((info.shape[0], window_size, window_size, 3), dtype=torch.uint8, device=img_dev.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(img_dev.device).cuda_stream)
    L.check(L.lib().hvn_extract_patches(img_dev.data_ptr(), img_dev.shape[0], img_dev.shape[1], coords.data_ptr(), info.shape[0],
                                        window_size, pad_tl, pad_tl, out.data_ptr(), stream), "hvn_extract_patches")
    return out


def extract_patches(padded, info, window_size):
    """uint8 [P, win, win, 3] (infer_loader.py:59-72 without the worker processes)."""
    out = np.empty((info.shape[0], window_size, window_size, padded.shape[2]), padded.dtype)
    for k, (y, x, _r, _c) in enumerate(info):
        out[k] = padded[y:y + window_size, x:x + window_size]
    return out


def stitch(patch_maps, info, src_shape):
    """patch_maps: tensor/array [P, h, w, C] in `info` order -> [src_h, src_w, C]
    (infer/tile.py:111-131).  Works on torch tensors (device) and numpy alike."""
    order = sorted(range(info.shape[0]), key=lambda k: (int(info[k][0]), int(info[k][1])))
    nr_row = int(info[:, 2].max()) + 1
    nr_col = int(info[:, 3].max()) + 1
    h, w, c = patch_maps.shape[1:]
    pm = patch_maps[torch.as_tensor(order, device=patch_maps.device)] if torch.is_tensor(patch_maps) else patch_maps[order]
    pm = pm.reshape(nr_row, nr_col, h, w, c)
    pm = pm.permute(0, 2, 1, 3, 4) if torch.is_tensor(pm) else pm.transpose(0, 2, 1, 3, 4)
    pm = pm.reshape(nr_row * h, nr_col * w, c)
    return pm[:src_shape[0], :src_shape[1]]


# --------------------------------------------------------------------------------------------
# rank sharding (torch.distributed; no collective when world == 1)
def shard_range(n_items, rank, world):
    """Contiguous, balanced partition: the first n % world ranks take one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _dist():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def a2a_single(recv, send, out_splits, in_splits):
    """`dist.all_to_all_single` with uneven splits.  RCCL takes the device tensors as they are; under gloo (the CPU tests, and the
    2-ranks-on-one-GPU test that runs the REAL kernels with only the transport substituted) device tensors are staged through the
    host."""
    dist, _rank, _world = _dist()
    if send.is_cuda and dist.get_backend() != "nccl":
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu(), out_splits, in_splits)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, out_splits, in_splits)


def agree_min(value, device=None):
    """The minimum of an integer over the ranks (all_reduce MIN): every rank then takes the same decision from it, e.g. the RAM
    budget that cuts a file list into caching rounds -- ranks that cut differently would run mismatched collectives."""
    dist, _rank, world = _dist()
    if world == 1:
        return int(value)
    dev = torch.device(device) if (device is not None and dist.get_backend() == "nccl") else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def gather_shards(local, n_items):
    """local: this rank's [n_local, ...] slice (shard_range order) -> the full [n_items, ...] on
    EVERY rank with one all_gather (shards padded to equal length)."""
    dist, rank, world = _dist()
    if world == 1:
        return local
    per = -(-n_items // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        parts.append(out[r * per:r * per + (hi - lo)])
    return torch.cat(parts, 0)


def route_to_owners(local, n_items, owner):
    """local: this rank's [n_local, ...] rows for the items of `shard_range(n_items, rank, world)`; owner: int array [n_items],
    the rank that consumes each item.  Returns (idx, rows): the global indices this rank owns, ascending, and their rows --
    ONE `all_to_all_single` with uneven splits (every rank derives all split sizes from `owner`, nothing is negotiated).
    Each row crosses the fabric at most once and lands only where it is used: 1 / world of what the all_gather of
    `gather_shards` delivers to every rank."""
    dist, rank, world = _dist()
    owner = np.asarray(owner, np.int64)
    assert owner.shape == (n_items,) and (n_items == 0 or (0 <= owner.min() and owner.max() < world))
    mine_idx = np.flatnonzero(owner == rank)
    if world == 1:
        return mine_idx, local
    lo, hi = shard_range(n_items, rank, world)
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    dest = owner[lo:hi]
    order = np.argsort(dest, kind="stable")                     # grouped by destination, index order kept inside a group
    in_splits = np.bincount(dest, minlength=world).tolist()
    out_splits = [int(np.count_nonzero(owner[slice(*shard_range(n_items, s, world))] == rank)) for s in range(world)]
    send = local[torch.as_tensor(order, device=local.device)].contiguous() if len(order) else local[:0].contiguous()
    recv = torch.empty((sum(out_splits),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    row = int(np.prod(local.shape[1:], dtype=np.int64))
    a2a_single(recv.view(-1), send.view(-1), [c * row for c in out_splits], [c * row for c in in_splits])
    return mine_idx, recv


def gather_to_rank0(tensors, dst=0):
    """Fan-in of per-rank results to rank `dst`: `tensors` is a tuple of tensors (None entries pass through) whose
    shapes are the same on every rank; returns, on rank `dst`, the tuple of [world * n, ...] concatenations in rank
    order and None elsewhere.  One `dist.gather` per tensor (RCCL: point-to-point sends over xGMI into rank `dst`'s
    buffer; gloo in the CPU tests) -- the instance maps and the fixed-size record tables travel as tensors, nothing
    is pickled.  world == 1: returns `tensors` unchanged."""
    dist, rank, world = _dist()
    if world == 1:
        return tensors
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        t = t.contiguous()
        dev = t.device
        if t.is_cuda and dist.get_backend() != "nccl":      # gloo: staged through the host (see a2a_single)
            t = t.cpu()
        if rank == dst:
            full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.gather(t, list(full.chunk(world, 0)), dst=dst)
            out.append(full.to(dev))
        else:
            dist.gather(t, None, dst=dst)
    return tuple(out) if rank == dst else None


_DT = ("uint8", "int32", "int64", "float32", "float64")


def pack_arrays(arrays):
    """list of numpy arrays -> one uint8 buffer: [n][per array: dtype code, ndim, shape...] as int64, then the raw bytes
    (each 8-byte aligned).  The wire format of `gather_items_to_rank0`: plain bytes in a tensor, nothing is pickled."""
    head = [len(arrays)]
    for a in arrays:
        head += [_DT.index(str(a.dtype)), a.ndim] + list(a.shape)
    parts = [np.asarray(head, np.int64).view(np.uint8)]
    for a in arrays:
        b = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        parts.append(b)
        if b.size % 8:
            parts.append(np.zeros(8 - b.size % 8, np.uint8))
    return np.concatenate(parts)


def unpack_arrays(buf):
    """Inverse of pack_arrays on a uint8 array (views into `buf`, no copies)."""
    n = int(buf[:8].view(np.int64)[0])
    pos, metas = 8, []
    for _ in range(n):
        code, ndim = (int(v) for v in buf[pos:pos + 16].view(np.int64))
        shape = tuple(int(v) for v in buf[pos + 16:pos + 16 + 8 * ndim].view(np.int64))
        metas.append((np.dtype(_DT[code]), shape))
        pos += 16 + 8 * ndim
    out = []
    for dt, shape in metas:
        nb = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        out.append(buf[pos:pos + nb].view(dt).reshape(shape))
        pos += (nb + 7) // 8 * 8
    return out


def gather_items_to_rank0(mine, dst=0, device=None):
    """mine: {item index: [numpy arrays]} held by this rank (every index owned by exactly one rank).  Returns the union of all
    ranks' items on rank `dst`, None elsewhere.  Two collectives: an all_gather of the byte counts, then one `dist.gather` of
    the packed uint8 buffers padded to the longest (RCCL over xGMI on the GPU box -- the buffers are staged through HBM on
    `device`, the caller's model device --, gloo in the CPU tests).  Nothing is pickled."""
    dist, rank, world = _dist()
    if world == 1:
        return mine
    keys = sorted(mine)
    payload = pack_arrays([np.asarray(keys, np.int64)] + [np.asarray([len(mine[k]) for k in keys], np.int64)] + [a for k in keys for a in mine[k]])
    # the staging device of the collective: the caller's model device (one process per GPU: never "whatever device is current")
    if dist.get_backend() == "nccl":
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([payload.size], dtype=torch.int64, device=dev))
    longest = int(sizes.max().item())
    send = torch.zeros(longest, dtype=torch.uint8, device=dev)
    send[:payload.size] = torch.from_numpy(payload).to(dev)
    if rank != dst:
        dist.gather(send, None, dst=dst)
        return None
    recv = torch.empty((world, longest), dtype=torch.uint8, device=dev)
    dist.gather(send, list(recv.unbind(0)), dst=dst)
    recv_h = recv.cpu().numpy()
    out = {}
    for r in range(world):
        arrs = unpack_arrays(recv_h[r, :int(sizes[r].item())])
        ks, counts = arrs[0].tolist(), arrs[1].tolist()
        pos = 2
        for k, c in zip(ks, counts):
            out[int(k)] = arrs[pos:pos + c]
            pos += c
    return out


def result_to_arrays(inst_h, rec_h, nr_types):
    """One image's / tile's finished result as arrays: instance map, record table, contour points + offsets (traced HERE, on
    the rank that owns the item, so the host work is spread over the ranks)."""
    from . import post_proc

    rec_h = np.ascontiguousarray(rec_h)
    pts, offs = post_proc.trace_contours_flat(inst_h, rec_h) if rec_h.size else (np.zeros((0, 2), np.int32), np.zeros(1, np.int64))
    return [np.ascontiguousarray(inst_h, np.int32), rec_h.view(np.uint8).reshape(rec_h.shape[0], rec_h.dtype.itemsize), pts, offs]


def arrays_to_result(arrs, nr_types, with_info=True, shift_xy=None):
    from . import post_proc

    inst_h, rec_b, pts, offs = arrs
    if not with_info:
        return np.array(inst_h), None
    rec_h = np.ascontiguousarray(rec_b).view(post_proc._REC_DTYPE).reshape(-1)
    return np.array(inst_h), post_proc.records_to_dict(rec_h, nr_types, contours_flat=(np.array(pts), np.array(offs)), shift_xy=shift_xy)


def run_sharded(items, step_fn, batch_size, gather=True):
    """Apply `step_fn(batch) -> tensor [b, ...]` to this rank's contiguous share of `items`
    ([P, ...] tensor) in batches.  gather=True: the result for ALL items on every rank (one all_gather);
    gather=False: this rank's rows only (`route_to_owners` then sends each row to the one rank that needs it)."""
    _, rank, world = _dist()
    lo, hi = shard_range(items.shape[0], rank, world)
    outs = []
    for b0 in range(lo, hi, batch_size):
        outs.append(step_fn(items[b0:min(b0 + batch_size, hi)]).clone())
    if outs:
        local = torch.cat(outs, 0)
    else:  # more ranks than items: contribute an empty slice of the right trailing shape
        probe = step_fn(items[:1])
        local = probe[:0].clone()
    return gather_shards(local, items.shape[0]) if gather else local


# --------------------------------------------------------------------------------------------
def process_images(images, model, nr_types=None, batch_size=32, return_centroids=True, return_raw=False, max_patches=16384):
    """images: list of uint8 [H,W,3] arrays (RGB).  Returns a list of (pred_inst int32 [H,W] numpy, inst_info_dict | None)
    in input order: complete on rank 0 (the writer); the other ranks hold their own images' results and None elsewhere.
    `return_raw=True` appends the stitched float32 prediction map [H,W,3|4] to each tuple (`--save_raw_map`,
    infer/tile.py:193-194); it travels to rank 0 with the other arrays.
    The images are worked off in groups of at most `max_patches` network patches (3.6 GB of uint8 patches + 1.7 GB of maps at the
    default), so that a large cache round of `process_file_list` never has all its overlapping patches in HBM at once; every rank
    forms the same groups (the collectives inside stay matched)."""
    win = 270 if (model.module if hasattr(model, "module") and not hasattr(model, "engine") else model).mode == "original" else 256
    msk = 80 if win == 270 else 164
    groups, cur, cur_n = [], [], 0
    for i, img in enumerate(images):
        n = patch_grid(img.shape, win, msk)[0].shape[0]
        if cur and cur_n + n > max_patches:
            groups.append(cur)
            cur, cur_n = [], 0
        cur.append(i)
        cur_n += n
    if cur:
        groups.append(cur)
    out = [None] * len(images)
    for grp in groups:
        for i, res in zip(grp, _process_image_group([images[i] for i in grp], model, nr_types, batch_size, return_centroids, return_raw)):
            out[i] = res
    return out


def _process_image_group(images, model, nr_types, batch_size, return_centroids, return_raw):
    """One group of `process_images`.

    Pipeline per call: patch extraction -> sharded HIP network (`run_desc.infer_step_device`)
    -> one all_to_all of the per-patch maps to the images' owners (`route_to_owners`) -> per-image stitch on the GPU -> on-GPU instance
    separation + instance table (`post_proc.process_batch_device`) for the images this rank owns
    -> tensor gather of instance maps / record tables / contours to rank 0 (`gather_items_to_rank0`)."""
    from . import post_proc, run_desc

    net = model.module if hasattr(model, "module") and not hasattr(model, "engine") else model
    win = 270 if net.mode == "original" else 256
    msk = 80 if net.mode == "original" else 164   # run_infer.py:145-150
    dev = next(net.parameters()).device
    infos, patches = [], []
    for i, img in enumerate(images):
        if dev.type == "cuda":      # source image up once, reflect padding + overlapping crops on the GPU
            info, pad_tl = patch_grid(img.shape, win, msk)
            patches.append(extract_patches_device(torch.from_numpy(np.ascontiguousarray(img)).to(dev), info, win, pad_tl))
        else:                       # host geometry (CPU tests of the sharding logic)
            padded, info = prepare_patching(img, win, msk)
            patches.append(torch.from_numpy(extract_patches(padded, info, win)))
        infos.append(info)
    all_patches = torch.cat(patches, 0)
    _, rank, world = _dist()
    # network: contiguous patch shards; then every per-patch map goes to the ONE rank that stitches its image (image i -> rank
    # i % world) with a single all_to_all -- not to every rank
    local = run_sharded(all_patches, lambda b: run_desc.infer_step_device(b.to(dev), model), batch_size, gather=False)
    counts = [info.shape[0] for info in infos]
    owner = np.repeat(np.arange(len(images)) % world, counts)
    _idx, pred = route_to_owners(local, all_patches.shape[0], owner)
    mine = {}
    k = 0
    for i, img in enumerate(images):
        n = counts[i]
        if i % world == rank:
            full = stitch(pred[k:k + n], infos[i], img.shape).contiguous()
            inst, rec, _ = post_proc.process_batch_device(full.unsqueeze(0), nr_types, return_centroids)
            inst_h = inst[0].cpu().numpy()
            rec_h = rec[0].cpu().numpy().view(post_proc._REC_DTYPE).reshape(-1) if rec is not None else np.zeros(0, post_proc._REC_DTYPE)
            mine[i] = result_to_arrays(inst_h, rec_h, nr_types)
            if return_raw:
                mine[i].append(full.cpu().numpy())
            k += n
    every = gather_items_to_rank0(mine, device=dev)          # the instance maps + record tables + contours travel as tensors to rank 0
    if every is None:                            # not rank 0: keeps only what it computed itself
        every = mine
    out = []
    for i in range(len(images)):
        if i not in every:
            out.append(None)
            continue
        res = arrays_to_result(every[i][:4], nr_types, with_info=(return_centroids or nr_types is not None))
        out.append(res + (np.array(every[i][4]),) if return_raw else res)
    return out
