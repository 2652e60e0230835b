"""Tile-mode inference manager: the host loop around the hot path that `infer/base.py:21-94` (`InferManager`) and
`infer/tile.py:150-387` (`process_file_list`) implement in the reference -- file list, RAM-bounded caching, the network +
post-processing of every cached image through `infer_tile.process_images` (HIP network, on-GPU instance separation and
instance table, tensor gather to rank 0), and the four writers (`mat/`, `json/`, `overlay/`, optional `qupath/`).

What is different from the reference, on purpose:
  * no DataLoader / ProcessPoolExecutor: patches are cut on the GPU and post-processing runs there, so
    `nr_inference_workers` / `nr_post_proc_workers` are accepted and ignored;
  * the caching loop never loses a file: the reference pops the file that overflows the RAM budget and drops it
    (`infer/tile.py:255-264` -- it is neither cached nor pushed back); here it starts the next round, and a single file
    larger than the budget is still processed on its own;
  * images are read with PIL (`cv2.imread` + BGR2RGB in the reference, `infer/tile.py:249-250`); `.npy` uint8 arrays are accepted too;
  * the overlay is drawn by `viz.visualize_instances_dict` (numpy), not `cv2.drawContours`: same colours, contours and dots,
    line rasterisation not bit-pinned (cv2 is absent from this image).
Only rank 0 writes; with `torch.distributed` initialised every rank must call `process_file_list` (it contains collectives)."""
import glob
import json
import math
import os
import pathlib
import re
import shutil

import numpy as np

from . import io_utils, viz


def rm_n_mkdir(dir_path):
    """misc/utils.py:99-104."""
    if os.path.isdir(dir_path):
        shutil.rmtree(dir_path)
    os.makedirs(dir_path)


def hot_colours(n):
    """`(plt.get_cmap("hot")(np.arange(n, dtype=np.int32))[..., :3] * 255).astype(np.uint8)` (infer/base.py:45-48): INTEGER
    arguments index the 256-entry lookup table, so the colours are entries 0..n-1 of 'hot' (nearly black reds).  'hot' is
    piecewise linear; red runs 0.0416 -> 1 over [0, 0.365079], green and blue are 0 below 0.365079 / 0.746032."""
    x = np.arange(n, dtype=np.float64) / 255.0
    r = np.interp(x, [0.0, 0.365079, 1.0], [0.0416, 1.0, 1.0])
    g = np.interp(x, [0.0, 0.365079, 0.746032, 1.0], [0.0, 0.0, 1.0, 1.0])
    b = np.interp(x, [0.0, 0.746032, 1.0], [0.0, 0.0, 1.0])
    return (np.stack([r, g, b], -1) * 255).astype(np.uint8)


def load_type_info(nr_types, type_info_path):
    """infer/base.py:29-53: `{type_id: (name, (r, g, b))}`."""
    if nr_types is None:
        return {None: ["no label", [0, 0, 0]]}
    if type_info_path is not None:
        with open(type_info_path, "r") as handle:
            table = {int(k): (v[0], tuple(v[1])) for k, v in json.load(handle).items()}
        for k in range(nr_types):
            assert k in table, "Not detect type_id=%d defined in json." % k
        return table
    return {k: (str(k), tuple(int(c) for c in v)) for k, v in enumerate(hot_colours(nr_types))}


def list_files(input_dir):
    """infer/tile.py:159-162: every entry of the directory, `[`/`]` in the path escaped for glob, sorted."""
    pattern = re.sub(r"([\[\]])", "[\\1]", "%s/*" % input_dir)
    files = sorted(glob.glob(pattern))
    assert len(files) > 0, "Not Detected Any Files From Path"
    return files


def read_image(path):
    """uint8 RGB [H, W, 3]."""
    if str(path).endswith(".npy"):
        img = np.load(path)
    else:
        from PIL import Image

        with Image.open(path) as im:
            img = np.asarray(im.convert("RGB"))
    img = np.ascontiguousarray(img[..., :3])
    assert img.dtype == np.uint8 and img.ndim == 3, (path, img.dtype, img.shape)
    return img


def padded_nbytes(shape, win, msk):
    """Bytes of the reflect-padded image `_prepare_patching` builds (infer/tile.py:60-78: `pad_tl` before, `last + win - size`
    after, per axis) -- what the reference's cache budget is charged with, times 5 (infer/tile.py:256-258)."""
    pad_tl = (win - msk) // 2
    last_h = (math.ceil((int(shape[0]) - msk) / msk) + 1) * msk
    last_w = (math.ceil((int(shape[1]) - msk) / msk) + 1) * msk
    return (pad_tl + last_h + win) * (pad_tl + last_w + win) * 3


def to_qupath(path, centroids, types, type_info):
    """convert_format.py:17-49 (QuPath v0.2.3 TSV)."""
    centroids = np.asarray(centroids)
    types = np.asarray(types)
    assert centroids.shape[0] == types.shape[0]
    with open(path, "w") as f:
        f.write("x\ty\tclass\tname\tcolor\n")
        for pos, t in zip(centroids, types):
            name, (r, g, b) = type_info[t if t is None else int(t)]
            f.write("{x}\t{y}\t{c}\t{n}\t{col}\n".format(x=pos[0], y=pos[1], c="", n=name, col=(int(r) << 16) + (int(g) << 8) + int(b)))


def mat_dict(pred_inst, inst_info, nr_types, raw_map=None):
    """infer/tile.py:178-196: singleton trailing axes "to make matlab happy"; no `inst_type` without a type branch."""
    vals = list(inst_info.values())
    out = {
        "inst_map": pred_inst,
        "inst_uid": np.array(list(inst_info.keys()))[:, None] if vals else np.zeros((0, 1), np.int64),
        "inst_type": np.array([v["type"] for v in vals])[:, None] if vals else np.zeros((0, 1), np.int64),
        "inst_centroid": np.array([v["centroid"] for v in vals]) if vals else np.zeros((0, 2)),
    }
    if nr_types is None:
        out.pop("inst_type")
    if raw_map is not None:
        out["raw_map"] = raw_map
    return out


class InferManager:
    """`InferManager(method={"model_args": {"nr_types": .., "mode": ..}, "model_path": ..}, type_info_path=..)`
    (infer/base.py:22-27 takes the same keywords).  `model=` hands over an already built network instead of a checkpoint
    (tests, in-process callers); `process_fn` replaces `infer_tile.process_images` (CPU tests of the host loop)."""

    def __init__(self, method, type_info_path=None, device="cuda", model=None, process_fn=None, load=True):
        self.method = method
        self.nr_types = method["model_args"]["nr_types"]
        self.mode = method["model_args"].get("mode", "original")
        self.type_info_path = type_info_path
        self.type_info_dict = load_type_info(self.nr_types, type_info_path)
        self.process_fn = process_fn
        self.model = model
        if model is None and process_fn is None and load:
            self.model = self._load_model(device)

    def _load_model(self, device):
        """infer/base.py:56-78: create, strict-load `["desc"]` (DataParallel `module.` prefix stripped), move to the GPU."""
        import torch

        from . import net_desc, train

        net = net_desc.create_model(**self.method["model_args"])
        saved = torch.load(self.method["model_path"], map_location="cpu")["desc"]
        net.load_state_dict(train.convert_checkpoint_keys(saved), strict=True)
        return net.to(device).eval()

    def process_file_list(self, run_args):
        """run_args (run_infer.py:134-172): input_dir, output_dir, batch_size, mem_usage, draw_dot, save_qupath, save_raw_map,
        patch_input_shape, patch_output_shape (+ nr_inference_workers, nr_post_proc_workers: ignored).
        Returns the list of image names written, in processing order (rank 0; [] on the other ranks)."""
        import psutil

        from . import infer_tile

        input_dir, output_dir = run_args["input_dir"], run_args["output_dir"]
        batch_size = int(run_args.get("batch_size", 32))
        mem_usage = float(run_args.get("mem_usage", 0.2))
        assert 0.0 < mem_usage < 1.0
        draw_dot = bool(run_args.get("draw_dot", False))
        save_qupath = bool(run_args.get("save_qupath", False))
        save_raw_map = bool(run_args.get("save_raw_map", False))
        win = int(run_args.get("patch_input_shape", 270 if self.mode == "original" else 256))
        msk = int(run_args.get("patch_output_shape", 80 if self.mode == "original" else 164))
        assert (win, msk) == ((270, 80) if self.mode == "original" else (256, 164)), "patch shapes are fixed by the model mode"
        _, rank, _world = infer_tile._dist()

        pending = list_files(input_dir)
        if rank == 0:
            for sub in ("json", "mat", "overlay") + (("qupath",) if save_qupath else ()):
                rm_n_mkdir("%s/%s/" % (output_dir, sub))
        process = self.process_fn or (lambda images: infer_tile.process_images(
            images, self.model, nr_types=self.nr_types, batch_size=batch_size, return_centroids=True, return_raw=save_raw_map))
        done = []
        self.rounds = []                                   # files per caching round (inspected by the tests)
        while pending:
            budget = int(psutil.virtual_memory().available * mem_usage) if "ram_budget_bytes" not in run_args else int(run_args["ram_budget_bytes"])
            # every rank must cut the list into the SAME rounds (the collectives inside `process` are matched by round): the
            # ranks sample their free RAM at different moments, so they agree on the smallest reading
            budget = infer_tile.agree_min(budget, self._collective_device())
            paths, images = [], []
            while pending:
                img = read_image(pending[0])
                budget -= 5 * padded_nbytes(img.shape, win, msk)
                if budget < 0 and images:
                    break                                  # opens the next round
                paths.append(pending.pop(0))
                images.append(img)
            self.rounds.append(len(paths))
            results = process(images)
            if rank != 0:
                continue
            for path, img, res in zip(paths, images, results):
                name = pathlib.Path(path).stem
                pred_inst, inst_info = res[0], res[1]
                raw_map = res[2] if save_raw_map else None
                self._write(output_dir, name, img, pred_inst, inst_info, raw_map, draw_dot, save_qupath)
                done.append(name)
        return done

    def _collective_device(self):
        try:
            net = self.model.module if hasattr(self.model, "module") and not hasattr(self.model, "engine") else self.model
            return next(net.parameters()).device
        except (AttributeError, StopIteration, TypeError):
            return None

    def _write(self, output_dir, name, img, pred_inst, inst_info, raw_map, draw_dot, save_qupath):
        """proc_callback, infer/tile.py:169-208."""
        import scipy.io as sio

        sio.savemat("%s/mat/%s.mat" % (output_dir, name), mat_dict(pred_inst, inst_info, self.nr_types, raw_map))
        overlay = viz.visualize_instances_dict(img, inst_info, draw_dot=draw_dot, type_colour=self.type_info_dict, line_thickness=2)
        viz.save_png("%s/overlay/%s.png" % (output_dir, name), overlay)
        if save_qupath:
            vals = list(inst_info.values())
            to_qupath("%s/qupath/%s.tsv" % (output_dir, name), np.array([v["centroid"] for v in vals]).reshape(-1, 2),
                      np.array([v["type"] for v in vals]), self.type_info_dict)
        io_utils.save_json("%s/json/%s.json" % (output_dir, name), inst_info, None)


# ----------------------------------------------------------------------------------------------
def open_slide(path):
    """Slide backend by extension.  `.npy` -> memory-mapped `ArraySlide`; plain images -> in-memory `ArraySlide`.
    OpenSlide formats (`.svs`, `.ndpi`, ... -- misc/wsi_handler.py:84-99) need the openslide binding, which this image does
    not have: they raise, and `process_wsi_list` logs the slide as crashed like the reference does (infer/wsi.py:744-749)."""
    from .infer_wsi import ArraySlide

    ext = pathlib.Path(path).suffix.lower()
    if ext == ".npy":
        return ArraySlide(np.load(path, mmap_mode="r"))
    if ext in (".png", ".jpg", ".jpeg", ".tif", ".tiff", ".bmp"):
        return ArraySlide(read_image(path))
    raise ValueError("no slide backend for %r in this build (OpenSlide is not available)" % ext)


def read_mask(path):
    """infer/wsi.py:477-480: grey image, > 0 -> 1."""
    from PIL import Image

    with Image.open(path) as im:
        return (np.asarray(im.convert("L")) > 0).astype(np.uint8)


class WsiManager(InferManager):
    """`infer/wsi.py:438-750` around `infer_wsi.WsiInference`: slide list, skip-if-done, mask file or the 1.25x heuristic,
    `json/<name>.json` with `mag`, optional `thumb/` and `mask/`.  No cache directory: the prediction map lives in HBM and
    the instance map in host RAM (`cache_path` is accepted and ignored).  `wsi_fn(slide, mask) -> (inst_map, inst_info)`
    replaces the GPU run in CPU tests."""

    def __init__(self, method, type_info_path=None, device="cuda", model=None, wsi_fn=None):
        super().__init__(method, type_info_path, device, model, load=wsi_fn is None)
        self.wsi_fn = wsi_fn

    def process_wsi_list(self, run_args):
        """run_args (run_infer.py:173-187): input_dir, output_dir, input_mask_dir, proc_mag, ambiguous_size, chunk_shape,
        tile_shape, save_thumb, save_mask, batch_size.  Returns {name: "done" | "skip" | "empty mask" | "crash"}."""
        import logging

        from . import infer_tile, infer_wsi, tissue_mask

        input_dir, output_dir = run_args["input_dir"], run_args["output_dir"]
        mask_dir = run_args.get("input_mask_dir") or ""
        save_thumb, save_mask = bool(run_args.get("save_thumb", False)), bool(run_args.get("save_mask", False))
        proc_mag = run_args.get("proc_mag", 40)
        _, rank, _world = infer_tile._dist()
        nested = save_thumb or save_mask                     # infer/wsi.py:700-703: json goes under json/ only then
        if rank == 0:
            os.makedirs(output_dir + "/json/", exist_ok=True)
            if save_thumb:
                os.makedirs(output_dir + "/thumb/", exist_ok=True)
            if save_mask:
                os.makedirs(output_dir + "/mask/", exist_ok=True)
        status = {}
        for wsi_path in sorted(glob.glob(input_dir + "/*")):
            if os.path.isdir(wsi_path):
                continue
            name = pathlib.Path(wsi_path).stem
            json_path = ("%s/json/%s.json" if nested else "%s/%s.json") % (output_dir, name)
            if os.path.exists(json_path):
                status[name] = "skip"
                continue
            try:
                slide = open_slide(wsi_path)
                msk_path = "%s/%s.png" % (mask_dir, name)
                mask = read_mask(msk_path) if os.path.isfile(msk_path) else tissue_mask.simple_get_mask(slide.thumbnail(32))
                if int(np.sum(mask)) == 0:
                    status[name] = "empty mask"
                    continue
                if rank == 0 and save_mask:
                    viz.save_png("%s/mask/%s.png" % (output_dir, name), np.repeat((mask * 255).astype(np.uint8)[..., None], 3, -1))
                if rank == 0 and save_thumb:
                    viz.save_png("%s/thumb/%s.png" % (output_dir, name), slide.thumbnail(32))
                if self.wsi_fn is not None:
                    _inst_map, inst_info = self.wsi_fn(slide, mask)
                else:
                    wsi = infer_wsi.WsiInference(self.model, nr_types=self.nr_types, batch_size=int(run_args.get("batch_size", 32)),
                                                 chunk_shape=int(run_args.get("chunk_shape", 10000)), tile_shape=int(run_args.get("tile_shape", 2048)),
                                                 ambiguous_size=int(run_args.get("ambiguous_size", 128)))
                    _inst_map, inst_info = wsi.run(slide, mask)
                if rank == 0:
                    io_utils.save_json(json_path, inst_info, mag=proc_mag)
                status[name] = "done"
            except Exception:                                 # noqa: BLE001  (the reference logs and moves on to the next slide)
                logging.exception("Crash")
                status[name] = "crash"
                if _world > 1:
                    # swallow-and-continue is only safe in one process: a rank that moves on to the next slide while the others
                    # wait in this slide's exchange / gather deadlocks the job -- fail the rank (and with it the launcher)
                    raise
        return status
