"""Drop-in for `models.hovernet.post_proc.process`
(/root/reference/models/hovernet/post_proc.py:94-186) on the GPU.

* `process(pred_map, nr_types=None, return_centroids=False)` keeps the reference's
  signature and return contract `(pred_inst int32 [H,W], inst_info_dict | None)` for one
  host map; it is a thin wrapper over the batched device path.
* `process_batch_device(pred_dev, nr_types)` is the north-star path: `[N,h,w,3|4]` float32
  maps already in HBM (straight from `run_desc.infer_step_device`) -> int32 instance maps
  and the per-instance table in HBM, no CPU round trip per tile.

Instance separation (`__proc_np_hv`, post_proc.py:26-90) and the array half of the
per-instance loop (bbox / centroid / type vote, post_proc.py:119-181) are HIP kernels in
libhvn_hip.so.  Contour tracing (`cv2.findContours`, post_proc.py:132-135) is an O(perimeter) host
routine of the same library over each instance's bbox crop (csrc/hvn_contour.cpp).  No CPU fallback
exists for the GPU stages.
"""
import ctypes

import numpy as np
import torch

from . import lib as L


class PostProc:
    """Owns the device workspace for `n` maps of `h x w` (grown on demand)."""

    def __init__(self, device="cuda"):
        L.require_gpu()
        self.device = torch.device(device)
        self._ws = None
        self._tws = None

    def _workspace(self, n, h, w):
        need = L.lib().hvn_postproc_workspace_bytes(n, h, w)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def separate(self, pred, taps=False):
        """pred: float32 device tensor [N,h,w,3|4] -> int32 device tensor [N,h,w]
        (+ (blb, dist, marker) stage taps when taps=True)."""
        assert pred.dtype == torch.float32 and pred.dim() == 4 and pred.is_cuda
        pred = pred.contiguous()
        n, h, w, c = pred.shape
        if c not in (3, 4):
            raise ValueError("prediction map must have 3 ([p,h,v]) or 4 ([type,p,h,v]) channels, got %d" % c)
        ws = self._workspace(n, h, w)
        self._last_nhw = (n, h, w)
        inst = torch.empty((n, h, w), dtype=torch.int32, device=self.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if not taps:
            L.check(L.lib().hvn_postproc(pred.data_ptr(), n, h, w, c, c - 3, inst.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                    "hvn_postproc")
            return inst
        blb = torch.empty((n, h, w), dtype=torch.int32, device=self.device)
        dist = torch.empty((n, h, w), dtype=torch.float64, device=self.device)
        marker = torch.empty((n, h, w), dtype=torch.int32, device=self.device)
        L.check(L.lib().hvn_postproc_taps(pred.data_ptr(), n, h, w, c, c - 3, inst.data_ptr(), blb.data_ptr(), dist.data_ptr(),
                                          marker.data_ptr(), ws.data_ptr(), ws.numel(), stream), "hvn_postproc_taps")
        return inst, blb, dist, marker

    def flood_stats(self, stream=None):
        """Which replay the marker-controlled watershed of the LAST `separate` call took, summed over its maps (hvn_postproc_stats):
        dict of component counts; `whole_tile_replays` are the exact one-lane whole-tile fallbacks."""
        if self._ws is None or getattr(self, "_last_nhw", None) is None:
            return None
        n, h, w = self._last_nhw
        out = (ctypes.c_longlong * 10)()
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        L.check(L.lib().hvn_postproc_stats(self._ws.data_ptr(), self._ws.numel(), n, h, w, out, ctypes.c_void_p(st.cuda_stream)), "hvn_postproc_stats")
        keys = ("components", "small_window", "bitmap_window", "hbm_window", "to_heap_by_marker_tie", "to_heap_by_full_frontier",
                "component_heap_replays", "whole_tile_replays", "maps_flagged", "largest_component_box")
        return {k: int(v) for k, v in zip(keys, out)}

    def table(self, inst, pred, nr_types):
        """-> (records uint8 view [N,max_inst,sizeof(rec)], counts int32 [N]) on the device."""
        n, h, w = inst.shape
        max_inst = h * w // 13 + 1   # an opened marker component holds at least one 13-px element
        nt = int(nr_types or 0)
        need = L.lib().hvn_instance_table_workspace_bytes(n, max_inst, nt)
        if self._tws is None or self._tws.numel() < need:
            self._tws = torch.empty(need, dtype=torch.uint8, device=self.device)
        rec = torch.empty((n, max_inst, ctypes.sizeof(L.hvn_inst_rec)), dtype=torch.uint8, device=self.device)
        counts = torch.empty((n,), dtype=torch.int32, device=self.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(L.lib().hvn_instance_table(inst.data_ptr(), pred.data_ptr() if nt else None, n, h, w, pred.shape[-1], nt,
                                           rec.data_ptr(), counts.data_ptr(), max_inst, self._tws.data_ptr(),
                                           self._tws.numel(), stream), "hvn_instance_table")
        return rec, counts


_REC_DTYPE = np.dtype([("label", "<i4"), ("area", "<i4"), ("rmin", "<i4"), ("rmax", "<i4"), ("cmin", "<i4"), ("cmax", "<i4"),
                       ("sum_x", "<f8"), ("sum_y", "<f8"), ("type", "<i4"), ("type_count", "<i4")])
assert _REC_DTYPE.itemsize == ctypes.sizeof(L.hvn_inst_rec)

_DEFAULT = {}


def _pp(device):
    key = str(device)
    if key not in _DEFAULT:
        _DEFAULT[key] = PostProc(device)
    return _DEFAULT[key]


def process_batch_device(pred_dev, nr_types=None, return_centroids=False):
    """pred_dev [N,h,w,3|4] float32 on the GPU -> (inst int32 [N,h,w] device tensor,
    records device tensor | None, counts device tensor | None)."""
    pp = _pp(pred_dev.device)
    inst = pp.separate(pred_dev)
    if return_centroids or nr_types is not None:
        rec, counts = pp.table(inst, pred_dev.contiguous(), nr_types)
        return inst, rec, counts
    return inst, None, None


def trace_contours_flat(inst_host, rec_host):
    """Host: contours[0] of every record (cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0] semantics, see
    csrc/hvn_contour.cpp) as flat arrays: (pts int32 [P,2] of (x, y), offs int64 [n_rec + 1]); record i owns
    pts[offs[i]:offs[i+1]] (empty for absent labels).  The flat form is what travels between ranks."""
    inst_host = np.ascontiguousarray(inst_host, np.int32)
    rec_host = np.ascontiguousarray(rec_host)
    h, w = inst_host.shape
    n = rec_host.shape[0]
    # a border pixel is visited at most 4 times (once per 4-neighbour side that faces the background)
    max_pts = 4 * int(rec_host["area"].sum()) + 8 * n + 16
    pts = np.empty((max_pts, 2), np.int32)
    offs = np.empty(n + 1, np.int64)
    tot = L.lib().hvn_trace_contours(inst_host.ctypes.data, h, w, rec_host.ctypes.data, n, pts.ctypes.data, max_pts, offs.ctypes.data)
    if tot < 0:
        raise L.HvnError("hvn_trace_contours failed (%d)" % tot)
    return pts[:tot].copy(), offs


def trace_contours(inst_host, rec_host):
    """-> dict label -> int32 [K,2] array of (x, y) (see trace_contours_flat)."""
    rec_host = np.ascontiguousarray(rec_host)
    pts, offs = trace_contours_flat(inst_host, rec_host)
    return {int(rec_host["label"][i]): pts[offs[i]:offs[i + 1]].copy() for i in range(rec_host.shape[0]) if rec_host["area"][i] > 0}


def records_to_dict(rec_host, nr_types, inst_host=None, contours_flat=None, shift_xy=None):
    """One tile's records (numpy structured array) -> the reference's inst_info_dict.  With `inst_host`
    the contours are traced too (or taken from `contours_flat` = trace_contours_flat's result, e.g. traced on another
    rank) and, like the reference (post_proc.py:140-143), instances whose contour
    has fewer than 3 points are left out of the dict (they stay in the instance map).  The per-instance fields are
    computed for the whole tile at once; only the dict assembly is a python loop (a WSI has ~10^6 instances).
    `shift_xy` = (x0, y0): the tile's origin in the slide, added to bbox, centroid and contour the way the WSI merge
    callbacks do (wsi.py:580-584: `+ top_left` with top_left = (x, y) on all three, i.e. x is added to the bbox ROWS --
    the reference's quirk, kept) -- vectorised here instead of three small-array additions per instance there."""
    r = rec_host[rec_host["area"] > 0]
    if contours_flat is not None:
        pts, offs = contours_flat
        if shift_xy is not None:
            pts = pts + np.asarray(shift_xy, pts.dtype)
        contours = {int(rec_host["label"][i]): pts[offs[i]:offs[i + 1]] for i in np.nonzero(rec_host["area"] > 0)[0]}
    else:
        contours = trace_contours(inst_host, rec_host) if inst_host is not None else None
    area = r["area"].astype(np.float64)
    bbox = np.stack([np.stack([r["rmin"], r["cmin"]], -1), np.stack([r["rmax"], r["cmax"]], -1)], 1).astype(np.int64)   # [n,2,2]
    # m10/m00 on the crop, then + offset (post_proc.py:145-152)
    cent = np.stack([r["sum_x"] / area + r["cmin"], r["sum_y"] / area + r["rmin"]], -1)
    if shift_xy is not None:
        if contours_flat is None:
            raise ValueError("shift_xy needs contours_flat")
        bbox = bbox + np.asarray(shift_xy, bbox.dtype)
        cent = cent + np.asarray(shift_xy, cent.dtype)
    labels = r["label"].tolist()
    types = r["type"].tolist() if nr_types is not None else None
    tprob = (r["type_count"] / (area + 1.0e-6)).tolist() if nr_types is not None else None
    out = {}
    for i, lab in enumerate(labels):
        contour = None
        if contours is not None:
            contour = contours[lab]
            if contour.shape[0] < 3:
                continue
        out[lab] = {"bbox": bbox[i], "centroid": cent[i], "contour": contour,
                    "type_prob": None if tprob is None else tprob[i], "type": None if types is None else types[i]}
    return out


def process(pred_map, nr_types=None, return_centroids=False):
    """Reference signature (post_proc.py:94): one host map [H,W,3|4] float32."""
    pred = torch.from_numpy(np.ascontiguousarray(pred_map, np.float32)).unsqueeze(0).to("cuda")
    inst, rec, _ = process_batch_device(pred, nr_types, return_centroids)
    pred_inst = inst[0].cpu().numpy()
    info = None
    if rec is not None:
        info = records_to_dict(rec[0].cpu().numpy().view(_REC_DTYPE).reshape(-1), nr_types, pred_inst)
    return pred_inst, info
