"""CPU restatement of the reference's `process()` -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/models/hovernet/post_proc.py:94-186: instance separation (`__proc_np_hv`,
restated in oracle/hvn_oracle.c) followed by the per-instance loop -- bounding box
(/root/reference/misc/utils.py:18-28), `cv2.moments` centroid, `cv2.findContours(RETR_TREE,
CHAIN_APPROX_SIMPLE)[0][0]` with the "< 3 points" skip (:140-143), majority type with the
background-runner-up rule (:162-181).  moments / findContours are the python restatement of
oracle/cv2_shim/_suzuki.py.  Pinned by tests/test_oracle_process.py to tests/golden/proc_*.npz, which
the reference's own `process()` made (oracle/make_golden_process.py).
"""
import os
import sys

import numpy as np
from scipy import ndimage

_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cv2_shim")
if _SHIM not in sys.path:
    sys.path.insert(0, _SHIM)
import _suzuki  # noqa: E402

from . import postproc as O  # noqa: E402


def instance_info(pred_inst, pred_type=None):
    """The per-instance loop on a finished instance map (any labelling, ids need not be contiguous)."""
    info = {}
    slices = ndimage.find_objects(pred_inst)
    for inst_id in np.unique(pred_inst):
        if inst_id == 0:
            continue
        sl = slices[inst_id - 1]
        r0, r1, c0, c1 = sl[0].start, sl[0].stop, sl[1].start, sl[1].stop
        crop = (pred_inst[r0:r1, c0:c1] == inst_id).astype(np.uint8)
        mom = _suzuki.moments(crop)
        contour = np.squeeze(_suzuki.find_contours_tree(crop)[0][0].astype("int32"))
        if contour.shape[0] < 3 or contour.ndim != 2:
            continue
        contour = contour + np.array([c0, r0], np.int32)
        e = {"bbox": np.array([[r0, c0], [r1, c1]]),
             "centroid": np.array([mom["m10"] / mom["m00"] + c0, mom["m01"] / mom["m00"] + r0]),
             "contour": contour, "type_prob": None, "type": None}
        if pred_type is not None:
            votes = pred_type[r0:r1, c0:c1][crop > 0]
            kinds, counts = np.unique(votes, return_counts=True)
            ranked = sorted(zip(kinds, counts), key=lambda kc: kc[1], reverse=True)   # stable: ties keep ascending type
            t = ranked[0][0]
            if t == 0 and len(ranked) > 1:
                t = ranked[1][0]
            e["type"] = int(t)
            e["type_prob"] = float(dict(ranked)[t] / (crop.sum() + 1.0e-6))
        info[int(inst_id)] = e
    return info


def process(pred_map, nr_types=None, return_centroids=False):
    pred_map = np.asarray(pred_map, np.float32)
    pred_type = pred_map[..., 0].astype(np.int32) if nr_types is not None else None
    pred_inst = O.proc_np_hv(pred_map[..., -3:])
    if not (return_centroids or nr_types is not None):
        return pred_inst, None
    return pred_inst, instance_info(pred_inst, pred_type)
