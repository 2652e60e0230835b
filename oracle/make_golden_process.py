"""Generate tests/golden/proc_*.npz with the REFERENCE's own `process()`
(/root/reference/models/hovernet/post_proc.py:94-186), imported unmodified.

Run with the secondary interpreter (real scipy 1.7.1 + scikit-image 0.18.3):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore oracle/make_golden_process.py

`cv2` resolves to oracle/cv2_shim/cv2.py.  For the per-instance loop (post_proc.py:119-181) that means
`cv2.moments` / `cv2.findContours` are the independent python restatement in oracle/cv2_shim/_suzuki.py
(OpenCV is absent from the box); `get_bounding_box`, `np.unique`, the type vote and every dtype
promotion are the reference's own code.  What the fixtures pin: the instance map, the KEY SET of
inst_info_dict (incl. the < 3 contour points skip), bbox, centroid, contour point ORDER, type, type_prob.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from refimport import out_dir, ref_import, selected, use_reference  # noqa: E402

use_reference(first=[os.path.join(HERE, "cv2_shim")])
pp = ref_import("models.hovernet.post_proc")  # the reference, unmodified (asserted to live under /root/reference)
from hover_net_amd.synth import synth_pred_maps  # noqa: E402

out_dir = out_dir()


def flatten(results):
    """[(inst, info)] -> dict of flat arrays (no pickles in the fixtures)."""
    inst = np.stack([r[0].astype(np.int32) for r in results])
    ids, bbox, cent, typ, tprob, cpts, coff, moff = [], [], [], [], [], [], [0], [0]
    for _, info in results:
        for k in info.keys():                      # insertion order = ascending instance id (np.unique)
            e = info[k]
            ids.append(int(k))
            bbox.append(np.asarray(e["bbox"], np.int64))
            cent.append(np.asarray(e["centroid"], np.float64))
            typ.append(-1 if e["type"] is None else int(e["type"]))
            tprob.append(np.nan if e["type_prob"] is None else float(e["type_prob"]))
            assert e["contour"].dtype == np.int32 and e["contour"].ndim == 2
            cpts.append(e["contour"])
            coff.append(coff[-1] + len(e["contour"]))
        moff.append(len(ids))
    return dict(inst=inst, ids=np.asarray(ids, np.int32), bbox=np.asarray(bbox, np.int64).reshape(-1, 2, 2),
                centroid=np.asarray(cent, np.float64).reshape(-1, 2), type=np.asarray(typ, np.int32),
                type_prob=np.asarray(tprob, np.float64), contour_pts=np.concatenate(cpts).astype(np.int32) if cpts else
                np.zeros((0, 2), np.int32), contour_off=np.asarray(coff, np.int64), map_off=np.asarray(moff, np.int64))


cases = {
    # name: (pred maps [N,H,W,3|4], nr_types)
    "s80t": (synth_pred_maps(8, 80, 80, 5, seed=51)[0], 5),
    "s80": (synth_pred_maps(6, 80, 80, None, seed=52)[0], None),
    "s164t": (synth_pred_maps(2, 164, 164, 6, seed=53)[0], 6),
    "s270": (synth_pred_maps(1, 270, 270, None, seed=54)[0], None),
}
# irregular instances: structured maps whose nucleus-probability channel is punched with smooth-noise holes, so that
# instances get holes and concavities (hole borders must not disturb contours[0]; the centroid is no longer the bbox centre)
rng = np.random.Generator(np.random.PCG64(61))


def _smooth(a, it=2):
    for _ in range(it):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5.0
    return a


punched = synth_pred_maps(4, 120, 120, 5, seed=55, k_lo=10, k_hi=30)[0]
for p in punched:
    n = _smooth(rng.normal(0, 1, (120, 120)))
    p[..., 1][n / n.std() > 1.3] = 0.1
cases["punched120t"] = (punched, 5)

for name in selected(cases):
    pred, nt = cases[name]
    res = [pp.process(p, nr_types=nt, return_centroids=True) for p in pred]
    z = flatten(res)
    np.savez_compressed(os.path.join(out_dir, "proc_%s.npz" % name), pred=pred, nr_types=np.int32(-1 if nt is None else nt), **z)
    n_inst = [len(np.unique(i)) - 1 for i in z["inst"]]
    print(name, pred.shape, "instances:", n_inst, "dict entries:", np.diff(z["map_off"]).tolist(),
          "contour points:", int(z["contour_off"][-1]))
