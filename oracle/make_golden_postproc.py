"""Generate tests/golden/pp_*.npz with the REFERENCE's own post_proc.py.

Run with the secondary interpreter (real scipy 1.7.1 + scikit-image 0.18.3):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore oracle/make_golden_postproc.py

It imports /root/reference/models/hovernet/post_proc.py unmodified; `cv2` resolves to
oracle/cv2_shim/cv2.py (OpenCV is absent from the box -- see hvn_oracle.c header).
What the goldens pin: scipy.ndimage.label / binary_fill_holes, skimage watershed,
remove_small_objects and every numpy dtype promotion in post_proc.py:26-90.
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

proc_np_hv = getattr(pp, "__proc_np_hv")
out_dir = out_dir()


def run(pred):
    return np.stack([proc_np_hv(p[..., -3:]).astype(np.int32) for p in pred])


cases = {}
# structured synthetic nuclei, the three tile sizes of SURVEY section 4
cases["s80"] = synth_pred_maps(12, 80, 80, None, seed=11)[0]
cases["s164"] = synth_pred_maps(3, 164, 164, None, seed=12)[0]
cases["s270"] = synth_pred_maps(1, 270, 270, None, seed=13)[0]
cases["s80t"] = synth_pred_maps(4, 80, 80, 5, seed=14)[0]
# ragged / non-square
cases["s57x91"] = synth_pred_maps(2, 57, 91, None, seed=15)[0]
# smooth random fields (no nucleus structure): irregular blobs, markers from noise
rng = np.random.Generator(np.random.PCG64(21))


def _smooth(a, it=6):
    for _ in range(it):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5.0
    return a


f = np.stack([_smooth(rng.normal(0, 1, (4, 80, 80)).transpose(1, 2, 0)).transpose(2, 0, 1) for _ in range(3)], -1)
f = f / f.std()
f[..., 0] = 0.5 + 0.5 * f[..., 0]
cases["noise80"] = f.astype(np.float32)
# quantised maps: few grey levels -> many equal-valued heap entries (tie order)
q = synth_pred_maps(4, 80, 80, None, seed=22, noise=0.0)[0]
q[..., 1:] = np.round(q[..., 1:] * 4) / 4
cases["quant80"] = q
# large connected blobs (components far beyond one nucleus: clumps), many noise markers inside: map 0 is one tile-filling
# blob, map 1 a ~21k-pixel blob next to ordinary nuclei -- the floods whose heap / label window outgrow on-chip memory
rb = np.random.Generator(np.random.PCG64(31))
fb = np.stack([_smooth(rb.normal(0, 1, (2, 200, 200)).transpose(1, 2, 0), it=4).transpose(2, 0, 1) for _ in range(3)], -1)
fb = fb / fb.std()
pb = np.zeros((2, 200, 200), np.float32)
pb[0, 4:196, 4:196] = 0.9
pb[1, 30:170, 25:175] = 0.9
small = synth_pred_maps(1, 200, 200, None, seed=32)[0][0]
keep = np.ones((200, 200), bool)
keep[24:176, 19:181] = False
pb[1][keep] = small[..., 0][keep]
fb[..., 0] = pb
fb[1, ..., 1:][keep] = small[..., 1:][keep]
cases["blob200"] = fb.astype(np.float32)
# degenerate: empty, full, constant h/v
e = np.zeros((3, 40, 40, 3), np.float32)
e[1, ..., 0] = 1.0
e[2, 5:30, 5:30, 0] = 1.0
cases["degenerate40"] = e

for name in selected(cases):
    pred = cases[name]
    inst = run(pred)
    np.savez_compressed(os.path.join(out_dir, "pp_%s.npz" % name), pred=pred, inst=inst)
    print(name, pred.shape, "instances per map:", [int(len(np.unique(i)) - 1) for i in inst])
