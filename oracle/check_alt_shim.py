"""Re-pin the cv2 leg of the post-processing goldens with the SECOND cv2 stand-in -- TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore oracle/check_alt_shim.py [--json out.json]

Runs the reference's own `__proc_np_hv` (/root/reference/models/hovernet/post_proc.py:26-90) over
oracle/cv2_shim_scipy/cv2.py (scipy.ndimage restatement of normalize / Sobel / GaussianBlur / morphologyEx, written
without the C oracle) on every committed tests/golden/pp_*.npz / proc_*.npz input and compares the instance maps with the
committed ones (which were made over oracle/cv2_shim/cv2.py -> hvn_oracle.c).  Reports per case: identical or not, and the
reference's own panoptic quality (metrics/stats_utils.py:178 get_fast_pq after :360 remap_label; [1, 1, 1] = same
partition) for every non-empty map.
"""
import glob
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from refimport import ref_import, use_reference  # noqa: E402

use_reference(first=[os.path.join(HERE, "cv2_shim_scipy")])
sys.modules.setdefault("matplotlib", types.ModuleType("matplotlib"))
sys.modules.setdefault("matplotlib.pyplot", types.ModuleType("matplotlib.pyplot"))

import cv2  # noqa: E402
pp = ref_import("models.hovernet.post_proc")  # the reference, unmodified (asserted to live under /root/reference)
_stats = ref_import("metrics.stats_utils")       # the reference's own metric
get_fast_pq, remap_label = _stats.get_fast_pq, _stats.remap_label

assert "cv2_shim_scipy" in cv2.__file__
proc_np_hv = getattr(pp, "__proc_np_hv")
report = {}
for path in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "pp_*.npz")) + glob.glob(os.path.join(REPO, "tests", "golden", "proc_*.npz"))):
    z = np.load(path)
    same, pqs = [], []
    for pred, want in zip(z["pred"], z["inst"]):
        got = proc_np_hv(pred[..., -3:]).astype(np.int32)
        same.append(bool(np.array_equal(got, want)))
        if want.max() > 0 and got.max() > 0:
            pqs.append([float(v) for v in get_fast_pq(remap_label(want), remap_label(got))[0]])
        else:
            pqs.append([1.0, 1.0, 1.0] if want.max() == got.max() else [0.0, 0.0, 0.0])
    report[os.path.basename(path)] = {"maps": len(same), "identical": int(sum(same)), "min_pq": float(min(p[2] for p in pqs))}
    print(os.path.basename(path), report[os.path.basename(path)])
if "--json" in sys.argv:
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
