"""Generate tests/golden/targets.npz with the REFERENCE's own target generation.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore oracle/make_golden_targets.py

Runs under the secondary interpreter (real scipy 1.7.1 + scikit-image 0.18.3 + matplotlib); imports
/root/reference/models/hovernet/targets.py unmodified -- `cv2` resolves to oracle/cv2_shim (imported by misc/utils.py
and dataloader/augs.py, unused on this path) and `torch` (imported at targets.py:4-5, unused by gen_targets) to an
empty stub, since that interpreter has no torch.  Stores the synthetic instance maps and the reference's
hv_map / np_map for them.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from refimport import out_dir, ref_import, use_reference  # noqa: E402

use_reference(first=[os.path.join(HERE, "cv2_shim")])
for name in ("torch", "torch.nn", "torch.nn.functional"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["torch"].nn = sys.modules["torch.nn"]
sys.modules["torch.nn"].functional = sys.modules["torch.nn.functional"]

import importlib.util  # noqa: E402

ref_targets = ref_import("models.hovernet.targets")  # the reference, unmodified (asserted to live under /root/reference)

spec = importlib.util.spec_from_file_location("targets_np", os.path.join(HERE, "targets_np.py"))
tn = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tn)

CASES = [(270, 80, 60, True, 1), (270, 80, 25, False, 2), (256, 164, 80, True, 3), (270, 80, 140, True, 4)]
out = {}
for k, (size, crop, n_inst, mirror, seed) in enumerate(CASES):
    ann = tn.synth_ann(np.random.default_rng(seed), size, n_inst, mirror)
    t = ref_targets.gen_targets(ann.copy(), (crop, crop))
    out["ann%d" % k] = ann.astype(np.int16)
    out["hv%d" % k] = t["hv_map"].astype(np.float32)
    out["np%d" % k] = t["np_map"].astype(np.uint8)
    out["crop%d" % k] = crop
    print(k, ann.shape, int(ann.max()), float(np.abs(t["hv_map"]).sum()), int(t["np_map"].sum()))
np.savez_compressed(os.path.join(out_dir(), "targets.npz"), n=len(CASES), **out)
