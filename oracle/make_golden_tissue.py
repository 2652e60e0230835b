"""Generate tests/golden/tissue_mask.npz with real scikit-image (secondary interpreter):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore oracle/make_golden_tissue.py

The morphology chain of infer/wsi.py:494-498 (remove_small_objects(min_size=256, connectivity=2) ->
remove_small_holes(area_threshold=128*128) -> binary_dilation(disk(16))) applied by skimage 0.18.3 to seeded boolean
images; pins hover_net_amd/tissue_mask.py's scipy restatement of those three calls (the two cv2 calls in front of them
cannot be pinned: OpenCV is absent)."""
import os

import numpy as np
from skimage import morphology

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for k, (seed, size, dens) in enumerate(((1, 400, 0.5), (2, 517, 0.35), (3, 300, 0.62))):
    rng = np.random.default_rng(seed)
    base = rng.random((size // 8 + 1, size // 8 + 1)) < dens            # blobby structure: 8x8 blocks + salt noise
    m = np.kron(base, np.ones((8, 8), bool))[:size, :size]
    m ^= rng.random((size, size)) < 0.02
    m[:40, :200] = True
    m[60:260, 60:260] &= ~(np.hypot(*np.mgrid[-100:100, -100:100]) < 70)  # a large hole (kept) ...
    m[300:330, 20:50] = True
    m[310:316, 30:36] = False                                            # ... and a small one (filled)
    a = morphology.remove_small_objects(m, min_size=16 * 16, connectivity=2)
    b = morphology.remove_small_holes(a, area_threshold=128 * 128)
    c = morphology.binary_dilation(b, morphology.disk(16))
    out["in%d" % k], out["a%d" % k], out["b%d" % k], out["c%d" % k] = m, a, b, c
    print(k, m.sum(), a.sum(), b.sum(), c.sum())
np.savez_compressed(os.path.join(REPO, "tests", "golden", "tissue_mask.npz"), n=3, **out)
