"""Generate tests/golden/augs.npz with the REFERENCE's own image-augmentation functions (/root/reference/dataloader/augs.py:36-113,
imported unmodified) under the secondary interpreter:

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore oracle/make_golden_augs.py

`cv2` resolves to oracle/cv2_shim/cv2.py, whose 8-bit GaussianBlur / medianBlur / cvtColor are oracle/augment_np.py's restatements
(OpenCV is absent), so these fixtures pin the REFERENCE'S GLUE around them: how `random_state` draws become kernel sizes and offsets, the
dtype promotions (`img * value`, `img + value` in float64), the `% 180` on the hue plane stored back into uint8, clip + truncating
`astype(uint8)`, and `add_to_contrast` returning its input.  They do not pin OpenCV's own arithmetic (header of oracle/augment_np.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from refimport import out_dir, ref_import, use_reference  # noqa: E402

use_reference(first=[os.path.join(HERE, "cv2_shim"), HERE])
ref = ref_import("dataloader.augs")  # the reference, unmodified (asserted to live under /root/reference)


class Draws:
    """Stands in for imgaug's `random_state`: hands out preset values and records what was asked."""

    def __init__(self, ints=(), floats=()):
        self.ints, self.floats, self.asked = list(ints), list(floats), []

    def randint(self, lo, hi, size=None):
        self.asked.append(("randint", lo, hi, size))
        if size is None:
            return self.ints.pop(0)
        return np.array([self.ints.pop(0) for _ in range(int(np.prod(size)))]).reshape(size)

    def uniform(self, lo, hi):
        self.asked.append(("uniform", lo, hi))
        return self.floats.pop(0)


rng = np.random.default_rng(2024)
img = rng.integers(0, 256, (6, 33, 41, 3), dtype=np.uint8)
img[1, :6] = 255
img[2, :6] = 0
img[3, :6] = 128
out = {"img": img}
g_in, g_out, g_k = [], [], []
for k, (a, b) in enumerate([(0, 0), (1, 0), (0, 2), (2, 1), (2, 2), (1, 1)]):
    d = Draws(ints=[a, b])
    g_out.append(ref.gaussian_blur([img[k]], d, None, None, max_ksize=3)[0])
    g_k.append((a * 2 + 1, b * 2 + 1))
    assert d.asked == [("randint", 0, 3, (2,))]
out["gauss_k"], out["gauss_out"] = np.array(g_k), np.stack(g_out)
m_out = []
for k, a in enumerate([0, 1, 2]):
    d = Draws(ints=[a])
    m_out.append(ref.median_blur([img[k]], d, None, None, max_ksize=3)[0])
    assert d.asked == [("randint", 0, 3, None)]
out["median_k"], out["median_out"] = np.array([1, 3, 5]), np.stack(m_out)
for name, fn, rng_arg, vals in (("hue", ref.add_to_hue, (-8, 8), [-8.0, -3.3, 0.0, 2.5, 7.999, 4.0]),
                                ("sat", ref.add_to_saturation, (-0.2, 0.2), [-0.2, -0.07, 0.0, 0.05, 0.2, 0.13]),
                                ("bright", ref.add_to_brightness, (-26, 26), [-26.0, -10.7, 0.0, 0.4, 25.5, 13.9]),
                                ("contrast", ref.add_to_contrast, (0.75, 1.25), [0.75, 0.9, 1.0, 1.1, 1.25, 1.2])):
    res = []
    for k, v in enumerate(vals):
        d = Draws(floats=[v])
        res.append(fn([img[k]], d, None, None, range=rng_arg)[0])
        assert d.asked == [("uniform",) + rng_arg] and res[-1].dtype == np.uint8
    out[name + "_val"], out[name + "_out"] = np.array(vals), np.stack(res)
np.savez_compressed(os.path.join(out_dir(), "augs.npz"), **out)
print("wrote tests/golden/augs.npz", {k: v.shape for k, v in out.items()})
