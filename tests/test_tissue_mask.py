"""hover_net_amd.tissue_mask: the morphology chain against goldens made by real scikit-image, Otsu / grey against
independent formulas."""
import os

import numpy as np
from scipy import ndimage

from hover_net_amd import tissue_mask as TM

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tissue_mask.npz")


def test_morphology_chain_matches_skimage_goldens():
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        a = TM.remove_small_objects(g["in%d" % k], 16 * 16, 2)
        assert np.array_equal(a, g["a%d" % k])
        b = TM.remove_small_holes(a, 128 * 128)
        assert np.array_equal(b, g["b%d" % k])
        c = ndimage.binary_dilation(b, structure=TM.disk(16))
        assert np.array_equal(c, g["c%d" % k])


def test_otsu_and_gray():
    rng = np.random.default_rng(0)
    gray = np.concatenate([rng.normal(60, 10, 5000), rng.normal(190, 15, 8000)]).clip(0, 255).astype(np.uint8)
    t = TM.otsu_threshold(gray)
    # brute force: maximise the between-class variance over all thresholds
    best, bt = -1.0, -1
    for th in range(256):
        lo, hi = gray[gray <= th], gray[gray > th]
        if len(lo) == 0 or len(hi) == 0:
            continue
        v = len(lo) * len(hi) * (lo.mean() - hi.mean()) ** 2
        if v > best:
            best, bt = v, th
    assert t == bt and 80 < t < 160
    rgb = rng.integers(0, 256, (50, 60, 3), dtype=np.uint8)
    ref = np.round(0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2])
    assert np.abs(TM.rgb_to_gray(rgb).astype(np.int32) - ref).max() <= 1


def test_simple_get_mask_end_to_end():
    thumb = np.full((600, 700, 3), 235, np.uint8)            # bright background
    thumb[100:400, 150:500] = (150, 90, 160)                 # tissue
    thumb[200:230, 250:280] = 235                            # small hole: filled
    thumb[500:505, 600:605] = (150, 90, 160)                 # dust: removed
    m = TM.simple_get_mask(thumb)
    assert m.dtype == np.uint8 and set(np.unique(m)) == {0, 1}
    assert m[250, 300] == 1 and m[215, 265] == 1 and m[502, 602] == 0
    assert m[90, 300] == 1 and m[80, 300] == 0              # dilated by 16 px
