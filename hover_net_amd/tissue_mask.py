"""Tissue mask at 1.25x for whole-slide inference -- host preprocessing of `infer/wsi.py:486-500`
(`simple_get_mask`): grey conversion, Otsu threshold, tissue = dark side, drop objects < 16x16 px (8-connected), fill holes
< 128x128 px, dilate with a radius-16 disk.  A few Mpix on the host per slide; not on the GPU path.

The two OpenCV calls are restated (OpenCV is not on the box, so they are unpinned against it): `cvtColor(RGB2GRAY)` is the
8-bit fixed-point form (R*4899 + G*9617 + B*1868 + 8192) >> 14, `threshold(THRESH_OTSU)` maximises the between-class
variance over the 256-bin histogram (first maximum wins) and marks pixels > t.  The three skimage.morphology calls are
restated with scipy.ndimage and pinned against skimage 0.18.3 (tests/golden/tissue_mask.npz, oracle/make_golden_tissue.py).
"""
import numpy as np
from scipy import ndimage


def rgb_to_gray(rgb):
    r, g, b = (rgb[..., k].astype(np.int32) for k in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def otsu_threshold(gray):
    hist = np.bincount(gray.reshape(-1), minlength=256).astype(np.float64)
    n = hist.sum()
    p = hist / n
    omega = np.cumsum(p)                       # class-0 probability for threshold t (values <= t)
    mu = np.cumsum(p * np.arange(256))
    mu_t = mu[-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        sigma = (mu_t * omega - mu) ** 2 / (omega * (1.0 - omega))
    sigma[~np.isfinite(sigma)] = 0.0
    return int(np.argmax(sigma))


def remove_small_objects(mask, min_size, connectivity):
    """skimage.morphology.remove_small_objects on a boolean image (connectivity 1 = 4-, 2 = 8-neighbourhood)."""
    lab, _ = ndimage.label(mask, ndimage.generate_binary_structure(2, connectivity))
    sizes = np.bincount(lab.reshape(-1))
    too_small = sizes < min_size
    too_small[0] = False
    out = mask.copy()
    out[too_small[lab]] = False
    return out


def remove_small_holes(mask, area_threshold):
    """skimage.morphology.remove_small_holes (default connectivity 1): small objects of the complement are filled."""
    return ~remove_small_objects(~mask, area_threshold, 1)


def disk(radius):
    yy, xx = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    return (xx * xx + yy * yy) <= radius * radius


def simple_get_mask(thumb_rgb):
    """thumb_rgb: uint8 [h,w,3] thumbnail at 1.25x -> uint8 {0,1} tissue mask (wsi.py:489-500)."""
    gray = rgb_to_gray(thumb_rgb)
    t = otsu_threshold(gray)
    mask = ~(gray > t)                                               # cv2.threshold -> 255 where > t; tissue = (mask == 0)
    mask = remove_small_objects(mask, 16 * 16, 2)
    mask = remove_small_holes(mask, 128 * 128)
    mask = ndimage.binary_dilation(mask, structure=disk(16))
    return mask.astype(np.uint8)
