"""SECOND, independent stand-in `cv2` module -- TEST INFRASTRUCTURE ONLY.

oracle/cv2_shim/cv2.py routes the four image-filter calls of the reference's post_proc.py
(/root/reference/models/hovernet/post_proc.py:49-54 normalize, :56-57 Sobel, :59-68 normalize,
:76 GaussianBlur, :83-84 getStructuringElement / morphologyEx) to the C restatement oracle/hvn_oracle.c, so
goldens made with it pin the C code only to itself for those calls.  This module restates the same five
functions a second time, from the documented semantics only, with numpy / scipy.ndimage in float64 and
without looking at the C code's operation order:

  normalize(NORM_MINMAX, 0, 1, CV_32F)   (x - min) / (max - min), 0 when max - min <= DBL_EPSILON
  Sobel(CV_64F, ksize=21)                separable correlation, BORDER_REFLECT_101 (= scipy "mirror"),
                                         smoothing taps = binomial(20), derivative taps = binomial(18) * [-1, 0, 1]
  GaussianBlur((3, 3), 0)                separable [1, 2, 1] / 4, BORDER_REFLECT_101
  getStructuringElement(ELLIPSE, (5,5))  OpenCV's row-span formula dx = round(c * sqrt((r^2 - dy^2) / r^2))
  morphologyEx(OPEN)                     erosion then dilation, the border never wins (erode: outside = 1, dilate: 0)

tests/test_oracle_cv2_independent.py holds the two against each other: integer results bit-equal, floating-point
results within the stated ulp bound (the two differ in summation order only), and the reference's __proc_np_hv run
over THIS module gives the same instance maps as the committed goldens.  moments / findContours are shared with the
first shim (they are python already, oracle/cv2_shim/_suzuki.py).
"""
import os
import sys

import numpy as np
from scipy import ndimage, special

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cv2_shim"))
from _suzuki import find_contours_tree as _find_contours_tree, moments as _moments  # noqa: E402

NORM_MINMAX = 32
CV_8U, CV_32F, CV_64F = 0, 5, 6
MORPH_OPEN = 2
MORPH_ELLIPSE = 2
RETR_TREE = 3
CHAIN_APPROX_SIMPLE = 2
COLOR_BGR2RGB = 4


def normalize(src, dst=None, alpha=0, beta=1, norm_type=NORM_MINMAX, dtype=CV_32F):
    assert norm_type == NORM_MINMAX and alpha == 0 and beta == 1 and dtype == CV_32F
    x = np.asarray(src, np.float64)
    lo, hi = x.min(), x.max()
    if hi - lo <= np.finfo(np.float64).eps:
        return np.zeros(x.shape, np.float32)
    return ((x - lo) / (hi - lo)).astype(np.float32)


def sobel_taps(ksize, order):
    if order == 0:
        return special.comb(ksize - 1, np.arange(ksize), exact=False)
    assert order == 1
    return np.convolve(special.comb(ksize - 3, np.arange(ksize - 2), exact=False), [-1.0, 0.0, 1.0])


def Sobel(src, ddepth, dx, dy, ksize=3):
    assert ddepth == CV_64F and src.dtype == np.float32 and (dx, dy) in ((1, 0), (0, 1))
    x = src.astype(np.float64)
    # taps are in correlation order (the derivative kernel has its negative side first, like OpenCV's [-1, 0, 1])
    kx, ky = sobel_taps(ksize, dx), sobel_taps(ksize, dy)
    x = ndimage.correlate1d(x, kx, axis=1, mode="mirror")
    return ndimage.correlate1d(x, ky, axis=0, mode="mirror")


def GaussianBlur(src, ksize, sigmaX):
    assert tuple(ksize) == (3, 3) and sigmaX == 0 and src.dtype == np.float64
    k = np.array([0.25, 0.5, 0.25])
    return ndimage.correlate1d(ndimage.correlate1d(src, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")


def getStructuringElement(shape, ksize):
    assert shape == MORPH_ELLIPSE
    w, h = ksize
    r, c = h // 2, w // 2
    out = np.zeros((h, w), np.uint8)
    for i in range(h):
        d = i - r
        if abs(d) <= r:
            half = int(np.rint(c * np.sqrt((r * r - d * d) / float(r * r))))
            out[i, max(c - half, 0):min(c + half + 1, w)] = 1
    return out


def morphologyEx(src, op, kernel):
    assert op == MORPH_OPEN and src.dtype == np.uint8
    fg = src != 0
    er = ndimage.binary_erosion(fg, structure=kernel.astype(bool), border_value=1)
    return ndimage.binary_dilation(er, structure=kernel.astype(bool), border_value=0).astype(np.uint8) * src.max()


def moments(array, binaryImage=False):
    assert not binaryImage
    return _moments(array)


def findContours(image, mode, method):
    assert mode == RETR_TREE and method == CHAIN_APPROX_SIMPLE and image.dtype == np.uint8
    return _find_contours_tree(image)
