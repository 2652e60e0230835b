"""Stand-in `cv2` module -- TEST INFRASTRUCTURE ONLY (golden-vector generation).

OpenCV is not installed on the build box.  To run the reference's own, unmodified
models/hovernet/post_proc.py (under /opt/conda/bin/python3.9, which has real scipy
and scikit-image) this module supplies the cv2 entry points that file touches
(post_proc.py:49-54,56-57,59-68,76,83-84), implemented by the C restatement in
oracle/hvn_oracle.c.  The scipy / skimage / numpy parts of the golden vectors are
therefore the real thing; the cv2 image filters are the C restatement, cross-checked by a second,
scipy-based stand-in (oracle/cv2_shim_scipy/cv2.py, tests/test_oracle_cv2_independent.py);
moments / findContours (post_proc.py:131-135) are python (_suzuki.py).
For the reference's dataloader/augs.py (oracle/make_golden_augs.py) the 8-bit GaussianBlur / medianBlur / cvtColor entry points
resolve to oracle/augment_np.py: the numpy glue of the six augmentation functions is then the reference's own code.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import postproc as _o  # noqa: E402  (oracle/postproc.py)

NORM_MINMAX = 32
CV_8U, CV_32F, CV_64F = 0, 5, 6
MORPH_OPEN = 2
MORPH_ELLIPSE = 2
RETR_TREE = 3
CHAIN_APPROX_SIMPLE = 2
COLOR_BGR2RGB = 4
COLOR_RGB2GRAY, COLOR_RGB2HSV, COLOR_HSV2RGB = 7, 41, 55
BORDER_REPLICATE = 1


def normalize(src, dst=None, alpha=0, beta=1, norm_type=NORM_MINMAX, dtype=CV_32F):
    assert norm_type == NORM_MINMAX and alpha == 0 and beta == 1 and dtype == CV_32F
    if src.dtype == np.float32:
        return _o.normalize_32f(src)
    assert src.dtype == np.float64
    return _o.normalize_64f32f(src)


def Sobel(src, ddepth, dx, dy, ksize=3):
    assert ddepth == CV_64F and ksize == 21 and src.dtype == np.float32
    assert (dx, dy) in ((1, 0), (0, 1))
    return _o.sobel21(src, dx)


def GaussianBlur(src, ksize, sigmaX, sigmaY=0, borderType=None):
    if src.dtype == np.uint8:          # dataloader/augs.py:42-44 (8-bit image, ksize in {1,3,5}^2, BORDER_REPLICATE): oracle/augment_np.py
        import augment_np as _a
        assert sigmaX == 0 and sigmaY == 0 and borderType == BORDER_REPLICATE
        return _a.gaussian_blur(src, int(ksize[0]), int(ksize[1]))
    assert tuple(ksize) == (3, 3) and sigmaX == 0 and src.dtype == np.float64
    return _o.gauss3_64f(src)


def medianBlur(src, ksize):
    """dataloader/augs.py:56."""
    import augment_np as _a
    assert src.dtype == np.uint8
    return _a.median_blur(src, int(ksize))


def cvtColor(src, code):
    """dataloader/augs.py:66,74,83 on 8-bit images."""
    import augment_np as _a
    assert src.dtype == np.uint8
    return {COLOR_RGB2HSV: _a.rgb2hsv_u8, COLOR_HSV2RGB: _a.hsv2rgb_u8, COLOR_RGB2GRAY: _a.rgb2gray_u8}[code](src)


def getStructuringElement(shape, ksize):
    assert shape == MORPH_ELLIPSE and tuple(ksize) == (5, 5)
    return np.array(
        [[0, 0, 1, 0, 0], [1] * 5, [1] * 5, [1] * 5, [0, 0, 1, 0, 0]], np.uint8
    )


def morphologyEx(src, op, kernel):
    assert op == MORPH_OPEN and src.dtype == np.uint8
    assert np.array_equal(kernel, getStructuringElement(MORPH_ELLIPSE, (5, 5)))
    return _o.morph_open5(src)


# --- post_proc.py:131-135 (the per-instance loop of `process`): independent python restatement, NOT the C oracle and
# --- not the product's tracer (see _suzuki.py)
from _suzuki import find_contours_tree as _find_contours_tree, moments as _moments  # noqa: E402


def moments(array, binaryImage=False):
    assert not binaryImage
    return _moments(array)


def findContours(image, mode, method):
    """OpenCV >= 4 protocol: (contours, hierarchy)."""
    assert mode == RETR_TREE and method == CHAIN_APPROX_SIMPLE and image.dtype == np.uint8
    return _find_contours_tree(image)
