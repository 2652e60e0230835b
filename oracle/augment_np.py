"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the reference's training-time augmentation, parameterised by
EXPLICIT parameters (no random draws here), for the parity tests of `hover_net_amd/augment.py` / `csrc/hvn_augment.hip`.

What it follows:
  * dataloader/train_loader.py:76-109 `FileLoader.__getitem__` and :111-199 `__get_augmentation`: shape augmentations
    (imgaug `Affine` scale / translate / shear / rotate with order 0 = nearest, constant 0 outside; `CropToFixedSize`
    centre; `Fliplr`, `Flipud`) on image AND annotation, then input augmentations on the image only: one of {Gaussian
    blur, median blur, additive Gaussian noise}, then hue / saturation / brightness / contrast in random order;
  * dataloader/augs.py:36-113: the explicit functions those `iaa.Lambda`s call.

PARITY UNPINNED: imgaug 0.4.0 and opencv-python 4.3.0.36 (requirements.txt:3,6) are in neither interpreter of this image,
so nothing here could be checked against the real libraries.  The cv2 arithmetic is restated from OpenCV 4.3's sources as
remembered (imgproc/src/color_hsv.simd.hpp `RGB2HSV_b` / `HSV2RGB_native`, color_yuv `RGB2Gray<uchar>`, smooth.dispatch.cpp
fixed-point `GaussianBlur` for 8-bit, median_blur.simd.hpp), the affine geometry from imgaug's documented convention
(rotation about the image centre, skimage's scale/rotation/shear parameterisation); the sub-pixel rounding of
`cv2.warpAffine(INTER_NEAREST)` (10-bit fixed point) is replaced by round-half-up of the float64 source coordinate.
Deliberately reproduced reference behaviour: `add_to_contrast` returns its input unchanged (augs.py:96-97 clips `img`, not `ret`)."""
import numpy as np


# ------------------------------------------------------------------------------------------------------------------
# geometry
def affine_matrix(h, w, scale_xy, translate_px, shear_deg, rotate_deg):
    """Forward 3x3 matrix source -> destination: scale, shear and rotation (skimage `AffineTransform` parameterisation) about the
    image centre ((w-1)/2, (h-1)/2), then the translation."""
    sx, sy = scale_xy
    rot, sh = np.deg2rad(rotate_deg), np.deg2rad(shear_deg)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    m = np.array([[sx * np.cos(rot), -sy * np.sin(rot + sh), 0.0],
                  [sx * np.sin(rot), sy * np.cos(rot + sh), 0.0],
                  [0.0, 0.0, 1.0]])
    to_origin = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    back = np.array([[1, 0, cx + translate_px[0]], [0, 1, cy + translate_px[1]], [0, 0, 1.0]])
    return back @ m @ to_origin


def shape_augment(img, ann, inv, out_hw, flip_lr, flip_ud):
    """img uint8 [H,W,3], ann int32 [H,W,C]; `inv` = inverse of `affine_matrix` (destination -> source, rows 0..1 used).
    Output = centre crop of size `out_hw` of the warped image, then the flips (train_loader.py:131-151)."""
    h, w = img.shape[:2]
    oh, ow = out_hw
    y0, x0 = int((h - oh) * 0.5), int((w - ow) * 0.5)
    ys, xs = np.mgrid[0:oh, 0:ow]
    if flip_lr:
        xs = ow - 1 - xs
    if flip_ud:
        ys = oh - 1 - ys
    xd = (xs + x0).astype(np.float64)
    yd = (ys + y0).astype(np.float64)
    sxf = inv[0, 0] * xd + inv[0, 1] * yd + inv[0, 2]
    syf = inv[1, 0] * xd + inv[1, 1] * yd + inv[1, 2]
    sx = np.floor(sxf + 0.5).astype(np.int64)
    sy = np.floor(syf + 0.5).astype(np.int64)
    ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    sxc, syc = np.clip(sx, 0, w - 1), np.clip(sy, 0, h - 1)
    out_img = np.where(ok[..., None], img[syc, sxc], 0).astype(np.uint8)
    out_ann = np.where(ok[..., None], ann[syc, sxc], 0).astype(np.int32)
    return out_img, out_ann


# ------------------------------------------------------------------------------------------------------------------
# input augmentations (image only)
_G3 = np.array([1, 2, 1], np.int64)           # getGaussianKernel(3, sigma<=0) * 4   (small_gaussian_tab)
_G5 = np.array([1, 4, 6, 4, 1], np.int64)     # getGaussianKernel(5, sigma<=0) * 16


def _taps(k):
    return {1: (np.array([1], np.int64), 1), 3: (_G3, 4), 5: (_G5, 16)}[k]


def gaussian_blur(img, kx, ky):
    """augs.py:36-47: `cv2.GaussianBlur(img, (kx, ky), 0, 0, BORDER_REPLICATE)` on uint8.  The 8-bit path works in fixed point;
    with power-of-two tap sums every intermediate is exact, so the result is round-half-up of the exact rational."""
    tx, dx = _taps(kx)
    ty, dy = _taps(ky)
    a = img.astype(np.int64)
    h, w = a.shape[:2]
    rx, ry = kx // 2, ky // 2
    row = np.zeros_like(a)
    for i, t in enumerate(tx):
        idx = np.clip(np.arange(w) + i - rx, 0, w - 1)
        row += t * a[:, idx]
    col = np.zeros_like(a)
    for j, t in enumerate(ty):
        idx = np.clip(np.arange(h) + j - ry, 0, h - 1)
        col += t * row[idx]
    den = dx * dy
    return ((2 * col + den) // (2 * den)).astype(np.uint8)


def median_blur(img, k):
    """augs.py:51-58: `cv2.medianBlur(img, k)`, per channel, BORDER_REPLICATE; k = 1 copies."""
    if k == 1:
        return img.copy()
    r = k // 2
    h, w = img.shape[:2]
    p = np.pad(img, ((r, r), (r, r), (0, 0)), "edge")
    stack = np.stack([p[dy:dy + h, dx:dx + w] for dy in range(k) for dx in range(k)], 0)
    return np.sort(stack, 0)[(k * k) // 2].astype(np.uint8)


def additive_noise(img, noise):
    """imgaug `AdditiveGaussianNoise` on uint8 with the samples GIVEN (`noise` float [H,W,1 or 3], already scaled): samples are
    rounded to integers, added, saturated."""
    return np.clip(img.astype(np.int64) + np.rint(noise).astype(np.int64), 0, 255).astype(np.uint8)


def _cv_round(x):
    return np.rint(x).astype(np.int64)          # cvRound: round half to even


_HSV_SHIFT = 12
_SDIV = np.zeros(256, np.int64)
_HDIV = np.zeros(256, np.int64)
_SDIV[1:] = _cv_round((255 << _HSV_SHIFT) / (1.0 * np.arange(1, 256)))
_HDIV[1:] = _cv_round((180 << _HSV_SHIFT) / (6.0 * np.arange(1, 256)))


def rgb2hsv_u8(img):
    """cv2.cvtColor(uint8, COLOR_RGB2HSV): H in 0..179 (color_hsv.simd.hpp RGB2HSV_b, table division with 12-bit shift)."""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    vr = v == r
    vg = v == g
    s = (diff * _SDIV[v] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    hh = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    hh = (hh * _HDIV[diff] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    hh = np.where(hh < 0, hh + 180, hh)
    return np.stack([hh, s, v], -1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def hsv2rgb_u8(hsv):
    """cv2.cvtColor(uint8, COLOR_HSV2RGB): float32 sector arithmetic of HSV2RGB_native on (h, s/255, v/255), `saturate_cast<uchar>(x*255)`."""
    f = np.float32
    h = hsv[..., 0].astype(f) * f(6.0 / 180.0)
    s = hsv[..., 1].astype(f) * f(1.0 / 255.0)
    v = hsv[..., 2].astype(f) * f(1.0 / 255.0)
    h = np.where(h >= f(6), h - f(6), h)
    sector = np.floor(h).astype(np.int64)
    hf = (h - sector.astype(f)).astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hf = np.where(bad, f(0), hf)
    one = f(1)
    tab = np.stack([v, (v * (one - s)).astype(f), (v * (one - (s * hf).astype(f))).astype(f), (v * (one - (s * (one - hf)).astype(f))).astype(f)], -1)
    idx = _SECTOR[sector]                                     # [..., 3] -> (b, g, r) table slots
    bgr = np.take_along_axis(tab, idx, -1)
    grey = (hsv[..., 1] == 0)[..., None]
    bgr = np.where(grey, v[..., None], bgr)
    rgb = bgr[..., ::-1]
    return np.clip(_cv_round((rgb * f(255.0)).astype(f)), 0, 255).astype(np.uint8)


def add_to_hue(img, hue):
    """augs.py:62-75: RGB -> HSV (8-bit, H 0..179), `hsv[..., 0] = (hsv[..., 0] + hue) % 180` (float, stored back truncated), HSV -> RGB."""
    hsv = rgb2hsv_u8(img)
    hsv[..., 0] = ((hsv[..., 0].astype(np.float64) + hue) % 180).astype(np.uint8)
    return hsv2rgb_u8(hsv)


def rgb2gray_u8(img):
    """cv2.cvtColor(uint8, COLOR_RGB2GRAY): (4899 R + 9617 G + 1868 B + 2^13) >> 14."""
    a = img.astype(np.int64)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def add_to_saturation(img, value):
    """augs.py:79-87 with `value` = 1 + the uniform draw."""
    gray = rgb2gray_u8(img)
    ret = img * value + (gray * (1 - value))[:, :, np.newaxis]
    return np.clip(ret, 0, 255).astype(np.uint8)


def add_to_contrast(img, value):
    """augs.py:91-99: the function clips and returns `img`, not the contrast-adjusted array -- identity on uint8 input."""
    return np.clip(img, 0, 255).astype(np.uint8)


def add_to_brightness(img, value):
    """augs.py:103-109."""
    return np.clip(img + value, 0, 255).astype(np.uint8)


COLOUR_OPS = {0: add_to_hue, 1: add_to_saturation, 2: add_to_brightness, 3: add_to_contrast}


def input_augment(img, kind, p0, p1, noise, order, hue, sat, bright, contrast):
    """kind 0 = Gaussian blur (kx = p0, ky = p1), 1 = median blur (k = p0), 2 = additive noise (`noise` array), 3 = none;
    then the four colour ops in `order` (a permutation of 0..3: hue, saturation, brightness, contrast)."""
    if kind == 0:
        img = gaussian_blur(img, int(p0), int(p1))
    elif kind == 1:
        img = median_blur(img, int(p0))
    elif kind == 2:
        img = additive_noise(img, noise)
    vals = {0: hue, 1: sat, 2: bright, 3: contrast}
    for op in order:
        img = COLOUR_OPS[int(op)](img, vals[int(op)])
    return img
