"""CPU: sanity of the augmentation oracle (oracle/augment_np.py) against independent references that ARE available here --
numpy's own rot90 / flips, python's colorsys for the HSV conversions (to the 8-bit rounding), exact identities.  imgaug and cv2
themselves are absent: the oracle's header says "parity unpinned" for their sub-pixel / rounding conventions."""
import colorsys

import numpy as np

from oracle import augment_np as A


def _img(seed=0, shape=(40, 50)):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, shape + (3,), dtype=np.uint8), rng.integers(0, 5, shape + (2,)).astype(np.int32)


def test_shape_identity_crop_flips_and_quarter_turn():
    img, ann = _img()
    o, a = A.shape_augment(img, ann, np.eye(3), (20, 30), False, False)
    assert np.array_equal(o, img[10:30, 10:40]) and np.array_equal(a, ann[10:30, 10:40])      # cropping_center
    o2, a2 = A.shape_augment(img, ann, np.eye(3), (20, 30), True, True)
    assert np.array_equal(o2, o[::-1, ::-1]) and np.array_equal(a2, a[::-1, ::-1])
    sq = np.random.default_rng(1).integers(0, 256, (31, 31, 3), dtype=np.uint8)
    fwd = A.affine_matrix(31, 31, (1, 1), (0, 0), 0, 90)
    o3, _ = A.shape_augment(sq, np.zeros((31, 31, 1), np.int32), np.linalg.inv(fwd), (31, 31), False, False)
    assert np.array_equal(o3, np.rot90(sq, 3))                # +90 degrees with y pointing down = clockwise
    # translation moves content by whole pixels; what comes from outside is 0
    fwd = A.affine_matrix(31, 31, (1, 1), (3, -2), 0, 0)
    o4, _ = A.shape_augment(sq, np.zeros((31, 31, 1), np.int32), np.linalg.inv(fwd), (31, 31), False, False)
    assert np.array_equal(o4[:-2, 3:], sq[2:, :-3]) and not o4[-2:].any() and not o4[:, :3].any()
    # scale 2 about the centre: nearest-neighbour pixel doubling around the centre pixel
    fwd = A.affine_matrix(31, 31, (2, 2), (0, 0), 0, 0)
    o5, _ = A.shape_augment(sq, np.zeros((31, 31, 1), np.int32), np.linalg.inv(fwd), (31, 31), False, False)
    assert np.array_equal(o5[15, 15], sq[15, 15]) and np.array_equal(o5[15, 18], sq[15, 17]) and np.array_equal(o5[11, 15], sq[13, 15])


def test_hsv_conversions_agree_with_colorsys_to_8bit_rounding():
    img, _ = _img(2)
    hsv = A.rgb2hsv_u8(img)
    ref = np.array([[colorsys.rgb_to_hsv(*(p / 255.0)) for p in row] for row in img])
    dh = np.abs(((hsv[..., 0] - ref[..., 0] * 180 + 90) % 180) - 90)
    sat = ref[..., 1] > 0.05                                   # hue of near-grey pixels is ill-conditioned
    assert dh[sat].max() <= 1.0 and np.abs(hsv[..., 1] - ref[..., 1] * 255).max() <= 1.0 and np.array_equal(hsv[..., 2], img.max(-1))
    back = A.hsv2rgb_u8(hsv)
    ref_back = np.array([[colorsys.hsv_to_rgb(h / 180.0, s / 255.0, v / 255.0) for h, s, v in row] for row in hsv]) * 255
    assert np.abs(back - ref_back).max() <= 0.5 + 1e-3          # exact HSV -> RGB of the quantised triple, rounded
    grey = np.full((4, 4, 3), 90, np.uint8)
    assert np.array_equal(A.rgb2hsv_u8(grey)[..., :2], np.zeros((4, 4, 2))) and np.array_equal(A.hsv2rgb_u8(A.rgb2hsv_u8(grey)), grey)
    assert np.abs(A.rgb2gray_u8(img) - img @ np.array([0.299, 0.587, 0.114])).max() <= 0.51


def test_colour_ops_identities_and_reference_quirks():
    img, _ = _img(3)
    assert np.array_equal(A.add_to_saturation(img, 1.0), img) and np.array_equal(A.add_to_brightness(img, 0.0), img)
    assert np.array_equal(A.add_to_contrast(img, 0.75), img)   # augs.py:96-97: the contrast op returns its input
    assert np.array_equal(A.add_to_saturation(img, 0.0), np.repeat(A.rgb2gray_u8(img)[..., None], 3, -1))
    b = A.add_to_brightness(img, 300.0)
    assert b.min() == 255 and A.add_to_brightness(img, -300.0).max() == 0
    assert np.array_equal(A.add_to_brightness(img, 10.7), np.clip(img.astype(int) + 10, 0, 255))      # truncation, not rounding
    h0 = A.rgb2hsv_u8(img)
    h1 = A.rgb2hsv_u8(A.add_to_hue(img, 7.0))
    satur = h0[..., 1] > 60
    d = (h1[..., 0].astype(int) - h0[..., 0].astype(int)) % 180
    assert np.median(d[satur]) == 7


def test_blurs():
    img, _ = _img(4)
    assert np.array_equal(A.gaussian_blur(img, 1, 1), img) and np.array_equal(A.median_blur(img, 1), img)
    flat = np.full((9, 9, 3), 77, np.uint8)
    assert np.array_equal(A.gaussian_blur(flat, 5, 3), flat) and np.array_equal(A.median_blur(flat, 5), flat)
    f = img.astype(np.float64)
    p = np.pad(f, ((0, 0), (1, 1), (0, 0)), "edge")
    want = np.floor((p[:, :-2] + 2 * p[:, 1:-1] + p[:, 2:]) / 4 + 0.5)
    assert np.array_equal(A.gaussian_blur(img, 3, 1), want.astype(np.uint8))
    spike = flat.copy()
    spike[4, 4] = 255
    assert np.array_equal(A.median_blur(spike, 3), flat)
    assert np.array_equal(A.additive_noise(flat, np.full((9, 9, 1), 2.5, np.float32)), flat + 2)       # rint: half to even
    assert A.additive_noise(flat, np.full((9, 9, 3), -500.0, np.float32)).max() == 0


def test_oracle_equals_the_references_own_functions():
    """tests/golden/augs.npz: outputs of the reference's unmodified dataloader/augs.py (oracle/make_golden_augs.py, cv2 entry points =
    this oracle's restatements) -- pins the glue: draw -> kernel size, float64 promotion, `% 180`, clip + truncation, the contrast no-op."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augs.npz"))
    img = g["img"]
    for k in range(6):
        assert np.array_equal(A.gaussian_blur(img[k], *g["gauss_k"][k]), g["gauss_out"][k])
        assert np.array_equal(A.add_to_hue(img[k], g["hue_val"][k]), g["hue_out"][k])
        assert np.array_equal(A.add_to_saturation(img[k], 1 + g["sat_val"][k]), g["sat_out"][k])
        assert np.array_equal(A.add_to_brightness(img[k], g["bright_val"][k]), g["bright_out"][k])
        assert np.array_equal(A.add_to_contrast(img[k], g["contrast_val"][k]), g["contrast_out"][k])
        assert np.array_equal(g["contrast_out"][k], img[k])
    for k in range(3):
        assert np.array_equal(A.median_blur(img[k], g["median_k"][k]), g["median_out"][k])


def test_affine_gather_agrees_with_scipy_affine_transform():
    """Independent check of the warp geometry: scipy.ndimage.affine_transform (order 0, constant 0) with the same destination -> source
    matrix in (row, col) order.  Random parameters never land on exact .5 source coordinates, where the two rounding rules could differ."""
    from scipy import ndimage

    rng = np.random.default_rng(9)
    img = rng.integers(1, 256, (64, 80, 3), dtype=np.uint8)
    ann = rng.integers(1, 9, (64, 80, 1)).astype(np.int32)
    for _ in range(6):
        fwd = A.affine_matrix(64, 80, rng.uniform(0.8, 1.2, 2), (rng.uniform(-3, 3), rng.uniform(-3, 3)), rng.uniform(-5, 5), rng.uniform(-179, 179))
        inv = np.linalg.inv(fwd)
        got_i, got_a = A.shape_augment(img, ann, inv, (64, 80), False, False)
        # (x, y) -> (row, col): swap the axes of the 2x2 part and of the offset
        m_rc = np.array([[inv[1, 1], inv[1, 0]], [inv[0, 1], inv[0, 0]]])
        off_rc = np.array([inv[1, 2], inv[0, 2]])
        for c in range(3):
            want = ndimage.affine_transform(img[..., c], m_rc, offset=off_rc, order=0, mode="constant", cval=0)
            both = (want != 0) & (got_i[..., c] != 0)             # scipy's "constant" mode also blanks the outer half-pixel band (source in (-0.5, 0))
            assert both.mean() > 0.4 and np.array_equal(want[both], got_i[..., c][both])
            assert ((want != 0) != (got_i[..., c] != 0)).mean() < 0.04 and not ((want != 0) & (got_i[..., c] == 0)).any()
        want_a = ndimage.affine_transform(ann[..., 0], m_rc, offset=off_rc, order=0, mode="constant", cval=0)
        both = (want_a != 0) & (got_a[..., 0] != 0)
        assert np.array_equal(want_a[both], got_a[..., 0][both])


def test_shape_augment_sampling_equals_scipy_affine_transform():
    """SURVEY 8f-4 / round-3 verdict item 8: the geometric half pinned against an independent library.  imgaug's `Affine(order=0,
    cval=0, backend="cv2")` (dataloader/train_loader.py:123-150) is `cv2.warpAffine(INTER_NEAREST, BORDER_CONSTANT)` over a matrix
    built about the image centre (w/2 - 0.5, h/2 - 0.5); neither library is on this box.  What CAN be pinned: for a given matrix, the
    oracle's (and therefore the kernel's: tests/test_gpu_augment.py, bit for bit) nearest-neighbour / constant-0 sampling + centre crop
    equals `scipy.ndimage.affine_transform(order=0, mode="grid-constant", cval=0)` -- the same "round the source coordinate to the
    nearest pixel, constant outside the image" rule -- on random scale / translate / shear / rotate draws from the reference's ranges.
    A pixel whose source coordinate lands within float rounding of x.5 may go either way (two summation orders): <= 3 per case.
    THE LIMIT, stated: imgaug's random stream, its shear parameterisation and cv2's 10-bit fixed-point coordinate rounding are not
    checked by anything here."""
    from scipy import ndimage

    rng = np.random.default_rng(7)
    h = w = 96
    out_hw = (64, 64)
    for case in range(12):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ann = rng.integers(0, 50, (h, w, 2)).astype(np.int32)
        m = A.affine_matrix(h, w, rng.uniform(0.8, 1.2, 2), rng.uniform(-0.01, 0.01, 2) * np.array([w, h]), rng.uniform(-5, 5), rng.uniform(-179, 179))
        inv = np.linalg.inv(m)
        got_img, got_ann = A.shape_augment(img, ann, inv, out_hw, False, False)
        # scipy works in (row, col): out[r, c] = in[M @ (r, c) + off]; the centre crop is an output offset
        y0, x0 = (h - out_hw[0]) // 2, (w - out_hw[1]) // 2
        mat = np.array([[inv[1, 1], inv[1, 0]], [inv[0, 1], inv[0, 0]]])
        off = np.array([inv[1, 2], inv[0, 2]]) + mat @ np.array([y0, x0], np.float64)
        bad = 0
        for c in range(3):
            want = ndimage.affine_transform(img[..., c], mat, offset=off, output_shape=out_hw, order=0, mode="grid-constant", cval=0)
            bad = max(bad, int((want != got_img[..., c]).sum()))
        for c in range(2):
            want = ndimage.affine_transform(ann[..., c], mat, offset=off, output_shape=out_hw, order=0, mode="grid-constant", cval=0)
            bad = max(bad, int((want != got_ann[..., c]).sum()))
        assert bad <= 3, (case, bad)
    # and the flips are numpy's, applied after the crop
    got_f, _ = A.shape_augment(img, ann, inv, out_hw, True, True)
    np.testing.assert_array_equal(got_f, got_img[::-1, ::-1])
