"""CPU: the host contour tracer (hover_net_amd/csrc/hvn_contour.cpp) on hand-made and random shapes.
OpenCV is not on the box, so the checks are the published conventions of cv2.findContours
(outer border, CHAIN_APPROX_SIMPLE) plus an independent definition of the border pixel SET."""
import numpy as np
import pytest
from scipy import ndimage

from hover_net_amd import post_proc as PP


def _recs(inst):
    recs = []
    for l in np.unique(inst):
        if l <= 0:
            continue
        ys, xs = np.nonzero(inst == l)
        recs.append((l, len(ys), ys.min(), ys.max() + 1, xs.min(), xs.max() + 1, 0., 0., -1, 0))
    return np.array(recs, dtype=PP._REC_DTYPE)


def _expand(pts):
    """Corner list -> every pixel on the closed 8-connected polyline."""
    out = []
    n = len(pts)
    for i in range(n):
        (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % n]
        steps = max(abs(x1 - x0), abs(y1 - y0))
        assert abs(x1 - x0) in (0, steps) and abs(y1 - y0) in (0, steps), "segments are horizontal, vertical or diagonal"
        for t in range(max(steps, 1)):
            out.append((x0 + (x1 - x0) * t // max(steps, 1), y0 + (y1 - y0) * t // max(steps, 1)))
    return out


def test_known_conventions():
    inst = np.zeros((8, 9), np.int32)
    inst[1:4, 2:5] = 1          # 3x3 square: OpenCV gives the 4 corners, starting top-left, going down first
    inst[5, 1] = 2              # isolated pixel: one point
    c = PP.trace_contours(inst, _recs(inst))
    assert c[1].tolist() == [[2, 1], [2, 3], [4, 3], [4, 1]]
    assert c[2].tolist() == [[1, 5]]
    line = np.zeros((5, 9), np.int32)
    line[2, 1:8] = 7            # 1-px line: two end points -> the reference drops it (< 3 points)
    c = PP.trace_contours(line, _recs(line))
    assert c[7].tolist() == [[1, 2], [7, 2]]
    info = PP.records_to_dict(_recs(line), None, line)
    assert info == {}


@pytest.mark.parametrize("seed", range(6))
def test_border_pixel_set_on_random_blobs(seed):
    rng = np.random.default_rng(seed)
    a = ndimage.gaussian_filter(rng.normal(size=(60, 70)), 3) > 0.02
    lab, n = ndimage.label(a, structure=np.ones((3, 3)))   # 8-connected components, like findContours' foreground
    inst = lab.astype(np.int32)
    cont = PP.trace_contours(inst, _recs(inst))
    assert sorted(cont) == list(range(1, n + 1))
    for l, pts in cont.items():
        m = inst == l
        # outer border = foreground pixels 4-adjacent to the background component that reaches the crop's outside
        filled = ndimage.binary_fill_holes(m)
        outer_bg = ~filled
        pad = np.pad(outer_bg, 1, constant_values=True)
        touch = pad[:-2, 1:-1] | pad[2:, 1:-1] | pad[1:-1, :-2] | pad[1:-1, 2:]
        want = set(zip(*np.nonzero(m & touch)[::-1]))
        poly = _expand([tuple(p) for p in pts.tolist()])
        assert set(poly) == want, l
        ys, xs = np.nonzero(m)
        y0 = ys.min()
        assert tuple(pts[0]) == (xs[ys == y0].min(), y0)      # starts at the first pixel in raster order
