"""cv2.moments / cv2.findContours for the stand-in cv2 module -- TEST INFRASTRUCTURE ONLY.

Independent of the product (hover_net_amd/csrc/hvn_contour.cpp) and of oracle/hvn_oracle.c: plain
python / numpy, written as a separate restatement so that the reference's own, unmodified
`process()` (/root/reference/models/hovernet/post_proc.py:94-186) can run under the secondary
interpreter and make the `tests/golden/proc_*.npz` fixtures.

What is restated (OpenCV 4.3, not on the box, so "from the published algorithm"):

* `findContours(img, RETR_TREE, CHAIN_APPROX_SIMPLE)`: Suzuki & Abe 1985 border following the way
  imgproc/src/contours.cpp does it -- the image is binarised and framed with one background pixel; rows
  are scanned left to right; an outer border starts where 0 -> 1(unvisited), a hole border where
  (>= 1) -> 0; the 8 neighbours are numbered counter-clockwise from east; the first neighbour is searched
  clockwise from west (outer) / east (hole), the following ones counter-clockwise from the one after
  the pixel we came from; visited border pixels are re-valued nbd, or nbd - 128 (signed char nbd | 0x80) when the border leaves
  them to the right ("right bound"), which is what keeps the scan from starting the same border twice;
  CHAIN_APPROX_SIMPLE keeps a point only where the chain code changes (and always for a one-pixel
  border).  The hierarchy (parent = last border met on the row, or its parent when both are of the same
  kind) is kept because the LIST ORDER depends on it: every finished border is pushed at the FRONT of its
  parent's child list and the list is emitted in pre-order, so `contours[0]` is the outer border of the
  top-level component that was found LAST.
* `moments(img)`: raw spatial moments of a uint8 image with the pixel values as weights, in double.
"""
import numpy as np

# chain code -> (dx, dy): 0 = E, then counter-clockwise on the screen (y down): NE, N, NW, W, SW, S, SE
_CODE = ((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))


class _Border:
    __slots__ = ("is_hole", "points", "parent", "children", "origin", "rect", "lval")

    def __init__(self, is_hole, parent, origin, lval):
        self.is_hole, self.parent, self.origin, self.lval = is_hole, parent, origin, lval
        self.points, self.children, self.rect = [], [], None


def _follow(a, y0, x0, is_hole, nbd, border):
    """Follows one border from (y0, x0) on the framed label image `a` (in place re-valuing), appends
    the CHAIN_APPROX_SIMPLE points (frame offset removed) to border.points."""
    pts = border.points
    s_end = s = 0 if is_hole else 4
    xs, ys = [x0], [y0]
    while True:                       # first neighbour: clockwise
        s = (s - 1) & 7
        dx, dy = _CODE[s]
        if a[y0 + dy, x0 + dx] != 0 or s == s_end:
            break
    if a[y0 + _CODE[s][1], x0 + _CODE[s][0]] == 0:          # came back to s_end: one isolated pixel
        a[y0, x0] = nbd - 128
        pts.append((x0 - 1, y0 - 1))
    else:
        y1, x1 = y0 + _CODE[s][1], x0 + _CODE[s][0]
        y3, x3 = y0, x0
        prev_s = s ^ 4
        while True:
            s_end = s
            while True:               # next neighbour: counter-clockwise, starting after where we came from
                s += 1
                dx, dy = _CODE[s & 7]
                if a[y3 + dy, x3 + dx] != 0:
                    break
            y4, x4 = y3 + dy, x3 + dx
            s &= 7
            if (s - 1) % (1 << 32) < s_end:      # unsigned compare: the east neighbour was examined and is background
                a[y3, x3] = nbd - 128
            elif a[y3, x3] == 1:
                a[y3, x3] = nbd
            if s != prev_s:
                pts.append((x3 - 1, y3 - 1))
                prev_s = s
            xs.append(x3)
            ys.append(y3)
            if y4 == y0 and x4 == x0 and y3 == y1 and x3 == x1:
                break
            y3, x3 = y4, x4
            s = (s + 4) & 7
    border.rect = (min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1)


def find_contours_tree(img):
    """-> (list of int32 [K,1,2] arrays in cv2's RETR_TREE list order, hierarchy int32 [1,n,4])."""
    src = np.asarray(img)
    assert src.ndim == 2
    H, W = src.shape
    a = np.zeros((H + 2, W + 2), np.int32)
    a[1:-1, 1:-1] = src != 0
    frame = _Border(True, None, None, 0)          # the image frame acts as a hole (its children are outer borders)
    table = {}                                    # lval -> borders drawn with that value, newest first
    nbd = 2
    for y in range(1, H + 1):
        lnbd_x = 0
        prev = 0
        row = a[y]
        x = 1
        while x <= W:           # the scanner excludes the frame's last column and row
            p = int(row[x])
            if p == prev:
                x += 1
                continue
            is_hole = 0
            start = True
            if not (prev == 0 and p == 1):
                if p != 0 or prev < 1:
                    start = False
                else:
                    if prev & -2:
                        lnbd_x = x - 1
                    is_hole = 1
            if start:
                if lnbd_x <= 0:
                    par = frame
                else:
                    lval = int(a[y, lnbd_x]) & 0x7F
                    cands = [b for b in table.get(lval, [])
                             if 0 <= lnbd_x - b.rect[0] < b.rect[2] and 0 <= y - b.rect[1] < b.rect[3]]
                    if len(cands) != 1:
                        # more than 126 borders in one image: OpenCV re-traces candidates to tell them apart; an
                        # instance crop never gets there
                        raise NotImplementedError("ambiguous border value (more than 126 borders)")
                    par = cands[0]
                    if par.is_hole == bool(is_hole):
                        par = par.parent if par.parent is not None else frame
                    assert par.is_hole != bool(is_hole)
                lnbd_x = x - is_hole
                b = _Border(bool(is_hole), par, (x - is_hole, y), nbd)
                lval = nbd
                nbd = (nbd + 1) & 127
                if nbd == 0:
                    nbd = 3
                _follow(a, y, x - is_hole, is_hole, lval, b)
                table.setdefault(lval, []).insert(0, b)
                par.children.insert(0, b)         # newest first
                # the scan resumes behind the start pixel with its NEW value as `prev`
                prev = int(row[x])
                x += 1
                continue
            prev = p
            if prev & -2:
                lnbd_x = x
            x += 1
    order = []

    def walk(b):
        order.append(b)
        for c in b.children:
            walk(c)

    for c in frame.children:
        walk(c)
    contours = [np.asarray(b.points, np.int32).reshape(-1, 1, 2) for b in order]
    idx = {id(b): i for i, b in enumerate(order)}
    hier = np.full((1, len(order), 4), -1, np.int32)
    for i, b in enumerate(order):
        sib = (b.parent.children if b.parent is not None else frame.children)
        k = sib.index(b)
        hier[0, i, 0] = idx[id(sib[k + 1])] if k + 1 < len(sib) else -1
        hier[0, i, 1] = idx[id(sib[k - 1])] if k > 0 else -1
        hier[0, i, 2] = idx[id(b.children[0])] if b.children else -1
        hier[0, i, 3] = idx[id(b.parent)] if b.parent is not frame else -1
    return contours, hier


def moments(img):
    a = np.asarray(img)
    assert a.ndim == 2 and a.dtype == np.uint8
    w = a.astype(np.float64)
    y, x = np.mgrid[0:a.shape[0], 0:a.shape[1]].astype(np.float64)
    m = {}
    for i in range(4):
        for j in range(4 - i):
            m["m%d%d" % (i, j)] = float((w * x ** i * y ** j).sum())
    return m
