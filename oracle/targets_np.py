"""numpy restatement of the reference's training-target generation -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/models/hovernet/targets.py:17-116 (`gen_instance_hv_map`, `gen_targets`) with its helpers
`fix_mirror_padding` (dataloader/augs.py:18-32: connected components of every instance id, scipy's default
4-connectivity), `cropping_center` (misc/utils.py:32-52), `get_bounding_box` (misc/utils.py:18-28) and
skimage's `remove_small_objects(min_size=30)` on the cropped label map (objects with FEWER than 30 pixels inside
the crop are dropped).  Pinned to goldens made by the reference itself (oracle/make_golden_targets.py,
tests/test_oracle_targets.py).

Reference behaviour kept on purpose: the bounding box is widened by 2 px WITHOUT clamping, so an instance whose box
starts within 2 px of the top / left border produces a negative slice start, i.e. an empty crop, and is skipped
(its HV target stays 0, targets.py:50-56); the box end simply truncates at the image border.
"""
import numpy as np
from scipy import ndimage


def split_instances(ann):
    """fix_mirror_padding as a partition: label map where every 4-connected component of equal id is one instance."""
    out = np.zeros(ann.shape, np.int32)
    nxt = 0
    for inst_id in np.unique(ann):
        if inst_id == 0:
            continue
        lab, n = ndimage.label(ann == inst_id)
        out[lab > 0] = lab[lab > 0] + nxt
        nxt += n
    return out


def gen_instance_hv_map(ann, crop_shape):
    comp = split_instances(ann)
    h, w = ann.shape[:2]
    h0, w0 = int((h - crop_shape[0]) * 0.5), int((w - crop_shape[1]) * 0.5)
    crop = comp[h0:h0 + crop_shape[0], w0:w0 + crop_shape[1]]
    x_map = np.zeros((h, w), np.float32)
    y_map = np.zeros((h, w), np.float32)
    ids, counts = np.unique(crop, return_counts=True)
    for inst_id, cnt in zip(ids, counts):
        if inst_id == 0 or cnt < 30:                      # remove_small_objects(min_size=30) on the crop
            continue
        m = comp == inst_id
        rows, cols = np.where(m.any(1))[0], np.where(m.any(0))[0]
        r0, r1, c0, c1 = rows[0] - 2, rows[-1] + 3, cols[0] - 2, cols[-1] + 3
        if r0 < 0 or c0 < 0:
            # negative slice start: python wraps it around; the crop is empty unless the instance also reaches the far
            # border, in which case the reference reads a wrapped window -- reproduce by literal slicing
            box = m[r0:r1, c0:c1]
            if box.shape[0] < 2 or box.shape[1] < 2:
                continue
            raise NotImplementedError("instance spanning the image from within 2 px of the near border: wrapped slice")
        box = m[r0:r1, c0:c1]
        if box.shape[0] < 2 or box.shape[1] < 2:
            continue
        com = ndimage.center_of_mass(box.astype(np.uint8))
        cr, cc = int(com[0] + 0.5), int(com[1] + 0.5)
        xs = np.arange(1, box.shape[1] + 1) - cc
        ys = np.arange(1, box.shape[0] + 1) - cr
        gx, gy = np.meshgrid(xs, ys)
        gx = np.where(box, gx, 0).astype(np.float32)
        gy = np.where(box, gy, 0).astype(np.float32)
        for g in (gx, gy):
            if g.min() < 0:
                g[g < 0] /= -g[g < 0].min()
            if g.max() > 0:
                g[g > 0] /= g[g > 0].max()
        x_map[r0:r1, c0:c1][box] = gx[box]
        y_map[r0:r1, c0:c1][box] = gy[box]
    return np.dstack([x_map, y_map])


def gen_targets(ann, crop_shape):
    """-> dict(hv_map float32 [ch,cw,2], np_map [ch,cw] in {0,1}) like targets.py:100-116."""
    hv = gen_instance_hv_map(ann, crop_shape)
    h, w = ann.shape[:2]
    h0, w0 = int((h - crop_shape[0]) * 0.5), int((w - crop_shape[1]) * 0.5)
    sl = (slice(h0, h0 + crop_shape[0]), slice(w0, w0 + crop_shape[1]))
    return {"hv_map": hv[sl], "np_map": (ann[sl] > 0).astype(ann.dtype)}


def synth_ann(rng, size=270, n_inst=60, mirror=True):
    """Synthetic instance-id map: ellipses (some overlapping the border, some tiny), plus mirrored duplicates that
    share an id with a disconnected twin (what the shape augmentation's reflect padding produces)."""
    ann = np.zeros((size, size), np.int32)
    yy, xx = np.mgrid[0:size, 0:size]
    for i in range(1, n_inst + 1):
        cy, cx = rng.integers(-5, size + 5, 2)
        ry, rx = rng.integers(2, 14, 2)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        ann[(u / rx) ** 2 + (v / ry) ** 2 <= 1.0] = i
    if mirror:      # reflect a band around the centre crop's edge so that some ids appear twice
        k = size // 2 - 20
        ann[:, k - 12:k] = ann[:, k:k + 12][:, ::-1]
    return ann
