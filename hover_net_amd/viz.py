"""Overlay writer of the tile manager (`misc/viz_utils.py:94-125` `visualize_instances_dict`): contours of every instance in
its type colour (or a random colour without a type table) and an optional centroid dot, drawn into a copy of the image.
Host glue behind the hot path; numpy only (cv2 is absent from this image), so the line rasterisation follows
`cv2.drawContours(thickness=2)` in intent -- closed polygon through the contour vertices, 2 px wide -- not bit for bit."""
import colorsys
import random
import struct
import zlib

import numpy as np


def random_colors(n, bright=True):
    """misc/viz_utils.py:22-33: n equally spaced hues, shuffled (python `random`: not reproducible by design)."""
    brightness = 1.0 if bright else 0.7
    colors = [colorsys.hsv_to_rgb(i / n, 1, brightness) for i in range(n)]
    random.shuffle(colors)
    return colors


def _segment_pixels(p0, p1):
    """Integer pixels of the straight segment p0 -> p1 (both inclusive), one per step along the major axis."""
    n = int(max(abs(p1[0] - p0[0]), abs(p1[1] - p0[1])))
    if n == 0:
        return np.array([[p0[0], p0[1]]], np.int64)
    t = np.arange(n + 1, dtype=np.float64) / n
    xs = np.floor(p0[0] + (p1[0] - p0[0]) * t + 0.5).astype(np.int64)
    ys = np.floor(p0[1] + (p1[1] - p0[1]) * t + 0.5).astype(np.int64)
    return np.stack([xs, ys], 1)


def draw_contour(img, contour, colour, thickness=2):
    """Closed polyline through `contour` (int [K,2] as (x, y)) into `img` (uint8 [H,W,3], modified in place)."""
    h, w = img.shape[:2]
    contour = np.asarray(contour, np.int64).reshape(-1, 2)
    if contour.shape[0] == 0:
        return img
    pts = [_segment_pixels(contour[i], contour[(i + 1) % contour.shape[0]]) for i in range(contour.shape[0])]
    pts = np.concatenate(pts, 0)
    lo, hi = -((thickness - 1) // 2), thickness // 2          # thickness 2 -> offsets {0, +1}; 3 -> {-1, 0, +1}
    offs = np.array([(dx, dy) for dy in range(lo, hi + 1) for dx in range(lo, hi + 1)], np.int64)
    allp = (pts[:, None, :] + offs[None]).reshape(-1, 2)
    ok = (allp[:, 0] >= 0) & (allp[:, 0] < w) & (allp[:, 1] >= 0) & (allp[:, 1] < h)
    allp = allp[ok]
    img[allp[:, 1], allp[:, 0]] = np.asarray(colour, np.uint8)
    return img


def draw_centroid_dot(img, centre, radius=3, colour=(255, 0, 0)):
    """Filled disc (`cv2.circle(..., 3, (255, 0, 0), -1)`, misc/viz_utils.py:121-124)."""
    h, w = img.shape[:2]
    cx, cy = int(centre[0]), int(centre[1])
    ys, xs = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    keep = xs * xs + ys * ys <= radius * radius
    px, py = xs[keep] + cx, ys[keep] + cy
    ok = (px >= 0) & (px < w) & (py >= 0) & (py < h)
    img[py[ok], px[ok]] = np.asarray(colour, np.uint8)
    return img


def visualize_instances_dict(input_image, inst_dict, draw_dot=False, type_colour=None, line_thickness=2):  # noqa: A002
    """Same signature as misc/viz_utils.py:94-96.  `type_colour`: {type_id: (name, (r, g, b))}."""
    overlay = np.array(input_image, copy=True)
    rng_colours = (np.array(random_colors(len(inst_dict))) * 255).astype(np.uint8) if len(inst_dict) else np.zeros((0, 3), np.uint8)
    for idx, (_inst_id, info) in enumerate(inst_dict.items()):
        if "type" in info and type_colour is not None and info["type"] in type_colour:
            colour = type_colour[info["type"]][1]
        else:
            colour = rng_colours[idx].tolist()
        if info.get("contour") is not None:
            draw_contour(overlay, info["contour"], colour, line_thickness)
        if draw_dot:
            draw_centroid_dot(overlay, info["centroid"])
    return overlay


def save_png(path, rgb):
    """uint8 [H,W,3] -> PNG.  PIL when present, else a minimal zlib writer (8-bit RGB, filter 0)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    try:
        from PIL import Image

        Image.fromarray(rgb).save(path)
        return
    except ImportError:
        pass
    h, w = rgb.shape[:2]
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
