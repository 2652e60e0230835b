"""Synthetic workloads of the shapes BASELINE.json names (pure numpy, no torch).

* `synth_tiles`     -- uint8 RGB network input patches (seeded noise).
* `synth_pred_maps` -- structured `[type?, p, h, v]` prediction maps the way a trained
  HoVer-Net emits them: random, partly overlapping elliptical nuclei, `p` high inside,
  `h`/`v` the per-instance horizontal/vertical distance maps in [-1, 1] (semantics of
  /root/reference/models/hovernet/targets.py:63-93), plus noise.  These give the
  post-processing a realistic instance load independent of any weights.
"""
import numpy as np


def synth_tiles(n, size=270, seed=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(n, size, size, 3), dtype=np.uint8)


def _one_map(rng, H, W, nr_types, k_lo, k_hi, noise):
    inst = np.zeros((H, W), np.int32)
    k = int(rng.integers(k_lo, k_hi + 1))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    for i in range(1, k + 1):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        ra, rb = rng.uniform(4, 12), rng.uniform(4, 12)
        th = rng.uniform(0, np.pi)
        c, s = np.cos(th), np.sin(th)
        u = (xx - cx) * c + (yy - cy) * s
        v = -(xx - cx) * s + (yy - cy) * c
        m = (u / ra) ** 2 + (v / rb) ** 2 <= 1.0
        inst[m] = i  # later nuclei overwrite earlier ones -> touching instances
    hmap = np.zeros((H, W), np.float64)
    vmap = np.zeros((H, W), np.float64)
    tmap = np.zeros((H, W), np.float64)
    for i in range(1, k + 1):
        m = inst == i
        if not m.any():
            continue
        ys, xs = np.nonzero(m)
        dx = xs - np.round(xs.mean())
        dy = ys - np.round(ys.mean())
        if dx.min() < 0:
            dx[dx < 0] /= -dx.min()
        if dx.max() > 0:
            dx[dx > 0] /= dx.max()
        if dy.min() < 0:
            dy[dy < 0] /= -dy.min()
        if dy.max() > 0:
            dy[dy > 0] /= dy.max()
        hmap[ys, xs] = dx
        vmap[ys, xs] = dy
        if nr_types:
            tmap[ys, xs] = 1 + (i % (nr_types - 1))
    mask = inst > 0
    p = 0.9 * mask + 0.05 + rng.normal(0, noise, (H, W))
    hmap = hmap + rng.normal(0, noise, (H, W))
    vmap = vmap + rng.normal(0, noise, (H, W))
    chans = [p, hmap, vmap]
    if nr_types:
        flip = rng.uniform(size=(H, W)) < 0.05  # some mis-typed pixels
        tmap = np.where(flip, rng.integers(0, nr_types, (H, W)), tmap)
        chans = [tmap] + chans
    return np.stack(chans, -1).astype(np.float32), inst


def synth_pred_maps(n, H=80, W=80, nr_types=None, seed=0, k_lo=5, k_hi=40, noise=0.02):
    """-> (pred [n,H,W,3|4] float32, painted instance maps [n,H,W] int32)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    scale = (H * W) / 6400.0
    lo, hi = max(1, int(k_lo * scale)), max(1, int(k_hi * scale))
    maps, insts = zip(*[_one_map(rng, H, W, nr_types, lo, hi, noise) for _ in range(n)])
    return np.stack(maps), np.stack(insts)


_GAINS = (("conv3.weight", 0.35), ("shortcut.weight", 0.6), ("conv_bot.weight", 0.6), ("conva.weight", 0.6),
          ("convf.weight", 0.6), ("u0.conv.weight", 1.5))


def _conv_gain(key):
    for suffix, g in _GAINS:
        if key.endswith(suffix):
            return g
    return 1.0


_SD_CACHE = {}


def synth_state_dict(mode="original", nr_types=None, seed=0, as_torch=True):
    """Seeded, platform-independent (numpy PCG64) checkpoint in the reference's key
    format: Kaiming fan_out convs like `Net.weights_init`
    (/root/reference/models/hovernet/net_utils.py:18-32) but with NON-trivial BatchNorm
    affine + running statistics so that BN folding / prologues are really exercised."""
    from .arch import param_table

    key_ = (mode, nr_types, int(seed))
    if key_ in _SD_CACHE:                       # (the last few checkpoints drawn in this process: 37 - 55 M normals take ~2 s; fresh copies are handed out)
        sd = {k: v.copy() for k, v in _SD_CACHE[key_].items()}
        if as_torch:
            import torch

            sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
        return sd
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, (kind, shape) in param_table(mode, nr_types).items():
        if kind == "conv":
            # He-normal over fan_in, with the residual / skip-summing convs damped so that
            # activations (and the logits) stay O(1..10) through the ~50 layers; absolute
            # tolerances on logits are only meaningful at that scale.
            fan_in = shape[1] * shape[2] * shape[3]
            a = rng.normal(0.0, np.sqrt(2.0 / fan_in) * _conv_gain(key), shape)
        elif kind == "bias":
            a = rng.normal(0.0, 0.1, shape)
        elif kind == "bn_w":
            a = rng.uniform(0.6, 1.4, shape)
        elif kind == "bn_b":
            a = rng.normal(0.0, 0.2, shape)
        elif kind == "bn_rm":
            a = rng.normal(0.0, 0.2, shape)
        elif kind == "bn_rv":
            a = rng.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            a = np.array(0, np.int64)
        elif kind == "ones":
            a = np.ones(shape)
        else:
            raise KeyError(kind)
        sd[key] = a.astype(np.int64 if kind == "bn_nbt" else np.float32)
    _SD_CACHE[key_] = {k: v.copy() for k, v in sd.items()}
    while len(_SD_CACHE) > 6:
        _SD_CACHE.pop(next(iter(_SD_CACHE)))
    if as_torch:
        import torch

        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    return sd


def synth_train_batch(n, mode="original", nr_types=None, seed=0):
    """Synthetic training batch in the reference's loader format (dataloader/train_loader.py:109-137 feeds
    run_desc.train_step): img uint8 [n,S,S,3], np_map int64 [n,h,w] in {0,1}, hv_map float32 [n,h,w,2] in [-1,1],
    tp_map int64 [n,h,w] in [0,nr_types) (only with nr_types).  Nuclei are painted ellipses (synth_pred_maps)."""
    size, out = (270, 80) if mode == "original" else (256, 164)
    pm, inst = synth_pred_maps(n, out, out, nr_types, seed=seed + 17, noise=0.0)
    c0 = 0 if nr_types is None else 1
    batch = {"img": synth_tiles(n, size, seed=seed),
             "np_map": (inst > 0).astype(np.int64),
             "hv_map": np.ascontiguousarray(np.clip(pm[..., c0 + 1:c0 + 3], -1.0, 1.0), np.float32)}
    if nr_types is not None:
        batch["tp_map"] = np.clip(pm[..., 0].round().astype(np.int64), 0, nr_types - 1) * (inst > 0)
    return batch
