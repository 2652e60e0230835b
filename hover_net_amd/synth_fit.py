"""A 'trained-like' HoVer-Net checkpoint made on the GPU box with the repository's OWN trainer -- the synthetic stand-in for the
checkpoints nobody can download here (no network).

A random-init network emits noise maps (0 nuclei in 'original' mode, tile-filling blobs in 'fast' mode), so neither a tolerance
statement about the segmentation nor a benchmark of the instance separation means much on it.  `fit` paints H&E-like tiles (partly
touching elliptical nuclei, darker and bluer than a noisy pink background, tinted by nucleus type), derives the targets with the
product's own `gen_targets_device` (bit-exact with the reference's targets.py) and runs a few hundred steps of
`run_desc.train_step` (HIP training path, FusedAdam) -- /root/reference/models/hovernet/run_desc.py:12-109 is the step,
opt.py:47-51 the loss table.  Used by bench.py (the timed step segments the network's OWN output) and by the tolerance tests
(tests/fit_util.py re-exports this module).

`init="kaiming"` starts from the reference's own initialisation (`Net.weights_init`, net_utils.py:18-32: Kaiming-normal convs,
BatchNorm weight 1 / bias 0) -- UN-damped residual branches, so the pre-activation blocks' running sums grow from unit to unit as
they do in a real checkpoint (net_utils.py:250-266); `init="synth"` starts from `synth.synth_state_dict` (damped residual convs)."""
import numpy as np
import torch

# per-type tint added to the nucleus colour (type 0 = background; indices wrap for more types)
_TINTS = np.array([[0, 0, 0], [0, 0, 0], [40, -10, -30], [-30, 25, 10], [20, 30, -40], [-25, -20, 35], [35, 15, 25]], np.float32)


def painted_tiles(n, size, seed, k_lo=6, k_hi=22, nr_types=None):
    """-> (img uint8 [n,size,size,3], ann int32 [n,size,size] instance ids[, typ int32 [n,size,size] nucleus types 1..nr_types-1]).
    Nuclei: ellipses of radius 5..11 px."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    imgs = np.empty((n, size, size, 3), np.uint8)
    anns = np.zeros((n, size, size), np.int32)
    typs = np.zeros((n, size, size), np.int32)
    for t in range(n):
        img = np.array([228.0, 190.0, 214.0], np.float32)[None, None] + rng.normal(0, 6.0, (size, size, 3)).astype(np.float32)
        # slow background shading
        img += (12.0 * np.sin(xx / 37.0 + rng.uniform(0, 6)) * np.cos(yy / 41.0 + rng.uniform(0, 6)))[..., None]
        k = int(rng.integers(k_lo, k_hi + 1))
        for i in range(1, k + 1):
            cy, cx = rng.uniform(8, size - 8, 2)
            a, b = rng.uniform(5, 11, 2)
            th = rng.uniform(0, np.pi)
            u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
            v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
            d = (u / a) ** 2 + (v / b) ** 2
            m = (d <= 1.0) & (anns[t] == 0)
            anns[t][m] = i
            col = np.array([92.0, 60.0, 150.0], np.float32) + rng.normal(0, 10.0, 3).astype(np.float32)
            if nr_types is not None:
                ty = int(rng.integers(1, nr_types))
                typs[t][m] = ty
                col = col + _TINTS[1 + (ty - 1) % (len(_TINTS) - 1)]
            shade = (0.75 + 0.25 * d[m])[:, None]                       # darker centre
            img[m] = col[None] * shade + rng.normal(0, 5.0, (int(m.sum()), 3)).astype(np.float32)
        imgs[t] = np.clip(img, 0, 255).astype(np.uint8)
    if nr_types is not None:
        return imgs, anns, typs
    return imgs, anns


def consep_density(size):
    """(k_lo, k_hi) nuclei per painted size x size tile at about CoNSeP's density: 24 319 nuclei in 41 images of 1000 x 1000 px =
    0.59 per 1000 px^2 = 3.8 per 80 x 80 output tile; painted 0.6 .. 1.4 x that."""
    mean = 24319.0 / 41.0 / 1.0e6 * size * size
    return max(1, int(0.6 * mean)), max(2, int(1.4 * mean))


def fit(mode="fast", nr_types=None, steps=240, batch=8, lr=1e-3, seed=0, log=None, init="synth", density=None, pool=None, local=True,
        deterministic=True):
    """Returns a trained-like network (eval mode, on the GPU) and its loss curve.
    deterministic=True (round 6): the training engine reduces every cross-workgroup sum in a fixed order and keeps the static weight-
    gradient split (`TrainEngine(deterministic=True)`), so the SAME seed gives the SAME weights on every run and every box -- a test or a
    benchmark that fits its own checkpoint is then a statement about one checkpoint, not about a sample of the atomics' orderings (round 5:
    one fit in two left the declared bf16 tolerance on the driver's box, the other did not).
    density: (k_lo, k_hi) nuclei per painted input tile (default: painted_tiles' 6..22).
    local=True: this process fits ALONE even when torch.distributed is initialised (bench.py --gpus N: every rank makes its own
    checkpoint; without this `train_step` would all-reduce the gradients of the N fits, i.e. run data-parallel training inside a
    benchmark's set-up)."""
    import os

    from . import net_desc, run_desc, targets
    from .optim import FusedAdam
    from .synth import synth_state_dict

    size, out = (270, 80) if mode == "original" else (256, 164)
    if init == "kaiming":
        torch.manual_seed(seed)
        net = net_desc.create_model(mode=mode, nr_types=nr_types, input_ch=3, freeze=False)     # weights_init: Kaiming-normal convs, BN 1 / 0
    else:
        net = net_desc.create_model(mode=mode, nr_types=nr_types, input_ch=3, freeze=False)
        net.load_state_dict(synth_state_dict(mode, nr_types, seed=seed), strict=True)
    net = net.to("cuda")
    net.train_deterministic = bool(deterministic)
    opt = FusedAdam(net.parameters(), lr=lr, betas=(0.9, 0.999))
    loss = {"np": {"bce": 1, "dice": 1}, "hv": {"mse": 1, "msge": 1}}
    if nr_types is not None:
        loss["tp"] = {"bce": 1, "dice": 1}
    run_info = [{"net": {"desc": net, "optimizer": opt, "extra_info": {"loss": loss}}}, {}]
    k_lo, k_hi = density if density is not None else (6, 22)
    painted = painted_tiles(pool if pool is not None else 8 * batch, size, seed=seed + 1, k_lo=k_lo, k_hi=k_hi, nr_types=nr_types)
    pool_img, pool_ann = painted[0], painted[1]
    pool_img_d = torch.from_numpy(pool_img).cuda()
    tg = targets.gen_targets_device(torch.from_numpy(pool_ann).cuda(), (out, out))
    tp = None
    if nr_types is not None:
        o = (size - out) // 2
        tp = torch.from_numpy(np.ascontiguousarray(painted[2][:, o:o + out, o:o + out]).astype(np.int64)).cuda()
    rng = np.random.default_rng(seed + 2)
    curve = []
    saved_dist, saved_share = run_desc._dist, os.environ.get("HVN_TILE_SHARE")
    if local:
        run_desc._dist = lambda: None                  # no gradient all-reduce
        os.environ["HVN_TILE_SHARE"] = "0"             # no launch-shape broadcast at engine build
    try:
        for it in range(steps):
            idx = torch.from_numpy(rng.choice(pool_img.shape[0], batch, replace=False)).cuda()
            feed = {"img": pool_img_d[idx], "np_map": tg["np_map"][idx], "hv_map": tg["hv_map"][idx]}
            if tp is not None:
                feed["tp_map"] = tp[idx]
            res = run_desc.train_step(feed, run_info)
            curve.append(float(res["EMA"]["overall_loss"]))
            if log is not None and (it % 40 == 0 or it == steps - 1):
                log("step %4d loss %.4f" % (it, curve[-1]))
    finally:
        run_desc._dist = saved_dist
        if saved_share is None:
            os.environ.pop("HVN_TILE_SHARE", None)
        else:
            os.environ["HVN_TILE_SHARE"] = saved_share
    net.eval()
    return net, curve
