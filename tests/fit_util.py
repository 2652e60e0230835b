"""A 'trained-like' HoVer-Net checkpoint for tolerance tests, made on the GPU box with the repository's OWN trainer.

A random-init network emits noise maps; the bf16 tolerance (BASELINE cfg 3) is about what happens to the SEGMENTATION of a
network that actually segments.  No checkpoint can be downloaded, so one is fitted here: synthetic H&E-like tiles (painted,
partly touching elliptical nuclei, darker and bluer than a noisy pink background) with targets from the product's own
`gen_targets_device` (bit-exact with the reference's targets.py, tests/test_gpu_targets.py), a few hundred steps of
`run_desc.train_step` (HIP training path, FusedAdam).  Test infrastructure only."""
import numpy as np
import torch


def painted_tiles(n, size, seed, k_lo=6, k_hi=22):
    """-> (img uint8 [n,size,size,3], ann int32 [n,size,size] instance ids).  Nuclei: ellipses of radius 5..11 px."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    imgs = np.empty((n, size, size, 3), np.uint8)
    anns = np.zeros((n, size, size), np.int32)
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
            shade = (0.75 + 0.25 * d[m])[:, None]                       # darker centre
            img[m] = col[None] * shade + rng.normal(0, 5.0, (int(m.sum()), 3)).astype(np.float32)
        imgs[t] = np.clip(img, 0, 255).astype(np.uint8)
    return imgs, anns


def fit(mode="fast", nr_types=None, steps=240, batch=8, lr=1e-3, seed=0, log=None):
    """Returns a trained-like network (eval mode, on the GPU) and its loss curve."""
    from hover_net_amd import net_desc, run_desc, targets
    from hover_net_amd.optim import FusedAdam
    from hover_net_amd.synth import synth_state_dict

    size, out = (270, 80) if mode == "original" else (256, 164)
    net = net_desc.create_model(mode=mode, nr_types=nr_types, input_ch=3, freeze=False)
    net.load_state_dict(synth_state_dict(mode, nr_types, seed=seed), strict=True)
    net = net.to("cuda")
    opt = FusedAdam(net.parameters(), lr=lr, betas=(0.9, 0.999))
    run_info = [{"net": {"desc": net, "optimizer": opt, "extra_info": {"loss": {"np": {"bce": 1, "dice": 1}, "hv": {"mse": 1, "msge": 1}}}}}, {}]
    pool_img, pool_ann = painted_tiles(8 * batch, size, seed=seed + 1)
    pool_img_d = torch.from_numpy(pool_img).cuda()
    tg = targets.gen_targets_device(torch.from_numpy(pool_ann).cuda(), (out, out))
    rng = np.random.default_rng(seed + 2)
    curve = []
    for it in range(steps):
        idx = torch.from_numpy(rng.choice(pool_img.shape[0], batch, replace=False)).cuda()
        feed = {"img": pool_img_d[idx], "np_map": tg["np_map"][idx], "hv_map": tg["hv_map"][idx]}
        res = run_desc.train_step(feed, run_info)
        curve.append(float(res["EMA"]["overall_loss"]))
        if log is not None and (it % 40 == 0 or it == steps - 1):
            log("step %4d loss %.4f" % (it, curve[-1]))
    net.eval()
    return net, curve
