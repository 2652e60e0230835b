"""Training input pipeline on the GPU: a resident patch set, per-sample augmentation and target generation without the
reference's 16 DataLoader workers.

Mirrors `dataloader/train_loader.py:27-199` (`FileLoader`): the `.npy` patches `[H, W, 5]` (RGB, instance id, type) that
`extract_patches.py` writes are uploaded ONCE into HBM; a batch is then
    draw parameters on the host (numpy Generator; same distributions as `__get_augmentation`, train_loader.py:111-199)
    -> `hvn_augment_shape`  affine (scale 0.8-1.2 per axis, translate +-1 %, shear +-5 deg, rotate +-179 deg, nearest, constant 0)
                            + centre crop to `input_shape` + flips, image and annotation in one gather
    -> `hvn_augment_input`  one of {Gaussian blur, median blur, additive Gaussian noise}, then hue +-8 / saturation +-0.2 /
                            brightness +-26 / contrast 0.75-1.25 in random order (image only)
    -> `targets.gen_targets_device` on the instance plane (`np_map`, `hv_map` at `mask_shape`), type plane centre-cropped
and comes back as the feed dict `run_desc.train_step` takes, already on the device.  "valid" mode is the centre crop only.

The draws come from this module's own generator, not imgaug's: the augmentation DISTRIBUTION is the reference's, the random
stream is not (imgaug is not in this image, and per-worker seeding makes the reference's own stream irreproducible anyway).
Kernel parity: tests/test_gpu_augment.py against oracle/augment_np.py with the same explicit parameters, bit for bit.
No CPU fallback."""
import ctypes

import numpy as np
import torch

from . import lib as L
from . import targets

AUG_DTYPE = np.dtype([("inv", "<f8", (6,)), ("src", "<i4"), ("flip_lr", "<i4"), ("flip_ud", "<i4"), ("kind", "<i4"), ("p0", "<i4"), ("p1", "<i4"),
                      ("per_channel", "<i4"), ("noise_scale", "<f4"), ("order", "<i4", (4,)), ("hue", "<f8"), ("sat", "<f8"), ("bright", "<f8"),
                      ("contrast", "<f8")], align=True)
assert AUG_DTYPE.itemsize == 128, AUG_DTYPE.itemsize      # == sizeof(hvn_aug_sample), include/hvn.h


def affine_matrix(h, w, scale_xy, translate_px, shear_deg, rotate_deg):
    """Forward matrix source -> destination: scale, shear, rotation (skimage `AffineTransform` parameterisation, which is what
    imgaug's `Affine` composes) about the image centre ((w-1)/2, (h-1)/2), then the translation."""
    sx, sy = scale_xy
    rot, sh = np.deg2rad(rotate_deg), np.deg2rad(shear_deg)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    m = np.array([[sx * np.cos(rot), -sy * np.sin(rot + sh), 0.0], [sx * np.sin(rot), sy * np.cos(rot + sh), 0.0], [0.0, 0.0, 1.0]])
    to_origin = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    back = np.array([[1, 0, cx + translate_px[0]], [0, 1, cy + translate_px[1]], [0, 0, 1.0]])
    return back @ m @ to_origin


def identity_params(n, src=None):
    """Records that only centre-crop ("valid" mode, train_loader.py:190-197)."""
    prm = np.zeros(n, AUG_DTYPE)
    prm["inv"][:] = (1, 0, 0, 0, 1, 0)
    prm["src"] = np.arange(n) if src is None else src
    prm["kind"] = 3
    prm["order"][:] = -1
    prm["sat"] = 1.0
    return prm


def draw_params(rng, src, h, w):
    """One record per entry of `src` with the distributions of train_loader.py:123-187."""
    n = len(src)
    prm = identity_params(n, src)
    for i in range(n):
        fwd = affine_matrix(h, w, rng.uniform(0.8, 1.2, 2), (rng.uniform(-0.01, 0.01) * w, rng.uniform(-0.01, 0.01) * h),
                            rng.uniform(-5, 5), rng.uniform(-179, 179))
        prm["inv"][i] = np.linalg.inv(fwd)[:2].reshape(-1)
    prm["flip_lr"] = rng.random(n) < 0.5                    # iaa.Fliplr(0.5), iaa.Flipud(0.5)
    prm["flip_ud"] = rng.random(n) < 0.5
    prm["kind"] = rng.integers(0, 3, n)                      # iaa.OneOf([gaussian_blur, median_blur, AdditiveGaussianNoise])
    prm["p0"] = rng.integers(0, 3, n) * 2 + 1                # random_state.randint(0, max_ksize=3) * 2 + 1   (augs.py:39-40, 54-55)
    prm["p1"] = rng.integers(0, 3, n) * 2 + 1
    prm["noise_scale"] = rng.uniform(0.0, 0.05 * 255, n)     # scale=(0.0, 0.05 * 255)
    prm["per_channel"] = rng.random(n) < 0.5                 # per_channel=0.5
    for i in range(n):
        prm["order"][i] = rng.permutation(4)                 # iaa.Sequential([...], random_order=True)
    prm["hue"] = rng.uniform(-8, 8, n)
    prm["sat"] = 1 + rng.uniform(-0.2, 0.2, n)               # augs.py:82
    prm["bright"] = rng.uniform(-26, 26, n)
    prm["contrast"] = rng.uniform(0.75, 1.25, n)
    return prm


def _upload(prm, device):
    return torch.from_numpy(np.ascontiguousarray(prm).view(np.uint8).reshape(-1)).to(device)


def augment_shape(img_dev, ann_dev, prm, out_hw):
    """img_dev uint8 [P,H,W,3], ann_dev int32 [P,H,W,C] (resident set, device); prm: AUG_DTYPE records -> (uint8 [n,oh,ow,3], int32 [n,oh,ow,C])."""
    L.require_gpu()
    assert img_dev.is_cuda and img_dev.dtype == torch.uint8 and img_dev.dim() == 4 and img_dev.shape[-1] == 3 and img_dev.is_contiguous()
    assert ann_dev.is_cuda and ann_dev.dtype == torch.int32 and ann_dev.dim() == 4 and ann_dev.shape[:3] == img_dev.shape[:3] and ann_dev.is_contiguous()
    p, h, w, _ = img_dev.shape
    c = ann_dev.shape[-1]
    n = len(prm)
    if n and (prm["src"].min() < 0 or prm["src"].max() >= p):
        raise ValueError("augment: source index outside the resident set of %d patches" % p)
    oh, ow = int(out_hw[0]), int(out_hw[1])
    prm_dev = _upload(prm, img_dev.device)
    oimg = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=img_dev.device)
    oann = torch.empty((n, oh, ow, c), dtype=torch.int32, device=img_dev.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(img_dev.device).cuda_stream)
    rc = L.lib().hvn_augment_shape(img_dev.data_ptr(), ann_dev.data_ptr(), p, h, w, c, prm_dev.data_ptr(), n, oh, ow, oimg.data_ptr(), oann.data_ptr(), stream)
    if rc:
        raise L.HvnError("hvn_augment_shape failed (%d): %s" % (rc, L.lib().hvn_train_last_error().decode()))
    return oimg, oann


def augment_input(img_dev, prm, noise=None):
    """img_dev uint8 [n,h,w,3] (device) -> augmented copy.  `noise`: float32 [n,h,w,3] standard-normal samples (drawn here with
    torch's device generator when a record asks for additive noise and none is given)."""
    L.require_gpu()
    assert img_dev.is_cuda and img_dev.dtype == torch.uint8 and img_dev.dim() == 4 and img_dev.shape[-1] == 3 and img_dev.is_contiguous()
    n, h, w, _ = img_dev.shape
    assert len(prm) == n
    if noise is None and (prm["kind"] == 2).any():
        noise = torch.randn((n, h, w, 3), dtype=torch.float32, device=img_dev.device)
    if noise is not None:
        assert noise.is_cuda and noise.dtype == torch.float32 and tuple(noise.shape) == (n, h, w, 3) and noise.is_contiguous()
    prm_dev = _upload(prm, img_dev.device)
    out = torch.empty_like(img_dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(img_dev.device).cuda_stream)
    rc = L.lib().hvn_augment_input(img_dev.data_ptr(), prm_dev.data_ptr(), noise.data_ptr() if noise is not None else None, n, h, w, out.data_ptr(), stream)
    if rc:
        raise L.HvnError("hvn_augment_input failed (%d): %s" % (rc, L.lib().hvn_train_last_error().decode()))
    return out


def _cropping_center(x, crop_shape):
    """misc/utils.py:32-52 on [N,H,W,...]."""
    h0 = int((x.shape[1] - crop_shape[0]) * 0.5)
    w0 = int((x.shape[2] - crop_shape[1]) * 0.5)
    return x[:, h0:h0 + crop_shape[0], w0:w0 + crop_shape[1]]


class DevicePatchLoader:
    """`FileLoader` + `DataLoader` of the training run (run_train.py:106-133) as one object over a patch set resident in HBM.

    patches: list of `.npy` paths or one uint8/int32 array [P,H,W,5] (RGB, instance id, type -- train_loader.py:84-89).
    Iterating yields `batch_size` feed dicts per step on the device: {"img" uint8 [B,ih,iw,3], "np_map" int32 [B,mh,mw],
    "hv_map" float32 [B,mh,mw,2], "tp_map" int32 [B,mh,mw] (with_type)} -- what `run_desc.train_step` / `valid_step` consume.
    mode "train": shuffled every epoch, ragged last batch dropped (DataLoader(shuffle=True, drop_last=True)); "valid": in order,
    centre crop only.  With `world > 1` every rank takes the slice `rank::world` of the epoch's permutation (same seed on all
    ranks), i.e. what a DistributedSampler would hand it."""

    def __init__(self, patches, input_shape, mask_shape, batch_size, mode="train", with_type=False, seed=0, device="cuda", rank=0, world=1):
        assert mode in ("train", "valid")
        self.device = torch.device(device)
        if isinstance(patches, (list, tuple)):          # one file at a time: the set never exists as one host array
            first = np.load(patches[0])
            assert first.ndim == 3 and first.shape[-1] >= 4, "patch files: [H, W, 5] = RGB + instance id (+ type)"
            h, w, c = first.shape
            self.img = torch.empty((len(patches), h, w, 3), dtype=torch.uint8, device=self.device)
            self.ann = torch.empty((len(patches), h, w, min(c - 3, 2)), dtype=torch.int32, device=self.device)
            for i, path in enumerate(patches):
                d = first if i == 0 else np.load(path)
                if d.shape != first.shape:
                    raise ValueError("patch %s has shape %s, the set's is %s" % (path, d.shape, first.shape))
                self.img[i] = torch.from_numpy(np.ascontiguousarray(d[..., :3]).astype(np.uint8)).to(self.device)
                self.ann[i] = torch.from_numpy(np.ascontiguousarray(d[..., 3:5]).astype(np.int32)).to(self.device)
        else:
            data = np.asarray(patches)
            assert data.ndim == 4 and data.shape[-1] >= 4, "patches: [P, H, W, 5] = RGB + instance id (+ type)"
            self.img = torch.from_numpy(np.ascontiguousarray(data[..., :3]).astype(np.uint8)).to(self.device)
            self.ann = torch.from_numpy(np.ascontiguousarray(data[..., 3:5]).astype(np.int32)).to(self.device)
        self.with_type = bool(with_type)
        if self.with_type:
            assert self.ann.shape[-1] == 2, "with_type needs the type plane (channel 4)"
        self.input_shape, self.mask_shape = tuple(int(v) for v in input_shape), tuple(int(v) for v in mask_shape)
        self.batch_size, self.mode = int(batch_size), mode
        self.seed, self.rank, self.world = int(seed), int(rank), int(world)
        self.epoch = 0

    def _share(self):
        """Patches per epoch on this rank: train = the same count on every rank (P // world, like DistributedSampler(drop_last=True) --
        `train.run_phases` refuses ranks that disagree on the number of steps: their collectives would not pair up); valid = every patch."""
        p = self.img.shape[0]
        return p // self.world if self.mode == "train" else len(range(self.rank, p, self.world))

    def __len__(self):
        mine = self._share()
        return mine // self.batch_size if self.mode == "train" else -(-mine // self.batch_size)

    def batch(self, prm, noise=None, generator=None):
        """The device pipeline for one batch of parameter records.  `noise` / `generator`: the N(0,1) samples of the additive-noise
        records, or the torch device generator to draw them with (the loader seeds one per epoch and rank)."""
        img, ann = augment_shape(self.img, self.ann, prm, self.input_shape)
        if (prm["kind"] != 3).any() or (prm["order"] >= 0).any():
            if noise is None and (prm["kind"] == 2).any():
                noise = torch.randn((len(prm),) + self.input_shape + (3,), dtype=torch.float32, device=self.device, generator=generator)
            img = augment_input(img, prm, noise)
        feed = {"img": img}
        if self.with_type:
            feed["tp_map"] = _cropping_center(ann[..., 1], self.mask_shape).contiguous()
        feed.update(targets.gen_targets_device(ann[..., 0].contiguous(), self.mask_shape))
        return feed

    def __iter__(self):
        p, h, w = self.img.shape[:3]
        order_rng = np.random.default_rng([self.seed, self.epoch])                 # the same permutation on every rank
        rng = np.random.default_rng([self.seed, self.epoch, self.rank + 1])        # this rank's augmentation draws
        gen = None
        if self.device.type == "cuda":                                              # ... and its noise samples
            gen = torch.Generator(device=self.device)
            gen.manual_seed(int(rng.integers(0, 2 ** 62)))
        self.epoch += 1
        order = order_rng.permutation(p) if self.mode == "train" else np.arange(p)
        mine = order[self.rank::self.world][:self._share()]
        for b in range(len(self)):
            src = mine[b * self.batch_size:(b + 1) * self.batch_size]
            prm = draw_params(rng, src, h, w) if self.mode == "train" else identity_params(len(src), src)
            yield self.batch(prm, generator=gen)
