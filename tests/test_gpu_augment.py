"""-m gpu: the augmentation kernels (hover_net_amd/csrc/hvn_augment.hip through include/hvn.h) against oracle/augment_np.py on the
same explicit parameter records, bit for bit, and the resident-set loader end to end (dataloader/train_loader.py:76-199)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _resident(p=3, h=96, w=88, seed=0):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (p, h, w, 3), dtype=np.uint8)
    ann = np.zeros((p, h, w, 2), np.int32)
    for k in range(p):                         # a few rectangular "nuclei" with ids and types
        for i in range(1, 9):
            y, x = rng.integers(0, h - 14), rng.integers(0, w - 14)
            ann[k, y:y + rng.integers(6, 14), x:x + rng.integers(6, 14)] = (i, rng.integers(1, 5))
    return img, ann


def test_shape_kernel_matches_oracle():
    from hover_net_amd import augment as G
    from oracle import augment_np as A

    img, ann = _resident()
    rng = np.random.default_rng(5)
    src = rng.integers(0, 3, 10)
    prm = G.draw_params(rng, src, 96, 88)
    prm[0] = G.identity_params(1, [2])[0]                       # one plain centre crop
    prm["inv"][1] = np.linalg.inv(G.affine_matrix(96, 88, (1, 1), (30, -40), 0, 0))[:2].reshape(-1)     # mostly outside -> zeros
    oi, oa = G.augment_shape(torch.from_numpy(img).cuda(), torch.from_numpy(ann).cuda(), prm, (64, 70))
    oi, oa = oi.cpu().numpy(), oa.cpu().numpy()
    assert oi.shape == (10, 64, 70, 3) and oa.shape == (10, 64, 70, 2) and oa.dtype == np.int32
    for i in range(10):
        inv = np.vstack([prm["inv"][i].reshape(2, 3), [0, 0, 1]])
        wi, wa = A.shape_augment(img[prm["src"][i]], ann[prm["src"][i]], inv, (64, 70), bool(prm["flip_lr"][i]), bool(prm["flip_ud"][i]))
        np.testing.assert_array_equal(oi[i], wi)
        np.testing.assert_array_equal(oa[i], wa)
    np.testing.assert_array_equal(oi[0], img[2, 16:80, 9:79])
    assert (oi[1] == 0).mean() > 0.3
    with pytest.raises(ValueError, match="outside the resident set"):
        bad = prm.copy()
        bad["src"][3] = 3
        G.augment_shape(torch.from_numpy(img).cuda(), torch.from_numpy(ann).cuda(), bad, (64, 70))


def test_input_kernel_matches_oracle():
    from hover_net_amd import augment as G
    from oracle import augment_np as A

    rng = np.random.default_rng(7)
    cases = [(0, 1, 1), (0, 3, 1), (0, 1, 5), (0, 5, 3), (0, 5, 5), (1, 1, 0), (1, 3, 0), (1, 5, 0), (2, 0, 0), (2, 1, 0), (3, 0, 0), (0, 3, 3)]
    n, h, w = len(cases), 37, 45
    img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    img[3, :5] = 255                                            # saturated / black / grey regions for the colour conversions
    img[4, :5] = 0
    img[5, :8] = 128
    prm = G.draw_params(rng, np.arange(n), h, w)
    for i, (kind, p0, p1) in enumerate(cases):
        prm["kind"][i], prm["p0"][i], prm["p1"][i] = kind, max(p0, 1), max(p1, 1)
        prm["per_channel"][i] = p0 if kind == 2 else 0
    prm["order"][10] = -1                                       # nothing at all: copy
    prm["order"][11] = (0, -1, 0, 2)                            # repeated / skipped entries are legal for the kernel
    prm["hue"][2], prm["hue"][3] = -8.0, 7.999
    prm["bright"][4], prm["bright"][5] = 26.0, -25.5
    noise = torch.randn((n, h, w, 3), generator=torch.Generator().manual_seed(3))
    out = G.augment_input(torch.from_numpy(img).cuda(), prm, noise.cuda()).cpu().numpy()
    z = noise.numpy()
    for i in range(n):
        nz = None
        if prm["kind"][i] == 2:
            zz = z[i] if prm["per_channel"][i] else z[i][..., :1]
            nz = (zz.astype(np.float32) * np.float32(prm["noise_scale"][i])).astype(np.float32)
        want = A.input_augment(img[i], prm["kind"][i], prm["p0"][i], prm["p1"][i], nz, [o for o in prm["order"][i] if o >= 0],
                               prm["hue"][i], prm["sat"][i], prm["bright"][i], prm["contrast"][i])
        np.testing.assert_array_equal(out[i], want, err_msg="case %d %s" % (i, cases[i]))
    np.testing.assert_array_equal(out[10], img[10])


def test_input_kernel_equals_the_references_own_functions():
    """Each op alone against tests/golden/augs.npz = outputs of the reference's unmodified dataloader/augs.py functions."""
    import os

    from hover_net_amd import augment as G

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augs.npz"))
    img = torch.from_numpy(g["img"]).cuda()

    def run(n, **fields):
        prm = G.identity_params(n)
        for k, v in fields.items():
            prm[k] = v
        return G.augment_input(img[:n].contiguous(), prm).cpu().numpy()

    np.testing.assert_array_equal(run(6, kind=0, p0=g["gauss_k"][:, 0], p1=g["gauss_k"][:, 1]), g["gauss_out"])
    np.testing.assert_array_equal(run(3, kind=1, p0=g["median_k"]), g["median_out"])
    np.testing.assert_array_equal(run(6, order=[[0, -1, -1, -1]] * 6, hue=g["hue_val"]), g["hue_out"])
    np.testing.assert_array_equal(run(6, order=[[-1, 1, -1, -1]] * 6, sat=1 + g["sat_val"]), g["sat_out"])
    np.testing.assert_array_equal(run(6, order=[[-1, -1, 2, -1]] * 6, bright=g["bright_val"]), g["bright_out"])
    np.testing.assert_array_equal(run(6, order=[[-1, -1, -1, 3]] * 6, contrast=g["contrast_val"]), g["contrast_out"])


def test_resident_loader_end_to_end():
    from hover_net_amd import augment as G
    from hover_net_amd import targets as T

    img, ann = _resident(p=10, h=120, w=120, seed=2)
    data = np.concatenate([img.astype(np.int32), ann], -1)
    ld = G.DevicePatchLoader(data, (100, 100), (40, 40), batch_size=4, mode="valid", with_type=True)
    batches = list(ld)
    assert len(batches) == 3 and [b["img"].shape[0] for b in batches] == [4, 4, 2]
    b0 = batches[0]
    np.testing.assert_array_equal(b0["img"].cpu().numpy(), img[:4, 10:110, 10:110])           # centre crop only
    np.testing.assert_array_equal(b0["tp_map"].cpu().numpy(), ann[:4, 40:80, 40:80, 1])
    want = T.gen_targets_device(torch.from_numpy(np.ascontiguousarray(ann[:4, 10:110, 10:110, 0])).cuda(), (40, 40))
    assert torch.equal(b0["hv_map"], want["hv_map"]) and torch.equal(b0["np_map"], want["np_map"])
    assert b0["img"].dtype == torch.uint8 and b0["np_map"].dtype == torch.int32 and b0["hv_map"].shape == (4, 40, 40, 2)

    tr = G.DevicePatchLoader(data, (100, 100), (40, 40), batch_size=4, mode="train", with_type=True, seed=11)
    e0 = list(tr)
    e1 = list(tr)
    assert len(e0) == 2 and all(b["img"].shape == (4, 100, 100, 3) for b in e0)                 # drop_last
    assert not torch.equal(e0[0]["img"], e1[0]["img"])                                          # a new epoch draws new augmentations
    tr2 = G.DevicePatchLoader(data, (100, 100), (40, 40), batch_size=4, mode="train", with_type=True, seed=11)
    e0b = list(tr2)
    assert all(torch.equal(a["img"], b["img"]) and torch.equal(a["hv_map"], b["hv_map"]) for a, b in zip(e0, e0b))   # same seed, same epoch
    for b in e0:
        assert set(np.unique(b["np_map"].cpu().numpy())) <= {0, 1} and float(b["hv_map"].abs().max()) <= 1.0
        tp = b["tp_map"].cpu().numpy()
        assert tp.min() >= 0 and tp.max() <= 4


def test_loader_feeds_train_step():
    """The feed dict goes straight into run_desc.train_step (device tensors, reference key names and dtypes)."""
    from hover_net_amd import augment as G
    from hover_net_amd import net_desc, optim, run_desc
    from hover_net_amd.synth import synth_state_dict

    img, ann = _resident(p=2, h=300, w=300, seed=4)          # one training batch: the engine build dominates this test
    data = np.concatenate([img.astype(np.int32), ann], -1)
    ld = G.DevicePatchLoader(data, (270, 270), (80, 80), batch_size=2, mode="train", with_type=True, seed=1)
    net = net_desc.create_model(mode="original", nr_types=5, input_ch=3, freeze=True)
    net.load_state_dict(synth_state_dict("original", 5, seed=0), strict=True)
    net = net.to("cuda").train()
    opt = optim.FusedAdam(filter(lambda p: p.requires_grad, net.parameters()), lr=1e-4)
    loss_tab = {"np": {"bce": 1, "dice": 1}, "hv": {"mse": 1, "msge": 1}, "tp": {"bce": 1, "dice": 1}}
    for feed in ld:
        out = run_desc.train_step(feed, [{"net": {"desc": net, "optimizer": opt, "extra_info": {"loss": loss_tab}}}, {}])
        assert np.isfinite(out["EMA"]["overall_loss"]) and out["raw"]["img"].shape == (2, 270, 270, 3)
    vl = G.DevicePatchLoader(data, (270, 270), (80, 80), batch_size=2, mode="valid", with_type=True)
    res = run_desc.valid_step(next(iter(vl)), [{"net": {"desc": net}}, {}])["raw"]
    assert res["true_tp"].shape == (2, 80, 80) and res["prob_np"].shape == (2, 80, 80) and res["imgs"].shape == (2, 270, 270, 3)
