"""Drop-in for `models.hovernet.run_desc.infer_step`
(/root/reference/models/hovernet/run_desc.py:171-197).

Same signature and return contract -- `infer_step(batch_data, model)` with `batch_data`
a uint8 `[N,H,W,3]` tensor, returning a float32 numpy array `[N,h,w,3|4]` =
`[type?, p_nuc, h, v]` on the host -- but the uint8 bytes go straight to HBM (no float
NCHW copy), the whole forward + softmax/argmax/concat epilogue is one launch plan, and
the only host sync is the final D2H of the 102 KB/tile map.

`train_step` / `valid_step` (run_desc.py:12-167) run the training-mode forward, the losses, the backward pass
and the optimizer step on the HIP path (hover_net_amd.train_engine).

`infer_step_device` is the same step without the D2H: it returns the device tensor so
`post_proc.process_batch_device` can run the instance separation on-GPU with no CPU
round trip per tile (the north-star path; bench.py times this one).
"""
import torch


def _unwrap(model):
    return model.module if hasattr(model, "module") and not hasattr(model, "engine") else model


def infer_step_device(batch_data, model):
    """uint8 [N,H,W,3] (host or device) -> float32 device tensor [N,h,w,3|4]; aliases an
    engine buffer that the next call overwrites."""
    net = _unwrap(model)
    net.eval()
    if batch_data.dtype != torch.uint8:
        batch_data = batch_data.to(torch.uint8)
    dev = next(net.parameters()).device
    imgs = batch_data.to(dev, non_blocking=True)
    eng = net.engine(imgs.shape[0])
    _, pred = eng.run(imgs)
    return pred


def infer_step(batch_data, model):
    pred = infer_step_device(batch_data, model)
    return pred.cpu().numpy()


def _dist():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def train_step(batch_data, run_info):
    """Drop-in for run_desc.py:12-109.  Same protocol: `run_info = [{"net": {"desc", "optimizer", "extra_info"}},
    state]`, `batch_data` = the loader's dict (img uint8 NHWC, np_map, hv_map, tp_map?), returns
    `{"EMA": {loss_<branch>_<term>, overall_loss}, "raw": {img, np: (true, pred), hv: (true, pred)}}`.

    What runs: forward in train() mode (batch-statistics BatchNorm, running stats updated), the reference's loss
    set (np bce+dice, hv mse+msge, tp bce+dice) with the weights of `extra_info["loss"]` (opt.py:47-51), backward
    and `optimizer.step()`, all on the HIP path (hover_net_amd.train_engine).  Multi-GPU is one process per GPU:
    when torch.distributed is initialised the loss partial sums and the flat gradient slab are SUM-all-reduced
    (RCCL), which reproduces the reference's single-process DataParallel step over the concatenated batch
    (full-batch dice / msge denominators, per-replica BatchNorm statistics)."""
    from . import train_engine

    run_info, _state_info = run_info
    model = run_info["net"]["desc"]
    optimizer = run_info["net"]["optimizer"]
    net = _unwrap(model)
    loss_opts = run_info["net"].get("extra_info", {}).get("loss")
    if loss_opts is not None:       # the branches the network does not have are ignored, like run_desc.py:66 (`for branch_name in pred_dict`)
        loss_opts = {k: dict(v) for k, v in loss_opts.items() if k in ("np", "hv") or (k == "tp" and net.nr_types is not None)}
    imgs = batch_data["img"]
    eng = train_engine.engine_for(net, imgs.shape[0])
    net.train()
    eng.set_loss_weights(loss_opts)
    eng.load_batch(batch_data)
    eng.forward()
    dist = _dist()
    if dist is None:
        eng.loss_and_backward()
    else:
        eng.loss_and_backward(world=dist.get_world_size(),
                              all_reduce=lambda t, async_op=False: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op))
    optimizer.step()
    result = {"EMA": dict(eng.loss_terms())}
    # two random samples for the visualisation protocol (run_desc.py:90-107)
    idx = torch.randint(0, imgs.shape[0], (2,))
    didx = idx.to(eng.device)
    prob_np = torch.softmax(eng.logits["np"][didx], 1)[:, 1].cpu().numpy()
    pred_hv = eng.logits["hv"][didx].permute(0, 2, 3, 1).cpu().numpy()
    true_np = torch.as_tensor(batch_data["np_map"])[idx].type(torch.int64).cpu().numpy()      # .cpu(): the feed may already live on the device
    true_hv = torch.as_tensor(batch_data["hv_map"])[idx].type(torch.float32).cpu().numpy()
    result["raw"] = {"img": torch.as_tensor(imgs)[idx].byte().cpu().numpy(), "np": (true_np, prob_np), "hv": (true_hv, pred_hv)}
    return result


def valid_step(batch_data, run_info):
    """Drop-in for run_desc.py:113-167: eval-mode forward of a validation batch on the HIP path; returns
    the same `{"raw": {...}}` protocol (prob_np = softmax(np)[..., 1], pred_hv, argmax type map)."""
    run_info, _state_info = run_info
    model = run_info["net"]["desc"]
    net = _unwrap(model)
    imgs = batch_data["img"]
    true_np = torch.squeeze(batch_data["np_map"]).type(torch.int64).cpu()        # .cpu(): a device-resident feed (augment.DevicePatchLoader) is fine too
    true_hv = torch.squeeze(batch_data["hv_map"]).type(torch.float32).cpu()
    pred = infer_step_device(imgs, model)            # [N,h,w,3|4] = [type?, p_nuc, h, v] on the device
    pred = pred.cpu()
    c0 = 0 if net.nr_types is None else 1
    result = {"raw": {"imgs": imgs.cpu().numpy(), "true_np": true_np.numpy(), "true_hv": true_hv.numpy(),
                      "prob_np": pred[..., c0].numpy().copy(), "pred_hv": pred[..., c0 + 1:c0 + 3].numpy().copy()}}
    if net.nr_types is not None:
        result["raw"]["true_tp"] = torch.squeeze(batch_data["tp_map"]).type(torch.int64).cpu().numpy()
        result["raw"]["pred_tp"] = pred[..., 0].numpy().copy()
    return result


def proc_valid_step_output(raw_data, nr_types=None):
    """Scalar half of run_desc.py:262-333: validation statistics over the accumulated `valid_step` outputs
    (`raw_data[name]` = list of per-patch arrays or one stacked array): nucleus-pixel accuracy and Dice at p > 0.5,
    per-type Dice, HV mean squared error per pixel.  Host numpy, computed over the whole set at once instead of patch
    by patch; the "image" half (the reference's matplotlib / cv2 visualisation) is not rebuilt and stays empty."""
    import numpy as np

    track = {"scalar": {}, "image": {}}
    prob_np = np.asarray(raw_data["prob_np"])
    true_np = np.asarray(raw_data["true_np"])
    pred_np = (prob_np > 0.5).astype(np.int32)

    def dice(true, pred, label):
        t, p = (true == label), (pred == label)
        return 2.0 * np.logical_and(t, p).sum() / (t.sum() + p.sum() + 1.0e-8)

    nr_pixels = true_np.size
    track["scalar"]["np_acc"] = (pred_np == true_np).sum() / nr_pixels
    track["scalar"]["np_dice"] = dice(true_np, pred_np, 1)
    if nr_types is not None:
        pred_tp, true_tp = np.asarray(raw_data["pred_tp"]), np.asarray(raw_data["true_tp"])
        for type_id in range(nr_types):
            track["scalar"]["tp_dice_%d" % type_id] = dice(true_tp, pred_tp, type_id)
    err = np.asarray(raw_data["pred_hv"], np.float64) - np.asarray(raw_data["true_hv"], np.float64)
    track["scalar"]["hv_mse"] = (err * err).sum() / nr_pixels
    return track
