"""Drop-in for `models.hovernet.run_desc.infer_step`
(/root/reference/models/hovernet/run_desc.py:171-197).

Same signature and return contract -- `infer_step(batch_data, model)` with `batch_data`
a uint8 `[N,H,W,3]` tensor, returning a float32 numpy array `[N,h,w,3|4]` =
`[type?, p_nuc, h, v]` on the host -- but the uint8 bytes go straight to HBM (no float
NCHW copy), the whole forward + softmax/argmax/concat epilogue is one launch plan, and
the only host sync is the final D2H of the 102 KB/tile map.

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


def train_step(batch_data, run_info):  # run_desc.py:12-109
    raise NotImplementedError("hover_net_amd: the training step (SURVEY 8a T1-T5) is not built in this round")


def valid_step(batch_data, run_info):
    """Drop-in for run_desc.py:113-167: eval-mode forward of a validation batch on the HIP path; returns
    the same `{"raw": {...}}` protocol (prob_np = softmax(np)[..., 1], pred_hv, argmax type map)."""
    run_info, _state_info = run_info
    model = run_info["net"]["desc"]
    net = _unwrap(model)
    imgs = batch_data["img"]
    true_np = torch.squeeze(batch_data["np_map"]).type(torch.int64)
    true_hv = torch.squeeze(batch_data["hv_map"]).type(torch.float32)
    pred = infer_step_device(imgs, model)            # [N,h,w,3|4] = [type?, p_nuc, h, v] on the device
    pred = pred.cpu()
    c0 = 0 if net.nr_types is None else 1
    result = {"raw": {"imgs": imgs.numpy(), "true_np": true_np.numpy(), "true_hv": true_hv.numpy(),
                      "prob_np": pred[..., c0].numpy().copy(), "pred_hv": pred[..., c0 + 1:c0 + 3].numpy().copy()}}
    if net.nr_types is not None:
        result["raw"]["true_tp"] = torch.squeeze(batch_data["tp_map"]).type(torch.int64).numpy()
        result["raw"]["pred_tp"] = pred[..., 0].numpy().copy()
    return result
