"""Drop-in for `models.hovernet.targets.gen_targets` (/root/reference/models/hovernet/targets.py:100-116) on the GPU.

`gen_targets(ann, crop_shape)` keeps the reference's signature and return (`{"hv_map", "np_map"}` numpy arrays for one
instance-id map); `gen_targets_device(ann_dev, crop_shape)` is the batched form for a GPU-side input pipeline: int32
`[N,H,W]` instance maps in HBM -> float32 `[N,ch,cw,2]` HV targets and `[N,ch,cw]` nucleus masks in HBM, bit-exact with
the reference (tests/test_gpu_targets.py).  Kernels: csrc/hvn_targets.hip.  No CPU fallback."""
import ctypes

import numpy as np
import torch

from . import lib as L

_WS = {}


def gen_targets_device(ann_dev, crop_shape):
    L.require_gpu()
    assert ann_dev.is_cuda and ann_dev.dtype == torch.int32 and ann_dev.dim() == 3
    ann_dev = ann_dev.contiguous()
    n, h, w = ann_dev.shape
    ch, cw = int(crop_shape[0]), int(crop_shape[1])
    need = L.lib().hvn_gen_targets_workspace_bytes(n, h, w)
    key = str(ann_dev.device)
    if key not in _WS or _WS[key].numel() < need:
        _WS[key] = torch.empty(need, dtype=torch.uint8, device=ann_dev.device)
    hv = torch.empty((n, ch, cw, 2), dtype=torch.float32, device=ann_dev.device)
    npm = torch.empty((n, ch, cw), dtype=torch.int32, device=ann_dev.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(ann_dev.device).cuda_stream)
    rc = L.lib().hvn_gen_targets(ann_dev.data_ptr(), n, h, w, ch, cw, hv.data_ptr(), npm.data_ptr(), _WS[key].data_ptr(), _WS[key].numel(), stream)
    if rc:
        raise L.HvnError("hvn_gen_targets failed (%d): %s" % (rc, L.lib().hvn_train_last_error().decode()))
    return {"hv_map": hv, "np_map": npm}


def gen_targets(ann, crop_shape, **kwargs):
    """Reference signature (targets.py:100): one instance-id map [H,W] -> numpy hv_map [ch,cw,2] float32, np_map [ch,cw]."""
    out = gen_targets_device(torch.from_numpy(np.ascontiguousarray(ann, np.int32)).unsqueeze(0).to("cuda"), crop_shape)
    return {"hv_map": out["hv_map"][0].cpu().numpy(), "np_map": out["np_map"][0].cpu().numpy().astype(ann.dtype)}
