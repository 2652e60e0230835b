"""`FusedAdam`: torch.optim.Adam semantics (the optimizer opt.py:38-44 configures: lr 1e-4, betas (0.9, 0.999),
eps 1e-8, no weight decay, no amsgrad) as ONE HIP launch over the flat parameter slab of the training engine
(hover_net_amd.train_engine lays every parameter and its gradient out at the same offset of two slabs).

Constructor signature = torch's (`FusedAdam(params, lr=..., betas=...)`), so the reference's
`optimizer[0](net.parameters(), **optimizer[1])` (run_train.py:186-191) takes it unchanged, as do
`lr_scheduler.StepLR` (param_groups[...]['lr']) and state_dict()/load_state_dict().  Parameters that are not
slab-backed (the engine has not been built yet, or a foreign tensor) are updated with one launch per tensor.
Parameters without a gradient keep a zero gradient in the slab: m and v stay 0 and the update is exactly 0,
which is what torch.optim.Adam's "skip params with grad None" does."""
import ctypes

import torch

from . import lib as L


def _check(rc):
    if rc != 0:   # the training entry points report through hvn_train_last_error
        raise L.HvnError("hvn_adam_step failed (%d): %s" % (rc, L.lib().hvn_train_last_error().decode()))


def _bump(params):
    """The kernel writes the weights through raw pointers, behind torch's version counters; bump one so that anything keyed
    on `p._version` (HoVerNet._weights_version hashes every parameter's version -> the cached inference plan) sees the
    update.  One parameter per group is enough for that key and keeps the step free of a 400-iteration python loop."""
    for p in params[:1]:
        torch._C._increment_version(p)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.fused_launches = 0
        self.fallback_launches = 0

    def _slab(self, group):
        """(lo, hi) element range if every param with a grad shares one storage with its grad at equal offsets."""
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None
        ws, gs = ps[0].data.untyped_storage().data_ptr(), ps[0].grad.untyped_storage().data_ptr()
        lo, hi = None, None
        for p in group["params"]:
            if p.data.untyped_storage().data_ptr() != ws or p.dtype != torch.float32:
                return None
            if p.grad is not None and (p.grad.untyped_storage().data_ptr() != gs or p.grad.storage_offset() != p.data.storage_offset()
                                       or p.grad.stride() != p.data.stride()):
                return None
            o = p.data.storage_offset()
            lo = o if lo is None else min(lo, o)
            hi = o + p.numel() if hi is None else max(hi, o + p.numel())
        return ws, gs, lo, hi

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = L.lib()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            st = self.state.setdefault("group%d" % gi, {})
            st["step"] = st.get("step", 0) + 1
            slab = self._slab(group)
            dev = group["params"][0].device
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if slab is not None:
                ws, gs, lo, hi = slab
                lo = lo // 4 * 4
                n = hi - lo
                if "m" not in st or st["m"].numel() != n + 4:
                    st["m"] = torch.zeros(n + 4, dtype=torch.float32, device=dev)
                    st["v"] = torch.zeros(n + 4, dtype=torch.float32, device=dev)
                rc = lib.hvn_adam_step(ws + 4 * lo, gs + 4 * lo, st["m"].data_ptr(), st["v"].data_ptr(), n, group["lr"], b1, b2,
                                       group["eps"], st["step"], stream)
                _check(rc)
                self.fused_launches += 1
                _bump(group["params"])
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise L.HvnError("FusedAdam: a parameter outside the training slab must be contiguous")
                ps = self.state.setdefault(p, {})
                if "m" not in ps:
                    ps["m"], ps["v"] = torch.zeros_like(p.data), torch.zeros_like(p.data)
                rc = lib.hvn_adam_step(p.data_ptr(), p.grad.data_ptr(), ps["m"].data_ptr(), ps["v"].data_ptr(), p.numel(), group["lr"], b1, b2,
                                       group["eps"], st["step"], stream)
                _check(rc)
                self.fallback_launches += 1
                _bump([p])
        return loss
