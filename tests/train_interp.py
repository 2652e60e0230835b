"""torch-CPU interpreter of hover_net_amd.train_plan (TEST ONLY): executes the forward / backward op lists with
the semantics the HIP kernels implement (accumulating gradients, shared / dilated gradient buffers, dgrad as a
stride-1 conv with flipped transposed weights), so that (a) the lowering can be checked against the training
oracle on the CPU and (b) every HIP training kernel has a per-op reference."""
import torch
import torch.nn.functional as F

from hover_net_amd import arch

EPS, MOM = arch.BN_EPS, 0.1


def vslice(t, v):
    s = v.step
    return t[:, v.y0:v.y0 + (v.h - 1) * s + 1:s, v.x0:v.x0 + (v.w - 1) * s + 1:s, v.c0:v.c0 + v.c]


def nchw(x):
    return x.permute(0, 3, 1, 2)


def nhwc(x):
    return x.permute(0, 2, 3, 1)


def expand_groups(w, groups):
    """[cout, cin_g, kh, kw] -> block-diagonal dense [cout, cin_g*groups, kh, kw]."""
    if groups == 1:
        return w
    cout, cin_g, kh, kw = w.shape
    full = torch.zeros(cout, cin_g * groups, kh, kw, dtype=w.dtype)
    og = cout // groups
    for g in range(groups):
        full[g * og:(g + 1) * og, g * cin_g:(g + 1) * cin_g] = w[g * og:(g + 1) * og]
    return full


def dgrad_weights(w, groups):
    """Forward weights -> weights of the data-gradient conv: Wt[ci][co][r][s] = W[co][ci][KH-1-r][KW-1-s]."""
    return expand_groups(w, groups).flip(2, 3).transpose(0, 1).contiguous()


def conv_fwd_ref(op, x, w, res=None):
    lo, hi = op.pad
    y = nhwc(F.conv2d(F.pad(nchw(x), (lo, hi, lo, hi)), w, stride=op.stride, groups=op.groups))
    return y if res is None else y + res


def wgrad_ref(op, x, dy, wshape):
    lo, hi = op.pad
    xp = F.pad(nchw(x), (lo, hi, lo, hi))
    return torch.nn.grad.conv2d_weight(xp, wshape, nchw(dy).contiguous(), stride=op.stride, groups=op.groups)


def dgrad_ref(op, dy, w, out_hw):
    """dy: dense-cover view tensor [N,Hd,Wd,cout] -> [N,out_h,out_w,cin]."""
    wt = dgrad_weights(w, op.groups)
    lo = op.pad[0]
    k = op.kh
    hi_h = out_hw[0] - dy.shape[1] - lo + k - 1
    hi_w = out_hw[1] - dy.shape[2] - lo + k - 1
    assert hi_h >= 0 and hi_w >= 0
    return nhwc(F.conv2d(F.pad(nchw(dy), (lo, hi_w, lo, hi_h)), wt))


def bn_fwd_ref(z, gamma, beta, rm, rv):
    """-> a, mean, rstd; updates rm / rv in place (torch BatchNorm2d train semantics)."""
    n = z.shape[0] * z.shape[1] * z.shape[2]
    zd = z.double()
    mean = zd.mean((0, 1, 2))
    var = (zd * zd).mean((0, 1, 2)) - mean * mean
    rstd = 1.0 / torch.sqrt(var + EPS)
    rm.mul_(1 - MOM).add_(MOM * mean.to(rm.dtype))
    rv.mul_(1 - MOM).add_(MOM * (var * n / (n - 1)).to(rv.dtype))
    scale = (gamma.double() * rstd).to(z.dtype)
    shift = (beta.double() - mean * gamma.double() * rstd).to(z.dtype)
    a = F.relu(z * scale + shift)
    return a, mean.to(z.dtype), rstd.to(z.dtype)


def bn_bwd_ref(z, a, da, gamma, mean, rstd):
    """-> dz, dgamma, dbeta."""
    n = z.shape[0] * z.shape[1] * z.shape[2]
    g = da * (a > 0)
    xhat = (z - mean) * rstd
    s1 = g.double().sum((0, 1, 2))
    s2 = (g * xhat).double().sum((0, 1, 2))
    c1 = gamma * rstd
    dz = c1 * (g - (s1 / n).to(z.dtype) - xhat * (s2 / n).to(z.dtype))
    return dz, s2.to(z.dtype), s1.to(z.dtype)


def upadd_ref(lo, skip):
    return lo.repeat_interleave(2, 1).repeat_interleave(2, 2) + skip


def upadd_bwd_ref(dy):
    n, h, w, c = dy.shape
    return dy.reshape(n, h // 2, 2, w // 2, 2, c).sum((2, 4))


class Interp:
    def __init__(self, plan, sd, n):
        self.P, self.n = plan, n
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.data = {id(b): torch.zeros(n, b.h, b.w, b.c) for b in plan.bufs}
        self.gbuf = {id(g): torch.zeros(n, g.h, g.w, g.c) for g in plan.gbufs}
        self.grads = {k: torch.zeros_like(self.sd[k]) for k in plan.trainable}
        self.saved, self.logits, self.dlogits = {}, {}, {}

    def view(self, v):
        return vslice(self.data[id(v.buf)] if id(v.buf) in self.data else self.gbuf[id(v.buf)], v)

    def forward(self, imgs_u8):
        sd = self.sd
        for op in self.P.fwd:
            if op.kind == "conv0":
                x = imgs_u8.to(torch.get_default_dtype()) / 255.0
                p = op.pad
                y = nhwc(F.conv2d(F.pad(nchw(x), (p, p, p, p)), sd[op.wkey]))
                self.view(op.y).copy_(y)
                self.imgs = x
            elif op.kind == "conv":
                res = None if op.res is None else self.view(op.res)
                self.view(op.y).copy_(conv_fwd_ref(op, self.view(op.x), sd[op.wkey], res))
            elif op.kind == "bnrelu":
                k = op.bnkey
                a, mean, rstd = bn_fwd_ref(self.view(op.z), sd[k + ".weight"], sd[k + ".bias"], sd[k + ".running_mean"], sd[k + ".running_var"])
                self.view(op.a).copy_(a)
                self.saved[k] = (mean, rstd)
            elif op.kind == "upadd":
                self.view(op.y).copy_(upadd_ref(self.view(op.lo), self.view(op.skip)))
            elif op.kind == "head":
                self.logits[op.branch] = F.conv2d(nchw(self.view(op.x)), sd[op.wkey], sd[op.bkey])
            else:  # pragma: no cover
                raise KeyError(op.kind)
        return self.logits

    def backward(self, dlogits):
        sd = self.sd
        for gb in self.P.gbufs:                      # the engine clears only the buffers the plan marks (train_plan._first_writers):
            self.gbuf[id(gb)].fill_(0.0 if getattr(gb, "zero", True) else float("nan"))    # whatever the others hold must never be read
        for g in self.grads.values():
            g.zero_()
        for op in self.P.bwd:
            if op.kind == "head_bwd":
                dl = dlogits[op.branch]                                        # [N,C,h,w]
                a = self.view(op.x)
                self.grads[op.wkey] += torch.einsum("nchw,nhwk->ck", dl, a).reshape(sd[op.wkey].shape)
                self.grads[op.bkey] += dl.sum((0, 2, 3))
                self.view(op.dx).add_(torch.einsum("nchw,ck->nhwk", dl, sd[op.wkey][:, :, 0, 0]))
            elif op.kind == "bnrelu_bwd":
                k = op.bnkey
                mean, rstd = self.saved[k]
                dz, dg, db = bn_bwd_ref(self.view(op.z), self.view(op.a), self.view(op.da), sd[k + ".weight"], mean, rstd)
                self.grads[k + ".weight"] += dg
                self.grads[k + ".bias"] += db
                if op.dz is not None:
                    self.view(op.dz).copy_(dz) if getattr(op, "store", False) else self.view(op.dz).add_(dz)
            elif op.kind == "wgrad":
                self.grads[op.wkey] += wgrad_ref(op, self.view(op.x), self.view(op.dy), sd[op.wkey].shape)
            elif op.kind == "dgrad":
                dx = dgrad_ref(op, self.view(op.dy), sd[op.wkey], (op.dx.h, op.dx.w))
                self.view(op.dx).copy_(dx) if getattr(op, "store", False) else self.view(op.dx).add_(dx)
            elif op.kind == "upadd_bwd":
                dy = self.view(op.dy)
                if op.dlo is not None:
                    self.view(op.dlo).add_(upadd_bwd_ref(dy))
                if op.dskip is not None:
                    self.view(op.dskip).add_(dy)
            elif op.kind == "conv0_wgrad":
                p = op.pad
                xp = F.pad(nchw(self.imgs), (p, p, p, p))
                self.grads[op.wkey] += torch.nn.grad.conv2d_weight(xp, sd[op.wkey].shape, nchw(self.view(op.dy)).contiguous())
            else:  # pragma: no cover
                raise KeyError(op.kind)
        return self.grads
