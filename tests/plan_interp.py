"""Torch-CPU interpreter of a `hover_net_amd.plan.Plan` -- TEST INFRASTRUCTURE ONLY.

Executes the very same op list the product hands to libhvn_hip.so, but with plain
torch fp32 ops on NHWC tensors, honouring every view / stride / prologue / epilogue
field.  Used (a) on CPU to prove the lowering (BN folding, concat-by-offset, crops,
block-diagonal grouped convs, arena packing) against oracle/net_torch.py, and (b) on
the GPU box as the per-op reference for the HIP kernels.
"""
import torch
import torch.nn.functional as F

from hover_net_amd import plan as PL


class Arena:
    """Backs every plan buffer by a slice of ONE flat tensor per sample-batch, at the
    offsets Plan.pack() assigned -- so lifetime-overlap bugs in the packing show up."""

    def __init__(self, plan, n):
        self.n = n
        self.flat = torch.full((n, plan.arena_per_sample), float("nan"))
        self.ext = {}

    def tensor(self, buf):
        if buf.offset < 0:  # image / logits / pred_map live outside the arena
            if buf.name not in self.ext:
                self.ext[buf.name] = torch.full((self.n, buf.h, buf.w, buf.c), float("nan"))
            return self.ext[buf.name]
        return self.flat[:, buf.offset:buf.offset + buf.size].view(self.n, buf.h, buf.w, buf.c)

    def view(self, v):
        t = self.tensor(v.buf)
        return t[:, v.y0:v.y0 + v.h, v.x0:v.x0 + v.w, v.c0:v.c0 + v.c]


def conv_ref(op, x, res=None, x2=None):
    """x: [N,h,w,cin] NHWC view tensor -> [N,ho,wo,cout].  x2: optional second 1x1 input (strided)."""
    if op.pre is not None:
        x = F.relu(x * torch.from_numpy(op.pre[0]) + torch.from_numpy(op.pre[1]))
    if x2 is not None:
        s2 = op.extra["stride2"]
        x = torch.cat([x, x2[:, ::s2, ::s2][:, :x.shape[1], :x.shape[2]]], -1)
    w = torch.from_numpy(PL.unpack_conv(op.w, op.cout))
    w = w.view(op.cout, w.shape[1], op.kh, op.kw)
    xin = x.permute(0, 3, 1, 2)
    ho, wo = op.y.h, op.y.w
    pad_b = (ho - 1) * op.stride + op.kh - x.shape[1] - op.pad_t
    pad_r = (wo - 1) * op.stride + op.kw - x.shape[2] - op.pad_l
    xin = F.pad(xin, (op.pad_l, max(pad_r, 0), op.pad_t, max(pad_b, 0)))
    y = F.conv2d(xin, w, stride=op.stride)[:, :, :ho, :wo].permute(0, 2, 3, 1)
    if op.bias is not None:
        y = y + torch.from_numpy(op.bias)
    if op.relu:
        y = F.relu(y)
    if res is not None:
        y = y + res
    if op.post is not None:
        y = F.relu(y * torch.from_numpy(op.post[0]) + torch.from_numpy(op.post[1]))
    return y


def chain_ref(op, x, res=None, x2=None):
    """OP_CHAIN from its OWN fields (include/hvn.h): y = post(W.[x | x2] + res); y2 = relu(W2.a + bias2), a = relu(y*pre) or y."""
    if x2 is not None:
        s2 = op.extra["stride2"]
        x = torch.cat([x, x2[:, ::s2, ::s2][:, :x.shape[1], :x.shape[2]]], -1)
    w = torch.from_numpy(PL.unpack_conv(op.w, op.cout))[:, :, 0]            # [cout, k]
    y = x @ w.t()
    if res is not None:
        y = y + res
    if op.post is not None:
        y = F.relu(y * torch.from_numpy(op.post[0]) + torch.from_numpy(op.post[1]))
    a = y
    if op.pre is not None:
        a = F.relu(y * torch.from_numpy(op.pre[0]) + torch.from_numpy(op.pre[1]))
    w2 = torch.from_numpy(PL.unpack_conv(op.extra["w2"], op.extra["cout2"]))[:, :, 0]
    y2 = a @ w2.t()
    if op.extra.get("bias2") is not None:
        y2 = y2 + torch.from_numpy(op.extra["bias2"])
    return y, F.relu(y2)


def conv0_ref(op, img_u8):
    x = img_u8.float().permute(0, 3, 1, 2)
    w = torch.from_numpy(op.w).permute(3, 2, 0, 1)  # [7,7,3,64] -> [64,3,7,7]
    x = F.pad(x, (op.pad_l, op.pad_l, op.pad_t, op.pad_t))
    y = F.conv2d(x, w) + torch.from_numpy(op.bias).view(1, -1, 1, 1)
    return F.relu(y).permute(0, 2, 3, 1)


def upadd_ref(lo, skip):
    return lo.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2) + skip


def head_ref(op, x):
    y = x @ torch.from_numpy(op.w).t() + torch.from_numpy(op.bias)
    return y.permute(0, 3, 1, 2).contiguous()  # NCHW logits


def predmap_ref(logits, branches):
    ch = []
    if "tp" in branches:
        ch.append(torch.argmax(logits["tp"], dim=1).unsqueeze(-1).float())
    l = logits["np"]
    ch.append(torch.softmax(l, dim=1)[:, 1].unsqueeze(-1))
    ch.append(logits["hv"].permute(0, 2, 3, 1))
    return torch.cat(ch, -1)


def wino_in_ref(op, x):
    """x [N,h,w,c] -> V [N,n*n,T,c] with T = ty*tx tiles (row-major), n = m + 4."""
    ty, tx = op.extra["tiles"]
    m = op.extra["m"]
    n_ = m + op.extra.get("r", 5) - 1
    bt = torch.from_numpy(op.w)                       # [n,n]
    pt = op.pad_t
    need_h, need_w = m * ty + (n_ - m), m * tx + (n_ - m)
    xp = F.pad(x.permute(0, 3, 1, 2), (pt, need_w - x.shape[2] - pt, pt, need_h - x.shape[1] - pt))
    tiles = xp.unfold(2, n_, m).unfold(3, n_, m)      # [N,c,ty,tx,n,n]
    v = torch.einsum("ai,nctsij,bj->nabtsc", bt, tiles, bt)
    return v.reshape(x.shape[0], n_ * n_, ty * tx, x.shape[3])


def wino_gemm_ref(op, v):
    """v [N,n2,T,cin] -> m [N,n2,T,cout]."""
    w = torch.from_numpy(op.w[:, :op.cout]).reshape(op.w.shape[0], op.cout, -1)   # [n2,cout,cin]
    return torch.einsum("nxtc,xoc->nxto", v, w)


def wino_out_ref(op, mm):
    """mm [N,n2,T,cout] -> y [N,ho,wo,cout] (partial last tiles cropped)."""
    ty, tx = op.extra["tiles"]
    m = op.extra["m"]
    n_ = m + op.extra.get("r", 5) - 1
    at = torch.from_numpy(op.w)                        # [m,n]
    nb, _, _, co = mm.shape
    mm = mm.reshape(nb, n_, n_, ty, tx, co)
    y = torch.einsum("pa,nabtsc,qb->ntpsqc", at, mm, at).reshape(nb, m * ty, m * tx, co)[:, :op.y.h, :op.y.w]
    if op.bias is not None:
        y = y + torch.from_numpy(op.bias)
    if op.relu:
        y = F.relu(y)
    return y


def run(plan, imgs_u8, taps=None):
    """imgs_u8: uint8 [N,H,W,3] tensor -> (logits dict NCHW, pred_map or None)."""
    n = imgs_u8.shape[0]
    A = Arena(plan, n)
    logits = {}
    pred = None
    with torch.no_grad():
        for op in plan.ops:
            if op.kind == PL.OP_CONV0:
                A.view(op.y).copy_(conv0_ref(op, imgs_u8))
            elif op.kind == PL.OP_WINO_IN:
                xin = A.view(op.x).clone()
                if op.res is not None:            # UPADD fused into the transform: input = nearest2x(res) + x
                    xin = upadd_ref(A.view(op.res), xin)
                A.view(op.y).copy_(wino_in_ref(op, xin))
            elif op.kind == PL.OP_CONV and op.extra.get("nbatch"):
                A.tensor(op.y.buf).copy_(wino_gemm_ref(op, A.tensor(op.x.buf).clone()))
            elif op.kind == PL.OP_WINO_OUT:
                A.view(op.y).copy_(wino_out_ref(op, A.view(op.x).clone()))
            elif op.kind == PL.OP_CONV:
                res = A.view(op.res).clone() if op.res is not None else None
                x2 = A.view(op.extra["x2"]).clone() if op.extra.get("x2") is not None else None
                A.view(op.y).copy_(conv_ref(op, A.view(op.x).clone(), res, x2))
            elif op.kind == PL.OP_CHAIN:
                res = A.view(op.res).clone() if op.res is not None else None
                x2 = A.view(op.extra["x2"]).clone() if op.extra.get("x2") is not None else None
                y, y2 = chain_ref(op, A.view(op.x).clone(), res, x2)
                A.view(op.y).copy_(y)
                A.view(op.extra["y2"]).copy_(y2)
            elif op.kind == PL.OP_UPADD:
                A.view(op.y).copy_(upadd_ref(A.view(op.x), A.view(op.res)))
            elif op.kind == PL.OP_HEAD:
                logits[op.y.buf.name.split(".")[1]] = head_ref(op, A.view(op.x))
            elif op.kind == PL.OP_PREDMAP:
                pred = predmap_ref(logits, op.extra["branches"])
            if taps is not None and op.name in taps:
                taps[op.name] = A.view(op.y).clone()
            assert op.kind in (PL.OP_HEAD, PL.OP_PREDMAP) or not torch.isnan(A.view(op.y)).any(), op.name
    return logits, pred
