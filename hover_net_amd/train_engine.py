"""Binds a training plan (hover_net_amd.train_plan) to HBM and runs the training step through libhvn_hip.so
(/root/reference/models/hovernet/run_desc.py:12-109).

Memory (sized for 288 GB, nothing is recomputed):
* one parameter slab and one gradient slab with the SAME layout -- every trainable tensor of the module is
  re-pointed at its slice (conv weights in channels_last order [cout][kh][kw][cin_g], which is what the weight-
  gradient kernel writes and the packers read), so `optimizer.step()` (torch's Adam or optim.FusedAdam, one launch
  over the slab) updates the weights the kernels use, and a data-parallel run all-reduces ONE flat tensor;
* one activation arena (tensor-major: each tensor holds the whole batch) and one gradient arena, contiguous with
  the gradient slab so that a single memset clears every accumulation target of a step;
* per-step packed copies of the weights for the conv kernel (forward and data-gradient forms).

torch is used for allocation, streams and `torch.distributed` (RCCL) only.  There is no CPU fallback.
"""
import ctypes
import os

import torch

from . import arch
from . import lib as L
from . import train_plan as TP
from . import winograd as WG
from .plan import OP_CONV, OP_CONV0, OP_HEAD, OP_UPADD, OP_WINO_IN, OP_WINO_OUT, _tile_n

T_NET, T_PACK_W, T_BN_FWD, T_BN_BWD, T_WGRAD, T_CONV0_WGRAD, T_UPADD_BWD, T_HEAD_BWD, T_WINO_DY, T_WINO_DW = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
T_SPLIT_X3, T_PACK_MULTI = 11, 12
MOMENTUM = 0.1


_TILE_CHOICE = {}       # (batch, launch shape) -> (choice, ms of the static choice, ms of the alternative): see TrainEngine.autotune_tiles
WGRAD_TARGETS = (1536, 1024, 768, 512, 384)     # workgroups a weight-gradient launch may aim at; the first is the untuned default


def static_wgrad_target(kh, kw):
    """Workgroups a DETERMINISTIC weight-gradient launch aims at (hvn_top.mode): a function of the launch SHAPE only, because a timed
    choice is a summation order that depends on the box and the run.  Read off `profiles/r06_wgrad_static_rule.txt` (tools/
    wgrad_static_rule.py: every weight-gradient launch of phase 0 / phase 1 / the fit's 'fast' batch 8 timed at 7 targets through the
    deterministic path): the 1x1 launches -- most of them, bf16x3 tiles of 128 x 128 -- are fastest at 768 on all three plans, the
    multi-tap ones (3x3, grouped 5x5) at 1024 - 1536; the launcher's own default of 1536 was tuned for the atomic form.  Sum over the
    launches of a step: 15.2 / 14.3 / 26.3 ms with this rule against 15.3 / 15.4 / 26.6 at 1536 everywhere and 14.5 / 13.7 / 24.6 with
    every launch at its own best."""
    return 1024 if kh * kw > 1 else 768


def _tune_reps(default):
    """Timings per candidate of the engines' timing passes (HVN_TUNE_REPS; tests/conftest.py sets 1: every candidate is bit-identical)."""
    return max(1, int(os.environ.get("HVN_TUNE_REPS", default)))


_STREAMS = {}


def _shared_streams(device, role, count):
    """The step's side streams, ONE set per process and device shared by every engine (engines run from one host thread, never
    concurrently): HIP multiplexes streams onto a few hardware queues, and a process that makes five new streams per engine -- a fit, then
    phase 0, then phase 1 -- ends up with two streams of one step on the same queue, i.e. without the concurrency they exist for (measured:
    phase 1 inside bench.py after the bench's own fit 41.4 ms, in a fresh process 37.6)."""
    have = _STREAMS.setdefault((str(device), role), [])
    while len(have) < count:
        have.append(torch.cuda.Stream(device=device))
    return have[:count]


def _align(n, a=64):
    return (n + a - 1) // a * a


class TrainEngine:
    def __init__(self, net, batch, device=None, share_shapes=True, deterministic=None):
        """deterministic (default: the module's `train_deterministic` attribute, else HVN_TRAIN_DETERMINISTIC, else True): every
        cross-workgroup sum of the step -- the weight gradients' split of the pixel sum, conv0's and the heads' weight gradients, the
        loss sums -- goes through per-workgroup partial results that a second launch adds in a fixed order
        (`hvn_run_train_plan_ws`, `hvn_loss.partials`) instead of fp32 / double atomics, and the weight-gradient split is the static
        one instead of a timed choice: the same step on the same data gives the same bits, run to run and box to box.
        share_shapes: data-parallel runs broadcast rank 0's measured launch shapes at the END of this constructor (one small
        collective, `_share_launch_shapes`): only legal when EVERY rank builds an engine at this point.  `engine_for` passes True for a
        module's first engine (every rank builds it at its first step) and False for rebuilds, which one rank may do alone."""
        L.require_gpu()
        self._share_shapes = bool(share_shapes)
        self.net = net
        if deterministic is None:
            deterministic = getattr(net, "train_deterministic", None)
        if deterministic is None:
            deterministic = os.environ.get("HVN_TRAIN_DETERMINISTIC", "1") != "0"
        self.deterministic = bool(deterministic)
        self.n = int(batch)
        self.device = torch.device(device) if device is not None else next(net.parameters()).device
        self.plan = P = TP.TrainPlan(net.mode, net.nr_types, net.freeze)
        self._keep = []                       # ctypes objects referenced by pointer from the op arrays
        self._params = dict(net.named_parameters())
        self._buffers = dict(net.named_buffers())
        self._build_slabs()
        data_elems, garena_elems = P.layout(self.n)
        dev = self.device
        self.arena = torch.empty(max(data_elems, 64), dtype=torch.float32, device=dev)
        # 5x5 stride-1 convs (decoder conva, 51 % of the forward FLOPs) run as Winograd F(4x4,5x5) like the inference
        # plan (plan.py:conv_winograd) in all three passes: forward, data gradient, and the weight gradient in the
        # transform domain; HVN_TRAIN_WINOGRAD=0 keeps them direct
        self.use_wino = os.environ.get("HVN_TRAIN_WINOGRAD", "1") != "0"
        self._du_off, du_total = {}, 0
        for key, c in P.convs.items():
            if self._is_wino(c) and c["train"]:
                self._du_off[key] = du_total
                du_total += _align(64 * c["cout"] * c["cin_g"])
        # gradient slab + Winograd-domain weight gradients + gradient arena in one allocation; ONE memset per step over the part that
        # accumulates: the slab, the transform-domain gradients and the arena's first `gzero_elems` -- the rest of the arena are buffers
        # whose first writer of the step overwrites them completely (train_plan._first_writers; round 6: 73 % of the arena in phase 1).
        # HVN_TRAIN_FIRST_STORE=0: every backward launch accumulates and everything is cleared (rounds 3-5).
        self.first_store = os.environ.get("HVN_TRAIN_FIRST_STORE", "1") != "0"
        self._gtotal = self._slab_elems + garena_elems + du_total
        self.gmem = torch.zeros(self._gtotal, dtype=torch.float32, device=dev)
        self.gslab = self.gmem[:self._slab_elems]
        self.wino_du = self.gmem[self._slab_elems:self._slab_elems + du_total]
        self.garena = self.gmem[self._slab_elems + du_total:]
        self._gzero = self.gmem[:self._slab_elems + du_total + P.gzero_elems] if self.first_store else self.gmem
        self._point_grads()
        g = P.geo
        self.img = torch.empty((self.n, g["inp"], g["inp"], 3), dtype=torch.uint8, device=dev)
        ho = g["out"]
        self.logits = {b: torch.empty((self.n, c, h, w), dtype=torch.float32, device=dev) for b, (c, h, w) in P.logits.items()}
        self.dlogits = {b: torch.zeros_like(v) for b, v in self.logits.items()}
        self.true_np = torch.zeros((self.n, ho, ho), dtype=torch.int32, device=dev)
        self.true_tp = torch.zeros((self.n, ho, ho), dtype=torch.int32, device=dev)
        self.true_hv = torch.zeros((self.n, ho, ho, 2), dtype=torch.float32, device=dev)
        self.sums = torch.zeros(64, dtype=torch.float64, device=dev)
        self.sobel_ws = torch.empty((self.n, ho, ho, 2), dtype=torch.float32, device=dev)
        cmax = max(P.bns.values())
        # Branch streams (round 6): the decoder branches (np / hv / tp, net_desc.py:77-97) are independent between the encoder's output and
        # the loss, and at the reference's batch of 4 per GPU (opt.py:75-76) one branch's launches fill a fraction of the 256 CUs -- each
        # branch's section of the forward and of the backward list runs on its own HIP stream, with its own scratch set (BN partial sums
        # and coefficients, transform-domain tensors, the deterministic-reduce workspace).  What the branches ADD to shared gradient
        # buffers (the skips' and conv_bot's output gradients, `upadd_bwd`) is deferred to after the join and runs in the single-stream
        # order: same sums in the same order, same bits (tests/test_gpu_train.py).  HVN_TRAIN_BRANCH_STREAMS=0: one stream.
        self.branches = arch.branch_names(net.nr_types)
        self.branch_streams = os.environ.get("HVN_TRAIN_BRANCH_STREAMS", "1") != "0"
        self._nsets = len(self.branches) if self.branch_streams else 1
        self._side = _shared_streams(dev, "side", self._nsets - 1)
        # Weight-gradient streams (round 6): a conv's weight gradient and its data gradient are independent, and nothing in the backward
        # pass waits for a weight gradient -- every plain `wgrad` launch floats on a second stream of its section, from the moment its
        # output gradient is complete (an event) to the first later launch that writes that gradient's buffer again (`_floats`: the
        # residual sums and the dense blocks' concats are accumulated further), at the latest the section's join.  Own deterministic-
        # reduce workspace per stream; same launches, same bits.  Measured (profiles/r06_wgrad_stream_ab.txt): phase 1 (batch 4, every
        # layer trains) 42.8 -> 39.2 ms, phase 0 (batch 16, frozen encoder: launches that fill the chip alone) 61.3 -> 62.9 ms -- so the
        # default `HVN_TRAIN_WGRAD_STREAM=auto` times the backward list both ways once at engine build (`_choose_wgrad_stream`; the bits
        # do not depend on the answer); 1 / 0 force it on / in list order on the section's stream.
        self._wgrad_mode = os.environ.get("HVN_TRAIN_WGRAD_STREAM", "auto")
        self.wgrad_stream = self._wgrad_mode != "0"
        self._wside = _shared_streams(dev, "wgrad", self._nsets) if self.wgrad_stream else []
        self.bn_ws = [torch.zeros(256 * 2 * cmax, dtype=torch.float64, device=dev) for _ in range(self._nsets)]     # HVN_BN_MAX_PARTS partial sums
        self.bn_coef = [torch.empty(3 * cmax, dtype=torch.float32, device=dev) for _ in range(self._nsets)]
        self.bn_save = torch.empty(sum(4 * c for c in P.bns.values()), dtype=torch.float32, device=dev)
        self._bn_save_off, off = {}, 0
        for k, c in P.bns.items():
            self._bn_save_off[k] = off
            off += 4 * c
        self.zero_bias = torch.zeros(64, dtype=torch.float32, device=dev)
        at, gm, bt = WG.MATS[4]
        self.wino_mats = torch.tensor(list(bt.reshape(-1)) + list(at.reshape(-1)) + list(gm.reshape(-1)), dtype=torch.float32, device=dev)
        self._bt_ptr, self._at_ptr, self._g_ptr = (self.wino_mats.data_ptr() + 4 * o for o in (0, 64, 96))
        self._alloc_packs()
        self._alloc_wino_scratch()
        self._nbt = [self._buffers[k + ".num_batches_tracked"] for k in P.bns]
        fwd_groups = [self._pack_ops()] + [self._lower_fwd(op) for op in P.fwd]
        self.fwd_ops = self._lower(fwd_groups)
        self._fwd_runs = self._runs([-1] + [self._set_of(op.name, True) for op in P.fwd], fwd_groups)
        bwd_groups, deferred = [], []
        for op in P.bwd:
            if op.kind == "upadd_bwd" and self.branch_streams:
                mine, later = self._lower_upadd_bwd_split(op)
                bwd_groups.append(mine)
                deferred += later
            else:
                bwd_groups.append(self._lower_bwd(op))
        # gradient buckets for data-parallel training: the backward list finishes the decoder branches first and their
        # parameters are the tail of the slab, so their all-reduce can run under the encoder's backward pass
        first_enc = next((i for i, op in enumerate(P.bwd) if not op.name.startswith("decoder.")), len(P.bwd))
        self._bwd_split = sum(len(g) for g in bwd_groups[:first_enc])
        tags = [self._set_of(op.name, True) for op in P.bwd]
        bwd_groups.insert(first_enc, deferred)          # the branches' sums into shared buffers: after the join, in list order
        tags.insert(first_enc, -1)
        self.bwd_ops = self._lower(bwd_groups)
        self._bwd_runs = self._runs(tags, bwd_groups)
        self._floats = self._floating_wgrads(bwd_groups, first_enc) if self.wgrad_stream else {}
        assert not self.branch_streams or all(hi <= self._bwd_split for k, lo, hi in self._bwd_runs if k >= 0)
        self._use_x3()
        dec_keys = [k for k in self._poff if k.startswith("decoder.")]
        self._dec_off = min(self._poff[k] for k in dec_keys)
        self.det_ws = self.loss_parts = None
        if self.deterministic:
            lib = L.lib()
            need = max(int(lib.hvn_train_workspace_bytes(ctypes.addressof(self.bwd_ops), len(self.bwd_ops), self.n)), 64)
            self.det_ws = [torch.empty((need + 3) // 4, dtype=torch.float32, device=dev) for _ in range(self._nsets)]
            self.det_ws_w = [torch.empty((need + 3) // 4, dtype=torch.float32, device=dev) for _ in self._wside]
            ho = P.geo["out"]
            self.loss_parts = torch.empty(max(int(lib.hvn_loss_partials_count(self.n, ho, ho)), 64), dtype=torch.float64, device=dev)
        self._loss = self._loss_desc()
        self.last_terms = None
        self.autotune_tiles()
        self._choose_wgrad_stream()

    # -- parameter / gradient slabs ------------------------------------------------------------------
    def _build_slabs(self):
        """Flat fp32 slab holding every float parameter in state_dict order; conv weights channels_last."""
        P = self.plan
        offs, total = {}, 0
        for key, (kind, shape) in P.table.items():
            if kind in ("conv", "bias", "bn_w", "bn_b"):
                numel = 1
                for s in shape:
                    numel *= s
                offs[key] = total
                total += _align(numel)
        self._slab_elems = total
        self._poff = offs
        self.wslab = torch.zeros(total, dtype=torch.float32, device=self.device)
        for key, off in offs.items():
            p = self._params[key]
            view = self._param_view(self.wslab, key, off)
            view.copy_(p.data.to(self.device))
            p.data = view

    def _param_view(self, slab, key, off):
        kind, shape = self.plan.table[key]
        if kind == "conv":
            cout, cin_g, kh, kw = shape
            return torch.as_strided(slab, shape, (kh * kw * cin_g, 1, kw * cin_g, cin_g), off)
        numel = 1
        for s in shape:
            numel *= s
        return slab[off:off + numel].view(shape)

    def _point_grads(self):
        for key, off in self._poff.items():
            p = self._params[key]
            p.grad = self._param_view(self.gslab, key, off) if key in self.plan.trainable else None

    def wptr(self, key):
        return self.wslab.data_ptr() + 4 * self._poff[key]

    def gptr(self, key):
        return self.gslab.data_ptr() + 4 * self._poff[key]

    # -- packed weights --------------------------------------------------------------------------------
    def _is_wino(self, c):
        return self.use_wino and c["kh"] == 5 and c["kw"] == 5 and c["groups"] == 1

    def _alloc_packs(self):
        """Per conv: the packed forward weights (mode 0) and, if its input needs a gradient, the transposed /
        flipped data-gradient weights (mode 1); Winograd convs hold the transformed U instead (modes 3 / 4)."""
        P = self.plan
        self._pack_off, total = {}, 0
        for key, c in P.convs.items():
            cin = c["cin_g"] * c["groups"]
            taps = 64 if self._is_wino(c) else c["kh"] * c["kw"]
            fm, dm = (3, 4) if self._is_wino(c) else (0, 1)
            lead = _align(c["cout"], _tile_n(c["cout"]))
            self._pack_off[(key, fm)] = (total, lead)
            total += _align(lead * cin * taps)
            if c["dgrad"]:
                lead = _align(cin, _tile_n(cin))
                self._pack_off[(key, dm)] = (total, lead)
                total += _align(lead * c["cout"] * taps)
        self._pack_off[("conv0./.weight", 2)] = (total, 64)
        total += _align(7 * 7 * 3 * 64)
        self.packs = torch.zeros(total, dtype=torch.float32, device=self.device)
        # bf16x3 (csrc/hvn_conv_x3.hip): the forward / data-gradient convs form their products on the bf16 matrix pipe from exact
        # three-way bf16 splits; the planes of the step's weight packings are made on the device right after the packing
        # (HVN_TRAIN_X3 = 6 (default) | 9 partial products per product, 0 = every conv on the fp32 pipe)
        self.x3_terms = int(os.environ.get("HVN_TRAIN_X3", "6"))
        # round 5: the weight gradients too (csrc/hvn_wgrad_x3.hip: both operands split on the fly; HVN_TRAIN_WGRAD_X3=0 keeps the fp32 pipe)
        wg = os.environ.get("HVN_TRAIN_WGRAD_X3", "1")            # 1: as the forward convs | 6 | 9 | 0: fp32 pipe
        self.wgrad_x3 = 0 if (wg == "0" or not self.x3_terms) else (int(wg) if wg in ("6", "9") else self.x3_terms)
        self.packs_x3 = torch.zeros(3 * total, dtype=torch.int16, device=self.device) if self.x3_terms else None

    def pack_ptr(self, key, mode):
        return self.packs.data_ptr() + 4 * self._pack_off[(key, mode)][0]

    def _pack_ops(self):
        """The step's weight packings: every plain (forward / data-gradient / conv0) packing in ONE launch over a device table
        (HVN_T_PACK_MULTI; HVN_TRAIN_PACK_MULTI=0 keeps one launch per packing), the Winograd transforms one launch each."""
        import numpy as np

        ops, table, blocks = [], [], [0]
        multi = os.environ.get("HVN_TRAIN_PACK_MULTI", "1") != "0"
        for (key, mode), (off, lead) in self._pack_off.items():
            if mode == 2:
                cout, cin_g, groups, kh, kw = 64, 3, 1, 7, 7
            else:
                c = self.plan.convs[key]
                cout, cin_g, groups, kh, kw = c["cout"], c["cin_g"], c["groups"], c["kh"], c["kw"]
            if multi and mode in (0, 1, 2):
                d = L.hvn_pack_desc()
                d.src, d.dst = self.wptr(key), self.packs.data_ptr() + 4 * off
                d.cout, d.cin_g, d.groups, d.taps, d.mode, d.lead_pad = cout, cin_g, max(1, groups), kh * kw, mode, lead
                total = 64 * kh * kw * 3 if mode == 2 else (lead * cin_g * max(1, groups) * kh * kw if mode == 0 else lead * cout * kh * kw)
                # HVN_T_PACK_MULTI runs the table on the device without HVN_T_PACK_W's per-call checks: the same checks, here
                cin = cin_g * max(1, groups)
                rows = cout if mode == 0 else cin
                if (not d.src or not d.dst or d.dst % 16 or (mode == 0 and cin % 32) or (mode == 1 and cout % 32) or
                        (mode != 2 and lead < rows) or total <= 0):
                    raise ValueError("weight packing %s mode %d: cout %d cin %d groups %d lead_pad %d is not a shape hvn_pack_w_multi may run"
                                     % (key, mode, cout, cin, groups, lead))
                table.append(d)
                blocks.append(blocks[-1] + (total + 255) // 256)
                continue
            t = L.hvn_top()
            t.kind, t.mode, t.lead_pad = T_PACK_W, mode, lead
            t.cout, t.cin_g, t.groups, t.kh, t.kw = cout, cin_g, groups, kh, kw
            t.p[0] = self.wptr(key)
            t.p[1] = self.packs.data_ptr() + 4 * off
            t.p[2] = self._g_ptr
            ops.append(t)
        if table:
            arr = (L.hvn_pack_desc * len(table))(*table)
            raw = np.frombuffer(arr, dtype=np.uint8).copy()
            self._pack_table = torch.from_numpy(raw).to(self.device)
            self._pack_first = torch.tensor(blocks, dtype=torch.int32, device=self.device)
            t = L.hvn_top()
            t.kind, t.cout = T_PACK_MULTI, len(table)
            t.p[0], t.p[1] = self._pack_table.data_ptr(), self._pack_first.data_ptr()
            t.batch_stride[0] = blocks[-1]
            ops.insert(0, t)
        if self.x3_terms:
            t = L.hvn_top()
            t.kind = T_SPLIT_X3
            t.p[0], t.p[1] = self.packs.data_ptr(), self.packs_x3.data_ptr()
            t.batch_stride[0] = self.packs.numel() // 32
            ops.append(t)
        return ops

    def _use_x3(self):
        """Point every eligible conv launch (dense, 128- or 64-wide column tile, weights in the step's packing buffer) at the bf16
        planes of its weights and at csrc/hvn_conv_x3.hip (act_dtype 3 | 2).  Grouped convs, conv0 and the heads stay as they are."""
        if not self.x3_terms:
            return
        lo, hi = self.packs.data_ptr(), self.packs.data_ptr() + 4 * self.packs.numel()
        for o in self._keep:
            if not isinstance(o, L.hvn_op) or o.kind != OP_CONV or o.groups > 1 or o.tile_n not in (128, 64) or not o.w or not (lo <= o.w < hi):
                continue
            off = (o.w - lo) // 4
            o.w = self.packs_x3.data_ptr() + 2 * 3 * off
            o.act_dtype = 3 if self.x3_terms == 6 else 2
            if o.nbatch > 1:
                o.batch_stride[1] = 3 * o.batch_stride[1]

    def _alloc_wino_scratch(self):
        """Transform-domain tensors: every trained Winograd conv keeps its forward V (the weight gradient reads it
        again); one more V and one M are shared by the data gradients / products of the step."""
        v = m = 64
        self.wino_vs = {}
        for op in self.plan.fwd:
            if op.kind == "conv" and self._is_wino(self.plan.convs[op.wkey]):
                t = -(-op.y.h // 4) * -(-op.y.w // 4)
                m = max(m, 64 * t * op.y.c)
                if op.train:
                    self.wino_vs[op.wkey] = torch.empty(self.n * 64 * t * op.x.c, dtype=torch.float32, device=self.device)
                else:
                    v = max(v, 64 * t * op.x.c)
                if op.dx:
                    t = -(-op.x.h // 4) * -(-op.x.w // 4)
                    v, m = max(v, 64 * t * op.y.c), max(m, 64 * t * op.x.c)
        self.wino_v = [torch.empty(self.n * v, dtype=torch.float32, device=self.device) for _ in range(self._nsets)]
        self.wino_m = [torch.empty(self.n * m, dtype=torch.float32, device=self.device) for _ in range(self._nsets)]

    @staticmethod
    def _tview(ptr, t1, h, c):
        """[64][tiles][c]-per-sample transform-domain tensor as a view (h = 64: all positions, h = 1: one position)."""
        v = L.hvn_view()
        v.base, v.sn, v.sy, v.sx, v.h, v.w, v.c, v.sc = ptr, 64 * t1 * c, t1 * c, c, h, t1, c, 1
        return v

    def _set_of(self, name, tag=False):
        """Scratch set / stream of an op: the decoder branch's index with branch streams on, else 0 (tag=True: -1 for ops outside the
        decoder branches -- the sections of `_runs`)."""
        if self.branch_streams and name.startswith("decoder."):
            return self.branches.index(name.split(".")[1])
        return -1 if tag else 0

    @staticmethod
    def _runs(tags, groups):
        """[(tag, first launch, end launch)] for the maximal runs of equal tags over the lowered groups."""
        runs, pos = [], 0
        for t, g in zip(tags, groups):
            if len(g):
                if runs and runs[-1][0] == t:
                    runs[-1][2] = pos + len(g)
                else:
                    runs.append([t, pos, pos + len(g)])
            pos += len(g)
        return [tuple(r) for r in runs]

    def _wino_conv(self, xin, yout, pad, wptr, lead, accumulate, vbuf=None, scr=0):
        """WINO_IN -> 64 batched GEMMs on the conv kernel -> WINO_OUT for one 5x5 stride-1 convolution of the view
        `xin` (hvn_view, zero padding `pad`) into the view `yout`; `scr`: the scratch set (branch stream) it runs on."""
        ty, tx = -(-yout.h // 4), -(-yout.w // 4)
        t1, cin, cout = ty * tx, xin.c, yout.c

        def tview(ptr, h, c):
            return self._tview(ptr, t1, h, c)
        vp, mp = (self.wino_v[scr] if vbuf is None else vbuf).data_ptr(), self.wino_m[scr].data_ptr()
        ops = [self._net(kind=OP_WINO_IN, kh=ty, kw=tx, stride=4, _rsv=5, pad_t=pad, pad_l=pad, x=xin, y=tview(vp, 64, cin), w=self._bt_ptr)]
        g = dict(kind=OP_CONV, kh=1, kw=1, stride=1, pad_t=0, pad_l=0, relu=0, cout=cout, tile_n=_tile_n(cout), groups=1,
                 x=tview(vp, 1, cin), y=tview(mp, 1, cout), w=wptr, nbatch=64)
        t = self._net(**g)
        o = self._keep[-1]
        o.batch_stride[0], o.batch_stride[1], o.batch_stride[2] = t1 * cin, lead * cin, t1 * cout
        ops.append(t)
        kw = dict(kind=OP_WINO_OUT, kh=ty, kw=tx, stride=4, _rsv=5, relu=0, cout=cout, x=tview(mp, 64, cout), y=yout, w=self._at_ptr)
        if accumulate:
            kw["res"] = yout
        ops.append(self._net(**kw))
        return ops

    # -- views -------------------------------------------------------------------------------------------
    def _view(self, v):
        if v is None:
            return L.hvn_view()
        b = v.buf
        base = (self.garena if isinstance(b, TP.GBuf) else self.arena).data_ptr()
        s = L.hvn_view()
        s.base = base + 4 * (b.off + (v.y0 * b.w + v.x0) * b.c + v.c0)
        s.sn, s.sy, s.sx = b.h * b.w * b.c, v.step * b.w * b.c, v.step * b.c
        s.h, s.w, s.c, s.sc = v.h, v.w, v.c, 1
        return s

    def _net(self, **kw):
        o = L.hvn_op()
        for k, v in kw.items():
            setattr(o, k, v)
        self._keep.append(o)
        t = L.hvn_top()
        t.kind = T_NET
        t.net = ctypes.pointer(o)
        return t

    # -- lowering -----------------------------------------------------------------------------------------
    def _lower_fwd(self, op):
        if op.kind == "conv0":
            x = L.hvn_view()
            g = self.plan.geo["inp"]
            x.base, x.sn, x.sy, x.sx, x.h, x.w, x.c, x.sc = self.img.data_ptr(), g * g * 3, g * 3, 3, g, g, 3, 1
            return [self._net(kind=OP_CONV0, kh=7, kw=7, stride=1, pad_t=op.pad, pad_l=op.pad, relu=0, cout=64, x_dtype=0, x=x,
                              y=self._view(op.y), w=self.pack_ptr(op.wkey, 2),
                              bias=self.zero_bias.data_ptr())]
        if op.kind == "conv" and self._is_wino(self.plan.convs[op.wkey]) and op.stride == 1 and op.res is None:
            off, lead = self._pack_off[(op.wkey, 3)]
            return self._wino_conv(self._view(op.x), self._view(op.y), op.pad[0], self.packs.data_ptr() + 4 * off, lead, False,
                                   vbuf=self.wino_vs.get(op.wkey), scr=self._set_of(op.name))
        if op.kind == "conv":
            kw = dict(kind=OP_CONV, kh=op.kh, kw=op.kw, stride=op.stride, pad_t=op.pad[0], pad_l=op.pad[0], relu=0, cout=op.y.c,
                      tile_n=_tile_n(op.y.c), groups=op.groups, x=self._view(op.x), y=self._view(op.y),
                      w=self.pack_ptr(op.wkey, 0), nbatch=1)
            if op.res is not None:
                kw["res"] = self._view(op.res)
            return [self._net(**kw)]
        if op.kind == "bnrelu":
            t = L.hvn_top()
            t.kind = T_BN_FWD
            t.x, t.y = self._view(op.z), self._view(op.a)
            k = op.bnkey
            t.p[0] = self.bn_ws[self._set_of(op.name)].data_ptr()
            t.p[1] = self.bn_save.data_ptr() + 4 * self._bn_save_off[k]
            t.p[2], t.p[3] = self.wptr(k + ".weight"), self.wptr(k + ".bias")
            t.p[4], t.p[5] = self._buffers[k + ".running_mean"].data_ptr(), self._buffers[k + ".running_var"].data_ptr()
            t.eps, t.momentum = arch.BN_EPS, MOMENTUM
            return [t]
        if op.kind == "upadd":
            return [self._net(kind=OP_UPADD, x=self._view(op.lo), res=self._view(op.skip), y=self._view(op.y))]
        if op.kind == "head":
            y = L.hvn_view()
            y.base = self.logits[op.branch].data_ptr()
            y.h, y.w, y.c = op.x.h, op.x.w, op.cout
            return [self._net(kind=OP_HEAD, cout=op.cout, x=self._view(op.x), y=y,
                              w=self.wptr(op.wkey),
                              bias=self.wptr(op.bkey))]
        raise KeyError(op.kind)

    def _lower_bwd(self, op):
        t = L.hvn_top()
        if op.kind == "head_bwd":
            t.kind, t.cout = T_HEAD_BWD, op.cout
            t.x, t.dx = self._view(op.x), self._view(op.dx)
            t.p[0], t.p[1] = self.dlogits[op.branch].data_ptr(), self.wptr(op.wkey)
            t.p[2], t.p[3] = self.gptr(op.wkey), self.gptr(op.bkey)
            return [t]
        if op.kind == "bnrelu_bwd":
            k = op.bnkey
            t.kind = T_BN_BWD
            t.x, t.y, t.dy, t.dx = self._view(op.z), self._view(op.a), self._view(op.da), self._view(op.dz)
            t.p[0] = self.bn_ws[self._set_of(op.name)].data_ptr()
            t.p[1] = self.bn_save.data_ptr() + 4 * self._bn_save_off[k]
            t.p[2] = self.wptr(k + ".weight")
            t.p[3], t.p[4] = self.gptr(k + ".weight"), self.gptr(k + ".bias")
            t.p[5] = self.bn_coef[self._set_of(op.name)].data_ptr()
            t.mode = 1 if (self.first_store and op.store) else 0          # grad z = instead of += (first writer of the step)
            return [t]
        if op.kind == "wgrad" and op.wkey in self._du_off and op.stride == 1:
            # Winograd-domain weight gradient: dM = A dY A^T, dU[pos] = dM[pos]^T V[pos] (64 batched problems), dg += G^T dU G
            ty, tx = -(-op.dy.h // 4), -(-op.dy.w // 4)
            t1, cin, cout = ty * tx, op.x.c, op.dy.c
            du = self.wino_du.data_ptr() + 4 * self._du_off[op.wkey]
            t.kind, t.kh, t.kw = T_WINO_DY, ty, tx
            wm = self.wino_m[self._set_of(op.name)].data_ptr()
            t.x, t.y = self._view(op.dy), self._tview(wm, t1, 64, cout)
            t.p[0] = self._at_ptr
            g = L.hvn_top()
            g.kind, g.kh, g.kw, g.stride, g.groups, g.nbatch = T_WGRAD, 1, 1, 1, 1, 64
            g.mode = static_wgrad_target(1, 1) if self.deterministic else 0
            g._pad = self.wgrad_x3
            g.x = self._tview(self.wino_vs[op.wkey].data_ptr(), t1, 1, cin)
            g.dy = self._tview(wm, t1, 1, cout)
            g.p[0] = du
            g.batch_stride[0], g.batch_stride[1], g.batch_stride[2] = t1 * cin, t1 * cout, cout * cin
            w = L.hvn_top()
            w.kind, w.cout, w.cin_g = T_WINO_DW, cout, cin
            w.p[0], w.p[1], w.p[2] = du, self.gptr(op.wkey), self._g_ptr
            return [t, g, w]
        if op.kind == "wgrad":
            t.kind = T_WGRAD
            t.mode = static_wgrad_target(op.kh, op.kw) if self.deterministic else 0
            t._pad = self.wgrad_x3 if op.groups <= 1 else 0
            t.kh, t.kw, t.stride, t.pad_t, t.pad_l, t.groups = op.kh, op.kw, op.stride, op.pad[0], op.pad[0], op.groups
            t.x, t.dy = self._view(op.x), self._view(op.dy)
            t.p[0] = self.gptr(op.wkey)
            return [t]
        if op.kind == "dgrad" and self._is_wino(self.plan.convs[op.wkey]):
            off, lead = self._pack_off[(op.wkey, 4)]
            return self._wino_conv(self._view(op.dy), self._view(op.dx), op.pad[0], self.packs.data_ptr() + 4 * off, lead,
                                   not (self.first_store and op.store), scr=self._set_of(op.name))
        if op.kind == "dgrad":
            dx = self._view(op.dx)
            kw = dict(kind=OP_CONV, kh=op.kh, kw=op.kw, stride=1, pad_t=op.pad[0], pad_l=op.pad[0], relu=0, cout=op.dx.c,
                      tile_n=_tile_n(op.dx.c), groups=1, x=self._view(op.dy), y=dx, w=self.pack_ptr(op.wkey, 1), nbatch=1)
            if not (self.first_store and op.store):
                kw["res"] = dx                     # accumulate: the conv epilogue's residual operand is the destination itself
            return [self._net(**kw)]
        if op.kind == "upadd_bwd":
            t.kind = T_UPADD_BWD
            t.dy, t.dx, t.y = self._view(op.dy), self._view(op.dlo), self._view(op.dskip)
            return [t]
        if op.kind == "conv0_wgrad":
            t.kind, t.pad_t = T_CONV0_WGRAD, op.pad
            g = self.plan.geo["inp"]
            t.x.base, t.x.sn, t.x.sy, t.x.sx, t.x.h, t.x.w, t.x.c, t.x.sc = self.img.data_ptr(), g * g * 3, g * 3, 3, g, g, 3, 1
            t.dy = self._view(op.dy)
            t.p[0] = self.gptr(op.wkey)
            return [t]
        raise KeyError(op.kind)

    def _floating_wgrads(self, bwd_groups, first_enc):
        """{launch index of a plain weight-gradient launch: launch index it must have finished before (None: the section's join)} --
        the first later op of the backward list that WRITES the buffer its output gradient lives in (a deferred `upadd_bwd` sum runs
        later than its op's position: the position is the conservative bound).  Winograd-domain weight gradients stay in list order
        (they share the transform-domain scratch with the data gradients)."""
        bwd = self.plan.bwd
        starts, pos = [], 0
        for g in bwd_groups:
            starts.append(pos)
            pos += len(g)

        def group(pi):
            return pi if pi < first_enc else pi + 1        # the deferred group sits at `first_enc`
        out = {}
        for pi, pj in self.plan.wgrad_windows().items():
            op = bwd[pi]
            if op.wkey in self._du_off and op.stride == 1:
                continue
            assert len(bwd_groups[group(pi)]) == 1
            out[starts[group(pi)]] = None if pj is None else starts[group(pj)]
        return out

    def _lower_upadd_bwd_split(self, op):
        """`upadd_bwd` of a decoder branch with branch streams: -> (launches of the branch's own section, deferred launches).  The
        low-resolution input of u2 / u1 belongs to the branch (its gradient is needed at once); the skips (d0 .. d2) and conv_bot's
        output are shared by the branches: those sums wait for the join."""
        def private(v):
            return v is not None and v.buf.name.startswith("decoder.")
        mine, later = [], []
        for keep, dst in ((private, mine), (lambda v: v is not None and not private(v), later)):
            dlo, dskip = (op.dlo if keep(op.dlo) else None), (op.dskip if keep(op.dskip) else None)
            if dlo is None and dskip is None:
                continue
            t = L.hvn_top()
            t.kind = T_UPADD_BWD
            t.dy, t.dx, t.y = self._view(op.dy), self._view(dlo), self._view(dskip)
            dst.append(t)
        return mine, later

    def _lower(self, groups):
        flat = [t for g in groups for t in g]
        arr = (L.hvn_top * len(flat))()
        for i, t in enumerate(flat):
            ctypes.memmove(ctypes.addressof(arr[i]), ctypes.addressof(t), ctypes.sizeof(L.hvn_top))
        return arr

    def autotune_tiles(self, reps=3, margin=0.985):
        """Measured column-tile selection for the step's CONV launches (forward convs, data gradients, the Winograd-domain products),
        like `engine.Engine.autotune_tiles` for the inference plan: the static choice of `plan._tile_n` fits batch 32 of the
        inference path, while a phase-1 step carries 4 samples per GPU (opt.py:75-76) and most of its launches are one or two rounds
        of workgroups, where the narrower tile's finer quantisation wins.  Candidates: 128 x 128 | 128 x 64 for cout >= 128,
        128 x 64 | 256 x 64 for cout = 64 -- same packed weights, same k order per output element, hence the same bits.  Timed
        on whatever the arenas hold (min of `reps` HIP-event timings after a warm-up launch); one choice per launch shape and
        batch, shared by every engine of the process.  The weight-gradient launches get the same treatment for the split of
        their pixel sum (fewer, longer workgroups pay fewer atomics: best for the encoder's few-tile launches at batch 4; the
        decoder's 400-tile 5x5 launches want the opposite).  HVN_TILE_SELECT=0 | model keeps the static choices."""
        if os.environ.get("HVN_TILE_SELECT", "auto") in ("0", "model"):
            return
        lib = L.lib()
        stream = self._stream()
        reps = _tune_reps(reps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def time_op(o):
            best = float("inf")
            for r in range(reps + 1):
                e0.record()
                L.check(lib.hvn_run_op(ctypes.addressof(o), self.n, stream), "hvn_run_op (autotune)")
                e1.record()
                e1.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1))
            return best

        for o in self._keep:
            if not isinstance(o, L.hvn_op) or o.kind != OP_CONV or o.groups > 1 or o.tile_n not in (128, 64):
                continue
            if o.tile_n == 64 and (o.x2.base or o.act_dtype in (2, 3)):
                continue                                   # the fused-shortcut / bf16x3 instantiations exist for 128 x 128 and 128 x 64 tiles only (as in Engine.autotune_tiles)
            key = (self.n, o.kh, o.kw, o.stride, o.pad_t, o.x.c, o.cout, o.y.h, o.y.w, o.x.h, o.x.w, bool(o.res.base), int(o.nbatch),
                   bool(o.pre_scale), int(o.x2.c) if o.x2.base else 0, int(o.act_dtype))
            cands = (128, 64) if o.tile_n == 128 else (64, 320)
            if (o.tile_n == 128 and o.act_dtype in (2, 3) and o.cout >= 128 and not o.pre_scale and os.environ.get("HVN_X3G", "1") != "0"):
                cands = cands + (896, 640)                 # + the LDS-DMA forms of the bf16x3 convolution (csrc/hvn_conv_x3g.hip): same packing, same bits
            key = key + (cands, str(self.device))          # (round-5 advisor: another candidate set / device is another question)
            if key not in _TILE_CHOICE:
                t = {}
                for tn in cands:
                    o.tile_n = tn
                    try:
                        t[tn] = time_op(o)
                    except L.HvnError:
                        if tn in (128, 64, 320):
                            raise
                        t[tn] = float("inf")               # a form the launcher refuses for this geometry
                best = min(cands[1:], key=lambda tn: t[tn])
                _TILE_CHOICE[key] = (best if t[best] < margin * t[cands[0]] else cands[0], t[cands[0]], t[best])
            o.tile_n = _TILE_CHOICE[key][0]
        # weight gradients: the split of the pixel sum (hvn_top.mode = workgroups aimed at; csrc/hvn_train.hip: launch_wgrad)
        tsz = ctypes.sizeof(L.hvn_top)
        base = ctypes.addressof(self.bwd_ops)

        def time_top(i):
            best = float("inf")
            for r in range(reps + 1):
                e0.record()
                rc = lib.hvn_run_train_plan(base + i * tsz, 1, self.n, stream)
                if rc:
                    raise L.HvnError("hvn_run_train_plan (autotune) failed (%d): %s" % (rc, lib.hvn_train_last_error().decode()))
                e1.record()
                e1.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1))
            return best

        for i in range(len(self.bwd_ops)):
            t = self.bwd_ops[i]
            if t.kind != T_WGRAD or self.deterministic:      # a timed split = a summation order that depends on the box and the run
                continue
            key = ("wgrad", self.n, t.kh, t.kw, t.stride, t.pad_t, t.x.c, t.dy.c, t.dy.h, t.dy.w, t.x.h, t.x.w, t.groups, int(t.nbatch), int(t._pad))
            if key not in _TILE_CHOICE:
                ms = {}
                for want in WGRAD_TARGETS:
                    t.mode = want
                    ms[want] = time_top(i)
                best = min(ms, key=ms.get)
                _TILE_CHOICE[key] = (best if ms[best] < margin * ms[WGRAD_TARGETS[0]] else WGRAD_TARGETS[0], ms[WGRAD_TARGETS[0]], ms[best])
            t.mode = _TILE_CHOICE[key][0]
        torch.cuda.synchronize(self.device)
        self.gmem.zero_()       # the data-gradient and weight-gradient launches accumulate
        self._share_launch_shapes()

    def _choose_wgrad_stream(self, reps=2):
        """HVN_TRAIN_WGRAD_STREAM=auto: the whole backward list timed with the weight gradients floating and in list order (on whatever
        the arenas hold, like `autotune_tiles`; the forward list is not run: it would move the running statistics), the faster one kept.
        Every rank may answer for itself: the answer changes no bit."""
        if self._wgrad_mode != "auto" or not self._floats or os.environ.get("HVN_TILE_SELECT", "auto") == "0":
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = {}
        for on in (True, False):
            self.wgrad_stream = on
            best = float("inf")
            for r in range(_tune_reps(reps) + 1):
                e0.record()
                self._run_range(self.bwd_ops, self._bwd_runs, 0, len(self.bwd_ops), "backward (timing)")
                e1.record()
                e1.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1))
            ms[on] = best
        self.wgrad_stream = ms[True] < 0.985 * ms[False]          # a tie keeps the list order (fewer events, fewer host calls)
        self.wgrad_stream_ms = (ms[True], ms[False])
        torch.cuda.synchronize(self.device)
        self.gmem.zero_()

    def _share_launch_shapes(self):
        """Data-parallel training: every rank times its own launches, and noise could give two ranks different weight-gradient splits,
        i.e. different fp32 summation orders of the same gradient (harmless after the all-reduce, but run-to-run variation nobody asked
        for).  Rank 0's choices are broadcast: one small int32 tensor at the build of a module's FIRST engine, which every rank does at
        its first training step (`engine_for`; train.run_phases makes a new module per phase and drops ragged batches, so each phase's
        engines are built in lock-step).  A REBUILD -- another batch size or freeze flag on a module that already has an engine, e.g. the
        ragged last batch of an external loader on ONE rank -- issues no collective (it would pair with the other ranks' gradient
        all-reduce) and keeps that rank's own timings.  HVN_TILE_SHARE=0 keeps the per-rank choices everywhere."""
        import torch.distributed as dist

        if (not self._share_shapes or os.environ.get("HVN_TILE_SHARE", "1") == "0" or
                not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)):
            return
        convs = [o for o in self._keep if isinstance(o, L.hvn_op) and o.kind == OP_CONV]
        tops = [self.bwd_ops[i] for i in range(len(self.bwd_ops)) if self.bwd_ops[i].kind == T_WGRAD]
        vals = torch.tensor([int(o.tile_n) for o in convs] + [int(t.mode) for t in tops], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            vals = vals.to(self.device)
        dist.broadcast(vals, 0)
        vals = vals.cpu().tolist()
        for o, v in zip(convs, vals[:len(convs)]):
            o.tile_n = v
        for t, v in zip(tops, vals[len(convs):]):
            t.mode = v

    def _loss_desc(self):
        d = L.hvn_loss()
        d.logits_np, d.logits_hv = self.logits["np"].data_ptr(), self.logits["hv"].data_ptr()
        d.grad_np, d.grad_hv = self.dlogits["np"].data_ptr(), self.dlogits["hv"].data_ptr()
        if "tp" in self.logits:
            d.logits_tp, d.grad_tp = self.logits["tp"].data_ptr(), self.dlogits["tp"].data_ptr()
            d.true_tp = self.true_tp.data_ptr()
        d.true_np, d.true_hv = self.true_np.data_ptr(), self.true_hv.data_ptr()
        d.sums, d.sobel_ws = self.sums.data_ptr(), self.sobel_ws.data_ptr()
        d.n, d.h, d.w = self.n, self.true_np.shape[1], self.true_np.shape[2]
        d.nr_types = self.net.nr_types or 0
        for i in range(6):
            d.weight[i] = 1.0
        if self.loss_parts is not None:
            d.partials, d.partials_cap = self.loss_parts.data_ptr(), self.loss_parts.numel()
        return d

    _WEIGHT_SLOT = {("np", "bce"): 0, ("np", "dice"): 1, ("hv", "mse"): 2, ("hv", "msge"): 3, ("tp", "bce"): 4, ("tp", "dice"): 5}

    def set_loss_weights(self, loss_opts):
        """loss_opts: the `extra_info["loss"]` table of the reference's config (opt.py:47-51), {branch: {term: weight}}; a
        term that is absent has weight 0 (run_desc.py:66-82 only visits the listed terms).  None = all ones."""
        w = [1.0] * 6 if loss_opts is None else [0.0] * 6
        for br, terms in (loss_opts or {}).items():
            for term, val in terms.items():
                if (br, term) not in self._WEIGHT_SLOT:
                    raise KeyError("unknown loss term %s/%s (the reference's loss_func_dict has bce, dice, mse, msge)" % (br, term))
                w[self._WEIGHT_SLOT[(br, term)]] = float(val)
        for i in range(6):
            self._loss.weight[i] = w[i]
        self._weights = w

    # -- one step ------------------------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def load_batch(self, batch):
        """batch: the reference loader's dict (img uint8 NHWC, np_map, hv_map, tp_map?) of tensors / arrays."""
        def put(dst, src, dtype):
            src = torch.as_tensor(src)
            if src.shape[0] != self.n:
                raise ValueError("the training engine was built for batch %d, got %d" % (self.n, src.shape[0]))
            dst.copy_(src.to(dtype).reshape(dst.shape), non_blocking=True)
        put(self.img, batch["img"], torch.uint8)
        put(self.true_np, batch["np_map"], torch.int32)
        put(self.true_hv, batch["hv_map"], torch.float32)
        if self.net.nr_types is not None:
            put(self.true_tp, torch.squeeze(torch.as_tensor(batch["tp_map"])).reshape(self.true_tp.shape), torch.int32)

    def _run_plan(self, ops_addr, n_ops, what, scr=0, w=False):
        """One hvn_top list on the current stream; with the deterministic-reduce workspace (of scratch set `scr`; w: of that set's
        weight-gradient stream) when the engine has one."""
        lib = L.lib()
        if self.det_ws is not None:
            ws = (self.det_ws_w if w else self.det_ws)[scr]
            rc = lib.hvn_run_train_plan_ws(ops_addr, n_ops, self.n, self._stream(), ws.data_ptr(), 4 * ws.numel())
        else:
            rc = lib.hvn_run_train_plan(ops_addr, n_ops, self.n, self._stream())
        if rc:
            raise L.HvnError("hvn_run_train_plan(%s) failed (%d): %s" % (what, rc, lib.hvn_train_last_error().decode()))

    def _run_section(self, ops, a, b, what, scr=0):
        """Launches [a, b) of a lowered list on the current stream, its floating weight gradients (`_floats`, backward list only) on the
        weight-gradient stream of scratch set `scr` between their two events; joined before returning."""
        base, osz = ctypes.addressof(ops), ctypes.sizeof(L.hvn_top)
        fl = self._floats if (ops is self.bwd_ops and self.wgrad_stream) else {}
        mine = [i for i in fl if a <= i < b]
        if not mine:
            return self._run_plan(base + a * osz, b - a, what, scr)
        S, W = torch.cuda.current_stream(self.device), self._wside[scr]
        cuts = sorted(set(mine) | {fl[i] for i in mine if fl[i] is not None and fl[i] < b})
        pos, pending = a, {}
        for c in cuts + [b]:
            if c > pos:
                self._run_plan(base + pos * osz, c - pos, what, scr)
            pos = c
            if c == b:
                break
            for ev in pending.pop(c, ()):                    # the launch at c writes a buffer a floating weight gradient reads
                S.wait_event(ev)
            if c in fl:
                ready = torch.cuda.Event()
                ready.record(S)                              # its output gradient is complete
                W.wait_event(ready)
                with torch.cuda.stream(W):
                    self._run_plan(base + c * osz, 1, what, scr, w=True)
                    if fl[c] is not None and fl[c] < b:
                        done = torch.cuda.Event()
                        done.record(W)
                        pending.setdefault(fl[c], []).append(done)
                pos = c + 1
        S.wait_stream(W)

    def _run_range(self, ops, runs, lo, hi, what):
        """Launches [lo, hi) of a lowered list: sections tagged with a branch (`_runs`) on that branch's stream -- branch 0 on the
        current one -- between a fork and a join, everything else on the current stream in list order."""
        base, osz = ctypes.addressof(ops), ctypes.sizeof(L.hvn_top)
        main = torch.cuda.current_stream(self.device)
        runs = [(k, max(a, lo), min(b, hi)) for k, a, b in runs if max(a, lo) < min(b, hi)]
        i = 0
        while i < len(runs):
            k, a, b = runs[i]
            if k < 0 or not self.branch_streams:
                self._run_section(ops, a, b, what)
                i += 1
                continue
            j = i
            while j < len(runs) and runs[j][0] >= 0:
                j += 1
            used = []
            for k, a, b in sorted(runs[i:j], key=lambda r: -r[0]):       # side streams first, branch 0 on the current stream last
                if k == 0:
                    self._run_section(ops, a, b, what)
                    continue
                side = self._side[k - 1]
                if side not in used:
                    side.wait_stream(main)
                    used.append(side)
                with torch.cuda.stream(side):
                    self._run_section(ops, a, b, what, scr=k)
            for side in used:
                main.wait_stream(side)
            i = j

    def forward(self):
        self._run_range(self.fwd_ops, self._fwd_runs, 0, len(self.fwd_ops), "forward")
        torch._foreach_add_(self._nbt, 1)
        self.net._train_version = getattr(self.net, "_train_version", 0) + 1    # invalidates the cached inference plan
        return self.logits

    def loss_forward(self):
        """Zero the gradient memory, run loss stage 1 -> this rank's partial sums (device float64 [64])."""
        lib = L.lib()
        self._gzero.zero_()
        self.sums.zero_()
        rc = lib.hvn_loss_forward(ctypes.byref(self._loss), self._stream())
        if rc:
            raise L.HvnError("hvn_loss_forward failed (%d): %s" % (rc, lib.hvn_train_last_error().decode()))
        return self.sums

    def backward(self, world=1, all_reduce=None):
        """Loss stage 2 (logit gradients from the -- possibly all-reduced -- sums) and the backward plan.  With
        `all_reduce(tensor, async_op)` (data-parallel training) the gradient slab is reduced in two buckets: the
        decoder's (the tail of the slab, complete once the decoder branches' backward ops have run) is launched
        asynchronously and overlaps the encoder's backward pass, the encoder's follows at the end."""
        lib = L.lib()
        s = self._stream()
        self._loss.total_pixels = float(world * self.n * self._loss.h * self._loss.w)
        rc = lib.hvn_loss_backward(ctypes.byref(self._loss), s)
        if rc:
            raise L.HvnError("hvn_loss_backward failed (%d): %s" % (rc, lib.hvn_train_last_error().decode()))
        n = len(self.bwd_ops)

        def run(lo, hi):
            if hi > lo:
                self._run_range(self.bwd_ops, self._bwd_runs, lo, hi, "backward")

        split = self._bwd_split
        if all_reduce is None:
            run(0, n)
        elif 0 < split < n:
            run(0, split)                                              # decoder branches
            pending = [all_reduce(self.gslab[self._dec_off:], True)]   # their bucket travels under ...
            run(split, n)                                              # ... conv_bot and the encoder
            pending.append(all_reduce(self.gslab[:self._dec_off], True))
            for w in pending:
                if w is not None and hasattr(w, "wait"):
                    w.wait()
        else:
            run(0, n)
            w = all_reduce(self.gslab, True)
            if w is not None and hasattr(w, "wait"):
                w.wait()
        return self.gslab

    def backward_from(self, dlogits):
        """Backward plan from caller-supplied logit gradients (dict branch -> [n, c, h, w]): what torch autograd hands
        `HoVerNet.forward`'s graph node when the loss was computed in torch.  Fills the gradient slab (the parameters' .grad
        memory) like `backward`, without the fused loss stage."""
        self._gzero.zero_()
        for br, buf in self.dlogits.items():
            g = dlogits.get(br)
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g.to(buf.dtype).reshape(buf.shape))
        self._run_range(self.bwd_ops, self._bwd_runs, 0, len(self.bwd_ops), "backward")
        return self.gslab

    def loss_and_backward(self, world=1, all_reduce=None):
        """loss stage 1 -> (SUM all-reduce of the partial sums) -> backward with the bucketed gradient all-reduce.
        `all_reduce(tensor, async_op)`: torch.distributed's SUM all-reduce (RCCL), returning a work handle when
        async_op is True.  This is the reference's DataParallel step, one process per GPU."""
        self.loss_forward()
        if all_reduce is not None:
            all_reduce(self.sums, False)
        self.backward(world, all_reduce)
        return self.sums

    def loss_terms(self, sums=None):
        """Raw sums -> the reference's named loss terms (run_desc.py:66-82) as python floats (one D2H sync)."""
        s = (self.sums if sums is None else sums).cpu().numpy()
        m = self._loss.total_pixels
        smooth = 1e-3
        t = {}

        def dice(i0, l0, r0, c):
            return float(sum(1.0 - (2.0 * s[i0 + k] + smooth) / (s[l0 + k] + s[r0 + k] + smooth) for k in range(c)))
        if self.net.nr_types is not None:
            t["loss_tp_bce"] = float(s[1] / m)
            t["loss_tp_dice"] = dice(16, 32, 48, self.net.nr_types)
        t["loss_np_bce"] = float(s[0] / m)
        t["loss_np_dice"] = dice(8, 10, 12, 2)
        t["loss_hv_mse"] = float(s[2] / (2.0 * m))
        t["loss_hv_msge"] = float(s[3] / (s[4] + 1.0e-8))
        # run_desc.py:80-82: the tracked terms are unweighted, the overall loss is their weighted sum
        w = getattr(self, "_weights", [1.0] * 6)
        names = ("loss_np_bce", "loss_np_dice", "loss_hv_mse", "loss_hv_msge", "loss_tp_bce", "loss_tp_dice")
        overall = sum(w[i] * t[k] for i, k in enumerate(names) if k in t)
        for i, k in enumerate(names):
            if k in t and w[i] == 0.0:
                del t[k]                 # a term that is not in the table is neither computed nor tracked by the reference
        t["overall_loss"] = float(overall)
        return t


def engine_for(net, batch):
    """The module's training engine for this batch size (built on first use, rebuilt when the batch size changes)."""
    eng = getattr(net, "_train_engine", None)
    if eng is None or eng.n != int(batch) or eng.plan.freeze != net.freeze:
        eng = TrainEngine(net, batch, share_shapes=eng is None)      # a rebuild may happen on one rank alone: no collective in it
        net._train_engine = eng
    return eng
