"""Binds a launch plan (hover_net_amd.plan) to HBM and runs it through libhvn_hip.so.

torch is used for exactly three things here: allocating device memory, naming the
current HIP stream, and wrapping the outputs as tensors.  Weights are uploaded once
(one flat fp32 slab, 256-byte aligned sub-arrays); activations live in one arena of
`max_batch x arena_per_sample` floats that is re-used by every call (sized for HBM,
not for reuse in cache: 555 MB/tile with the Winograd transform-domain tensors -> 17.8 GB at batch 32).
`dtype="bf16"` binds the same plan to the bf16 kernels (bf16 arena and conv weights, fp32 logits).
"""
import ctypes

import numpy as np
import torch

from . import lib as L
from . import plan as PL


def _view_struct(v, base_ptr, sn, itemsize=4):
    b = v.buf
    sx = b.c
    sy = b.w * b.c
    s = L.hvn_view()
    s.base = base_ptr + (v.y0 * sy + v.x0 * sx + v.c0) * itemsize
    s.sn, s.sy, s.sx = sn, sy, sx
    s.h, s.w, s.c, s.sc = v.h, v.w, v.c, 1
    return s


WG_SLOTS = 512            # resident 128x128 conv workgroups: 2 per CU (VGPRs) x 256 CUs
WG_SLOTS_NARROW = 768     # resident 128x64 workgroups: 3 per CU (48 KB of LDS each with the swizzled layout, <= 168 VGPRs)
NARROW_TILE_COST = 0.45   # time of a round of 128x64 tiles relative to a round of 128x128 tiles (3 x 1/2 vs 2 x 1 tiles of MFMA work per CU)


def pick_tile_n(op, batch):
    """Column tile (128 or 64) of a CONV launch by its wave quantisation -- the MODEL behind `HVN_TILE_SELECT=model` and the
    starting point of the measured selection (`Engine.autotune_tiles`, the default): a launch of W workgroups runs in
    ceil(W / slots) rounds, so 1092 workgroups of 128x128 tiles (2.13 rounds -> 3) lose 29 % to the last round.  The packed
    weights are the same for both widths (cout is padded to 128) and so is every output bit (same k order per element).  Only
    plain launches with cout >= 128 are re-tiled; HVN_TILE_SELECT=0 keeps the static choice of plan._tile_n.
    Experiment knobs: HVN_FORCE_TILE_N=64|128, HVN_WG_SLOTS_64, HVN_NARROW_COST."""
    import math
    import os

    if op.tile_n != 128 or os.environ.get("HVN_TILE_SELECT", "auto") == "0":
        return op.tile_n
    force = os.environ.get("HVN_FORCE_TILE_N")
    if force:
        return int(force)
    slots64 = int(os.environ.get("HVN_WG_SLOTS_64", WG_SLOTS_NARROW))
    cost64 = float(os.environ.get("HVN_NARROW_COST", NARROW_TILE_COST))
    m_tiles = math.ceil(batch * op.y.h * op.y.w / 128.0)
    nb = int(op.extra.get("nbatch", 1))
    wide = math.ceil(m_tiles * math.ceil(op.cout / 128.0) * nb / WG_SLOTS)
    narrow = math.ceil(m_tiles * math.ceil(op.cout / 64.0) * nb / slots64) * cost64
    return 64 if narrow < wide else 128


_STREAM_POOL, _STREAM_CHOICE = {}, {}   # device -> [streams]; (device, sub-batches, lanes, dtype) -> (pool offset, {offset: ms})


def lane_stream_pool(device, count):
    """The inference engines' side streams: one list per process and device, grown on demand, never replaced (see Engine.autotune_streams)."""
    have = _STREAM_POOL.setdefault(str(device), [])
    while len(have) < count:
        have.append(torch.cuda.Stream(device))
    return have


_TILE_CACHE = {}        # device -> {launch shape key: (choice, ms of the default, ms of the choice, {candidate: ms})}
X3G_256, X3G_128 = 128 + 0x300, 128 + 0x200     # hvn_op.tile_n of the LDS-DMA forms of the bf16x3 convolution (include/hvn.h)
X3R = 128 + 0x400                               # hvn_op.tile_n of the bf16x3 CHAIN with a register-resident input tile (include/hvn.h)


def x3g_forms_for(op):
    """The LDS-DMA workgroup shapes (csrc/hvn_conv_x3g.hip) a bf16x3 CONV launch may run on besides hvn_conv_x3.hip's, as tile_n codes:
    they need >= 128 output channels and -- the 256-row form with a prologue -- its two per-channel vectors next to the operand rings
    in the CU's 160 KB of LDS.  Same packing, same bits (tests/test_gpu_x3.py): which one runs is a timing decision (`Engine.autotune_tiles`).  HVN_X3G=0
    keeps hvn_conv_x3.hip everywhere; HVN_X3G=896 | 640 offers one form only."""
    import os

    want = os.environ.get("HVN_X3G", "1")
    if want == "0" or op.cout < 128 or int(op.extra.get("groups", 1)) != 1:
        return ()
    cin = op.x.c
    forms = []
    for code, bm, na in ((X3G_256, 256, 3), (X3G_128, 128, 2)):
        if want not in ("1", str(code)):
            continue
        if op.pre is not None and bm == 256 and na * bm * 128 + 2 * 3 * 128 * 64 + 2 * cin * 4 > 160 * 1024:
            continue                                   # (the 128-row form reads the prologue's vectors from global memory)
        forms.append(code)
    return tuple(forms)


def to_bf16_bits(a):
    """float32 array -> bfloat16 bit patterns (uint16), round to nearest even."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7FFF)) >> 16).astype(np.uint16)


def repack_conv_bf16(w32):
    """[cout_pad, cin/32, taps, 32] fp32 (plan._pack_conv) -> [cout_pad, ceil(cin/64), taps, 64] bf16 bits: the bf16
    kernel's k-step is 64 channels; an odd number of 32-channel slabs gets a zero slab (the kernel loads zeros there)."""
    cp, s32, taps, _ = w32.shape
    if s32 % 2:
        w32 = np.concatenate([w32, np.zeros((cp, 1, taps, 32), np.float32)], 1)
        s32 += 1
    w = w32.reshape(cp, s32 // 2, 2, taps, 32).transpose(0, 1, 3, 2, 4).reshape(cp, s32 // 2, taps, 64)
    return to_bf16_bits(w)


def split_bf16x3(w32):
    """fp32 array -> its three bf16 planes (uint16 bit patterns) with h + m + l == w exactly: h = bf16(w), m = bf16(w - h),
    l = bf16(w - h - m), round to nearest even, every difference exact in fp32 (the device splits activations the same way)."""
    w = np.ascontiguousarray(w32, np.float32)

    def bf(a):
        bits = to_bf16_bits(a)
        return bits, (bits.astype(np.uint32) << 16).view(np.float32)

    h, hf = bf(w)
    r = w - hf
    m, mf = bf(r)
    l, _ = bf(r - mf)
    return h, m, l


def pack_conv_x3(w32):
    """[..., cout_pad, cin/32, taps, 32] fp32 (plan._pack_conv; a leading axis for the Winograd positions) ->
    [..., cout_pad, k-steps, 3, 32] bf16 bits: the three planes of every k-step (slab-major, tap inside) side by side."""
    lead = w32.shape[:-3]
    kt = w32.shape[-3] * w32.shape[-2]
    planes = split_bf16x3(w32.reshape(lead + (kt, 32)))
    return np.stack(planes, axis=-2)


def pack_conv_x3_device(w32, device):
    """`pack_conv_x3` computed on the GPU (round 6: the numpy split of the ~150 M packed weights of a cfg-2 plan -- Winograd-domain tensors
    included -- took 3 - 9 s of every engine build): the fp32 packing goes up as it is and the three planes are formed there with the same
    round-to-nearest-even conversions (torch's float32 -> bfloat16 is the bit trick of `to_bf16_bits`; both differences are exact in
    fp32).  -> flat int16 tensor [rows][3][32] on `device`, bit-equal to `pack_conv_x3(w32).ravel()` (tests/test_gpu_x3.py)."""
    w = torch.from_numpy(np.ascontiguousarray(w32, np.float32).reshape(-1, 32)).to(device)
    h = w.to(torch.bfloat16)
    r = w - h.float()
    m = r.to(torch.bfloat16)
    l = (r - m.float()).to(torch.bfloat16)
    return torch.stack((h, m, l), dim=1).view(torch.int16).reshape(-1)


class Engine:
    def __init__(self, plan, max_batch=32, device="cuda", n_split=None, dtype="fp32", n_lanes=None):
        L.require_gpu()
        assert dtype in ("fp32", "bf16")
        self.dtype = dtype
        self.isz = 4 if dtype == "fp32" else 2          # bytes per activation element
        if dtype == "bf16" and any(op.kind in (PL.OP_WINO_IN, PL.OP_WINO_OUT) for op in plan.ops):
            raise ValueError("the bf16 path runs direct convolutions: build the plan with winograd=0")
        self.plan = plan
        self.max_batch = int(max_batch)
        self.device = torch.device(device)
        # A forward is a dependent chain of ~150 launches, many of which fill the 512 workgroup slots
        # of the chip only 1.5-4.x times (tail quantisation up to 29 %).  Splitting the batch into
        # independent sub-batches on their own HIP streams lets the tail of one chain be filled by
        # the other's workgroups.
        # Default launch schedule (round 4): fp32 = two encoder sub-batches on two streams + the decoder branches on three
        # (measured +2.6 .. +3.3 %, bit-equal outputs: tests/test_gpu_chain.py); `n_split=1, n_lanes=0` (or HVN_SPLIT=1 HVN_LANES=0)
        # is the single launch stream on which every launch can be timed alone (bench.py's roofline leg, the PMC runs).
        import os
        dflt = ("2", "2") if dtype == "fp32" else ("1", "0")
        self.n_split = int(os.environ.get("HVN_SPLIT", dflt[0])) if n_split is None else int(n_split)
        self.n_lane_streams = int(os.environ.get("HVN_LANES", dflt[1])) if n_lanes is None else int(n_lanes)  # extra streams for the decoder branches (0: one launch stream)
        self.split_decoder = os.environ.get("HVN_SPLIT_DECODER", "0") != "0"
        self._streams = None
        self._stream_off = 0
        self._upload_params()
        self.arena = torch.empty((self.max_batch, plan.arena_per_sample), dtype=torch.float32 if dtype == "fp32" else torch.int16,
                                 device=self.device)
        self.logits = {br: torch.empty((self.max_batch, b.c, b.h, b.w), dtype=torch.float32, device=self.device)
                       for br, b in plan.logits.items()}
        self.pred_map = None
        if plan.pred_map is not None:
            pm = plan.pred_map
            self.pred_map = torch.empty((self.max_batch, pm.h, pm.w, pm.c), dtype=torch.float32, device=self.device)
        self.ops = (L.hvn_op * len(plan.ops))()
        self._bind()
        self._sub_ops = {}
        # measured launch-shape choices (column tile / kernel form), keyed by everything the timing depends on: shared by every engine of
        # the process on this device -- the forms give the same bits, so WHICH engine timed a shape first is invisible in the results
        self.tile_choice = _TILE_CACHE.setdefault(str(self.device), {})
        tile_file = os.environ.get("HVN_TILE_FILE")          # a list of per-op column tiles written by another engine of the same plan
        if tile_file and dtype == "fp32":                   # (bench.py hands its measured choices to the PMC child runs)
            import json
            tiles = json.load(open(tile_file))
            if len(tiles) != len(self.ops):
                raise ValueError("HVN_TILE_FILE holds %d entries for a plan of %d ops" % (len(tiles), len(self.ops)))
            for o, op, tn in zip(self.ops, plan.ops, tiles):
                if op.kind == PL.OP_CONV and ((op.tile_n == 128 and (tn in (64, 128) or (op.extra.get("x3") and tn in x3g_forms_for(op)))) or
                                              (op.tile_n == 64 and tn in (64, 320) and not op.extra.get("x3"))):
                    o.tile_n = tn
                if op.kind == PL.OP_CHAIN and (tn in (64, 128) or (tn == X3R and op.extra.get("x3"))):
                    o.tile_n = tn
        elif dtype == "fp32" and os.environ.get("HVN_TILE_SELECT", "auto") == "auto" and not os.environ.get("HVN_FORCE_TILE_N"):
            self.autotune_tiles()
        elif dtype == "bf16" and os.environ.get("HVN_TILE_SELECT", "auto") == "auto" and os.environ.get("HVN_BF16G", "1") != "0":
            self.autotune_bf16_forms()
        if os.environ.get("HVN_TILE_SELECT", "auto") == "auto" and os.environ.get("HVN_STREAM_SELECT", "1") != "0":
            self.autotune_streams()

    # ---------------------------------------------------------------------------------
    def _upload_params(self):
        arrays, offs, total = [], {}, 0

        def put(key, a):
            nonlocal total
            if a is None:
                return
            a = np.ascontiguousarray(a, np.float32).ravel()
            offs[key] = total
            arrays.append((total, a))
            total += (a.size + 63) // 64 * 64

        w16, off16, tot16 = [], {}, 0
        for i, op in enumerate(self.plan.ops):
            if self.dtype == "bf16" and op.kind in (PL.OP_CONV, PL.OP_CHAIN):
                wb = repack_conv_bf16(op.w).ravel()
                off16[i] = tot16
                w16.append((tot16, wb))
                tot16 += (wb.size + 127) // 128 * 128
                if op.kind == PL.OP_CHAIN:               # + the second conv's (csrc/hvn_conv_chain_bf16.hip)
                    wb2 = repack_conv_bf16(op.extra["w2"]).ravel()
                    off16[(i, "w2")] = tot16
                    w16.append((tot16, wb2))
                    tot16 += (wb2.size + 127) // 128 * 128
            elif self.dtype == "fp32" and op.kind in (PL.OP_CONV, PL.OP_CHAIN) and op.extra.get("x3"):
                # fp32 weights as three bf16 planes (csrc/hvn_conv_x3.hip), split on the device: (offset, fp32 packing) now, planes below
                off16[i] = tot16
                w16.append((tot16, op.w, True))
                tot16 += (3 * op.w.size + 127) // 128 * 128
                if op.kind == PL.OP_CHAIN:               # + the second conv's (csrc/hvn_conv_chain_x3.hip)
                    off16[(i, "w2")] = tot16
                    w16.append((tot16, op.extra["w2"], True))
                    tot16 += (3 * op.extra["w2"].size + 127) // 128 * 128
            else:
                put((i, "w"), op.w)
            put((i, "bias"), op.bias)
            if not (op.kind == PL.OP_CHAIN and (self.dtype == "bf16" or op.extra.get("x3"))):
                put((i, "w2"), op.extra.get("w2"))
            put((i, "bias2"), op.extra.get("bias2"))
            if op.pre is not None:
                put((i, "pre_s"), op.pre[0])
                put((i, "pre_b"), op.pre[1])
            if op.post is not None:
                put((i, "post_s"), op.post[0])
                put((i, "post_b"), op.post[1])
        host = np.zeros(max(total, 64), np.float32)
        for off, a in arrays:
            host[off:off + a.size] = a
        self.params = torch.from_numpy(host).to(self.device)
        self._poff = offs
        self._poff16 = off16
        self.params16 = None
        if tot16:
            if any(len(e) == 3 for e in w16):            # bf16x3: planes made on the device, one conv at a time
                self.params16 = torch.zeros(tot16, dtype=torch.int16, device=self.device)
                for off, w32, _ in w16:
                    pl = pack_conv_x3_device(w32, self.device)
                    self.params16[off:off + pl.numel()] = pl
            else:
                h16 = np.zeros(tot16, np.uint16)
                for off, a in w16:
                    h16[off:off + a.size] = a
                self.params16 = torch.from_numpy(h16.view(np.int16)).to(self.device)

    def _pptr(self, i, name):
        off = self._poff.get((i, name))
        return None if off is None else self.params.data_ptr() + 4 * off

    def _bind(self):
        P = self.plan
        abase = self.arena.data_ptr()
        sn = P.arena_per_sample
        for i, op in enumerate(P.ops):
            o = self.ops[i]
            o.act_dtype = 1 if self.dtype == "bf16" else 0
            o.kind, o.kh, o.kw, o.stride = op.kind, op.kh, op.kw, op.stride
            o.pad_t, o.pad_l, o.relu, o.cout, o.tile_n, o.x_dtype = op.pad_t, op.pad_l, op.relu, op.cout, op.tile_n, 0
            if op.kind == PL.OP_CONV and self.dtype == "fp32":
                o.tile_n = pick_tile_n(op, self.max_batch)
            o.groups = int(op.extra.get("groups", 1))
            o._rsv = int(op.extra.get("stride2", 1))
            o.nbatch = int(op.extra.get("nbatch", 1))
            for bi, bs in enumerate(op.extra.get("batch_strides", (0, 0, 0))):
                o.batch_stride[bi] = int(bs)
            if op.kind in (PL.OP_WINO_IN, PL.OP_WINO_OUT):
                o.kh, o.kw = op.extra["tiles"]
                o.stride, o._rsv = op.extra["m"], op.extra.get("r", 5)      # F(m x m, r x r)
            if op.kind == PL.OP_PREDMAP:
                o.x.base = self.logits["np"].data_ptr()
                o.res.base = self.logits["hv"].data_ptr()
                o.w = self.logits["tp"].data_ptr() if "tp" in self.logits else None
                o.cout = P.nr_types or 0
                o.y.base = self.pred_map.data_ptr()
                o.y.h, o.y.w, o.y.c = P.pred_map.h, P.pred_map.w, P.pred_map.c
                continue
            if op.kind == PL.OP_CHAIN:
                o.cout2 = int(op.extra["cout2"])
                o.w2, o.bias2 = self._pptr(i, "w2"), self._pptr(i, "bias2")
                if self.dtype == "bf16":                 # csrc/hvn_conv_chain_bf16.hip: both packings in hvn_conv_bf16.hip's layout
                    o.w2 = self.params16.data_ptr() + 2 * self._poff16[(i, "w2")]
                elif op.extra.get("x3"):
                    o.w2 = self.params16.data_ptr() + 2 * self._poff16[(i, "w2")]
                    o.act_dtype = 2 if int(op.extra["x3"]) == 9 else 3
                    o.tile_n = 128
            for fld, v in (("x", op.x), ("res", op.res), ("y", op.y), ("x2", op.extra.get("x2")), ("y2", op.extra.get("y2"))):
                if v is None:
                    continue
                if v.buf.offset >= 0:
                    setattr(o, fld, _view_struct(v, abase + self.isz * v.buf.offset, sn, self.isz))
            if op.kind == PL.OP_CONV0:
                o.x.h, o.x.w, o.x.c = op.x.h, op.x.w, 3  # base / strides set per call
            if op.kind == PL.OP_HEAD:
                br = op.y.buf.name.split(".")[1]
                o.y.base = self.logits[br].data_ptr()
                o.y.h, o.y.w, o.y.c = op.y.h, op.y.w, op.y.c
            o.w = self._pptr(i, "w") if i not in self._poff16 else self.params16.data_ptr() + 2 * self._poff16[i]
            if i in self._poff16 and self.dtype == "bf16":
                o.groups = 1        # grouped convs run as block-diagonal dense GEMMs on the bf16 pipe
            if self.dtype == "fp32" and op.kind == PL.OP_CONV and op.extra.get("x3"):
                o.act_dtype = 2 if int(op.extra["x3"]) == 9 else 3
                if o.nbatch > 1:    # Winograd positions: the plane packing's stride between problems, in bf16 elements
                    o.batch_stride[1] = int(op.w.shape[-4]) * int(op.w.shape[-3]) * int(op.w.shape[-2]) * 96
            o.bias = self._pptr(i, "bias")
            o.pre_scale, o.pre_shift = self._pptr(i, "pre_s"), self._pptr(i, "pre_b")
            o.post_scale, o.post_shift = self._pptr(i, "post_s"), self._pptr(i, "post_b")

    # ---------------------------------------------------------------------------------
    def autotune_streams(self, reps=3):
        """Which streams of the process-wide pool this engine's launch schedule runs on (round 6).  HIP multiplexes a process's streams
        onto a few hardware queues in creation order, and two lanes of one step that share a queue run one after the other: with k idle
        streams created before the engine's own, the cfg-2 network step measured 41.5 / 42.4 / 41.7 / 43.4 / 41.5 ms for k = 0 .. 4 against
        42.4 on one stream (`profiles/r06_stream_map_probe.txt`) -- the schedule's gain depended on what the process had done before (in
        bench.py: a fit).  So the streams come from one pool per process and device (`lane_stream_pool`), and the engine times its whole
        schedule on the pool's four rotations once per (device, schedule) and process; outputs do not depend on the answer
        (tests/test_gpu_chain.py: every schedule gives the same bits)."""
        split = self.n_split if (self.n_split > 1 and self.max_batch >= 2 * self.n_split) else 1
        need = (split - 1) + split * self.n_lane_streams
        if need == 0 or self.plan.ops[0].kind != PL.OP_CONV0 or not getattr(self.plan, "geo", None):
            return
        g = int(self.plan.geo.get("inp", 0))
        if g <= 0:
            return
        key = (str(self.device), split, self.n_lane_streams, self.dtype)
        if key not in _STREAM_CHOICE:
            import os
            reps = max(1, int(os.environ.get("HVN_TUNE_REPS", reps)))
            imgs = torch.zeros((self.max_batch, g, g, 3), dtype=torch.uint8, device=self.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ms = {}
            for off in range(4):
                self._stream_off, self._streams = off, None
                best = float("inf")
                for r in range(reps + 1):
                    e0.record()
                    self.run(imgs)
                    self.run(imgs)
                    e1.record()
                    e1.synchronize()
                    if r:
                        best = min(best, e0.elapsed_time(e1) / 2)
                ms[off] = best
            _STREAM_CHOICE[key] = (min(ms, key=ms.get), ms)
        self._stream_off, self._streams = _STREAM_CHOICE[key][0], None

    def autotune_tiles(self, reps=3, margin=0.985):
        """Measured column-tile selection (the default; `HVN_TILE_SELECT=model` keeps `pick_tile_n`'s rounds model): every
        re-tileable CONV launch is timed once per distinct shape with 128x128 and with 128x64 tiles at this engine's batch
        (min of `reps` HIP-event timings after a warm-up launch, on whatever the arena holds -- MFMA time does not depend
        on the values) and gets the narrow tile when that is faster by more than 1.5 %.  Both widths produce the same bits
        (identical k order per output element), so the choice is invisible in the results.  ~1 s per engine."""
        import os

        lib = L.lib()
        reps = max(1, int(os.environ.get("HVN_TUNE_REPS", reps)))       # (tests/conftest.py: 1 -- every candidate gives the same bits)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        osz = ctypes.sizeof(L.hvn_op)
        base = ctypes.addressof(self.ops)

        # the encoder launches of a split engine run on sub-batches: their shapes are timed at the sub-batch size
        lanes = getattr(self.plan, "lanes", None)
        enc_end = lanes[0][2] if (lanes and lanes[0][0] == "main" and len(lanes) > 1) else 0
        split = self.n_split if (self.n_split > 1 and self.max_batch >= 2 * self.n_split) else 1
        sub_n = -(-self.max_batch // split)
        if os.environ.get("HVN_TUNE_SUB", "1") == "0":      # A/B knob: time every launch at the full batch (round 3's selection)
            sub_n = self.max_batch

        def time_op(i):
            best = float("inf")
            nb = sub_n if (i < enc_end or self.split_decoder) else self.max_batch
            for r in range(reps + 1):
                e0.record()
                L.check(lib.hvn_run_op(base + i * osz, nb, ctypes.c_void_p(stream)), "hvn_run_op (autotune)")
                e1.record()
                e1.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1))
            return best

        for i, op in enumerate(self.plan.ops):
            if op.kind == PL.OP_CHAIN and op.extra.get("x3"):
                # bf16x3 chain: csrc/hvn_conv_chain_x3.hip (tile_n 128) or, where it exists (conv3 with 64 input channels), the form with the
                # input tile resident in registers and every other operand a chunk ahead in flight (csrc/hvn_conv_chain_x3r.hip, tile_n
                # X3R) -- same bits (tests/test_gpu_chain.py), picked by time.  HVN_CHAIN_X3R=0 | force: never | wherever it exists.
                mode = os.environ.get("HVN_CHAIN_X3R", "1")
                x2 = op.extra.get("x2")
                if mode == "0" or op.x.c != 64 or (x2 is not None and (x2.c != 64 or op.res is not None or op.extra["cout2"] != 64)):
                    continue
                key = ("chain_x3", sub_n if (i < enc_end or self.split_decoder) else self.max_batch, op.x.c, x2.c if x2 is not None else 0, op.cout, op.extra["cout2"],
                       op.y.h, op.y.w, op.res is not None, op.post is not None, op.pre is not None, int(op.extra.get("x3", 0)), mode)
                if key not in self.tile_choice:
                    o = self.ops[i]
                    t = {}
                    for tn in ((X3R,) if mode == "force" else (128, X3R)):
                        o.tile_n = tn
                        t[tn] = time_op(i)
                    best = X3R if (mode == "force" or t[X3R] < margin * t[128]) else 128
                    self.tile_choice[key] = (best, t.get(128, float("nan")), t[X3R], dict(t))
                self.ops[i].tile_n = self.tile_choice[key][0]
                continue
            if op.kind == PL.OP_CHAIN:                     # chained 1x1 convs: 128 or 64 pixels per workgroup (same bits)
                if os.environ.get("HVN_CHAIN_BM"):         # A/B runs: force one
                    self.ops[i].tile_n = int(os.environ["HVN_CHAIN_BM"])
                    continue
                x2 = op.extra.get("x2")
                key = ("chain", sub_n if (i < enc_end or self.split_decoder) else self.max_batch, op.x.c, x2.c if x2 is not None else 0, op.cout, op.extra["cout2"], op.y.h, op.y.w, op.res is not None,
                       op.post is not None, op.pre is not None)
                if key not in self.tile_choice:
                    o = self.ops[i]
                    t = {}
                    for tn in (128, 64):
                        o.tile_n = tn
                        t[tn] = time_op(i)
                    self.tile_choice[key] = (64 if t[64] < margin * t[128] else 128, t[128], t[64])
                self.ops[i].tile_n = self.tile_choice[key][0]
                continue
            if op.kind != PL.OP_CONV or op.tile_n not in (128, 64) or int(op.extra.get("groups", 1)) != 1:
                continue
            x2 = op.extra.get("x2")
            key = (sub_n if (i < enc_end or self.split_decoder) else self.max_batch, op.kh, op.kw, op.stride, op.x.c, op.cout, op.y.h, op.y.w, op.x.h, op.x.w, op.res is not None, op.pre is not None,
                   op.post is not None, int(op.extra.get("nbatch", 1)), x2.c if x2 is not None else 0)
            # 128-channel-wide plans: 128 x 128 or 128 x 64 tiles; 64-wide ones: 128 x 64 or 256 x 64 (tile_n 320 = 64 | 0x100)
            if op.tile_n == 64 and (x2 is not None or op.extra.get("x3")):
                continue                                   # the fused-shortcut and bf16x3 instantiations exist for 128 x 128 and 128 x 64 tiles only
            key = key + (int(op.extra.get("x3", 0)),)
            cands = (128, 64) if op.tile_n == 128 else (64, 320)
            if op.tile_n == 128 and op.extra.get("x3") and x3g_forms_for(op):
                # + the LDS-DMA forms of the bf16x3 kernel (csrc/hvn_conv_x3g.hip): 256 | 128 pixels x 128 channels, same bits
                forced = os.environ.get("HVN_X3G_FORCE")       # tests / A-B runs: that form wherever it exists, no timing
                if forced and int(forced) in x3g_forms_for(op):
                    self.ops[i].tile_n = int(forced)
                    try:
                        time_op(i)
                        continue
                    except L.HvnError:
                        self.ops[i].tile_n = 128
                cands = cands + x3g_forms_for(op)
            key = key + (cands,)                           # (the cache is shared between engines: another candidate set is another question)
            if key not in self.tile_choice:
                o = self.ops[i]
                t = {}
                for tn in cands:
                    o.tile_n = tn
                    try:
                        t[tn] = time_op(i)
                    except L.HvnError:
                        if tn in (128, 64, 320):
                            raise
                        t[tn] = float("inf")       # an LDS-DMA form the launcher refuses for this geometry (32-bit reach of a 256-row tile)
                best = min(cands[1:], key=lambda tn: t[tn])
                self.tile_choice[key] = (best if t[best] < margin * t[cands[0]] else cands[0], t[cands[0]], t[best], dict(t))
            self.ops[i].tile_n = self.tile_choice[key][0]
        torch.cuda.synchronize(self.device)

    def autotune_bf16_forms(self, reps=3, margin=0.985):
        """bf16 engine (BASELINE cfg 3): every CONV launch with >= 128 output channels and no prologue is timed once per distinct
        shape on csrc/hvn_conv_bf16.hip (tile_n 128) and on the two LDS-DMA forms of csrc/hvn_conv_bf16g.hip (256 | 128 pixels x 128
        channels) and keeps the fastest -- same packing and same bits for all three (tests/test_gpu_bf16.py).  HVN_BF16G=0 keeps
        hvn_conv_bf16.hip everywhere, HVN_BF16G_FORCE=896 | 640 takes that form wherever it exists (tests, A/B runs)."""
        import os

        lib = L.lib()
        reps = max(1, int(os.environ.get("HVN_TUNE_REPS", reps)))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        osz = ctypes.sizeof(L.hvn_op)
        base = ctypes.addressof(self.ops)
        forced = os.environ.get("HVN_BF16G_FORCE")

        def time_op(i):
            best = float("inf")
            for r in range(reps + 1):
                e0.record()
                L.check(lib.hvn_run_op(base + i * osz, self.max_batch, ctypes.c_void_p(stream)), "hvn_run_op (autotune)")
                e1.record()
                e1.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1))
            return best

        for i, op in enumerate(self.plan.ops):
            if op.kind != PL.OP_CONV or self.ops[i].tile_n != 128 or op.pre is not None or op.cout < 128 or int(op.extra.get("nbatch", 1)) > 1:
                continue
            x2 = op.extra.get("x2")
            key = ("bf16", self.max_batch, op.kh, op.kw, op.stride, op.x.c, op.cout, op.y.h, op.y.w, op.x.h, op.x.w, op.res is not None,
                   op.post is not None, x2.c if x2 is not None else 0, int(op.extra.get("groups", 1)))
            key = key + (forced,)
            if key not in self.tile_choice:
                t = {}
                for tn in ((int(forced),) if forced else (128, X3G_256, X3G_128)):
                    self.ops[i].tile_n = tn
                    try:
                        t[tn] = time_op(i)
                    except L.HvnError:
                        if tn == 128:
                            raise
                        t[tn] = float("inf")           # a form the launcher refuses for this geometry
                best = min(t, key=t.get)
                if t[best] == float("inf") or (not forced and best != 128 and t[best] >= margin * t[128]):
                    best = 128
                self.tile_choice[key] = (best, t.get(128, float("nan")), t[best], dict(t))
            self.ops[i].tile_n = self.tile_choice[key][0]
        torch.cuda.synchronize(self.device)

    # ---------------------------------------------------------------------------------
    def _set_input(self, imgs):
        o = self.ops[0]
        assert self.plan.ops[0].kind == PL.OP_CONV0
        n = imgs.shape[0]
        if n > self.max_batch:
            raise ValueError("batch %d exceeds the engine's max_batch %d" % (n, self.max_batch))
        g = self.plan.geo["inp"]
        if imgs.dtype == torch.uint8:       # infer_step hands uint8 NHWC (run_desc.py:176)
            if tuple(imgs.shape[1:]) != (g, g, 3):
                raise ValueError("expected uint8 [N,%d,%d,3] patches, got %s" % (g, g, tuple(imgs.shape)))
            imgs = imgs.contiguous()
            o.x_dtype = 0
            o.x.sn, o.x.sy, o.x.sx, o.x.sc = g * g * 3, g * 3, 3, 1
        elif imgs.dtype == torch.float32:   # HoVerNet.forward contract: float NCHW in 0..255
            if tuple(imgs.shape[1:]) != (3, g, g):
                raise ValueError("expected float32 [N,3,%d,%d] images, got %s" % (g, g, tuple(imgs.shape)))
            imgs = imgs.contiguous()
            o.x_dtype = 1
            o.x.sn, o.x.sy, o.x.sx, o.x.sc = 3 * g * g, g, 1, g * g
        else:
            raise TypeError("images must be uint8 NHWC or float32 NCHW, got %s" % imgs.dtype)
        if imgs.device != self.device:
            imgs = imgs.to(self.device, non_blocking=True)
        o.x.base = imgs.data_ptr()
        return imgs, n

    def run_raw(self, n):
        """Run the bound plan on whatever the arena holds (per-kernel tests: no CONV0 input)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(L.lib().hvn_run_plan(ctypes.addressof(self.ops), len(self.ops), n, ctypes.c_void_p(stream)), "hvn_run_plan")

    def _shifted_ops(self, first):
        """A copy of the bound descriptors whose per-sample views start at sample `first`."""
        key = first
        if key not in self._sub_ops:
            ops = (L.hvn_op * len(self.ops))()
            ctypes.memmove(ops, self.ops, ctypes.sizeof(self.ops))
            for i, op in enumerate(self.plan.ops):
                o = ops[i]
                for fld in ("x", "res", "y", "x2", "y2"):
                    v = getattr(o, fld)
                    if v.base:
                        if op.kind == PL.OP_HEAD and fld == "y":
                            v.base += 4 * first * op.y.c * op.y.h * op.y.w
                        elif op.kind == PL.OP_PREDMAP:
                            c = {"x": 2, "res": 2, "y": self.plan.pred_map.c}[fld]
                            v.base += 4 * first * c * self.plan.pred_map.h * self.plan.pred_map.w
                        elif op.kind == PL.OP_CONV0 and fld == "x":
                            pass  # set per call
                        else:
                            v.base += self.isz * first * v.sn
                if op.kind == PL.OP_PREDMAP and o.w:
                    o.w += 4 * first * (self.plan.nr_types or 0) * self.plan.pred_map.h * self.plan.pred_map.w
            self._sub_ops[key] = ops
        return self._sub_ops[key]

    def _launch(self, ops, n_ops, cnt, stream, lane_streams, start=0):
        """One sub-batch from op `start`: encoder on `stream`, decoder branches fanned out over `lane_streams`."""
        lib = L.lib()
        base = ctypes.addressof(ops)
        osz = ctypes.sizeof(L.hvn_op)
        lanes = getattr(self.plan, "lanes", None)
        if not lanes or not lane_streams or n_ops != len(self.ops):
            L.check(lib.hvn_run_plan(base + start * osz, n_ops - start, cnt, ctypes.c_void_p(stream.cuda_stream)), "hvn_run_plan")
            return
        branch_i = 0
        pending = []
        for lane, lo, hi in lanes:
            lo = max(lo, start)
            if lo >= hi:
                continue
            if lane == "main":
                for ev in pending:          # join before the shared epilogue
                    stream.wait_event(ev)
                pending = []
                L.check(lib.hvn_run_plan(base + lo * osz, hi - lo, cnt, ctypes.c_void_p(stream.cuda_stream)), "hvn_run_plan")
            else:
                st = stream if branch_i == 0 else lane_streams[(branch_i - 1) % len(lane_streams)]
                if st is not stream:
                    fork = torch.cuda.Event()
                    fork.record(stream)
                    st.wait_event(fork)
                L.check(lib.hvn_run_plan(base + lo * osz, hi - lo, cnt, ctypes.c_void_p(st.cuda_stream)), "hvn_run_plan")
                if st is not stream:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    pending.append(ev)
                branch_i += 1

    def run(self, imgs, upto=None):
        """imgs: uint8 [N,H,W,3] or float32 [N,3,H,W] -> (logits dict of [N,C,h,w] views, pred_map [N,h,w,3|4] or None).
        The returned tensors alias engine-owned buffers that the next call overwrites."""
        imgs, n = self._set_input(imgs)
        main = torch.cuda.current_stream(self.device)
        n_ops = len(self.ops) if upto is None else upto
        split = self.n_split if (n >= 2 * self.n_split and self.n_split > 1) else 1
        n_lane = self.n_lane_streams
        need = (split - 1) + split * n_lane
        if self._streams is None or len(self._streams) < need:
            self._streams = lane_stream_pool(self.device, self._stream_off + need)[self._stream_off:self._stream_off + need]
        lane_pool = self._streams[split - 1:]
        lanes = getattr(self.plan, "lanes", None)
        enc_end = lanes[0][2] if (lanes and lanes[0][0] == "main" and len(lanes) > 1) else 0
        if split == 1:
            self._launch(self.ops, n_ops, n, main, lane_pool[:n_lane])
        elif not self.split_decoder and enc_end and upto is None:
            # encoder (large launches, tail-quantised) on `split` sub-batch streams; the decoder's small
            # dense-unit launches run on the whole batch, fanned out over the branch lanes
            o0 = self.ops[0]
            esz = 1 if o0.x_dtype == 0 else 4
            fork = torch.cuda.Event()
            fork.record(main)
            bounds = [n * k // split for k in range(split + 1)]
            for k in range(split):
                first, cnt = bounds[k], bounds[k + 1] - bounds[k]
                ops = self._shifted_ops(first)
                ops[0].x = o0.x
                ops[0].x_dtype = o0.x_dtype
                ops[0].x.base = o0.x.base + esz * first * o0.x.sn
                st = main if k == 0 else self._streams[k - 1]
                if k:
                    st.wait_event(fork)
                L.check(L.lib().hvn_run_plan(ctypes.addressof(ops), enc_end, cnt, ctypes.c_void_p(st.cuda_stream)), "hvn_run_plan")
                if k:
                    join = torch.cuda.Event()
                    join.record(st)
                    main.wait_event(join)
            self._launch(self.ops, n_ops, n, main, lane_pool[:n_lane], start=enc_end)
        else:
            o0 = self.ops[0]
            esz = 1 if o0.x_dtype == 0 else 4
            fork = torch.cuda.Event()
            fork.record(main)
            bounds = [n * k // split for k in range(split + 1)]
            for k in range(split):
                first, cnt = bounds[k], bounds[k + 1] - bounds[k]
                ops = self._shifted_ops(first)
                ops[0].x = o0.x
                ops[0].x_dtype = o0.x_dtype
                ops[0].x.base = o0.x.base + esz * first * o0.x.sn
                st = main if k == 0 else self._streams[k - 1]
                if k:
                    st.wait_event(fork)
                self._launch(ops, n_ops, cnt, st, lane_pool[k * n_lane:(k + 1) * n_lane])
                if k:
                    join = torch.cuda.Event()
                    join.record(st)
                    main.wait_event(join)
        self._keepalive = imgs
        logits = {br: t[:n] for br, t in self.logits.items()}
        return logits, (None if self.pred_map is None else self.pred_map[:n])

    def buffer(self, view, n):
        """Test hook: the activation window `view` of the first n samples as a tensor view."""
        b = view.buf
        arena = self.arena if self.dtype == "fp32" else self.arena.view(torch.bfloat16)
        t = arena[:n, b.offset:b.offset + b.size].view(n, b.h, b.w, b.c)
        return t[:, view.y0:view.y0 + view.h, view.x0:view.x0 + view.w, view.c0:view.c0 + view.c]
