"""Two-stream tile pipeline: the instance separation of batch i runs on a side HIP stream while the
network of batch i+1 runs on the main stream.

The reference overlaps these two stages with a CPU process pool next to the GPU loop
(/root/reference/infer/tile.py:232-234, 361-386).  Here both stages are on the GPU: the network
saturates the matrix cores, the post-processing is a chain of small latency-bound launches (its
watershed uses one workgroup per tile), so running them concurrently hides the latter almost entirely.
Hand-over is a 3.3 MB device copy of the prediction maps into one of two ping-pong buffers, ordered with
HIP events (no host sync anywhere in `submit`).
"""
import torch

from . import post_proc, run_desc


class TilePipeline:
    def __init__(self, model, nr_types=None, return_centroids=True, device=None):
        self.model = model
        self.nr_types = nr_types
        self.return_centroids = return_centroids
        net = model.module if hasattr(model, "module") and not hasattr(model, "engine") else model
        self.device = torch.device(device) if device is not None else next(net.parameters()).device
        self.side = torch.cuda.Stream(self.device)
        self._pp = post_proc.PostProc(self.device)   # own workspace: used on the side stream only
        self._buf = [None, None]
        self._free = [None, None]                    # event: the side stream is done reading _buf[k]
        self._n = 0
        self._last = None
        # host batches (the reference contract: infer_step takes a CPU uint8 tensor) go up on their own copy stream into one of
        # two device slots, so the H2D of batch i+1 runs under the network of batch i
        self.h2d = torch.cuda.Stream(self.device)
        self._in = [None, None]
        self._in_free = [None, None]                 # event: the network is done reading _in[k]
        # optional diagnosis (bench.py --gpus N): HIP events around every `gather` call on the side stream
        self.time_gather = False
        self.gather_events = []

    def submit(self, tiles_u8, extra_maps=None, gather=None, to_host=False):
        """Network on the current stream, post-processing on the side stream.  Returns
        (inst, records, counts) tensors that are valid after `wait()` (or after the side
        stream is otherwise synchronised).  `extra_maps`: optional additional [N,h,w,C] maps to
        post-process in the same side-stream slot (bench.py's structured workload).
        `gather`: optional callable applied to the result tuple on the side stream (the multi-GPU
        path hands `infer_tile.gather_to_rank0` here, so the RCCL gather overlaps the next network
        pass too).  `to_host=True`: the results are copied to pinned host memory on the side stream
        (the reference's contract ends on the host: infer/tile.py:308-316); the returned tensors are
        then the pinned buffers of this slot, overwritten two submits later."""
        main = torch.cuda.current_stream(self.device)
        k = self._n & 1
        self._n += 1
        if not tiles_u8.is_cuda:
            tiles_u8 = self._upload(tiles_u8, k, main)
        pred = run_desc.infer_step_device(tiles_u8, self.model)       # aliases the engine's buffer
        if self._in[k] is not None and tiles_u8 is self._in[k]:
            self._in_free[k] = torch.cuda.Event()
            self._in_free[k].record(main)
        if self._free[k] is not None:
            main.wait_event(self._free[k])                             # ping-pong slot k is free again
        if self._buf[k] is None or self._buf[k].shape != pred.shape:
            self._buf[k] = torch.empty_like(pred)
        self._buf[k].copy_(pred)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            out = self._run_pp(self._buf[k])
            if extra_maps is not None:
                out = self._run_pp(extra_maps)
            if gather is not None:
                if self.time_gather:
                    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g0.record(self.side)
                out = gather(out)
                if self.time_gather:
                    g1.record(self.side)
                    self.gather_events.append((g0, g1))
            if to_host and out is not None:
                out = self._to_host(out, k)
            self._free[k] = torch.cuda.Event()
            self._free[k].record(self.side)
        self._last = out
        return out

    def _upload(self, tiles_host, k, main):
        if tiles_host.dtype != torch.uint8:
            tiles_host = tiles_host.to(torch.uint8)
        if self._in[k] is None or self._in[k].shape != tiles_host.shape:
            self._in[k] = torch.empty(tiles_host.shape, dtype=torch.uint8, device=self.device)
            self._in_free[k] = None
        with torch.cuda.stream(self.h2d):
            if self._in_free[k] is not None:
                self.h2d.wait_event(self._in_free[k])
            self._in[k].copy_(tiles_host, non_blocking=True)          # asynchronous when the source is pinned
            up = torch.cuda.Event()
            up.record(self.h2d)
        main.wait_event(up)
        return self._in[k]

    def _run_pp(self, maps):
        inst = self._pp.separate(maps)
        if self.return_centroids or self.nr_types is not None:
            rec, counts = self._pp.table(inst, maps, self.nr_types)
            return inst, rec, counts
        return inst, None, None

    def _to_host(self, out, k):
        """D2H of (inst, records, counts) into this slot's pinned buffers, asynchronous on the side stream."""
        if not hasattr(self, "_pin"):
            self._pin = [None, None]
        bufs = self._pin[k]
        if bufs is None or any((b is None) != (t is None) or (t is not None and (b.shape != t.shape or b.dtype != t.dtype))
                               for b, t in zip(bufs, out)):
            bufs = tuple(None if t is None else torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in out)
            self._pin[k] = bufs
        for b, t in zip(bufs, out):
            if t is not None:
                b.copy_(t, non_blocking=True)
        return bufs

    def flood_stats(self):
        """Replay statistics of the watershed of the last network output this pipeline post-processed (waits for the side stream)."""
        self.side.synchronize()
        return self._pp.flood_stats(self.side)

    def gather_ms(self):
        """Mean duration (ms) of the timed `gather` calls since the last call of this method (side stream synchronised first)."""
        self.side.synchronize()
        ev, self.gather_events = self.gather_events, []
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else 0.0

    def wait(self):
        self.side.synchronize()
        return self._last
