"""The paths a USER runs (train.py / eval.py) on the fast forms the benchmark measures: host batches staged through pinned memory and
copied to the device on a side stream while the previous batch computes (DevicePrefetcher), the inference forward as one hipGraph
replay per batch shape (SegmentRunner -> GraphedSegment), the training step as one hipGraph replay with the gradient exchange as a node
of it (make_train_step -> GraphedTrainStep + FlatGradSync + fused AdamW), each with a LOUD eager fallback.

The reference's recipe these serve: /root/reference/website/src/pages/[lang]/reprod/index.astro:238-264 (two GPUs under a
torch.distributed launcher, batch_size 8, learning_rate 1e-4, num_iterations 3000).  Host plumbing only -- no arithmetic of the memory
path lives here; the one computation is the uint8 -> [0, 1] cast of frames a loader delivers as bytes (done on the GPU: a quarter of
the float32 bytes cross PCIe)."""
from __future__ import annotations

import os
import sys
import warnings
from typing import Iterable, Iterator, Optional, Tuple

import torch


class DevicePrefetcher:
    """Iterate a loader of (frames, target) HOST batches as device tensors, `slots` batches deep: batch i+1 is staged into pinned memory
    and copied host-to-device on a side stream while batch i computes; the consumer's stream waits only for its own batch's copy event,
    and a slot's device buffers are not overwritten before the consumer of their previous batch has been passed (an event recorded on the
    consumer's stream when it asks for the next batch).  uint8 frames are cast to `frames_dtype` and scaled to [0, 1] on the device;
    float frames are cast; targets keep their dtype unless `target_dtype` is given -- the casts run on the COPY stream, right behind the
    batch's transfer, into per-slot buffers: the consumer's stream receives finished tensors and spends nothing on them.  Yields
    (frames, target) device tensors that stay valid until the NEXT next().
    ONE slot is kept free: slots - 1 batches are staged ahead, so the slot a new batch is staged into was handed out TWO iterations ago.
    (Round 6: with every slot staged or in use, the copy into a slot waited for the forward launched a moment earlier; the host sat in that
    wait -- the runtime resolves a copy's wait on an unfinished event on the HOST -- and launched the next forward only when the previous one
    had finished: one launch latency of GPU idle time per batch, 1.04 against 0.96 ms.)
    threaded=True: the loader is iterated and the batches are staged by a WORKER THREAD (a slot goes back to the worker when the consumer asks
    for the next batch; an exception in the loader or the staging is re-raised in the consumer).  Off by default: measured at the EchoNet
    shape it is SLOWER (1.23 against 1.05 ms per batch, profiles/r06_v_pipeline_probe.txt) -- the loop is not bound by host work (a graph
    launch is 0.17 ms, the staging calls ~0.1 ms) but by the host-to-device copies themselves, which take 2-3x their stand-alone time
    beside the forward's kernels; a second thread only adds hand-offs.  Useful when the LOADER is slow (decoding in the main process)."""

    def __init__(self, loader: Iterable, device: torch.device, slots: int = 3, frames_dtype: Optional[torch.dtype] = None,
                 target_dtype: Optional[torch.dtype] = None, threaded: bool = False):
        if slots < 2:
            raise ValueError("DevicePrefetcher needs at least two slots")
        self.loader, self.device, self.slots, self.threaded = loader, device, slots, threaded
        self.frames_dtype, self.target_dtype = frames_dtype, target_dtype
        self.stream = torch.cuda.Stream(device=device)
        # the casts run on a stream of their OWN (round 6): behind the copy on the copy stream, every batch made the DMA engine wait for a
        # small kernel that itself waits for room beside the forward's kernels -- copy, cast, copy in one queue ran at 0.53 ms per 19 MB batch
        # with nothing else on the GPU and ~0.75 ms beside the forward, against 0.36 ms for the copies alone
        self.cast_stream = torch.cuda.Stream(device=device) if os.environ.get("GDKVM_PREFETCH_CAST_STREAM", "1") != "0" else self.stream
        self._pinned = [None] * slots           # per slot: the page-locked (frames, target) its copy in flight reads (kept alive)
        self._stage_buf = [None] * slots        # per slot: this class's own pinned staging buffers, re-used while the shape holds
        self._dev = [None] * slots
        self._free = [None] * slots             # event: the consumer is done with this slot's device buffers
        self._copied = [None] * slots           # event: the slot's last host-to-device copy
        self._conv = [None] * slots             # per slot: the cast frames / target (written on the cast stream)
        self._cast_done = [None] * slots        # event: the slot's last cast (reads the raw device buffers the next copy overwrites)
        self.h2d_bytes = 0

    def _stage(self, slot: int, batch) -> Tuple[torch.Tensor, torch.Tensor, torch.cuda.Event]:
        frames, target = batch
        dev = self._dev[slot]
        if dev is None or dev[0].shape != frames.shape or dev[0].dtype != frames.dtype or dev[1].shape != target.shape or dev[1].dtype != target.dtype:
            dev = self._dev[slot] = (torch.empty(frames.shape, dtype=frames.dtype, device=self.device),
                                     torch.empty(target.shape, dtype=target.dtype, device=self.device))
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()    # (the slot's previous host-to-device copy has left its host buffers)
        if frames.is_pinned() and target.is_pinned():
            # already page-locked (DataLoader(pin_memory=True) stages in its own thread): copy straight from it, keep it alive until copied
            pin = (frames, target)
        else:
            pin = self._stage_buf[slot]
            if pin is None or pin[0].shape != frames.shape or pin[0].dtype != frames.dtype or pin[1].shape != target.shape or pin[1].dtype != target.dtype:
                pin = self._stage_buf[slot] = (torch.empty(frames.shape, dtype=frames.dtype, pin_memory=True),
                                               torch.empty(target.shape, dtype=target.dtype, pin_memory=True))
            pin[0].copy_(frames)                # (host memcpy into page-locked memory: what makes the H2D copy asynchronous)
            pin[1].copy_(target)
        self._pinned[slot] = pin
        with torch.cuda.stream(self.stream):
            if self._free[slot] is not None:
                self.stream.wait_event(self._free[slot])
            if self._cast_done[slot] is not None and self.cast_stream is not self.stream:
                self.stream.wait_event(self._cast_done[slot])          # (the slot's raw buffers were last read by its previous cast)
            dev[0].copy_(pin[0], non_blocking=True)
            dev[1].copy_(pin[1], non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.stream)            # (the pinned buffers are free again once this has passed)
        with torch.cuda.stream(self.cast_stream):
            if self.cast_stream is not self.stream:
                self.cast_stream.wait_event(copied)
                if self._free[slot] is not None:
                    self.cast_stream.wait_event(self._free[slot])      # (the cast buffers of the slot are the consumer's inputs)
            out_f, out_t = self._convert(slot, dev[0], dev[1])
            ev = torch.cuda.Event()
            ev.record(self.cast_stream)
            self._cast_done[slot] = ev
        self._copied[slot] = copied
        self.h2d_bytes += frames.numel() * frames.element_size() + target.numel() * target.element_size()
        return out_f, out_t, ev

    def _convert(self, slot: int, frames: torch.Tensor, target: torch.Tensor):
        """The batch in the dtypes the consumer wants, in this slot's own buffers (called under the copy stream)."""
        fdt = (self.frames_dtype or torch.float32) if frames.dtype == torch.uint8 else (self.frames_dtype or frames.dtype)
        tdt = self.target_dtype or target.dtype
        if fdt == frames.dtype and tdt == target.dtype:
            return frames, target
        conv = self._conv[slot]
        fits = lambda buf, src, dt: (buf is None) == (dt == src.dtype) and (buf is None or (buf.shape == src.shape and buf.dtype == dt))
        if conv is None or not fits(conv[0], frames, fdt) or not fits(conv[1], target, tdt):
            conv = self._conv[slot] = (torch.empty(frames.shape, dtype=fdt, device=self.device) if fdt != frames.dtype else None,
                                       torch.empty(target.shape, dtype=tdt, device=self.device) if tdt != target.dtype else None)
        out_f, out_t = frames, target
        if conv[0] is not None:
            if frames.dtype == torch.uint8:
                torch.mul(frames, 1.0 / 255.0, out=conv[0])    # (ONE pass: fp32(byte) * fp32(1/255), rounded once to the buffer's dtype)
            else:
                conv[0].copy_(frames)             # (the cast)
            out_f = conv[0]
        if conv[1] is not None:
            conv[1].copy_(target)
            out_t = conv[1]
        return out_f, out_t

    def _hand_out(self, queue):
        s, f, t, ev = queue.pop(0)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)                       # the consumer's stream waits for ITS batch's copy (and casts) only
        yield f, t
        done = torch.cuda.Event()                # (the consumer came back for the next batch: everything it launched on slot s is in its stream)
        done.record(torch.cuda.current_stream(self.device))
        self._free[s] = done

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        if self.threaded:
            yield from self._iter_threaded()
            return
        queue = []                               # staged batches, oldest first: (slot, frames, target, copy event)
        slot = 0
        for batch in self.loader:
            queue.append((slot,) + self._stage(slot, batch))
            slot = (slot + 1) % self.slots
            if len(queue) < max(self.slots - 1, 1):
                continue                         # slots - 2 copies stay in flight behind the batch being computed, ONE slot stays free (below)
            yield from self._hand_out(queue)
        while queue:
            yield from self._hand_out(queue)

    def _iter_threaded(self):
        import queue as _q
        import threading
        ready: "_q.Queue" = _q.Queue()           # staged batches (slot, frames, target, event) | an exception | None at the end
        free = threading.Semaphore(self.slots)   # slots the worker may stage into (given back when the consumer moves on)
        stop = threading.Event()
        dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())

        def work():
            try:
                if dev.index is not None:
                    torch.cuda.set_device(dev)     # (the current device is per thread)
                slot = 0
                for batch in self.loader:
                    while not free.acquire(timeout=0.1):
                        if stop.is_set():
                            return
                    if stop.is_set():
                        return
                    ready.put((slot,) + self._stage(slot, batch))
                    slot = (slot + 1) % self.slots
                ready.put(None)
            except BaseException as e:           # noqa: BLE001 -- handed to the consumer
                ready.put(e)

        worker = threading.Thread(target=work, name="gdkvm-prefetch", daemon=True)
        worker.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                s, f, t, ev = item
                torch.cuda.current_stream(dev).wait_event(ev)
                yield f, t
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(dev))
                self._free[s] = done             # (written before the slot is released: the worker reads it after acquiring)
                free.release()
        finally:
            stop.set()
            worker.join(timeout=5.0)


class _Pending:
    """What SegmentRunner.submit returns: the batch's result once get() has made the caller's stream wait for it."""

    def __init__(self, out, event=None, clone=True):
        self._out, self._event, self._clone = out, event, clone

    def get(self):
        if self._event is not None:
            torch.cuda.current_stream(self._out[0].device).wait_event(self._event)
        mask, counts = self._out[0], self._out[1]
        return (mask.clone(), None if counts is None else counts.clone()) if self._clone else (mask, counts)


class SegmentRunner:
    """model.segment(frames, target) for a stream of batches: captured GraphedSegments per batch shape (the full batches of a split), an eager
    call -- said on stderr when a capture fails -- for shapes seen fewer than `min_repeats` times (the last, short batch).

    in_flight = 2 (default): TWO forwards in flight -- `in_flight` captures per shape (a single stream inside each, a memory pool of its own),
    used in turn, each replayed on a host stream of its own: batch i + 1 starts while batch i is still running.  submit() copies the batch
    into the graph's input buffers on that stream, makes the CALLER's stream wait for that copy only (so the batch's source -- a
    DevicePrefetcher slot -- is free again as soon as the copy has run, not when the forward has), replays, and returns a handle; get() on
    the handle makes the caller's stream wait for the result.  Keep one batch of lag between submit and get (eval.py does) or call the
    runner directly (`runner(frames, target)` = submit().get(): no overlap).  Batches already in HBM: 0.81 against 0.875 ms per 16 x 32
    frames (model.InFlightSegments, bench.py's timed loop); fed from pinned host memory through a DevicePrefetcher: 0.89 against 0.96-0.99
    (tools/pipeline_trace.py, profiles/r06_aj_pipeline_slack.txt).
    in_flight = 1: one forward at a time, two groups of clips on two streams inside the graph; zero_copy then captures a graph OVER each
    distinct input buffer (a DevicePrefetcher's slots) in one memory pool, so a batch is a bare replay with no copy (up to `max_graphs` per
    shape; beyond that the batch is copied into the first graph's buffers) -- but the slot stays taken until the forward has finished."""

    def __init__(self, model, graph: bool = True, min_repeats: int = 2, zero_copy: bool = True, max_graphs: int = 4, in_flight: int = 2):
        self.model, self.graph, self.min_repeats, self.zero_copy, self.max_graphs = model, graph, min_repeats, zero_copy, max_graphs
        self.in_flight = max(1, in_flight)
        self._graphs, self._seen, self._streams, self._turn = {}, {}, None, {}
        self.replays = self.eager_calls = self.captures = 0

    def _capture(self, shape_key, per_shape, key, frames, target, **kw):
        try:
            from .model import GraphedSegment
            g = GraphedSegment(self.model, frames, target, **kw)
            per_shape[key] = g
            self.captures += 1
            return g
        except Exception as e:                  # noqa: BLE001 -- a failed capture must not end an evaluation: say so, run eagerly
            print(f"[gdkvm] forward not captured for {shape_key[0]} ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            per_shape["failed"] = True
            return None

    def submit(self, frames: torch.Tensor, target: Optional[torch.Tensor] = None) -> _Pending:
        shape_key = (tuple(frames.shape), frames.dtype, None if target is None else target.dtype)
        self._seen[shape_key] = self._seen.get(shape_key, 0) + 1
        per_shape = self._graphs.setdefault(shape_key, {})
        usable = self.graph and self._seen[shape_key] >= self.min_repeats and per_shape.get("failed") is None
        if usable and self.in_flight > 1:
            dev = frames.device
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.in_flight)]
            turn = self._turn.get(shape_key, 0)
            self._turn[shape_key] = (turn + 1) % self.in_flight
            g = per_shape.get(turn)
            if g is None:
                g = self._capture(shape_key, per_shape, turn, frames.clone(), None if target is None else target.clone(), streams=1)
            if g is not None:
                s, cur = self._streams[turn], torch.cuda.current_stream(dev)
                s.wait_stream(cur)              # the batch is ready as far as the caller's stream knows
                with torch.cuda.stream(s):
                    g.frames.copy_(frames, non_blocking=True)
                    if target is not None:
                        g.target.copy_(target, non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record(s)
                    out = g(g.frames, g.target)
                    done = torch.cuda.Event()
                    done.record(s)
                cur.wait_event(copied)          # (the source buffers are free once this has passed: a prefetcher slot may be refilled)
                self.replays += 1
                return _Pending(out, done)
        elif usable:
            ptr_key = (frames.data_ptr(), None if target is None else target.data_ptr()) if self.zero_copy else "copy"
            g = per_shape.get(ptr_key)
            if g is None and len(per_shape) < self.max_graphs:
                first = next((v for k, v in per_shape.items() if k != "failed"), None)
                own = (frames, target) if self.zero_copy else (frames.clone(), None if target is None else target.clone())
                g = self._capture(shape_key, per_shape, ptr_key, own[0], own[1], pool=None if first is None else first.graph.pool())
            elif g is None:
                g = next(v for k, v in per_shape.items() if k != "failed")     # (more buffers than graphs: copy into the first graph's)
            if g is not None:
                self.replays += 1
                return _Pending(g(frames, target))
        self.eager_calls += 1
        with torch.no_grad():
            return _Pending(self.model.segment(frames, target=target), clone=False)

    def __call__(self, frames: torch.Tensor, target: Optional[torch.Tensor] = None):
        return self.submit(frames, target).get()


def make_train_step(model, opt_factory, frames: torch.Tensor, target: torch.Tensor, autocast_dtype, world: int, device,
                    graph: bool = True, opt_state: Optional[dict] = None):
    """The training step train.py runs, in the form bench.py measures: ONE hipGraph replay per step (GraphedTrainStep) over the bare module,
    the gradient all-reduce a node of the graph (FlatGradSync) when world > 1, fused capturable AdamW -- or, loudly, the eager step
    (DistributedDataParallel + the default AdamW) when the capture fails or graph=False.  opt_factory(params, fused: bool, capturable: bool)
    -> optimiser (opt_state: a checkpoint's optimiser state, loaded before the capture).  Returns (step(frames, target) -> loss, optimiser,
    description dict); the shapes are those of `frames` / `target`.  The capture's warm-up runs info["warmup_steps"] REAL optimiser steps on
    the batch given here: the caller counts them as iterations."""
    from .train import FlatGradSync, GraphedTrainStep, train_step, wrap_ddp
    info = {"launch": "eager", "grad_sync": "none", "optimizer": "AdamW"}
    if graph:
        sync = None
        try:
            if world > 1:
                sync = FlatGradSync(model)
                sync.broadcast_parameters()
            opt = opt_factory(model.parameters(), True, True)
            if opt_state is not None:
                opt.load_state_dict(opt_state)
            gstep = GraphedTrainStep(model, opt, frames, target, autocast_dtype, warmup=2, grad_sync=sync)
            info.update(launch="one hipGraph replay per step (GraphedTrainStep)", optimizer="AdamW(fused, capturable)", warmup_steps=gstep.eager_steps,
                        grad_sync="FlatGradSync: one flat-bucket all-reduce, a node of the graph" if sync is not None else "none (one rank)")
            return (lambda f, t: gstep(f, t)), opt, info
        except Exception as e:                  # noqa: BLE001
            warnings.warn(f"gdkvm_amd: the training step was not captured ({type(e).__name__}: {e}); falling back to the EAGER step "
                          "(DistributedDataParallel, ~1.6x slower per step on this model)", RuntimeWarning)
            print(f"[gdkvm] training step not captured ({type(e).__name__}: {e}); running the eager step", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
    ddp = wrap_ddp(model, device)
    opt = opt_factory(model.parameters(), False, False)
    if opt_state is not None:
        opt.load_state_dict(opt_state)
    info.update(grad_sync="DistributedDataParallel" if ddp is not model else "none (one rank)")
    return (lambda f, t: train_step(ddp, opt, f, t, autocast_dtype)), opt, info
