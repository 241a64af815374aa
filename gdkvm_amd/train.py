"""Training-side host code for the GDKVM module: loss, one optimisation step, DDP wrapping.

The reference's harness is not in the snapshot; its guide names the recipe this mirrors: batch_size 8,
learning_rate 1.0e-4, num_iterations 3000, two GPUs under a torch.distributed launcher
(/root/reference/website/src/pages/[lang]/reprod/index.astro:238-252).  One process per GPU; the only exchange per
step is the DDP gradient all-reduce (RCCL over xGMI on the GPU box, gloo in CPU tests) -- the memory path itself shards
over clips with no collective."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


def segmentation_loss(logits: torch.Tensor, target: torch.Tensor, dice_weight: float = 1.0, eps: float = 1.0) -> torch.Tensor:
    """Cross-entropy + soft Dice over [B,T,ncls,H,W] logits and [B,T,H,W] integer labels (fp32 math).  Labels outside
    [0, ncls) are unlabelled pixels (EchoNet-Dynamic: every frame but the two traced ones): left out of the cross-entropy mean and of every
    Dice sum -- intersection, target AND prediction mass (gdkvm_seg_loss_fwd does the same)."""
    B, T, C, H, W = logits.shape
    lg = logits.reshape(B * T, C, H, W).float()
    tg = target.reshape(B * T, H, W).long()
    labelled = (tg >= 0) & (tg < C)                 # anything else (255 in annotation masks) carries no class
    # sum over the labelled pixels / their count, with the count clamped to 1: a batch without a labelled pixel gives 0 (as
    # seg_loss_finalize_kernel does; F.cross_entropy's own mean would be NaN and reach AdamW) -- and no host branch, i.e. no
    # device-to-host synchronisation in front of the backward
    ce = F.cross_entropy(lg, torch.where(labelled, tg, torch.full_like(tg, -100)), ignore_index=-100, reduction="sum") \
        / labelled.sum().clamp_min(1)
    p = lg.softmax(1) * labelled.unsqueeze(1)       # an unlabelled pixel adds to no sum at all: not to the prediction mass P_c either
    oh = (F.one_hot(torch.where(labelled, tg, torch.zeros_like(tg)), C) * labelled.unsqueeze(-1)).permute(0, 3, 1, 2).float()
    inter = (p * oh).sum((0, 2, 3))
    dice = 1.0 - ((2 * inter + eps) / (p.sum((0, 2, 3)) + oh.sum((0, 2, 3)) + eps)).mean()
    return ce + dice_weight * dice


def segmentation_loss_lowres(lowres_logits: torch.Tensor, target: torch.Tensor, dice_weight: float = 1.0, eps: float = 1.0) -> torch.Tensor:
    """segmentation_loss(bilinear_upsample(lowres_logits -> target size), target) as one fused HIP forward / backward
    (ops.seg_loss): the full-resolution logits, their softmax and one-hot tensors never exist.  [B,T,ncls,h,w] logits."""
    from . import ops
    B, T, C, h, w = lowres_logits.shape
    return ops.seg_loss(lowres_logits.reshape(B * T, C, h, w), target.reshape(B * T, *target.shape[-2:]), dice_weight, eps)


def wrap_ddp(model: nn.Module, device: Optional[torch.device] = None, bucket_cap_mb: int = 25, force: bool = False) -> nn.Module:
    """DistributedDataParallel over the default process group (gradient all-reduce bucketed and overlapped with the
    backward).  xGMI is point-to-point, so buckets are kept large enough to amortise ring latency: the whole model is
    ~16 MB of fp32 gradients, i.e. one or two buckets.  A single process needs no wrapper and gets the bare model back;
    ``force`` wraps even at world size 1 (an initialised group is still required): the reducer's hooks, bucket views and
    the RCCL all-reduce then run for real on a one-rank group, which is how the one-GPU tests exercise this path."""
    import torch.distributed as dist
    if not dist.is_initialized():
        if force:
            raise RuntimeError("wrap_ddp(force=True) needs an initialised process group")
        return model
    if dist.get_world_size() == 1 and not force:
        return model
    ids = None if device is None or device.type != "cuda" else [device.index]
    return nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                               gradient_as_bucket_view=True)


class FlatGradSync:
    """The step's ONE exchange as ONE collective on the stream the step runs on: after the backward every gradient is copied into its slice
    of one flat fp32 bucket (a few multi-tensor copy launches), the bucket is all-reduced to the mean over the ranks (RCCL over xGMI on the GPU
    box, gloo in CPU tests), and the parameters' ``.grad`` become views of it (same shape AND strides as the parameter: the fused optimiser
    pairs elements by address) -- what DistributedDataParallel does with its reducer, hooks and bucket views, in a form a HIP graph can hold:
    no autograd hooks, no host-side bucket bookkeeping, one communication node.  The whole model is ~16 MB of fp32 gradients: one ring
    all-reduce of 2 (P-1)/P x 16 MB over 7 x 153 GB/s links is ~0.1 - 0.2 ms against a 5.7 ms step, so it is not overlapped with the backward.
    Use with the BARE module (not wrapped): ``sync = FlatGradSync(model); sync.broadcast_parameters(); train_step(..., grad_sync=sync)``.
    The reference's own recipe is a multi-process launch (/root/reference/website/src/pages/[lang]/reprod/index.astro:238-249).

    Two differences from DistributedDataParallel, both deliberate: (i) a trainable parameter that received NO gradient in a step raises
    (as DDP does on its next iteration without find_unused_parameters) -- zero-filling it would let the optimiser decay weights and
    moments of a parameter that took no part in the step, which a one-process step never does; ``allow_unused=True`` asks for exactly
    that zero-fill (the gradient is then a true zero on every rank).  (ii) buffers (BatchNorm running statistics) are broadcast once, by
    ``broadcast_parameters()``; DDP re-broadcasts rank 0's buffers before every forward.  Every rank's running statistics then follow its
    own shard (the trained weights are identical on all ranks regardless); call ``broadcast_buffers()`` before saving a checkpoint or
    evaluating if rank 0's statistics are wanted everywhere."""

    def __init__(self, model: nn.Module, group=None, allow_unused: bool = False):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("FlatGradSync needs an initialised process group")
        self.group = group
        self.allow_unused = allow_unused
        self.names = {id(p): n for n, p in model.named_parameters()}
        self.world = dist.get_world_size(group)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buffers = list(model.buffers())
        if not self.params:
            raise RuntimeError("FlatGradSync: the model has no trainable parameter")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise RuntimeError("FlatGradSync: parameters must share one device and dtype")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self.views, off = [], 0
        for p in self.params:
            n = p.numel()
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            self.views.append(self.flat[off:off + n].as_strided(p.shape, p.stride()) if dense else self.flat[off:off + n].view(p.shape))
            off += n
        # gloo has no AVG: SUM then a scale (CPU tests); RCCL averages inside the collective
        self._avg = dev.type == "cuda"

    def broadcast_parameters(self, src: int = 0) -> None:
        """Every rank starts from rank `src`'s weights and buffers (what DistributedDataParallel does at construction)."""
        import torch.distributed as dist
        for t in list(self.params) + self.buffers:
            dist.broadcast(t.data, src, group=self.group)

    def broadcast_buffers(self, src: int = 0) -> None:
        """Rank `src`'s buffers (BatchNorm running statistics, step counters) to every rank: DDP does this before each forward, this class
        only on request (checkpoints, evaluation)."""
        import torch.distributed as dist
        for t in self.buffers:
            dist.broadcast(t.data, src, group=self.group)

    def __call__(self) -> None:
        import torch.distributed as dist
        missing = [v for p, v in zip(self.params, self.views) if p.grad is None]
        if missing:
            if not self.allow_unused:
                names = [self.names.get(id(p), "?") for p in self.params if p.grad is None]
                raise RuntimeError(f"FlatGradSync: no gradient for {names[:4]}{' ...' if len(names) > 4 else ''} in this step; keep every trainable "
                                   "parameter in the graph (as DistributedDataParallel requires), freeze it, or pass allow_unused=True")
            torch._foreach_zero_(missing)                    # (allow_unused: a parameter outside this step's graph contributes zero on every rank)
        have = [(v, p.grad) for p, v in zip(self.params, self.views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if self._avg:
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v


def train_step(model: nn.Module, opt: torch.optim.Optimizer, frames: torch.Tensor, target: torch.Tensor,
               autocast_dtype: Optional[torch.dtype] = None, grad_sync=None) -> torch.Tensor:
    """forward -> loss -> backward (HIP backward kernels; DDP all-reduce if wrapped) -> [grad_sync(): FlatGradSync on a bare module] ->
    optimiser step."""
    opt.zero_grad(set_to_none=True)
    fused = frames.is_cuda                      # GPU: the objective is evaluated on the stride-4 logits by the HIP loss kernels
    kw = {"_lowres": True} if fused else {}
    if autocast_dtype is not None:
        with torch.autocast(frames.device.type, dtype=autocast_dtype):
            logits = model(frames, **kw)
    else:
        logits = model(frames, **kw)
    loss = segmentation_loss_lowres(logits, target) if fused else segmentation_loss(logits, target)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    opt.step()
    from .model import weights_changed
    weights_changed(model)     # (fused optimisers write the parameters without bumping their version counters: THIS model's pack caches key on it)
    return loss.detach()


def fit_synthetic(model: nn.Module, steps: int, clips: int = 8, frames: int = 8, size: int = 112, num_classes: int = 2, lr: float = 2e-3,
                  seed: int = 0, device: Optional[torch.device] = None, autocast_dtype: Optional[torch.dtype] = torch.bfloat16):
    """A short fit on the seeded synthetic echo clips (gdkvm_amd.data.SyntheticEchoClips): `steps` AdamW steps of `clips` fresh clips
    each.  Not a training recipe -- it exists so that parity figures (Dice of one build's masks against another's, against the CPU
    reference's, against the labels) are measured on a head that separates its classes by real margins instead of a random-init one
    whose logits differ by 1e-3.  Returns the list of losses."""
    from .data import SyntheticEchoClips
    dev = device or next(model.parameters()).device
    ds = SyntheticEchoClips(steps * clips, frames, size, num_classes, seed=seed)
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    model.train()
    losses = []
    for s in range(steps):
        items = [ds[s * clips + j] for j in range(clips)]
        x = torch.stack([a for a, _ in items]).to(dev)
        y = torch.stack([b for _, b in items]).to(dev)
        losses.append(float(train_step(model, opt, x, y, autocast_dtype if dev.type == "cuda" else None)))
    model.eval()
    return losses


class GraphedTrainStep:
    """train_step captured ONCE into a HIP graph and replayed: a training step of this model is ~420 kernel launches of 5 - 100 us, and with
    the step's kernels down to 6.5 ms the eager loop is bound by the HOST (Python autograd + launch calls: 6.5 - 8.6 ms per step depending
    on the box's CPU, measured round 4) -- the graph replays the same kernels in the same order with no host work in between.
    Several ranks: the bare module with grad_sync=FlatGradSync(model) -- the gradient all-reduce is then ONE collective node of the same graph
    (a DistributedDataParallel wrapper is refused: its reducer is host logic); shapes are fixed at construction; the optimiser must be
    capturable (torch.optim.AdamW(..., fused=True, capturable=True)).  The first `warmup` steps run eagerly on a side stream (library
    convolutions pick their solvers, the HIP library sets its kernel attributes, the optimiser creates its state), then one step is
    captured; every call copies the batch into the graph's input buffers and replays -- same arithmetic, same order, same results as
    train_step (tools/graph_step_diag.py prints both loss sequences side by side; tests/test_train_side_gpu.py asserts them equal)."""

    def __init__(self, model: nn.Module, opt: torch.optim.Optimizer, frames: torch.Tensor, target: torch.Tensor,
                 autocast_dtype: Optional[torch.dtype] = None, warmup: int = 3, grad_sync=None):
        if not frames.is_cuda:
            raise RuntimeError("GraphedTrainStep needs device tensors")
        if isinstance(model, nn.parallel.DistributedDataParallel):
            raise RuntimeError("GraphedTrainStep takes the bare module; several ranks: pass grad_sync=FlatGradSync(model) (one captured all-reduce)")
        self.model, self.opt, self.autocast_dtype, self.grad_sync = model, opt, autocast_dtype, grad_sync
        self.frames, self.target = frames.clone(), target.clone()
        side = torch.cuda.Stream(device=frames.device)
        side.wait_stream(torch.cuda.current_stream(frames.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.loss = train_step(model, opt, self.frames, self.target, autocast_dtype, grad_sync)
        torch.cuda.current_stream(frames.device).wait_stream(side)
        torch.cuda.synchronize(frames.device)
        self.graph = torch.cuda.CUDAGraph()
        # (with a collective in the step the process group's watchdog THREAD polls the events of the warm-up steps' all-reduces while this
        # thread captures; under the default "global" capture mode its hipEventQuery aborts the process with "operation not permitted when
        # stream is capturing" -- measured round 5 -- so only this thread's calls are policed)
        import torch.distributed as dist
        mode = {"capture_error_mode": "thread_local"} if (grad_sync is not None or (dist.is_available() and dist.is_initialized())) else {}
        with torch.cuda.graph(self.graph, **mode):
            self.loss = train_step(model, opt, self.frames, self.target, autocast_dtype, grad_sync)
        self.eager_steps = max(1, warmup)       # optimiser steps taken before the first replay (the capture itself runs no kernel)
        # the captured kernels address the per-weight packs (re-packed by the graph itself every replay) and other buffers the warm-up steps
        # allocated outside the graph's pool: held here, so that GDKVM.invalidate_packed_weights() (an eval() / train() toggle between steps)
        # cannot free what a replay writes
        from .model import _packs_held
        inner = getattr(model, "module", model)
        self._held = _packs_held(inner) if hasattr(inner, "invalidate_packed_weights") else []

    def __call__(self, frames: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if frames.shape != self.frames.shape or target.shape != self.target.shape:
            raise RuntimeError(f"GraphedTrainStep was captured for {tuple(self.frames.shape)} / {tuple(self.target.shape)}")
        if frames.data_ptr() != self.frames.data_ptr():
            self.frames.copy_(frames, non_blocking=True)
        if target.data_ptr() != self.target.data_ptr():
            self.target.copy_(target, non_blocking=True)
        self.graph.replay()
        from .model import weights_changed
        weights_changed(self.model)
        return self.loss.clone()                # (the graph's own output buffer is overwritten by the next replay)
