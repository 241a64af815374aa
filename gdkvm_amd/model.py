"""GDKVM nn.Module -- the Python seam of the drop-in boundary.

The reference's own module is not in the snapshot (/root/reference/README.md:1 points at an un-vendored
repo), so the surface below is the builder's SPEC-v0: ``forward(frames[B,T,C,H,W], mask0=None, state=None,
return_state=False) -> logits[B,T,ncls,H,W] (, state)`` with parameters grouped ``encoder.* / key_proj.* /
query_proj.* / value_proj.* / gate_proj.* / kpff.* / decoder.*`` and one ``_KEY_REMAP`` table applied in
``load_state_dict`` for the day real checkpoints are visible.

The memory path between encoder and decoder -- LKVA read, GDR write, KPFF -- is ``ops.scan_fwd`` / ``ops.kpff_fwd``
(hand-written HIP behind include/gdkvm.h).  The CNN either side of it is hand-written too in the
inference build (``fuse_for_inference()``: every convolution runs on csrc/conv3x3_*.hip, conv_igemm.hip, stem_conv_pool.hip
with its epilogue inside); the training build runs every convolution of the default model on hand-written kernels too since round 5 (the
stem since round 4; the two strided blocks on csrc/conv_s2_train.hip), forward and backward, deterministically.  There is no
eager fallback for the memory path: on a CPU tensor or without libgdkvm_hip.so the forward raises.
"""
from __future__ import annotations

from dataclasses import dataclass
import os
from typing import NamedTuple, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_RULES = {"gated_linear": ops.RULE_GATED_LINEAR, "delta_parallel": ops.RULE_DELTA_PARALLEL,
          "delta_sequential": ops.RULE_DELTA_SEQUENTIAL}

# checkpoint-key prefixes of the real model code -> ours; empty until that code is visible
_KEY_REMAP: dict = {}


@dataclass
class GDKVMConfig:
    in_channels: int = 3
    num_classes: int = 2            # EchoNet-Dynamic: background / LV.  CAMUS: 4.
    heads: int = 1                  # Hh
    key_dim: int = 64               # Dk per head (the HIP kernels are specialised for 64)
    value_dim: int = 256            # Dv per head
    pixel_dim: int = 256            # Cp, stride-16 encoder feature
    widths: Tuple[int, int, int] = (64, 128, 256)
    rule: str = "delta_sequential"
    stride: int = 16
    # Inference on long clips (BASELINE.json configs[4]): how the memory scan treats the time axis of ONE forward call.
    #   1 (default)  the serial recurrence, gdkvm_scan_fwd: a clip processed as consecutive calls with the state carried
    #                (GDKVM.segment_clip) is bit-identical to one call;
    #   0            gdkvm_scan_fwd_segmented with the segment count it picks for the shape (a power of two >= 4 when the serial grid
    #                leaves more than half the CUs idle -- 2 clips x 512 frames: 16 -- else the serial scan); n > 1: that many.
    #                Equal to the serial scan up to fp32 re-association through the segments' transition matrices, NOT bit for bit.
    scan_segments: int = 1
    # SURVEY.md A.1 flag: carry z [B,Hh,Dk] ("the same recurrence on v == 1") beside S and divide the read-out by |q . z| + normalizer_eps.
    # Inference only.  The module's `state` is then [B,Hh,Dk,Dv+1]: S with z as one more column.
    normalizer: bool = False
    normalizer_eps: float = 1e-6
    # per-frame `step` mode -- the value written for frame t carries the embedding (mask_embed) of the mask
    # PREDICTED for frame t (of mask0 for frame 0 when it is given), XMem-style: read -> KPFF -> decoder -> mask -> write, frame by frame.
    # Inference only; the time loop is then ~12 launches per frame instead of one scan launch per chunk (GDKVM.segment captures it in
    # one hipGraph all the same: GraphedSegment).
    mask_feedback: bool = False


def _bn(c):
    return nn.BatchNorm2d(c)


# Bumped by weights_changed(): part of every weight-pack key.  Version counters alone do not see every write -- in-place writes through
# ``.data`` bump nothing, and neither does torch.optim.AdamW(fused=True) (round 5) -- so whoever steps an optimiser says so here
# (gdkvm_amd.train.train_step and GraphedTrainStep do).
_WEIGHTS_EPOCH = [0]


def weights_changed(model: Optional[nn.Module] = None) -> None:
    """Tell the packed-weight caches of the inference paths that parameters were written (an optimiser step): the packs are rebuilt on
    their next use.  Cheap (a counter); call it after any write the version counters cannot show.  With `model` (a GDKVM, or a wrapper
    holding one as ``.module``) only THAT model's caches and graphs are told -- a frozen teacher / EMA copy in the same process keeps its
    packs and its captured GraphedSegment (train_step / GraphedTrainStep pass the model they stepped); without an argument every model
    in the process is told (the process-wide epoch)."""
    inner = getattr(model, "module", model)
    cell = None if inner is None else inner.__dict__.get("_epoch_cell")
    if cell is None:
        _WEIGHTS_EPOCH[0] += 1
    else:
        cell[0] += 1


def _epoch_of(owner) -> Tuple[int, int]:
    """(process-wide epoch, the owning model's epoch): GDKVM hands its ``_epoch_cell`` to the sub-modules that cache packs."""
    cell = None if owner is None else owner.__dict__.get("_epoch_cell")
    return (_WEIGHTS_EPOCH[0], 0 if cell is None else cell[0])


def _wkey(owner, *tensors):
    """Cache key of a weight pack: the weights epochs (weights_changed: process-wide and the owning model's), and version counter AND
    storage address of EVERY source tensor.
    In-place writes through ``.data`` (``p.data.copy_()``: EMA swaps, weight surgery) and fused optimiser steps bump no counter -- after
    such a write call ``weights_changed(model)`` or ``GDKVM.invalidate_packed_weights()`` (load_state_dict / .to() / a train() / eval()
    mode CHANGE / fuse_for_inference() do the latter themselves; train_step / GraphedTrainStep the former)."""
    return _epoch_of(owner) + tuple((t._version, t.data_ptr()) for t in tensors)


# Convolutions the hand-written kernels do not serve (other `widths`, odd sizes, fp32 inference) run on the framework's library: allowed
# (BASELINE.json north_star keeps the CNN on PyTorch-ROCm), but never silently on a GPU tensor -- one warning per layer and reason, and
# GDKVM_STRICT=1 turns it into an error (CI / benchmark runs: the measured path must be the hand-written one).
_STRICT = os.environ.get("GDKVM_STRICT", "0") == "1"
_FALLBACKS_SEEN: set = set()


def _library_fallback(x: torch.Tensor, layer: str, why: str) -> None:
    """Called right before a GPU convolution leaves the hand-written path."""
    if not x.is_cuda:
        return                                              # (the CPU reference module is plain torch by design)
    msg = f"gdkvm_amd: {layer} runs on the framework's library kernel, not on csrc/*.hip ({why})"
    if _STRICT:
        raise RuntimeError(msg + " [GDKVM_STRICT=1]")
    if (layer, why) not in _FALLBACKS_SEEN:
        _FALLBACKS_SEEN.add((layer, why))
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _describe(conv: nn.Conv2d, x: torch.Tensor) -> str:
    return (f"Conv2d({conv.in_channels}->{conv.out_channels}, k={tuple(conv.kernel_size)}, s={tuple(conv.stride)}) on "
            f"{tuple(x.shape)} {str(x.dtype).replace('torch.', '')}")


_BN_COUNTED_BY_MODEL = [False]
# The strided 3x3 layers of the inference build: gdkvm_conv_bias_act's general implicit-GEMM kernel with the epilogue inside
# (default), or the library convolution + one epilogue pass (GDKVM_CONV_IGEMM=0).  Measured equal on the EchoNet shapes -- 1.013 /
# 1.020 ms against 1.021 ms per cfg2 forward -- so the hand-written path is taken: no solver search, one launch.
_IGEMM_STRIDED = os.environ.get("GDKVM_CONV_IGEMM", "1") != "0"
# SURVEY.md §8f row n4 in the inference build: key / query / value projections, both gate logits and the key / query norms in ONE
# launch over the pixel feature (ops.proj_gates), the scan taking the norms as given -- or (GDKVM_PROJ_GATES=0, the A/B switch
# behind DESIGN.md §8 n4's numbers) the three-launch form: ops.proj_rows, ops.gate_logits, norms inside gdkvm_scan_prep.
_PROJ_GATES = os.environ.get("GDKVM_PROJ_GATES", "1") != "0"
# The inference stem reading the NCHW frames itself (round 4: ops.stem_conv_pool_nchw), or (GDKVM_STEM_NCHW=0, the A/B switch) the
# space-to-depth pass followed by the stem kernel; bit-identical.
_STEM_NCHW = os.environ.get("GDKVM_STEM_NCHW", "1") != "0"
# training stem: BatchNorm + ReLU + max-pool as one op in both directions (ops.bn_relu_pool); "0" = bn_act then maxpool3x3s2 (A/B switch)
_STEM_BN_POOL = os.environ.get("GDKVM_STEM_BN_POOL", "1") != "0"
# training: the decoder's 1x1 head on gdkvm_head_logits / gdkvm_head_bwd (deterministic gradients); "0" = the library convolution (A/B switch)
_TRAIN_HEAD_HIP = os.environ.get("GDKVM_TRAIN_HEAD_HIP", "1") != "0"
# training: a residual block's input as two outputs of its first convolution's node (ops.conv3x3_fork); "0" = the framework adds the gradients
_TRAIN_CONV_FORK = os.environ.get("GDKVM_TRAIN_CONV_FORK", "1") != "0"
# training: a strided block's 3x3 / stride-2 convolution + 1x1 branch on csrc/conv_s2_train.hip (deterministic); "0" = the library convolutions (A/B switch)
_TRAIN_CONV_S2 = os.environ.get("GDKVM_TRAIN_CONV_S2", "1") != "0"
# segment_clip(graph=True): the next chunk's encoder + projections beside the current chunk's memory path and decoder (PipelinedClip); "0" = one
# whole-forward graph per chunk, chunks strictly one after the other (A/B switch; same bits)
_CLIP_PIPELINE = os.environ.get("GDKVM_CLIP_PIPELINE", "1") != "0"
# training: key / query / value / gate projections as one stacked product (ops.token_projections); "0" = one product each (A/B switch)
_TRAIN_PROJ_STACKED = os.environ.get("GDKVM_TRAIN_PROJ_STACKED", "1") != "0"


def _bn_act(bn: nn.BatchNorm2d, x: torch.Tensor, relu: bool, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(bn(x) (+ residual)).  Training mode on the GPU runs the fused HIP passes of csrc/bn.hip in both directions
    (ops.bn_act: statistics, normalise + add + ReLU; the library BatchNorm kernels were 38 % of a training step); eval mode of
    the un-folded model and the CPU reference module use torch."""
    v = 8 if x.dtype == torch.bfloat16 else 4
    if (x.is_cuda and bn.training and bn.affine and x.dtype in (torch.bfloat16, torch.float32) and bn.weight.dtype == torch.float32
            and x.shape[1] % v == 0 and x.shape[1] // v <= 256):
        momentum = bn.momentum
        if bn.track_running_stats:
            if not _BN_COUNTED_BY_MODEL[0]:                 # GDKVM.forward bumps all counters in one foreach launch
                bn.num_batches_tracked.add_(1)
            if momentum is None:
                momentum = 1.0 / float(bn.num_batches_tracked)
        res = None if residual is None else residual.to(x.dtype)
        return ops.bn_act(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                          bn.running_var if bn.track_running_stats else None, res, momentum or 0.0, bn.eps, relu)
    if _BN_COUNTED_BY_MODEL[0] and bn.training and bn.track_running_stats:
        bn.num_batches_tracked.sub_(1)                      # torch's own layer counts this step itself
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y, inplace=True) if relu else y


def _conv(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """A training-build convolution: the stride-1 3x3 layers with channel counts in multiples of 64 run forward, data gradient
    and weight gradient on the hand-written kernels (ops.conv3x3: csrc/conv3x3_tile.hip, conv3x3_wgrad.hip), everything else is
    conv(x) (MIOpen)."""
    if (torch.is_grad_enabled() and isinstance(conv, nn.Conv2d) and conv.bias is None and conv.padding_mode == "zeros"
            and ops.conv3x3_train_served(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)):
        return ops.conv3x3(x, conv.weight)
    if torch.is_grad_enabled():
        _library_fallback(x, _describe(conv, x), "training build: only stride-1 3x3 layers with channel counts in multiples of 64, "
                          "the stem and the strided blocks of the default widths are hand-written")
    return conv(x)


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = _bn(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = _bn(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), _bn(cout))

    def forward(self, x):
        if isinstance(self.conv1, FusedConv):              # inference build: conv -> one fused epilogue pass
            pair = self.conv1.with_down(x, self.down) if self.down is not None else None
            if pair is not None:                            # first convolution and downsample branch in ONE launch
                return self.conv2(pair[0], pair[1])
            skip = x if self.down is None else self.down(x)
            return self.conv2(self.conv1(x), skip.contiguous(memory_format=torch.channels_last) if skip.is_cuda else skip)
        if (self.down is None and _TRAIN_CONV_FORK and torch.is_grad_enabled() and x.requires_grad and self.conv1.bias is None
                and self.conv1.padding_mode == "zeros"
                and ops.conv3x3_train_served(x, self.conv1.weight, self.conv1.stride, self.conv1.padding, self.conv1.dilation, self.conv1.groups)):
            # the block's input feeds the first convolution AND the skip: as two outputs of one node (ops.conv3x3_fork), so that the
            # skip's gradient is added in the data-gradient kernel's epilogue (a separate add was 23 us per stride-4 block)
            y1, xs = ops.conv3x3_fork(x, self.conv1.weight)
            y = _bn_act(self.bn1, y1, True)
            return _bn_act(self.bn2, _conv(self.conv2, y), True, xs)
        if (self.down is not None and _TRAIN_CONV_S2 and torch.is_grad_enabled() and isinstance(self.conv1, nn.Conv2d)
                and isinstance(self.down[0], nn.Conv2d) and ops.conv_s2_block_served(x, self.conv1, self.down[0])):
            # the strided convolution and the downsample branch as one node on the hand-written kernels (ops.conv_s2_block): one forward launch,
            # one data-gradient launch for both branches, deterministic weight gradients
            y1, yd = ops.conv_s2_block(x, self.conv1.weight, self.down[0].weight)
            y = _bn_act(self.bn1, y1, True)
            return _bn_act(self.bn2, _conv(self.conv2, y), True, _bn_act(self.down[1], yd, False))
        y = _bn_act(self.bn1, _conv(self.conv1, x), True)
        skip = x if self.down is None else _bn_act(self.down[1], self.down[0](x), False)
        return _bn_act(self.bn2, _conv(self.conv2, y), True, skip)


class Encoder(nn.Module):
    """ResNet-18-style trunk to stride 16; returns (f4, f8, f16)."""

    def __init__(self, cin, widths):
        super().__init__()
        w4, w8, w16 = widths
        self.stem = nn.Sequential(nn.Conv2d(cin, w4, 7, 2, 3, bias=False), _bn(w4), nn.ReLU(inplace=True),
                                  nn.MaxPool2d(3, 2, 1))
        self.layer1 = nn.Sequential(BasicBlock(w4, w4, 1), BasicBlock(w4, w4, 1))
        self.layer2 = nn.Sequential(BasicBlock(w4, w8, 2), BasicBlock(w8, w8, 1))
        self.layer3 = nn.Sequential(BasicBlock(w8, w16, 2), BasicBlock(w16, w16, 1))

    def forward(self, x):
        s = self.stem
        if len(s) == 4 and isinstance(s[1], nn.BatchNorm2d):                  # training build: conv, fused BN + ReLU, pool
            if torch.is_grad_enabled() and isinstance(s[0], nn.Conv2d) and ops.stem_conv_served(x, s[0]):
                x = ops.stem_conv(x, s[0].weight)         # the hand-written stem kernel, convolution only (csrc/stem_conv_pool.hip)
            else:
                if torch.is_grad_enabled():
                    _library_fallback(x, "encoder.stem: " + _describe(s[0], x), "training stem kernel serves 7x7 / stride 2 / <= 4 input channels / 64 outputs in bf16")
                x = s[0](x)
            p, bn = s[3], s[1]
            pool_ok = isinstance(p, nn.MaxPool2d) and (p.kernel_size, p.stride, p.padding, p.dilation, p.ceil_mode) == (3, 2, 1, 1, False)
            if (_STEM_BN_POOL and pool_ok and bn.training and bn.affine and bn.weight.dtype == torch.float32 and ops.bn_relu_pool_served(x)):
                # BatchNorm + ReLU + max-pool as one op both ways: the full-resolution activation and its gradient are never written
                momentum = bn.momentum
                if bn.track_running_stats:
                    if not _BN_COUNTED_BY_MODEL[0]:
                        bn.num_batches_tracked.add_(1)
                    if momentum is None:
                        momentum = 1.0 / float(bn.num_batches_tracked)
                x = ops.bn_relu_pool(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                     bn.running_var if bn.track_running_stats else None, momentum or 0.0, bn.eps)
                f4 = self.layer1(x)
                f8 = self.layer2(f4)
                return f4, f8, self.layer3(f8)
            x = _bn_act(bn, x, True)
            if (x.is_cuda and isinstance(p, nn.MaxPool2d) and (p.kernel_size, p.stride, p.padding, p.dilation, p.ceil_mode) == (3, 2, 1, 1, False)
                    and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0 and x.dtype in (torch.bfloat16, torch.float32)):
                x = ops.maxpool3x3s2(x)                   # HIP pool + gather backward
            else:
                x = p(x)
        else:
            x = s(x)
        f4 = self.layer1(x)
        f8 = self.layer2(f4)
        return f4, f8, self.layer3(f8)


def _rows_aligned16(x: torch.Tensor) -> torch.Tensor:
    """x [T, ...] -> the same values with every x[t] starting on a 16-byte boundary (rows padded inside one buffer; x itself when its rows
    already do).  Each x[t] stays a contiguous block: one copy for all frames instead of one per frame."""
    row = x[0].numel() * x.element_size()
    if row % 16 == 0 and x.data_ptr() % 16 == 0 and x.is_contiguous():
        return x
    per = 16 // x.element_size()
    n = x[0].numel()
    padded = (n + per - 1) // per * per
    buf = torch.empty((x.shape[0], padded), dtype=x.dtype, device=x.device)
    buf[:, :n].copy_(x.reshape(x.shape[0], n))
    return buf[:, :n].unflatten(1, x.shape[1:]) if x.dim() > 2 else buf[:, :n]


class UpBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin + cskip, cout, 3, 1, 1, bias=False), _bn(cout), nn.ReLU(inplace=True),
                                  nn.Conv2d(cout, cout, 3, 1, 1, bias=False), _bn(cout), nn.ReLU(inplace=True))

    def forward(self, x, skip):
        c = self.conv
        if (isinstance(c[0], FusedConv) and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16
                and skip.dtype == torch.bfloat16 and c[0].cat_served(x.shape[1], skip.shape[1], skip.shape[-1])):
            # inference build: only the enlarged feature is written; the first convolution reads it and the skip feature where they
            # lie (no concatenated copy: 154 MB less traffic per forward at cfg2)
            u = ops.upsample_bilinear(x.contiguous(memory_format=torch.channels_last), skip.shape[2:])
            y = c[0].forward_cat(u, skip)
            for m in list(c)[1:]:
                y = m(y)
            return y
        if (x.is_cuda and x.dtype == torch.bfloat16 and skip.dtype == torch.bfloat16
                and x.shape[1] % 8 == 0 and skip.shape[1] % 8 == 0):            # one fused HIP pass (differentiable)
            x = ops.upsample_cat(x.contiguous(memory_format=torch.channels_last),
                                 skip.contiguous(memory_format=torch.channels_last))
        else:
            x = torch.cat([F.interpolate(x, size=skip.shape[-2:], mode="bilinear", align_corners=False), skip], 1)
        if len(c) == 6 and isinstance(c[1], nn.BatchNorm2d):                    # training build
            return _bn_act(c[4], _conv(c[3], _bn_act(c[1], _conv(c[0], x), True)), True)
        return c(x)


class HeadFeature(NamedTuple):
    """What Decoder.forward(head_fused=True) hands to GDKVM.segment: the stride-4 decoder feature (channels_last) and the 1x1 head."""
    feature: torch.Tensor
    weight: torch.Tensor
    bias: torch.Tensor


class Decoder(nn.Module):
    def __init__(self, cp, widths, ncls):
        super().__init__()
        w4, w8, _ = widths
        self.up8 = UpBlock(cp, w8, w8)
        self.up4 = UpBlock(w8, w4, w4)
        self.head = nn.Conv2d(w4, ncls, 1)

    def forward(self, f, f8, f4, size, head_fused=False):
        """size = (H, W): full-resolution logits; size = None: the stride-4 logits (segment() upsamples them inside
        the fused argmax kernel instead of materialising them); head_fused (with size None, inference on the GPU): the stride-4
        FEATURE and the head's fp32 weights as a HeadFeature -- segment() folds the head into the argmax kernel too."""
        y = self.up4(self.up8(f, f8), f4)
        hd = self.head
        v = 8 if y.dtype == torch.bfloat16 else 4
        g = y.shape[1] // v
        if (y.is_cuda and not torch.is_grad_enabled() and y.dtype in (torch.bfloat16, torch.float32) and hd.kernel_size == (1, 1)
                and hd.stride == (1, 1) and hd.padding == (0, 0) and hd.groups == 1 and hd.bias is not None and y.shape[1] % v == 0
                and g <= 64 and g & (g - 1) == 0 and hd.out_channels <= g and y.is_contiguous(memory_format=torch.channels_last)):
            # 1x1 convolution + bias straight into the NCHW planes the argmax / loss kernels read (ops.head_logits): one pass
            key = _wkey(self, hd.weight, hd.bias) + (y.device,)
            cache = getattr(self, "_head_w32", None)
            if cache is None or cache[0] != key:
                cache = (key, hd.weight.detach().reshape(hd.out_channels, -1).float().contiguous(), hd.bias.detach().float().contiguous())
                self._head_w32 = cache
            if head_fused and size is None:
                return HeadFeature(y, cache[1], cache[2])
            x = ops.head_logits(y, cache[1], cache[2])
        elif _TRAIN_HEAD_HIP and torch.is_grad_enabled() and ops.head_served(y, hd):
            # training: gdkvm_head_logits forward, gdkvm_head_bwd backward (one pass, fixed summation order) -- the library's bf16 weight
            # gradient for this layer accumulates atomically and differed by several bf16 ulps from run to run
            x = ops.head(y, hd.weight, hd.bias)
        else:
            _library_fallback(y, "decoder.head: " + _describe(hd, y), "head kernels serve a 1x1 convolution with bias on a channels_last bf16 / fp32 feature of 8 ... 512 channels")
            x = hd(y)
        return x if size is None else F.interpolate(x, size=size, mode="bilinear", align_corners=False)


def _fold_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """conv -> BN(eval) == one conv with W' = W * g/sqrt(var+eps), b' = beta + (b - mean) * g/sqrt(var+eps)."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                      conv.dilation, conv.groups, bias=True).to(conv.weight.device, conv.weight.dtype)
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    fused.weight.data = conv.weight * scale.reshape(-1, 1, 1, 1)
    b0 = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
    fused.bias.data = bn.bias + (b0 - bn.running_mean) * scale
    return fused


class FusedConv(nn.Module):
    """Inference-only: a convolution with its folded-BatchNorm bias, optional residual add and optional ReLU.  On the GPU
    in bf16 every 3x3 layer of the model runs as ONE hand-written kernel with the epilogue inside (forward(): kernels 4 / 5 /
    9 of gdkvm_conv_bias_act); what those do not serve (odd channel counts, fp32) is a library convolution followed by ONE
    fused HIP epilogue pass instead of PyTorch's separate bias-add / add / clamp kernels.  The bias is kept in fp32."""

    def __init__(self, conv: nn.Conv2d, relu: bool):
        super().__init__()
        self.conv = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                              conv.dilation, conv.groups, bias=False).to(conv.weight.device, conv.weight.dtype)
        self.conv.weight.data = conv.weight.data
        self.epi = _EpilogueBias(conv.bias.data.float() if conv.bias is not None else torch.zeros(conv.out_channels, device=conv.weight.device))
        self.relu = relu

    @staticmethod
    def _tile(cin: int, cout: int, stride: int, has_res: bool, width: int = 0) -> Optional[int]:
        """Which hand-written kernel of gdkvm_conv_bias_act serves a 3x3 layer (include/gdkvm.h): 4 = 64 -> 64 channels, 5 = input
        channels in multiples of 64; None = neither does (strided layers, wide rows: the general kernel 9 takes those in forward();
        odd channel counts: the framework convolution + one epilogue pass).  Any choice computes the same result: this is speed only (tools/conv_probe.py)."""
        if stride != 1 or cin % 64 or cout % 16:
            return None
        if cin == 64 and cout == 64:
            return 4
        return 5 if width <= 64 else None                   # (the chunked kernel tiles rows of at most 64 pixels)

    def _packed(self, device, igemm: bool = False):
        """The fragment-ordered copy of the weights (ops.conv3x3_pack_weights, or ops.conv_igemm_pack_weights for the general
        kernel), kept until the weights change (_wkey)."""
        key = _wkey(self, self.conv.weight) + (device, igemm)
        ent = self.__dict__.get("_wpack")
        if ent is None or ent[0] != key:
            ent = (key, (ops.conv_igemm_pack_weights if igemm else ops.conv3x3_pack_weights)(self.conv.weight))
            self.__dict__["_wpack"] = ent
        return ent[1]

    def with_down(self, x, down):
        """(self(x), down(x)) from one launch of the general kernel (ops.conv_down_bias_act) when this is a strided 3x3 layer and
        `down` the block's folded 1x1 convolution of the same stride whose bias already sits in the consumer's epilogue; else None."""
        cv = self.conv
        d = down[0] if isinstance(down, nn.Sequential) and len(down) == 1 else None
        if not (_IGEMM_STRIDED and isinstance(d, FusedConv) and getattr(d, "bias_folded_downstream", False) and not d.relu
                and not getattr(self, "bias_folded_downstream", False) and x.is_cuda and x.dtype == torch.bfloat16
                and cv.kernel_size == (3, 3) and cv.padding == (1, 1) and cv.dilation == (1, 1) and cv.groups == 1
                and cv.stride[0] == cv.stride[1] and d.conv.kernel_size == (1, 1) and d.conv.stride == cv.stride
                and d.conv.padding == (0, 0) and d.conv.groups == 1 and d.conv.in_channels == cv.in_channels
                and d.conv.out_channels == cv.out_channels and cv.in_channels % 32 == 0 and cv.out_channels % 128 == 0
                and cv.weight.dtype == torch.bfloat16 and d.conv.weight.dtype == torch.bfloat16
                and cv.weight.is_contiguous(memory_format=torch.channels_last)
                and self._tile(cv.in_channels, cv.out_channels, cv.stride[0], False, x.shape[-1]) is None):
            return None
        dw = d.conv.weight if d.conv.weight.is_contiguous(memory_format=torch.channels_last) else d.conv.weight.contiguous(memory_format=torch.channels_last)
        return ops.conv_down_bias_act(x.contiguous(memory_format=torch.channels_last), cv.weight, self.epi.bias, self._packed(x.device, igemm=True),
                                      dw, d._packed(x.device, igemm=True), None, cv.stride[0], self.relu)

    def cat_served(self, c1: int, c2: int, width: int) -> bool:
        """Can forward_cat run this layer on [x1 (c1 channels) ; x2 (c2)] without the concatenated tensor (gdkvm_conv_cat_bias_act)?"""
        cv = self.conv
        return (not getattr(self, "bias_folded_downstream", False) and cv.kernel_size == (3, 3) and cv.groups == 1
                and cv.dilation == (1, 1) and cv.stride == (1, 1) and cv.padding == (1, 1) and cv.weight.is_cuda
                and cv.weight.dtype == torch.bfloat16 and cv.weight.is_contiguous(memory_format=torch.channels_last)
                and c1 % 64 == 0 and c2 % 64 == 0 and c1 + c2 == cv.in_channels and cv.out_channels % 16 == 0 and width <= 64)

    def forward_cat(self, x1, x2):
        """forward(torch.cat([x1, x2], 1)) with every 64-channel chunk read from the tensor it lies in."""
        cl = torch.channels_last
        return ops.conv_cat_bias_act(x1.contiguous(memory_format=cl), x2.contiguous(memory_format=cl), self.conv.weight, self.epi.bias,
                                     None, self.relu, 0, self._packed(x1.device))

    def forward(self, x, residual=None):
        cv = self.conv
        folded = getattr(self, "bias_folded_downstream", False)
        if (not folded and x.is_cuda and x.dtype == torch.bfloat16 and cv.kernel_size == (3, 3) and cv.groups == 1
                and cv.dilation == (1, 1) and cv.stride[0] == cv.stride[1] and cv.padding == (1, 1) and cv.weight.dtype == torch.bfloat16
                and cv.in_channels % 8 == 0 and cv.out_channels % 8 == 0 and cv.weight.is_contiguous(memory_format=torch.channels_last)):
            tile = self._tile(cv.in_channels, cv.out_channels, cv.stride[0], residual is not None, x.shape[-1])
            if tile is not None:                            # bias (+ residual) (+ ReLU) inside the implicit-GEMM kernel: one launch
                packed = self._packed(x.device) if tile in (4, 5) else None
                return ops.conv_bias_act(x.contiguous(memory_format=torch.channels_last), cv.weight, self.epi.bias, residual,
                                         cv.stride[0], 1, self.relu, tile, packed)
            if _IGEMM_STRIDED and cv.in_channels % 32 == 0 and cv.out_channels % 128 == 0:
                # what the two 3x3 / 1 / 1 kernels do not take -- the strided layers, rows wider than 64 pixels -- on the general
                # implicit-GEMM kernel, epilogue included (A/B switch: see _IGEMM_STRIDED)
                return ops.conv_bias_act(x.contiguous(memory_format=torch.channels_last), cv.weight, self.epi.bias, residual,
                                         cv.stride[0], 1, self.relu, ops.CONV_KERNEL_IGEMM, self._packed(x.device, igemm=True))
        _library_fallback(x, _describe(cv, x), "inference build: the hand-written kernels take bf16 channels_last 3x3 / pad 1 layers with channel counts "
                          "in multiples of 8 (strided: 32 in, 128 out) and the 1x1 branch fused into a strided block")
        y = self.conv(x)
        if folded:                                          # the bias was added to the consumer's epilogue bias: nothing to do here
            return y
        if not y.is_cuda:                                  # CPU reference module (oracle/model_ref.py): plain torch
            y = y + self.epi.bias.to(y.dtype).reshape(1, -1, 1, 1)
            y = y if residual is None else y + residual
            return F.relu(y) if self.relu else y
        y = y.contiguous(memory_format=torch.channels_last)
        return ops.bias_act_(y, self.epi.bias, residual, self.relu)


class FusedConvPool(FusedConv):
    """The stem: conv -> bias + ReLU + 3x3/2 max-pool as ONE HIP pass over the conv output (the full-resolution activation is
    read once and never written back).  A 7x7 / stride 2 / pad 3 stem additionally runs in space-to-depth form on the GPU:
    the same arithmetic as a 4x4 / stride 1 convolution on the (c, row parity, column parity) channels of the half-resolution
    image, which MIOpen executes 2.2x faster than the 3-channel 7x7 kernel (implicit-GEMM K = 147 and a zero-fill pass)."""

    S2D_CH = 16

    def enable_s2d(self):
        cv = self.conv
        if (cv.kernel_size, cv.stride, cv.padding, cv.dilation, cv.groups) != ((7, 7), (2, 2), (3, 3), (1, 1), 1) \
                or 4 * cv.in_channels > self.S2D_CH:
            return self
        w7 = cv.weight.data
        w4 = torch.zeros(cv.out_channels, self.S2D_CH, 4, 4, dtype=w7.dtype, device=w7.device)
        for a in range(-2, 2):                          # tap u of the 7x7 kernel = 2a + p + 3 on the half-resolution grid
            for p_ in range(2):
                u = 2 * a + p_ + 3
                if not 0 <= u <= 6:
                    continue
                for b in range(-2, 2):
                    for q_ in range(2):
                        v = 2 * b + q_ + 3
                        if 0 <= v <= 6:
                            w4[:, p_ * 2 + q_:4 * cv.in_channels:4, a + 2, b + 2] = w7[:, :, u, v]
        self.register_buffer("w_s2d", w4.contiguous(memory_format=torch.channels_last))
        return self

    def forward(self, x, residual=None):
        if x.is_cuda and getattr(self, "w_s2d", None) is not None and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            if (_STEM_NCHW and x.dtype == torch.bfloat16 and x.shape[1] <= 4 and self.w_s2d.dtype == torch.bfloat16
                    and tuple(self.w_s2d.shape) == (64, 16, 4, 4) and self.w_s2d.is_contiguous(memory_format=torch.channels_last)):
                # the stem kernel builds its space-to-depth band from the NCHW frames itself: one kernel, no 16-channel copy of the input
                return ops.stem_conv_pool_nchw(x.contiguous(), self.w_s2d, self.epi.bias)
            xs = ops.stem_s2d(x.contiguous(), self.S2D_CH)                       # NCHW frames -> NHWC space-to-depth, one pass
            if (xs.dtype == torch.bfloat16 and self.w_s2d.dtype == torch.bfloat16 and tuple(self.w_s2d.shape) == (64, 16, 4, 4)
                    and self.w_s2d.is_contiguous(memory_format=torch.channels_last)):
                return ops.stem_conv_pool(xs, self.w_s2d, self.epi.bias)         # convolution + bias + ReLU + max-pool: one kernel
            y = F.conv2d(xs, self.w_s2d, None, 1, 2)                             # [N, Cout, H/2 + 1, W/2 + 1]: last row/col unused
            return ops.bias_relu_maxpool(y.contiguous(memory_format=torch.channels_last)[:, :, :x.shape[2] // 2, :x.shape[3] // 2],
                                         self.epi.bias)
        _library_fallback(x, "encoder.stem: " + _describe(self.conv, x), "inference stem kernel serves 7x7 / stride 2 / pad 3, <= 4 input channels, 64 outputs, bf16, even sizes")
        y = self.conv(x)
        if not y.is_cuda:
            return F.max_pool2d(F.relu(y + self.epi.bias.to(y.dtype).reshape(1, -1, 1, 1)), 3, 2, 1)
        return ops.bias_relu_maxpool(y.contiguous(memory_format=torch.channels_last), self.epi.bias)


class _EpilogueBias(nn.Module):
    """fp32 bias that survives .to(bfloat16) on the parent."""

    def __init__(self, bias):
        super().__init__()
        self.bias = nn.Parameter(bias, requires_grad=False)

    def _apply(self, fn, recurse=True):
        def keep_fp32(t):
            r = fn(t)
            return r.float() if r.is_floating_point() else r
        return super()._apply(keep_fp32, recurse)


def _fold_sequential(seq: nn.Sequential) -> nn.Sequential:
    mods, out, i = list(seq), [], 0
    while i < len(mods):
        if isinstance(mods[i], nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
            out.append(_fold_bn(mods[i], mods[i + 1])); i += 2
        else:
            out.append(mods[i]); i += 1
    return nn.Sequential(*out)


class KPFFParams(nn.Module):
    """Weights of Key-Pixel Feature Fusion; always fp32 (the kernel reads fp32 weights)."""

    def _apply(self, fn, recurse=True):
        # .to(bfloat16) / .half() on the parent must not narrow these: device moves only
        def keep_fp32(t):
            r = fn(t)
            return r.float() if r.is_floating_point() else r
        return super()._apply(keep_fp32, recurse)

    def __init__(self, ck, cv, cp):
        super().__init__()
        cin = cp + ck + cv
        self.wa = nn.Parameter(torch.empty(2 * cp, cin))
        self.ba = nn.Parameter(torch.zeros(2 * cp))
        self.wl = nn.Parameter(torch.empty(cp, ck))
        self.wg = nn.Parameter(torch.empty(cp, cv))
        for p in (self.wa, self.wl, self.wg):
            nn.init.kaiming_uniform_(p, a=5 ** 0.5)


class GDKVM(nn.Module):
    def __init__(self, cfg: Optional[GDKVMConfig] = None):
        super().__init__()
        self.cfg = cfg = cfg or GDKVMConfig()
        if cfg.rule not in _RULES:
            raise ValueError(f"unknown rule {cfg.rule!r}")
        Hh, Dk, Dv, Cp = cfg.heads, cfg.key_dim, cfg.value_dim, cfg.pixel_dim
        if cfg.widths[2] != Cp:
            raise ValueError("pixel_dim must equal the stride-16 encoder width")
        if cfg.normalizer and (cfg.mask_feedback or cfg.scan_segments != 1):
            raise ValueError("normalizer=True runs on the serial scan only (no mask_feedback, scan_segments=1)")
        if cfg.normalizer and not 0.0 < cfg.normalizer_eps < 1.0:
            raise ValueError("normalizer_eps must lie in (0, 1)")
        self.encoder = Encoder(cfg.in_channels, cfg.widths)
        self.key_proj = nn.Conv2d(Cp, Hh * Dk, 1)
        self.query_proj = nn.Conv2d(Cp, Hh * Dk, 1)
        self.value_proj = nn.Conv2d(Cp, Hh * Dv, 1)
        self.mask_embed = nn.Conv2d(1, Hh * Dv, 1, bias=False)          # first-frame mask -> value (optional input)
        self.gate_proj = nn.Conv2d(Cp, Hh, 1)                            # beta logits, per token and head
        self.decay_proj = nn.Linear(Cp, Hh)                              # alpha logits, per frame and head
        nn.init.constant_(self.decay_proj.bias, 2.0)                     # sigmoid(2) ~ 0.88: remember by default
        self.kpff = KPFFParams(Hh * Dk, Hh * Dv, Cp)
        self.decoder = Decoder(Cp, cfg.widths, cfg.num_classes)
        # this model's weights epoch (weights_changed(model)), shared with every sub-module that caches a weight pack
        self.__dict__["_epoch_cell"] = [0]
        self._share_epoch_cell()

    def _share_epoch_cell(self):
        cell = self.__dict__["_epoch_cell"]
        for m in self.modules():
            if isinstance(m, (Decoder, FusedConv)):
                m.__dict__["_epoch_cell"] = cell

    # ------------------------------------------------------------------ packed-weight caches of the inference build
    def invalidate_packed_weights(self):
        """Drop the weight packs the inference forward keeps between calls (K/Q/V fragment pack, fp32 gate and head weights,
        KPFF bf16 pack, every FusedConv's fragment-ordered weight copy).  They are keyed on every source tensor's version counter and address (`_wkey`), which catches
        optimiser steps, ``load_state_dict`` and re-binding; an in-place write through ``.data`` changes neither, so code that
        does one must call this.  Called by load_state_dict(), _apply() (.to / .cuda / .half ...), train() and
        fuse_for_inference()."""
        for name in ("_qkv_pack", "_gate_w32", "_kpff_pack", "_clip_graphs", "_mask_w32"):
            self.__dict__.pop(name, None)
        # graphs captured over this module (GraphedSegment, GraphedTrainStep) hold the packs they read and compare this counter: a replay
        # after the packs were dropped raises instead of convolving with the weights of capture time
        self.__dict__["_pack_epoch"] = self.__dict__.get("_pack_epoch", 0) + 1
        ops.drop_train_packs(self.__dict__.pop("_train_pack_weights", None) or ())     # (the training step's per-weight packs, ops.conv3x3_train_packs)
        if "_modules" not in self.__dict__:
            return
        dec = self._modules.get("decoder")
        if dec is not None:
            dec.__dict__.pop("_head_w32", None)
        for m in self.modules():                            # the fragment-ordered weight copies of the fused convolutions
            if isinstance(m, FusedConv):
                m.__dict__.pop("_wpack", None)

    def _apply(self, fn, recurse=True):
        self.invalidate_packed_weights()
        return super()._apply(fn, recurse)

    def train(self, mode: bool = True):
        if mode != self.training:                           # (a defensive model.eval() on an eval-mode model changes nothing: graphs captured over it stay valid)
            self.invalidate_packed_weights()
        return super().train(mode)

    # ------------------------------------------------------------------ memory path (HIP; overridable hooks)
    def _memory_scan(self, q, k, v, alpha_logit, beta_logit, state, norms=None, readout=True):
        """q,k [B,T,N,Hh,Dk] v [B,T,N,Hh,Dv] alpha [B,T,Hh] beta [B,T,N,Hh] -> (R [B,T,N,Hh,Dv], S_T).  norms (inference): the
        inverse key / query norms ops.proj_gates produced with the projections.  readout=False (the write half of a per-frame step):
        R is not produced (None)."""
        flags = ops.FLAG_NORMALIZE_QK | ops.FLAG_GATE_LOGITS
        if self.cfg.normalizer:
            if torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v, alpha_logit, beta_logit)):
                raise NotImplementedError("GDKVMConfig(normalizer=True) is an inference flag: no backward through the normalised read-out")
            Dv = v.shape[-1]                                  # module state = [S | z]: [B,Hh,Dk,Dv+1]
            s0 = None if state is None else state[..., :Dv].contiguous()
            z0 = None if state is None else state[..., Dv].contiguous()
            r, s_out, z_out = ops.scan_fwd_normalizer(q, k, v, alpha_logit, beta_logit, s0, z0, _RULES[self.cfg.rule], flags, self.cfg.normalizer_eps)
            return r, torch.cat([s_out, z_out.unsqueeze(-1)], -1)
        if torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v, alpha_logit, beta_logit)):
            return ops.scan(q, k, v, alpha_logit, beta_logit, state, _RULES[self.cfg.rule], flags)   # saves history
        seg = self.cfg.scan_segments
        if seg != 1 and q.is_cuda and q.shape[1] > 1:
            # time segments of one long call run concurrently (SURVEY §8f n3 inside one GPU); segments = 0 lets the library choose and
            # falls back to the serial scan where segmenting cannot pay
            if seg > 1 and q.shape[1] % seg:
                raise ValueError(f"scan_segments={seg} must divide the call's {q.shape[1]} frames")
            return ops.scan_fwd_segmented(q, k, v, alpha_logit, beta_logit, state, segments=seg, rule=_RULES[self.cfg.rule], flags=flags)
        return ops.scan_fwd(q, k, v, alpha_logit, beta_logit, state, rule=_RULES[self.cfg.rule], flags=flags, norms=norms, readout=readout)

    def _mask_from_lowres(self, lowres, H, W, target=None, mask_out=None, counts_out=None):
        """stride-4 logits [F,ncls,hl,wl] -> (mask uint8 [F,H,W], Dice counts | None): bilinear x4 + argmax (ties -> lowest class)."""
        return ops.upsample_argmax_dice(lowres.contiguous(), H, W, target, mask_out, counts_out)

    def _memory_read(self, q, state, norms=None):
        """The read half of a per-frame step (mask_feedback): q [B,N,Hh,Dk], state [B,Hh,Dk,Dv] fp32 -> R [B,N,Hh,Dv] = Qn S."""
        return ops.lkva_read(q, state, ops.FLAG_NORMALIZE_QK, norms)

    def _embed_mask_(self, v, mask, h, w):
        """v [B,N,Hh*Dv] += mask_embed(adaptive_avg_pool(mask != 0)) in place; mask uint8 [B,H,W] (a predicted mask)."""
        key = _wkey(self, self.mask_embed.weight) + (v.device,)
        cache = self.__dict__.get("_mask_w32")
        if cache is None or cache[0] != key:
            cache = (key, self.mask_embed.weight.detach().reshape(-1).float().contiguous())
            self.__dict__["_mask_w32"] = cache
        return ops.mask_embed_add_(v, mask, cache[1], h, w)

    def _fuse(self, local, glob, pixel, h, w):
        p = self.kpff
        if torch.is_grad_enabled() and (pixel.requires_grad or p.wa.requires_grad):
            return ops.kpff(local, glob, pixel, p.wa, p.ba, p.wl, p.wg, h, w)
        # inference: keep the bf16 weight pack of the previous call while the weight tensors are unchanged
        key = _wkey(self, p.wa, p.ba, p.wl, p.wg) + (local.dtype, local.device)
        cache = getattr(self, "_kpff_pack", None)
        hit = cache is not None and cache[0] == key
        ws = cache[1] if hit else torch.empty(ops.load().gdkvm_kpff_workspace_bytes(local.shape[-1], glob.shape[-1], pixel.shape[-1],
                                                                             ops._io_dtype(local)), dtype=torch.uint8, device=local.device)
        out = ops.kpff_fwd(local, glob, pixel, p.wa, p.ba, p.wl, p.wg, h, w, workspace=ws, packed=hit)
        self._kpff_pack = (key, ws)
        return out

    # ------------------------------------------------------------------------------------------ forward
    @staticmethod
    def _tokens(x):
        """[BT,C,h,w] conv output -> [BT, h*w, C] token-major view (free when x is channels_last)."""
        return x.permute(0, 2, 3, 1).reshape(x.shape[0], x.shape[2] * x.shape[3], x.shape[1]).contiguous()

    def forward(self, frames: torch.Tensor, mask0: Optional[torch.Tensor] = None,
                state: Optional[torch.Tensor] = None, return_state: bool = False, _lowres: bool = False, _head_fused: bool = False):
        if frames.dim() != 5:
            raise ValueError("frames must be [B,T,C,H,W]")
        cfg = self.cfg
        B, T, C, H, W = frames.shape
        Hh, Dk, Dv = cfg.heads, cfg.key_dim, cfg.value_dim
        if cfg.mask_feedback:
            logits, _, _, s_out = self._forward_feedback(frames, mask0, state, lowres=_lowres)
            return (logits, s_out) if return_state else logits
        x = frames.reshape(B * T, C, H, W)
        dt = self.key_proj.weight.dtype
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype(x.device.type) if x.is_cuda else x.dtype
        stem0 = self.encoder.stem[0]
        if x.is_cuda and ((isinstance(stem0, FusedConvPool) and getattr(stem0, "w_s2d", None) is not None)
                          or (torch.is_grad_enabled() and isinstance(stem0, nn.Conv2d) and dt == torch.bfloat16 and ops.stem_conv_served(x, stem0))):
            x = x.to(dt)                                                         # (the stem kernels read NCHW frames themselves)
        else:
            x = x.to(dtype=dt, memory_format=torch.channels_last)                # cast + NHWC in one pass
        if self.training and x.is_cuda and torch.is_grad_enabled() and dt == torch.bfloat16:
            # the forward and data-gradient packs of every stride-1 3x3 layer's weights, ONE launch for the step (ops.conv3x3 finds them;
            # per layer it was a cast, a pack and, in the backward, a second pack)
            ws = getattr(self, "_train_pack_weights", None)
            if ws is None:
                ws = self._train_pack_weights = [m.weight for m in self.modules() if isinstance(m, nn.Conv2d) and m.bias is None
                                                 and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.groups == 1
                                                 and m.weight.shape[0] % 64 == 0 and m.weight.shape[1] % 64 == 0]
            # (EVERY training forward re-packs: nothing a cache could key on records a fused optimiser's step -- ops.conv3x3_train_packs)
            ops.conv3x3_train_packs([w for w in ws if w.is_cuda and w.dtype == torch.float32])
        counted = self.training and x.is_cuda
        if counted:                                                              # BatchNorm step counters: one launch, not 19
            nbt = [m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.track_running_stats]
            if nbt:
                torch._foreach_add_(nbt, 1)
        _BN_COUNTED_BY_MODEL[0] = counted
        try:
            f4, f8, f16 = self.encoder(x)
            return self._after_encoder(f4, f8, f16, mask0, state, return_state, _lowres, (B, T, H, W), _head_fused)
        finally:
            _BN_COUNTED_BY_MODEL[0] = False
            ops.end_train_packs()

    def _forward_feedback(self, frames, mask0=None, state=None, lowres=False, target=None, masks_only=False, mask_out=None, counts_out=None):
        """The per-frame step mode (cfg.mask_feedback).  Encoder and projections run once for all frames (they do
        not depend on the state), then frame by frame, every clip's frame t together:
            R_t = Qn_t S_{t-1}                         (_memory_read: gdkvm_lkva_read)
            F_t = KPFF(K_t, R_t, P_t); decoder; mask_t = argmax(upsample(logits_t))
            v_t = value_proj(f_t) + mask_embed(pool(mask_t != 0))      (mask0 instead of mask_0 when given)
            S_t = GDR(S_{t-1}, K_t, v_t)               (_memory_scan with T = 1, no read-out)
        The batch is laid out TIME-MAJOR ([T*B, ...]: one transposed copy of the frames) so that every per-frame operand is a contiguous
        block.  Returns (logits [B,T,ncls,.,.] | None when masks_only, mask uint8 [B,T,H,W], Dice counts [B,T,ncls,3] | None, S_T)."""
        cfg = self.cfg
        if torch.is_grad_enabled() and self.training:
            raise NotImplementedError("GDKVMConfig(mask_feedback=True) is an inference mode (the predicted mask is an argmax): eval() / torch.no_grad()")
        B, T, C, H, W = frames.shape
        Hh, Dk, Dv = cfg.heads, cfg.key_dim, cfg.value_dim
        with torch.no_grad():
            x = frames.transpose(0, 1).reshape(T * B, C, H, W)                   # time-major copy
            dt = self.key_proj.weight.dtype
            stem0 = self.encoder.stem[0]
            if x.is_cuda and isinstance(stem0, FusedConvPool) and getattr(stem0, "w_s2d", None) is not None:
                x = x.to(dt)
            else:
                x = x.to(dtype=dt, memory_format=torch.channels_last)
            f4, f8, f16 = self.encoder(x)
            h, w = f16.shape[-2:]
            N = h * w
            p_tok, k_tok, q, v, alpha, beta, norms = self._project(f16, T, B, None)      # q [T,B,N,Hh,Dk], alpha [T,B,Hh], ...
            v = v.reshape(T, B, N, Hh * Dv)
            k5 = k_tok.reshape(T, B, N, Hh, Dk)
            nrm = None if norms is None else norms.reshape(T, B * N, Hh, 2)
            if frames.is_cuda:                   # (the kernels want 16-byte aligned operands: frame t's gate / norm rows must start on one)
                alpha, beta = _rows_aligned16(alpha), _rows_aligned16(beta)
                nrm = None if nrm is None else _rows_aligned16(nrm)
            S = torch.zeros((B, Hh, Dk, Dv), dtype=torch.float32, device=frames.device) if state is None else state.to(torch.float32)
            tgt = None if target is None else target.transpose(0, 1).contiguous()          # [T,B,H,W]
            mask_tm = torch.empty((T, B, H, W), dtype=torch.uint8, device=frames.device)
            counts_tm = None if target is None else torch.empty((T, B, cfg.num_classes, 3), dtype=torch.int32, device=frames.device)
            lows = []
            for t in range(T):
                sl = slice(t * B, (t + 1) * B)
                r_t = self._memory_read(q[t], S, None if nrm is None else nrm[t])
                fused = self._fuse(k_tok[sl], r_t.reshape(B, N, Hh * Dv), p_tok[sl], h, w)
                fmap = fused.reshape(B, h, w, -1).permute(0, 3, 1, 2)
                out = self.decoder(fmap, f8[sl], f4[sl], None, head_fused=masks_only)
                tg_t = None if tgt is None else tgt[t]
                co_t = None if counts_tm is None else counts_tm[t]
                # (the mask kernel wants 16-byte aligned outputs: frame t's slab of the time-major result is, for the usual shapes)
                slab_ok = (B * H * W) % 16 == 0 and (co_t is None or (B * cfg.num_classes * 12) % 16 == 0)
                if isinstance(out, HeadFeature) and slab_ok:   # fused inference build: head + upsample + argmax (+ Dice) in one kernel
                    ops.head_upsample_argmax_dice(out.feature, out.weight, out.bias, H, W, tg_t, mask_tm[t], co_t)
                elif isinstance(out, HeadFeature):
                    m_t, c_t = ops.head_upsample_argmax_dice(out.feature, out.weight, out.bias, H, W, tg_t)
                    mask_tm[t].copy_(m_t)
                    if co_t is not None:
                        co_t.copy_(c_t)
                else:
                    if not masks_only:
                        lows.append(out)
                    m_t, c_t = self._mask_from_lowres(out, H, W, tg_t)
                    mask_tm[t].copy_(m_t)
                    if co_t is not None:
                        co_t.copy_(c_t)
                v_t = v[t]
                if t == 0 and mask0 is not None:
                    m0 = F.adaptive_avg_pool2d(mask0.to(v_t.dtype), (h, w))
                    v_t = v_t + self._tokens(self.mask_embed(m0))
                else:
                    v_t = self._embed_mask_(v_t, mask_tm[t], h, w)
                kw = {} if nrm is None else {"norms": nrm[t]}
                _, S = self._memory_scan(q[t].unsqueeze(1), k5[t].unsqueeze(1), v_t.reshape(B, 1, N, Hh, Dv), alpha[t].unsqueeze(1),
                                         beta[t].unsqueeze(1), S, readout=False, **kw)
            mask = mask_tm.transpose(0, 1)
            if mask_out is not None:
                mask_out.view(B, T, H, W).copy_(mask)
                mask = mask_out.view(B, T, H, W)
            else:
                mask = mask.contiguous()
            counts = None
            if counts_tm is not None:
                counts = counts_tm.transpose(0, 1)
                if counts_out is not None:
                    counts_out.view(B, T, cfg.num_classes, 3).copy_(counts)
                    counts = counts_out.view(B, T, cfg.num_classes, 3)
                else:
                    counts = counts.contiguous()
            logits = None
            if not masks_only:
                low = torch.stack(lows, 1)                                        # [B,T,ncls,hl,wl]
                logits = low if lowres else F.interpolate(low.reshape(B * T, *low.shape[2:]), size=(H, W), mode="bilinear",
                                                           align_corners=False).reshape(B, T, -1, H, W)
            return logits, mask, counts, S

    def _after_encoder(self, f4, f8, f16, mask0, state, return_state, _lowres, dims, _head_fused=False):
        cfg = self.cfg
        B, T, H, W = dims
        Hh, Dk, Dv = cfg.heads, cfg.key_dim, cfg.value_dim
        h, w = f16.shape[-2:]
        N = h * w
        return self._after_projection(f4, f8, f16, self._project(f16, B, T, mask0), state, return_state, _lowres, dims, _head_fused)

    def _after_projection(self, f4, f8, f16, projected, state, return_state, _lowres, dims, _head_fused=False):
        """Everything that depends on the memory state: the scan (read + write), KPFF, the decoder.  Split from _after_encoder so that a long
        clip's NEXT chunk can run its encoder and projections beside this (PipelinedClip)."""
        cfg = self.cfg
        B, T, H, W = dims
        Hh, Dk, Dv = cfg.heads, cfg.key_dim, cfg.value_dim
        h, w = f16.shape[-2:]
        N = h * w
        p_tok, k_tok, q, v, alpha, beta, norms = projected
        if norms is not None and not cfg.normalizer:
            r, s_out = self._memory_scan(q, k_tok.reshape(B, T, N, Hh, Dk), v, alpha, beta, state, norms=norms)
        else:
            r, s_out = self._memory_scan(q, k_tok.reshape(B, T, N, Hh, Dk), v, alpha, beta, state)
        fused = self._fuse(k_tok, r.reshape(B * T, N, Hh * Dv), p_tok, h, w)       # [BT,N,Cp]
        fmap = fused.reshape(B * T, h, w, -1).permute(0, 3, 1, 2)                  # channels_last view, no copy
        logits = self.decoder(fmap, f8, f4, None if _lowres else (H, W), head_fused=_head_fused and _lowres)
        if isinstance(logits, HeadFeature):
            return (logits, s_out) if return_state else logits
        logits = logits.reshape(B, T, cfg.num_classes, *logits.shape[-2:])
        return (logits, s_out) if return_state else logits

    def _project(self, f16, B, T, mask0=None):
        """Everything derived from the stride-16 feature per token: (p_tok [BT,N,Cp], k_tok [BT,N,Hh*Dk], q [B,T,N,Hh,Dk], v [B,T,N,Hh,Dv],
        alpha logits [B,T,Hh], beta logits [B,T,N,Hh], inverse key / query norms | None).  (B, T) only shape the views: the per-frame
        step mode passes (T, B) for its time-major batch."""
        cfg = self.cfg
        Hh, Dk, Dv = cfg.heads, cfg.key_dim, cfg.value_dim
        h, w = f16.shape[-2:]
        N = h * w
        # The four 1x1 projections are token-major GEMMs on the (free) [BT*N, Cp] view of the channels_last feature, with
        # the bias in the GEMM epilogue: their outputs ARE the contiguous [.., N, C] rows the memory path reads -- no
        # conv -> bias-add -> permute-copy chain per projection (done at the framework level).
        p_tok = self._tokens(f16)                                                # [BT,N,Cp] view, no copy
        tok2d = p_tok.reshape(B * T * N, -1)

        train_gpu = tok2d.is_cuda and torch.is_grad_enabled() and tok2d.requires_grad

        def proj(conv):
            w2 = conv.weight.reshape(conv.out_channels, -1)
            if train_gpu and conv.out_channels >= 8:         # weight gradient over the B*T*N token axis: split-K (ops.wgrad)
                return ops.token_linear(tok2d, w2, conv.bias)
            if train_gpu and tok2d.dtype == torch.bfloat16:
                # a projection with fewer than 8 outputs (the write gate: one per head) as the same product on 16 zero-padded rows: as a
                # library GEMM with N = 1 its weight gradient -- a weighted column sum over 25 088 tokens -- took 66 us of a training step
                pad = 16 - conv.out_channels
                wp = F.pad(w2, (0, 0, 0, pad))
                bp = None if conv.bias is None else F.pad(conv.bias, (0, pad))
                return ops.token_linear(tok2d, wp, bp)[:, :conv.out_channels]
            return F.linear(tok2d, w2, conv.bias)

        wk, wq, wv = Hh * Dk, Hh * Dk, Hh * Dv
        norms = beta = alpha = beta_stacked = alpha_stacked = None
        cp8 = tok2d.shape[1] // 8
        infer_bf16 = (tok2d.is_cuda and not train_gpu and not torch.is_grad_enabled() and tok2d.dtype == torch.bfloat16
                      and tok2d.shape[1] % 32 == 0 and tok2d.shape[1] <= 512 and wk % 16 == 0 and wv % 16 == 0)
        if infer_bf16 and _PROJ_GATES and Dk == ops.KERNEL_DK and cp8 & (cp8 - 1) == 0:
            # ONE launch for everything derived from the pixel feature: K / Q / V, both gate logits, the key / query norms
            projs = (self.key_proj, self.query_proj, self.value_proj)
            gp, dp = self.gate_proj, self.decay_proj
            key = _wkey(self, *(t for c in projs for t in (c.weight, c.bias)), gp.weight, gp.bias, dp.weight, dp.bias) + (tok2d.device,)
            cache = getattr(self, "_qkv_pack", None)
            if cache is None or cache[0] != key:
                w_all = torch.cat([c.weight.detach().reshape(c.out_channels, -1).float() for c in projs], 0)
                b_all = torch.cat([c.bias.detach().float() for c in projs], 0).contiguous()
                gw = tuple(t.detach().float().contiguous() for t in (gp.weight.reshape(Hh, -1), gp.bias, dp.weight, dp.bias))
                cache = (key, ops.pack_rows_weight(w_all), b_all, gw)
                self._qkv_pack = cache
            (k2d, q2d, v2d), (beta, alpha), norms = ops.proj_gates(p_tok, cache[1], cache[2], *cache[3], Hh, Dk, Dv)
            beta, alpha = beta.reshape(B, T, N, Hh), alpha.reshape(B, T, Hh)
            k_tok, q, v = k2d.reshape(B * T, N, wk), q2d.reshape(B, T, N, Hh, Dk), v2d.reshape(B, T, N, wv)
        elif infer_bf16:
            # the three projections in ONE pass over the tokens (ops.proj_rows: token tile in LDS, weights streamed in MFMA
            # fragment order); the packed weight is rebuilt only when a projection's parameters change
            projs = (self.key_proj, self.query_proj, self.value_proj)
            key = _wkey(self, *(t for c in projs for t in (c.weight, c.bias))) + (tok2d.device,)
            cache = getattr(self, "_qkv_pack", None)
            if cache is None or cache[0] != key:
                w_all = torch.cat([c.weight.detach().reshape(c.out_channels, -1).float() for c in projs], 0)
                b_all = torch.cat([c.bias.detach().float() for c in projs], 0).contiguous()
                cache = (key, ops.pack_rows_weight(w_all), b_all)
                self._qkv_pack = cache
            k2d, q2d, v2d = ops.proj_rows(tok2d, cache[1], cache[2], (wk, wq, wv))
            k_tok, q, v = k2d.reshape(B * T, N, wk), q2d.reshape(B, T, N, Hh, Dk), v2d.reshape(B, T, N, wv)
        elif train_gpu and tok2d.dtype == torch.bfloat16 and _TRAIN_PROJ_STACKED:
            # training: the four projections of the feature as ONE stacked product forward and two backward (ops.token_projections)
            # -- and the decay logit with them: W_d mean_n(x) + b_d = mean_n(W_d x + b_d), so the per-frame decay is the token mean of one
            # more stacked column (the mean taken in fp32 over bf16 per-token values; the separate path rounded the mean itself to bf16)
            k2d, q2d, v2d, b2d, a2d = ops.token_projections(tok2d, (self.key_proj, self.query_proj, self.value_proj, self.gate_proj, self.decay_proj))
            k_tok, q, v = k2d.reshape(B * T, N, wk), q2d.reshape(B, T, N, Hh, Dk), v2d.reshape(B, T, N, wv)
            beta_stacked = b2d.float().reshape(B, T, N, Hh)
            alpha_stacked = a2d.float().reshape(B * T, N, Hh).mean(1).reshape(B, T, Hh)
        else:
            k_tok = proj(self.key_proj).reshape(B * T, N, Hh * Dk)               # local key feature
            q = proj(self.query_proj).reshape(B, T, N, Hh, Dk)
            v = proj(self.value_proj).reshape(B, T, N, Hh * Dv)
        if mask0 is not None:
            m = F.adaptive_avg_pool2d(mask0.to(v.dtype), (h, w))
            me = self._tokens(self.mask_embed(m))                                # [B,N,Hh*Dv]
            v = torch.cat([v[:, :1] + me.unsqueeze(1), v[:, 1:]], 1)
        keep_mask_embed = mask0 is None and torch.is_grad_enabled() and self.mask_embed.weight.requires_grad
        v = v.reshape(B, T, N, Hh, Dv)
        v8 = 8 if p_tok.dtype == torch.bfloat16 else 4
        g_lanes = p_tok.shape[-1] // v8
        if beta is not None:
            pass                                                                 # (came with the projections)
        elif (p_tok.is_cuda and not train_gpu and not torch.is_grad_enabled() and p_tok.dtype in (torch.bfloat16, torch.float32)
                and p_tok.shape[-1] % v8 == 0 and g_lanes <= 64 and g_lanes & (g_lanes - 1) == 0):
            # both gate logits in one pass over the feature (token mean + two N = 1 projections + casts as framework ops: 6 launches)
            gp, dp = self.gate_proj, self.decay_proj
            key = _wkey(self, gp.weight, gp.bias, dp.weight, dp.bias) + (p_tok.device,)
            cache = getattr(self, "_gate_w32", None)
            if cache is None or cache[0] != key:
                cache = (key, tuple(t.detach().float().contiguous() for t in (gp.weight.reshape(Hh, -1), gp.bias, dp.weight, dp.bias)))
                self._gate_w32 = cache
            beta, alpha = ops.gate_logits(p_tok, *cache[1])
            beta, alpha = beta.reshape(B, T, N, Hh), alpha.reshape(B, T, Hh)
        else:
            beta = beta_stacked if beta_stacked is not None else proj(self.gate_proj).float().reshape(B, T, N, Hh)
            alpha = alpha_stacked if alpha_stacked is not None else self.decay_proj(p_tok.mean(1)).float().reshape(B, T, Hh)
        if keep_mask_embed:
            # keep every parameter in the graph (DDP: no unused params) -- through the B*T*Hh decay logits, not through v: a zero added to the
            # [B*T*N, Hh*Dv] values was a 29 us pass forward and a reduction over their gradient backward (round 4)
            alpha = alpha + 0.0 * self.mask_embed.weight.sum()
        return p_tok, k_tok, q, v, alpha, beta, norms

    @torch.no_grad()
    def fuse_for_inference(self):
        """Fold every eval-mode BatchNorm into the convolution in front of it (in place; inference only).
        The folded module computes the same function; BatchNorm kernels were 46 % of a cfg2 forward
        (profiles/r01_a_bench_cfg2_kernel_stats.csv)."""
        if self.training:
            raise RuntimeError("fuse_for_inference() needs eval() mode")
        self.invalidate_packed_weights()
        def fuse_seq(seq):
            """conv, bn, relu -> FusedConv(relu) ; conv, bn -> FusedConv(no relu)"""
            mods, out, i = list(_fold_sequential(seq)), [], 0
            while i < len(mods):
                if isinstance(mods[i], nn.Conv2d):
                    relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                    out.append(FusedConv(mods[i], relu)); i += 2 if relu else 1
                else:
                    out.append(mods[i]); i += 1
            return nn.Sequential(*out)

        for m in list(self.modules()):
            if isinstance(m, BasicBlock) and isinstance(m.bn1, nn.BatchNorm2d):
                m.conv1, m.bn1 = FusedConv(_fold_bn(m.conv1, m.bn1), True), nn.Identity()
                m.conv2, m.bn2 = FusedConv(_fold_bn(m.conv2, m.bn2), True), nn.Identity()   # + residual, then ReLU
                if m.down is not None:
                    m.down = fuse_seq(m.down)
                    d = m.down[0]
                    if len(m.down) == 1 and isinstance(d, FusedConv) and not d.relu:
                        # out = relu(conv2(.) + b2 + (down(x) + bd)): one bias, one epilogue pass instead of two
                        m.conv2.epi.bias.data += d.epi.bias.data
                        d.bias_folded_downstream = True
            elif isinstance(m, UpBlock):
                m.conv = fuse_seq(m.conv)
        stem = fuse_seq(self.encoder.stem)
        if (len(stem) == 2 and isinstance(stem[0], FusedConv) and stem[0].relu and isinstance(stem[1], nn.MaxPool2d)
                and (stem[1].kernel_size, stem[1].stride, stem[1].padding) == (3, 2, 1)):
            fused = FusedConvPool.__new__(FusedConvPool)
            nn.Module.__init__(fused)
            fused.conv, fused.epi, fused.relu = stem[0].conv, stem[0].epi, True
            stem = nn.Sequential(fused.enable_s2d())
        self.encoder.stem = stem
        self._share_epoch_cell()
        return self

    @torch.no_grad()
    def segment(self, frames, target=None, return_state: bool = False, _mask_out=None, _counts_out=None, **kw):
        """logits -> (mask uint8 [B,T,H,W], Dice counts int32 [B,T,ncls,3] | None) with the HIP argmax kernel
        (return_state: plus the memory state after the last frame, [B,Hh,Dk,Dv] fp32).  _mask_out / _counts_out: caller-owned contiguous
        outputs of those shapes (GraphedSegment: slices of ONE result for the parts of a batch it runs on several streams)."""
        B, T, _, H, W = frames.shape
        if self.cfg.mask_feedback:
            _, mask, counts, s_out = self._forward_feedback(frames, kw.get("mask0"), kw.get("state"), target=target, masks_only=True,
                                                            mask_out=_mask_out, counts_out=_counts_out)
            return (mask, counts, s_out) if return_state else (mask, counts)
        tgt = None if target is None else target.reshape(B * T, H, W).contiguous()
        mo = None if _mask_out is None else _mask_out.view(B * T, H, W)
        co = None if _counts_out is None else _counts_out.view(B * T, -1, 3)
        lowres = self.forward(frames, _lowres=True, _head_fused=True, return_state=return_state, **kw)   # [B,T,ncls,H/4,W/4], or the feature under the head
        s_out = None
        if return_state:
            lowres, s_out = lowres
        out = self._masks_of(lowres, (B, T, H, W), tgt, mo, co)
        return out + (s_out,) if return_state else out

    def _masks_of(self, lowres, dims, tgt, mo, co):
        """The last step of segment(): stride-4 logits (or the feature under the head) -> (mask [B,T,H,W], counts [B,T,ncls,3] | None)."""
        B, T, H, W = dims
        if isinstance(lowres, HeadFeature):
            # head + upsample + argmax + Dice in one kernel: the class planes never reach memory (bit-identical to the two-kernel form)
            mask, counts = ops.head_upsample_argmax_dice(lowres.feature, lowres.weight, lowres.bias, H, W, tgt, mo, co)
            ncls = lowres.weight.shape[0]
        else:
            ncls, hl, wl = lowres.shape[2:]
            mask, counts = ops.upsample_argmax_dice(lowres.reshape(B * T, ncls, hl, wl).contiguous(), H, W, tgt, mo, co)
        return (mask.reshape(B, T, H, W), (None if counts is None else counts.reshape(B, T, ncls, 3)))

    def _encode_project(self, frames):
        """The state-INDEPENDENT part of the inference forward of [B,T,C,H,W] frames: encoder and projections (what forward() does up to
        _after_projection).  Returns (f4, f8, f16, projected)."""
        if self.training or self.cfg.mask_feedback:
            raise RuntimeError("_encode_project serves the inference forward in scan mode")
        B, T, C, H, W = frames.shape
        x = frames.reshape(B * T, C, H, W)
        dt = self.key_proj.weight.dtype
        stem0 = self.encoder.stem[0]
        if x.is_cuda and isinstance(stem0, FusedConvPool) and getattr(stem0, "w_s2d", None) is not None:
            x = x.to(dt)                                                         # (as forward(): the stem kernels read NCHW frames themselves)
        else:
            x = x.to(dtype=dt, memory_format=torch.channels_last)
        f4, f8, f16 = self.encoder(x)
        return f4, f8, f16, self._project(f16, B, T, None)

    def _segment_from_features(self, feats, state, dims, target=None):
        """segment(..., state=, return_state=True) from _encode_project's result: (mask, counts | None, state after the last frame)."""
        B, T, H, W = dims
        f4, f8, f16, projected = feats
        lowres, s_out = self._after_projection(f4, f8, f16, projected, state, True, True, dims, True)
        tgt = None if target is None else target.reshape(B * T, H, W).contiguous()
        return self._masks_of(lowres, dims, tgt, None, None) + (s_out,)

    @torch.no_grad()
    def segment_clip(self, frames, chunk_frames: int, target=None, mask0=None, state=None, graph: bool = False):
        """A long clip as consecutive chunks of ``chunk_frames`` frames with the memory state carried from chunk to chunk
        ("GDR memory-state carry across chunks", BASELINE.json configs[4]): activations are held for one chunk at a time, and -- with
        the default ``scan_segments = 1`` -- masks, Dice counts and the final state are those of ONE call over the whole clip, and of
        any other chunk length, bit for bit (every kernel of the inference build is deterministic and works frame by frame; the scan is
        chunk-invariant by gdkvm_scan_fwd's contract).  Returns (mask [B,T,H,W] uint8, counts [B,T,ncls,3] | None, final state)."""
        B, T = frames.shape[:2]
        if chunk_frames <= 0:
            raise ValueError("chunk_frames must be positive")
        masks, counts = [], []
        if graph and frames.is_cuda and mask0 is None and T % chunk_frames == 0:
            # every chunk is ONE replay of a hipGraph captured for the chunk's shape (GraphedSegment with the state carried; cached on the
            # module): the same kernels and the same bits as the loop below, without the host's launch calls between them -- a long clip is
            # many short forwards, which is where a slow host shows (round 4: 17.5 against 12.0 ms per 1024 frames between two boxes)
            cfg = self.cfg
            # (keyed on everything the captured kernels were chosen by: the shape, the recurrence and its segmenting, and the weight /
            # pack epochs -- a graph reads the weight packs of capture time; invalidate_packed_weights() drops the whole cache)
            key = (B, chunk_frames) + tuple(frames.shape[2:]) + (frames.dtype, target is not None, frames.device, cfg.rule, cfg.scan_segments,
                                                                   _epoch_of(self), self.__dict__.get("_pack_epoch", 0))
            cache = self.__dict__.setdefault("_clip_graphs", {})
            for old in [k_ for k_ in cache if k_[:-3] == key[:-2] and k_[:-1] != key]:
                del cache[old]                              # the same shape under older weights: never replayed again
            pipelined = (_CLIP_PIPELINE and not cfg.mask_feedback and T // chunk_frames >= 2
                         and (state is None or state.shape[-1] == cfg.value_dim + int(cfg.normalizer)))
            key = key + (pipelined,)
            if key not in cache:
                if pipelined:
                    cache[key] = PipelinedClip(self, frames[:, :chunk_frames].clone(), None if target is None else target[:, :chunk_frames].clone())
                else:
                    s0 = torch.zeros((B, cfg.heads, cfg.key_dim, cfg.value_dim), dtype=torch.float32, device=frames.device)
                    cache[key] = GraphedSegment(self, frames[:, :chunk_frames].clone(), None if target is None else target[:, :chunk_frames].clone(), state=s0)
            g = cache[key]
            if pipelined:
                return g(frames, target, state)
            cur = torch.zeros_like(g.state) if state is None else state
            for t0 in range(0, T, chunk_frames):
                m, c, s_new = g(frames[:, t0:t0 + chunk_frames], None if target is None else target[:, t0:t0 + chunk_frames], state=cur)
                masks.append(m.clone())
                counts.append(None if c is None else c.clone())
                cur = s_new                                 # (the graph's output buffer: copied into its input buffer by the next call)
            return torch.cat(masks, 1), (None if target is None else torch.cat(counts, 1)), cur.clone()
        for t0 in range(0, T, chunk_frames):
            t1 = min(T, t0 + chunk_frames)
            m, c, state = self.segment(frames[:, t0:t1], None if target is None else target[:, t0:t1], return_state=True,
                                       mask0=mask0 if t0 == 0 else None, state=state)
            masks.append(m)
            counts.append(c)
        return torch.cat(masks, 1), (None if target is None else torch.cat(counts, 1)), state

    def graphed_segment(self, frames, target=None, warmup: int = 2, state=None, streams=None, pool=None):
        """segment() for ONE clip shape captured into a hipGraph (GraphedSegment): a serving loop that replays it spends no host time
        on the ~25 launches of a forward; `streams`, `pool`: see GraphedSegment."""
        return GraphedSegment(self, frames, target, warmup, state, streams, pool)

    # -------------------------------------------------------------------------------------- checkpoints
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        remapped = {}
        for key, val in state_dict.items():
            for old, new in _KEY_REMAP.items():
                if key.startswith(old):
                    key = new + key[len(old):]
                    break
            remapped[key] = val
        self.invalidate_packed_weights()
        return super().load_state_dict(remapped, strict=strict, **kw)


def _packs_held(model: "GDKVM"):
    """Every packed-weight tensor the module caches hold right now (what a captured graph reads by address)."""
    held = [model.__dict__.get(n) for n in ("_qkv_pack", "_gate_w32", "_kpff_pack", "_mask_w32")]
    dec = model._modules.get("decoder")
    if dec is not None:
        held.append(dec.__dict__.get("_head_w32"))
    for m in model.modules():
        if isinstance(m, FusedConv):
            held.append(m.__dict__.get("_wpack"))
    for p in model.parameters():
        held.append(p.__dict__.get("_gdkvm_train_packs"))
    return [h for h in held if h is not None]


class GraphedSegment:
    """GDKVM.segment(frames, target) captured ONCE into a hipGraph and replayed (inference build, fixed clip shape): the same kernels in the
    same order with no host work between them -- a forward is ~25 launches for ~0.95 ms of GPU time, so a slow or busy host (one launch call
    costing more than ~35 us) would otherwise bound the rate.  Construction runs `warmup` eager calls on a side stream (weight packs, kernel
    attributes), then captures; a call copies the batch into the graph's input buffers unless it IS those buffers, replays and returns the
    graph's output tensors (overwritten by the next call): (mask uint8 [B,T,H,W], Dice counts int32 [B,T,ncls,3] | None)."""

    def __init__(self, model: "GDKVM", frames: torch.Tensor, target: Optional[torch.Tensor] = None, warmup: int = 2,
                 state: Optional[torch.Tensor] = None, streams: Optional[int] = None, pool=None):
        """state (fp32 [B,Hh,Dk,Dv]): capture the state-carrying form -- calls then take `state=` (copied into the graph's buffer) and
        return (mask, counts, state after the last frame), as segment(..., state=, return_state=True) does (GDKVM.segment_clip).
        streams: the batch is cut into that many equal groups of clips whose forwards run on their own streams INSIDE the one graph (fork at
        the start, join at the end, every group writing its slice of the one result).  Clips never interact, so the masks are the same
        bits; what changes is the schedule: every kernel of a forward fills the chip, drains with a tail and runs its phases in lockstep,
        and a second, independent stream fills those gaps -- cfg2: 0.898 against 0.940 ms per 16 x 32 frames with two groups of eight,
        although each half-batch kernel alone is less efficient (the halves one after the other: 1.085 ms); three or four groups and
        unequal ones lose (profiles/r05_n_two_streams_in_one_graph.txt).  None = 2 for batches of at least 8 clips that halve, else 1 (GDKVM_SEGMENT_STREAMS overrides where it divides the batch).
        pool: another GraphedSegment's ``graph.pool()`` -- graphs of one pool share their activation memory (never replay two of them
        concurrently; a graph's outputs are valid until another graph of the pool replays): several captures over DIFFERENT input buffers
        cost one set of activations (bench.py rotates eight input batches this way)."""
        if not frames.is_cuda:
            raise RuntimeError("GraphedSegment needs device tensors")
        self.model, self.frames, self.target, self.state = model, frames, target, state
        B = frames.shape[0]
        def aligned(n):
            """Every group writes its slice of the ONE mask / counts result, and the mask kernel wants 16-byte aligned outputs: the
            per-group byte offsets (per x T x H x W mask bytes, per x T x ncls x 12 count bytes) must be multiples of 16."""
            per_ = B // n
            return n == 1 or ((per_ * frames.shape[1] * frames.shape[3] * frames.shape[4]) % 16 == 0
                              and (target is None or (per_ * frames.shape[1] * model.cfg.num_classes * 12) % 16 == 0))
        if streams is None:
            env = os.environ.get("GDKVM_SEGMENT_STREAMS", "")
            streams = int(env) if env.isdigit() and int(env) >= 1 and B % int(env) == 0 else (2 if (B >= 8 and B % 2 == 0) else 1)
            if not aligned(streams):
                streams = 1                                 # (e.g. 10 clips x 3 frames x 2 classes: 360-byte count slabs)
        if streams < 1 or B % streams:
            raise ValueError(f"GraphedSegment: streams={streams} must divide the {B} clips")
        if not aligned(streams):
            raise ValueError(f"GraphedSegment: with streams={streams} a group's slice of the mask / Dice-count result does not start on a 16-byte "
                             f"boundary for {B} clips x {frames.shape[1]} frames ({model.cfg.num_classes} classes); use streams=1")
        self.streams = streams
        kw = {} if state is None else {"state": state, "return_state": True}
        with torch.no_grad():
            side = torch.cuda.Stream(device=frames.device)
            side.wait_stream(torch.cuda.current_stream(frames.device))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    model.segment(self.frames, self.target, **kw)
                    if streams > 1:                        # (the groups' shapes too: nothing may be built or sized inside the capture)
                        kw_w = {} if state is None else {"state": state[: B // streams], "return_state": True}
                        model.segment(self.frames[: B // streams], None if target is None else self.target[: B // streams], **kw_w)
            torch.cuda.current_stream(frames.device).wait_stream(side)
            torch.cuda.synchronize(frames.device)
            self.graph = torch.cuda.CUDAGraph()
            # (with a process group alive its watchdog THREAD may poll events of earlier collectives while this thread captures: under the
            # default "global" capture mode that aborts the process -- train.GraphedTrainStep met it -- so only this thread is policed)
            mode = {} if pool is None else {"pool": pool}
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                mode["capture_error_mode"] = "thread_local"
            if streams == 1:
                with torch.cuda.graph(self.graph, **mode):
                    self.out = model.segment(self.frames, self.target, **kw)
            else:
                T, H, W = frames.shape[1], frames.shape[3], frames.shape[4]
                mask = torch.empty((B, T, H, W), dtype=torch.uint8, device=frames.device)
                counts = None if target is None else torch.empty((B, T, model.cfg.num_classes, 3), dtype=torch.int32, device=frames.device)
                self._side = [torch.cuda.Stream(device=frames.device) for _ in range(streams)]
                per = B // streams
                with torch.cuda.graph(self.graph, **mode):
                    cur = torch.cuda.current_stream(frames.device)
                    for s_ in self._side:
                        s_.wait_stream(cur)
                    states = []
                    for i, s_ in enumerate(self._side):
                        with torch.cuda.stream(s_):
                            lo, hi = i * per, (i + 1) * per
                            kw_i = {} if state is None else {"state": state[lo:hi], "return_state": True}
                            res = model.segment(self.frames[lo:hi], None if target is None else self.target[lo:hi], _mask_out=mask[lo:hi],
                                                _counts_out=None if counts is None else counts[lo:hi], **kw_i)
                            if state is not None:
                                states.append(res[2])
                    for s_ in self._side:
                        cur.wait_stream(s_)
                    self.out = (mask, counts) if state is None else (mask, counts, torch.cat(states, 0))     # (the groups' final states: one small copy)
        # The graph holds raw addresses of the weight packs the warm-up calls built OUTSIDE its memory pool: keep them alive here (a replay
        # must never read freed memory), and remember the epochs they belong to -- a replay after the weights or packs changed would segment
        # with the weights of capture time, so __call__ raises instead.
        self._held = _packs_held(model)
        self._epochs = (_epoch_of(model), model.__dict__.get("_pack_epoch", 0))

    def __call__(self, frames: torch.Tensor, target: Optional[torch.Tensor] = None, state: Optional[torch.Tensor] = None):
        if (frames.shape != self.frames.shape or frames.dtype != self.frames.dtype or (target is None) != (self.target is None)
                or (state is None) != (self.state is None)):
            raise RuntimeError(f"GraphedSegment was captured for frames {tuple(self.frames.shape)} {self.frames.dtype}"
                               f"{'' if self.target is None else ' with a target'}{'' if self.state is None else ' with a state'}")
        if self._epochs != (_epoch_of(self.model), self.model.__dict__.get("_pack_epoch", 0)):
            raise RuntimeError("GraphedSegment: the model's weights or weight packs changed since the capture (optimiser step, load_state_dict, "
                               ".to(), train() / eval()): the graph reads the packs of capture time -- capture a new one")
        if frames.data_ptr() != self.frames.data_ptr():
            self.frames.copy_(frames, non_blocking=True)
        if target is not None and target.data_ptr() != self.target.data_ptr():
            self.target.copy_(target, non_blocking=True)
        if state is not None and state.data_ptr() != self.state.data_ptr():
            self.state.copy_(state, non_blocking=True)
        self.graph.replay()
        return self.out


class PipelinedClip:
    """GDKVM.segment_clip's chunk loop (a long clip as consecutive chunks with the memory state carried: BASELINE.json configs[4]) with the NEXT
    chunk's encoder and projections running BESIDE the current chunk's memory path, KPFF, decoder and mask kernel (round 6).  Only the second
    half depends on the state, and a chunk of a long clip is few frames of few clips -- kernels that leave most of the chip idle -- so the two
    halves of consecutive chunks overlap: configs[4] (2 clips x 512 frames of 256 x 256 in 16 chunks) 12.3 -> 11.2 ms, one 112 x 112 clip of 256
    frames 2.68 -> 2.07 ms, four 3.45 -> 2.80 (tools/clip_pipeline_probe.py, profiles/r06_at_clip_pipeline.txt; GDKVM_CLIP_PIPELINE=0 = one
    whole-forward graph per chunk).  Two captured "front" graphs (encoder + projections, double-buffered outputs, a stream of their own) and two
    "back" graphs (scan with the state carried, KPFF, decoder, masks + Dice counts; the caller's stream), tied by events:
        front(i + 1) waits for back(i - 1) (which read the buffers front(i + 1) overwrites);  back(i) waits for front(i).
    The same kernels on the same operands in the same order per chunk: masks, counts and the final state are segment_clip's, bit for bit."""

    def __init__(self, model: "GDKVM", frames: torch.Tensor, target: Optional[torch.Tensor] = None, warmup: int = 2):
        if not frames.is_cuda:
            raise RuntimeError("PipelinedClip needs device tensors")
        cfg, dev = model.cfg, frames.device
        B, Tc, _, H, W = frames.shape
        self.model, self.shape, self.dtype, self.has_target = model, tuple(frames.shape), frames.dtype, target is not None
        self.dims = (B, Tc, H, W)
        self.state = torch.zeros((B, cfg.heads, cfg.key_dim, cfg.value_dim + int(cfg.normalizer)), dtype=torch.float32, device=dev)
        self.fin = [frames.clone(), frames.clone()]
        self.tgt = [None, None] if target is None else [target.clone(), target.clone()]
        self.front_stream = torch.cuda.Stream(device=dev)
        mode = {}
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            mode["capture_error_mode"] = "thread_local"
        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):          # (weight packs, kernel attributes, workspaces: nothing may be built inside a capture)
                    model._segment_from_features(model._encode_project(self.fin[0]), self.state, self.dims, self.tgt[0])
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.gfront, self.feats, self.gback, self.out = [], [], [], []
            for j in range(2):                           # front graphs: a pool EACH -- in a shared pool the second capture places its outputs where
                g = torch.cuda.CUDAGraph()               # the first keeps scratch, and front(i + 2) would write over what back(i + 1) still reads
                with torch.cuda.graph(g, **mode):
                    f = model._encode_project(self.fin[j])
                self.gfront.append(g)
                self.feats.append(f)
            for j in range(2):                           # back graphs: a pool of their own (a back graph runs while a front graph does)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **(mode if not self.gback else dict(mode, pool=self.gback[0].pool()))):
                    o = model._segment_from_features(self.feats[j], self.state, self.dims, self.tgt[j])
                self.gback.append(g)
                self.out.append(o)
        self._held = _packs_held(model)
        self._epochs = (_epoch_of(model), model.__dict__.get("_pack_epoch", 0))

    @torch.no_grad()
    def __call__(self, frames: torch.Tensor, target: Optional[torch.Tensor] = None, state: Optional[torch.Tensor] = None):
        """frames [B, n * chunk, C, H, W] -> (mask [B,T,H,W], counts | None, state after the last frame)."""
        B, Tc = self.shape[:2]
        if (frames.shape[0] != B or tuple(frames.shape[2:]) != self.shape[2:] or frames.dtype != self.dtype or frames.shape[1] % Tc
                or (target is None) != (not self.has_target)):
            raise RuntimeError(f"PipelinedClip was captured for chunks of {self.shape} {self.dtype}{' with a target' if self.has_target else ''}")
        if self._epochs != (_epoch_of(self.model), self.model.__dict__.get("_pack_epoch", 0)):
            raise RuntimeError("PipelinedClip: the model's weights or weight packs changed since the capture -- capture a new one")
        dev = frames.device
        cur, fs = torch.cuda.current_stream(dev), self.front_stream
        n = frames.shape[1] // Tc
        if state is None:
            self.state.zero_()
        else:
            self.state.copy_(state, non_blocking=True)
        fs.wait_stream(cur)                              # the clip is ready as far as the caller's stream knows
        front_done, back_done = [None, None], [None, None]

        def launch_front(i):
            j = i % 2
            with torch.cuda.stream(fs):
                if back_done[j] is not None:
                    fs.wait_event(back_done[j])          # back(i - 2) read the buffers this replay overwrites
                self.fin[j].copy_(frames[:, i * Tc:(i + 1) * Tc], non_blocking=True)
                self.gfront[j].replay()
                front_done[j] = torch.cuda.Event()
                front_done[j].record(fs)

        masks, counts = [], []
        launch_front(0)
        for i in range(n):
            j = i % 2
            if i + 1 < n:
                launch_front(i + 1)                      # queued BEFORE this chunk's second half: it runs beside it
            cur.wait_event(front_done[j])
            if target is not None:
                self.tgt[j].copy_(target[:, i * Tc:(i + 1) * Tc], non_blocking=True)
            self.gback[j].replay()
            m, c, s_new = self.out[j]
            self.state.copy_(s_new, non_blocking=True)   # (the next chunk's back graph reads the one state buffer)
            masks.append(m.clone())
            counts.append(None if c is None else c.clone())
            back_done[j] = torch.cuda.Event()
            back_done[j].record(cur)
        cur.wait_stream(fs)
        return torch.cat(masks, 1), (None if target is None else torch.cat(counts, 1)), self.state.clone()


class InFlightSegments:
    """Several forwards IN FLIGHT (round 6): one captured GraphedSegment (a single stream inside) per resident input batch, replayed round-robin
    on `in_flight` host streams, so that batch i + 1 starts while batch i is still running.  Whole-batch kernels are more efficient than the
    half-batch kernels of the two-groups-inside-one-graph form, and two forwards that drift apart overlap DIFFERENT kernels (a memory-bound
    mask kernel beside an MFMA-bound convolution) instead of the same kernel twice: cfg2 0.81-0.83 ms per forward with two in flight against
    0.89-0.90 for GraphedSegment(streams=2) one at a time and 0.93 for one stream one at a time; three or four in flight do not add
    (profiles/r06_y_forwards_in_flight.txt).  Throughput, not latency: a batch's masks are ready when ITS stream has finished.
    Graph i always runs on stream i % in_flight; the graphs of one stream share a memory pool (they never overlap), graphs of different streams
    do not.  launch(i) replays graph i on its stream and returns (mask, counts | None, event recorded behind it); wait(i) makes the caller's
    current stream wait for that replay; synchronize() waits for every stream on the host."""

    def __init__(self, model: "GDKVM", batches, targets=None, in_flight: int = 2):
        if in_flight < 1 or not batches or len(batches) % in_flight:
            raise ValueError(f"InFlightSegments: {len(batches)} batches must be a positive multiple of in_flight={in_flight}")
        dev = batches[0].device
        self.in_flight = in_flight
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(in_flight)]
        self.graphs, self.events = [], [None] * len(batches)
        for i, b in enumerate(batches):
            pool = None if i < in_flight else self.graphs[i % in_flight].graph.pool()
            self.graphs.append(GraphedSegment(model, b, None if targets is None else targets[i], streams=1, pool=pool))
        torch.cuda.synchronize(dev)

    def __len__(self):
        return len(self.graphs)

    def launch(self, i: int):
        g = self.graphs[i % len(self.graphs)]
        s = self.streams[(i % len(self.graphs)) % self.in_flight]
        with torch.cuda.stream(s):
            out = g(g.frames, g.target)
            ev = torch.cuda.Event()
            ev.record(s)
        self.events[i % len(self.graphs)] = ev
        return out[0], out[1], ev

    def wait(self, i: int) -> None:
        ev = self.events[i % len(self.graphs)]
        if ev is not None:
            torch.cuda.current_stream(self.graphs[0].frames.device).wait_event(ev)

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()
