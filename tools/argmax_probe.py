#!/usr/bin/env python3
"""Diagnostic: the mask kernels at the cfg2 shape (512 frames, 28x28 -> 112x112, 2 classes, bf16): head, upsample + argmax (+ Dice), fused."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from tools.config_sweep import ev_time

dev = torch.device("cuda")
BT, C, ncls, hl, H = 512, 64, 2, 28, 112
x = torch.randn(BT, C, hl, hl, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(ncls, C, device=dev) / 8
b = torch.randn(ncls, device=dev)
tgt = torch.randint(0, ncls, (BT, H, H), device=dev, dtype=torch.uint8)
lo = ops.head_logits(x, w, b)
print(f"head_logits                         {ev_time(lambda: ops.head_logits(x, w, b)):6.1f} us")
print(f"upsample_argmax (no target)         {ev_time(lambda: ops.upsample_argmax_dice(lo, H, H)):6.1f} us")
print(f"upsample_argmax_dice                {ev_time(lambda: ops.upsample_argmax_dice(lo, H, H, tgt)):6.1f} us")
print(f"head_upsample_argmax (no target)    {ev_time(lambda: ops.head_upsample_argmax_dice(x, w, b, H, H)):6.1f} us")
print(f"head_upsample_argmax_dice           {ev_time(lambda: ops.head_upsample_argmax_dice(x, w, b, H, H, tgt)):6.1f} us")
