#!/usr/bin/env python3
"""Times the inference forward (fused build, bf16) at a given clip shape.  usage: forward_shape.py B T H W [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402

B, T, H, W = (int(x) for x in sys.argv[1:5])
it = int(sys.argv[5]) if len(sys.argv) > 5 else 20
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
dev = torch.device("cuda")
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
frames = torch.rand(B, T, 3, H, W, device=dev).bfloat16()
with torch.no_grad():
    for _ in range(5):
        model.segment(frames)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        mask, _ = model.segment(frames)
    b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / it
print(f"B={B} T={T} {H}x{W}: forward {ms:.3f} ms = {B * T / ms * 1e3:.0f} frames/s; mask {tuple(mask.shape)}")
