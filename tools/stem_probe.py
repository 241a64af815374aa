#!/usr/bin/env python3
"""Probe: the stem conv (7x7/2 on 3 channels) vs its space-to-depth form (4x4/1 on 12 channels padded to 16 or not)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.backends.cudnn.benchmark = True
from tools.config_sweep import ev_time
dev = torch.device("cuda"); bf = torch.bfloat16
x = torch.randn(512, 3, 112, 112, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 3, 7, 7, device=dev) * 0.05).to(bf).contiguous(memory_format=torch.channels_last)
print(f"7x7/2 cin3      : {ev_time(lambda: F.conv2d(x, w, None, 2, 3)):8.1f} us")
for cin in (12, 16):
    xs = torch.randn(512, cin, 56, 56, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
    ws = (torch.randn(64, cin, 4, 4, device=dev) * 0.05).to(bf).contiguous(memory_format=torch.channels_last)
    print(f"4x4/1 cin{cin:<2d} p2  : {ev_time(lambda: F.conv2d(xs, ws, None, 1, 2)):8.1f} us")
    xp = torch.randn(512, cin, 59, 59, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
    print(f"4x4/1 cin{cin:<2d} p0  : {ev_time(lambda: F.conv2d(xp, ws, None, 1, 0)):8.1f} us  (pre-padded 59x59 input)")
