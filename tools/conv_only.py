#!/usr/bin/env python3
"""Runs one hand-written convolution layer a few times (for rocprofv3 counter passes).  usage: conv_only.py C H K [kernel] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402

C, H, K = (int(x) for x in sys.argv[1:4])
kern = int(sys.argv[4]) if len(sys.argv) > 4 else 5
it = int(sys.argv[5]) if len(sys.argv) > 5 else 5
x = torch.randn(512, C, H, H, device="cuda").relu().bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(K, C, 3, 3, device="cuda") / (C * 9) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
b = torch.randn(K, device="cuda")
for _ in range(it):
    ops.conv_bias_act(x, w, b, None, 1, 1, True, kern)
torch.cuda.synchronize()
