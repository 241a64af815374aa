#!/usr/bin/env python3
"""Runs gdkvm_scan_fwd_segmented alone at a given shape (for rocprofv3 --kernel-trace --stats).  usage: seg_only.py B T N Dv segments [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402

B, T, N, Dv, seg = (int(x) for x in sys.argv[1:6])
it = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
q, k = (torch.randn(B, T, N, 1, 64, device=dev, generator=g).bfloat16() for _ in range(2))
v = torch.randn(B, T, N, 1, Dv, device=dev, generator=g).bfloat16()
al = 2 + torch.randn(B, T, 1, device=dev, generator=g); be = torch.randn(B, T, N, 1, device=dev, generator=g)
for _ in range(it):
    ops.scan_fwd_segmented(q, k, v, al, be, segments=seg, flags=3)
torch.cuda.synchronize()
