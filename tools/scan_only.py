#!/usr/bin/env python3
"""Runs gdkvm_scan_fwd alone at a given shape (for rocprofv3 --kernel-trace --stats).  usage: scan_only.py B T N Dv [iters] [f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402

B, T, N, Dv = (int(x) for x in sys.argv[1:5])
it = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dt = torch.float32 if len(sys.argv) > 6 and sys.argv[6] == "f32" else torch.bfloat16
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
q, k = (torch.randn(B, T, N, 1, 64, device=dev, generator=g).to(dt) for _ in range(2))
v = torch.randn(B, T, N, 1, Dv, device=dev, generator=g).to(dt)
al = 2 + torch.randn(B, T, 1, device=dev, generator=g); be = torch.randn(B, T, N, 1, device=dev, generator=g)
ws = ops.new_workspace(B, T, 1, N, 64, Dv, dev)
r = torch.empty(B, T, N, 1, Dv, device=dev, dtype=dt); s = torch.empty(B, 1, 64, Dv, device=dev)
for _ in range(it):
    ops.scan_fwd(q, k, v, al, be, None, flags=3, workspace=ws, out=r, state_out=s)
torch.cuda.synchronize()
