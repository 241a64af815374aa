#!/usr/bin/env python3
"""Long-clip scan (cfg5 shape: B=2, T=512, N=256, bf16): one serial call vs ops.scan_fwd_segmented with 4..32 time segments."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from tools.config_sweep import ev_time

dev = torch.device("cuda")
B, T, N, Hh, Dk, Dv = 2, 512, 256, 1, 64, 256
g = torch.Generator(device=dev).manual_seed(1)
q, k = (torch.randn(B, T, N, Hh, Dk, device=dev, generator=g).bfloat16() for _ in range(2))
v = torch.randn(B, T, N, Hh, Dv, device=dev, generator=g).bfloat16()
al = 2 + torch.randn(B, T, Hh, device=dev, generator=g); be = torch.randn(B, T, N, Hh, device=dev, generator=g)
r0, s0 = ops.scan_fwd(q, k, v, al, be, flags=3)
print(f"serial            {ev_time(lambda: ops.scan_fwd(q, k, v, al, be, flags=3), iters=5):9.1f} us")
for seg in (4, 8, 16, 32):
    r, s = ops.scan_fwd_segmented(q, k, v, al, be, segments=seg, flags=3)
    err = (s - s0).abs().max().item()
    t = ev_time(lambda: ops.scan_fwd_segmented(q, k, v, al, be, segments=seg, flags=3), iters=5)
    print(f"segments={seg:2d}       {t:9.1f} us   max|dS| vs serial {err:.2e}")
