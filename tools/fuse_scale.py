#!/usr/bin/env python3
"""Diagnostic: scan_prep time against the number of frames (N = 256, Dv = 256, bf16), chunk-parallel + compose vs the fused chunk walk."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from tools.config_sweep import ev_time

dev = torch.device("cuda")
N, Dv = 256, 256
for FH in (64, 128, 256, 384, 512, 768, 1024, 2048):
    B, T = 2, FH // 2
    torch.manual_seed(FH)
    q, k = (torch.randn(B, T, N, 1, 64, device=dev).bfloat16() for _ in range(2))
    v = torch.randn(B, T, N, 1, Dv, device=dev).bfloat16()
    be = torch.randn(B, T, N, 1, device=dev)
    ws = torch.empty(ops.scan_workspace_bytes(B, T, 1, N, 64, Dv), dtype=torch.uint8, device=dev)
    t = {}
    for mode in ("0", "1"):
        os.environ["GDKVM_PREP_FUSE"] = mode
        t[mode] = ev_time(lambda: ops.scan_prep(q, k, v, be, ws, flags=3), iters=5)
    print(f"frames {FH:5d}: chunk-parallel + compose {t['0']:7.1f} us   fused {t['1']:7.1f} us", flush=True)
