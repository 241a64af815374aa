#!/usr/bin/env python3
"""One serial gdkvm_scan_fwd against ops.scan_fwd_segmented with forced time-segment counts, bf16.
    python3 tools/seg_bench.py [cfg2|cfg3|cfg5]       (default cfg5: B=2, T=512, N=256)
cfg2 (B=16, T=32, N=49) is the shape where the serial grid is already one workgroup per CU: segments there are co-resident
workgroups on the same CUs (the serial kernel needs 8 KB of LDS), which is what VERDICT r02 item 3(iii) asked to be measured."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from tools.config_sweep import ev_time

SHAPES = {"cfg2": (16, 32, 49, (2, 4, 8)), "cfg3": (8, 20, 256, (2, 4, 5, 10)), "cfg5": (2, 512, 256, (4, 8, 16, 32))}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
B, T, N, segs = SHAPES[name]
dev = torch.device("cuda")
Hh, Dk, Dv = 1, 64, 256
g = torch.Generator(device=dev).manual_seed(1)
q, k = (torch.randn(B, T, N, Hh, Dk, device=dev, generator=g).bfloat16() for _ in range(2))
v = torch.randn(B, T, N, Hh, Dv, device=dev, generator=g).bfloat16()
al = 2 + torch.randn(B, T, Hh, device=dev, generator=g); be = torch.randn(B, T, N, Hh, device=dev, generator=g)
r0, s0 = ops.scan_fwd(q, k, v, al, be, flags=3)
it = 20 if name != "cfg5" else 5
print(f"{name}: B={B} T={T} N={N}")
print(f"serial            {ev_time(lambda: ops.scan_fwd(q, k, v, al, be, flags=3), iters=it):9.1f} us")
for seg in segs:
    r, s = ops.scan_fwd_segmented(q, k, v, al, be, segments=seg, flags=3)
    err = (s - s0).abs().max().item()
    t = ev_time(lambda: ops.scan_fwd_segmented(q, k, v, al, be, segments=seg, flags=3), iters=it)
    print(f"segments={seg:2d}       {t:9.1f} us   max|dS| vs serial {err:.2e}")
