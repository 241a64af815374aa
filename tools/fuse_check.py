#!/usr/bin/env python3
"""Diagnostic: the fused chunk walk of gdr_prepm_kernel (GDKVM_PREP_FUSE=1) against the chunk-parallel + compose path (=0) on the
same inputs, and their timings at the cfg3 / cfg5 shapes.  python tools/fuse_check.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from tools.config_sweep import ev_time

dev = torch.device("cuda")
for (B, T, N, Dv) in [(1, 3, 100, 64), (2, 4, 256, 256), (3, 2, 130, 128), (1, 2, 65, 192), (8, 20, 256, 256), (2, 512, 256, 256)]:
    torch.manual_seed(B * T + N)
    q, k = (torch.randn(B, T, N, 1, 64, device=dev).bfloat16() for _ in range(2))
    v = torch.randn(B, T, N, 1, Dv, device=dev).bfloat16()
    al = 2 + torch.randn(B, T, 1, device=dev)
    be = torch.randn(B, T, N, 1, device=dev)
    out = {}
    for mode in ("0", "1"):
        os.environ["GDKVM_PREP_FUSE"] = mode
        r, s = ops.scan_fwd(q, k, v, al, be, flags=3)
        ws = torch.empty(ops.scan_workspace_bytes(B, T, 1, N, 64, Dv), dtype=torch.uint8, device=dev)
        t = ev_time(lambda: ops.scan_prep(q, k, v, be, ws, flags=3), iters=5)
        out[mode] = (r.float(), s, t)
    dr = (out["0"][0] - out["1"][0]).abs().max().item()
    ds = (out["0"][1] - out["1"][1]).abs().max().item()
    print(f"B={B} T={T} N={N} Dv={Dv}: |dR| {dr:.2e} |dS| {ds:.2e} (|S| {out['0'][1].abs().max().item():.2f})  prep {out['0'][2]:.1f} -> {out['1'][2]:.1f} us", flush=True)
