#!/usr/bin/env python3
"""Diagnostic: gdkvm_scan_fwd as one call vs two half-clips in time with the second half's prep on a side stream while the first half
is applied (state carried: bit-identical by the chunking contract).  cfg2 and cfg3 shapes, bf16."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from tools.config_sweep import ev_time

dev = torch.device("cuda")
side = torch.cuda.Stream()
for name, B, T, N in (("cfg2", 16, 32, 49), ("cfg3", 8, 20, 256), ("cfg5-chunk", 2, 32, 256)):
    Dv = 256
    g = torch.Generator(device=dev).manual_seed(1)
    q, k = (torch.randn(B, T, N, 1, 64, device=dev, generator=g).bfloat16() for _ in range(2))
    v = torch.randn(B, T, N, 1, Dv, device=dev, generator=g).bfloat16()
    al = 2 + torch.randn(B, T, 1, device=dev, generator=g); be = torch.randn(B, T, N, 1, device=dev, generator=g)
    h = T // 2
    halves = [[x[:, :h].contiguous() for x in (q, k, v, al, be)], [x[:, h:].contiguous() for x in (q, k, v, al, be)]]
    ws = [torch.empty(ops.scan_workspace_bytes(B, x[0].shape[1], 1, N, 64, Dv), dtype=torch.uint8, device=dev) for x in halves]
    wsf = torch.empty(ops.scan_workspace_bytes(B, T, 1, N, 64, Dv), dtype=torch.uint8, device=dev)
    r = torch.empty(B, T, N, 1, Dv, device=dev, dtype=torch.bfloat16); s = torch.empty(B, 1, 64, Dv, device=dev)
    ra, rb = torch.empty_like(r[:, :h].contiguous()), torch.empty_like(r[:, h:].contiguous())
    sa = torch.empty_like(s)

    def one():
        ops.scan_fwd(q, k, v, al, be, flags=3, workspace=wsf, out=r, state_out=s)

    def two():
        cur = torch.cuda.current_stream()
        a, b = halves
        ops.scan_prep(a[0], a[1], a[2], a[4], ws[0], flags=3)
        side.wait_stream(cur)                              # (inputs ready; in a product: an event recorded before the first prep)
        with torch.cuda.stream(side):
            ops.scan_prep(b[0], b[1], b[2], b[4], ws[1], flags=3)
        ops.scan_apply(a[0], a[3], ws[0], Dv, flags=3, out=ra, state_out=sa)
        cur.wait_stream(side)
        ops.scan_apply(b[0], b[3], ws[1], Dv, state=sa, flags=3, out=rb, state_out=s)

    one(); s1 = s.clone(); two()
    same = torch.equal(s1, s) and torch.equal(r[:, :h], ra) and torch.equal(r[:, h:], rb)
    print(f"{name}: one call {ev_time(one):6.1f} us   two halves, prep overlapped {ev_time(two):6.1f} us   bit-identical {same}", flush=True)
