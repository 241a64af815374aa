#!/usr/bin/env python3
"""Runs only the HIP hot-path ops at the cfg2 shapes (no torch convolutions) -- the command profiled for the PMC
passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE) and for clean per-kernel stats.
    python3 tools/hotpath_only.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda")
    B, T, N, Hh, Dk, Dv, Cp, S, ncls = 16, 32, 49, 1, 64, 256, 256, 112, 2
    g = torch.Generator(device=dev).manual_seed(1)
    q, k = (torch.randn(B, T, N, Hh, Dk, device=dev, generator=g).bfloat16() for _ in range(2))
    v = torch.randn(B, T, N, Hh, Dv, device=dev, generator=g).bfloat16()
    al = 2 + torch.randn(B, T, Hh, device=dev, generator=g)
    be = torch.randn(B, T, N, Hh, device=dev, generator=g)
    ws = torch.empty(ops.scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=dev)
    r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=torch.bfloat16)
    s = torch.empty(B, Hh, Dk, Dv, device=dev)
    L = torch.randn(B * T, N, Dk, device=dev, generator=g).bfloat16()
    P = torch.randn(B * T, N, Cp, device=dev, generator=g).bfloat16()
    cin = Cp + Dk + Dv
    wa = torch.randn(2 * Cp, cin, device=dev, generator=g) / cin ** 0.5
    ba = torch.zeros(2 * Cp, device=dev)
    wl = torch.randn(Cp, Dk, device=dev, generator=g) / Dk ** 0.5
    wg = torch.randn(Cp, Dv, device=dev, generator=g) / Dv ** 0.5
    f = torch.empty(B * T, N, Cp, device=dev, dtype=torch.bfloat16)
    logits = torch.randn(B * T, ncls, S, S, device=dev, generator=g).bfloat16()
    tgt = (torch.rand(B * T, S, S, device=dev, generator=g) > 0.5).to(torch.uint8)
    wqkv = torch.randn(2 * Dk + Dv, Cp, device=dev, generator=g) / Cp ** 0.5
    pack, bqkv = ops.pack_rows_weight(wqkv), torch.zeros(2 * Dk + Dv, device=dev)
    wgt, wdc = (torch.randn(Hh, Cp, device=dev, generator=g) / Cp ** 0.5 for _ in range(2))
    bg, bd = torch.zeros(Hh, device=dev), 2 + torch.zeros(Hh, device=dev)
    for _ in range(iters):
        ops.proj_gates(P, pack, bqkv, wgt, bg, wdc, bd, Hh, Dk, Dv)        # row n4: projections + gates + norms, one launch
        ops.scan_fwd(q, k, v, al, be, flags=3, workspace=ws, out=r, state_out=s)
        ops.kpff_fwd(L, r.reshape(B * T, N, Dv), P, wa, ba, wl, wg, 7, 7, out=f)
        ops.argmax_dice(logits, tgt)
    torch.cuda.synchronize()
    print("hotpath_only done", iters)


if __name__ == "__main__":
    main()
