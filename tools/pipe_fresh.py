#!/usr/bin/env python3
"""The concurrent form of gdkvm_scan_fwd on a FRESH workspace (garbage, then the previous call's results for different inputs): a stale
line anywhere in the hand-over shows as a mismatch against the plain sequence run on another workspace.
usage: pipe_fresh.py B T N [Hh Dv bf16|f32]   (GDKVM_PREP_FUSE as set by the caller)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402

B, T, N = (int(x) for x in sys.argv[1:4])
Hh = int(sys.argv[4]) if len(sys.argv) > 4 else 1
Dv = int(sys.argv[5]) if len(sys.argv) > 5 else 256
dt = torch.float32 if len(sys.argv) > 6 and sys.argv[6] == "f32" else torch.bfloat16
dev = torch.device("cuda")
ws_p = ops.new_workspace(B, T, Hh, N, 64, Dv, dev)
ws_p.random_(0, 255)
ws_r = ops.new_workspace(B, T, Hh, N, 64, Dv, dev)
for seed in range(4):
    g = torch.Generator(device=dev).manual_seed(seed)
    q, k = (torch.randn(B, T, N, Hh, 64, device=dev, generator=g).to(dt) for _ in range(2))
    v = (torch.randn(B, T, N, Hh, Dv, device=dev, generator=g) * (seed + 1)).to(dt)
    al = 2 + torch.randn(B, T, Hh, device=dev, generator=g)
    be = torch.randn(B, T, N, Hh, device=dev, generator=g)
    s0 = torch.randn(B, Hh, 64, Dv, device=dev, generator=g)
    os.environ["GDKVM_SCAN_PIPE"] = "1"
    rp, sp = ops.scan_fwd(q, k, v, al, be, s0, flags=3, workspace=ws_p)
    os.environ["GDKVM_SCAN_PIPE"] = "0"
    rr, sr = ops.scan_fwd(q, k, v, al, be, s0, flags=3, workspace=ws_r)
    torch.cuda.synchronize()
    d = (rp.float() - rr.float()).abs().amax(dim=(2, 3))            # [B, T, Dv]
    bad_t = (d != 0).any(2).any(0).nonzero().flatten().tolist()
    bad_c = (d != 0).any(1).any(0).nonzero().flatten().tolist()
    ds = (sp - sr).abs()
    print(f"{B}x{T}x{N} Hh {Hh} Dv {Dv} {dt} fuse {os.environ.get('GDKVM_PREP_FUSE')} seed {seed}: read-out equal {torch.equal(rp, rr)} (differing {int((rp != rr).sum())}, "
          f"nan {int(torch.isnan(rp.float()).sum())}, max |d| {float(d.max()):.3g}, frames {bad_t[:16]} of {len(bad_t)}, columns {bad_c[:8]}..{bad_c[-4:]} of {len(bad_c)}), "
          f"state equal {torch.equal(sp, sr)} (differing {int((sp != sr).sum())}, max |d| {float(ds.max()):.3g})", flush=True)
