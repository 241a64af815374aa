#!/usr/bin/env python3
"""Times gdkvm_scan_apply with and without the in-kernel read-out at a given shape (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402
from tools.config_sweep import ev_time  # noqa: E402


def main():
    B, T, N, Dv = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (16, 32, 49, 256)
    dev = torch.device("cuda")
    Hh, Dk = 1, 64
    for dt in (torch.bfloat16, torch.float32):
        g = torch.Generator(device=dev).manual_seed(1)
        q, k = (torch.randn(B, T, N, Hh, Dk, device=dev, generator=g).to(dt) for _ in range(2))
        v = torch.randn(B, T, N, Hh, Dv, device=dev, generator=g).to(dt)
        al = 2 + torch.randn(B, T, Hh, device=dev, generator=g); be = torch.randn(B, T, N, Hh, device=dev, generator=g)
        ws = ops.new_workspace(B, T, Hh, N, Dk, Dv, dev)
        r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=dt); s = torch.empty(B, Hh, Dk, Dv, device=dev)
        ops.scan_prep(q, k, v, be, ws, flags=3)
        t_r = ev_time(lambda: ops.scan_apply(q, al, ws, Dv, flags=3, out=r, state_out=s))
        t_n = ev_time(lambda: ops.scan_apply(q, al, ws, Dv, flags=3, state_out=s, want_readout=False))
        print(f"B={B} T={T} N={N} Dv={Dv} {str(dt)[6:]:9s} apply with read-out {t_r:8.1f} us   states only {t_n:8.1f} us   per frame {1e3 * t_r / T:7.1f} / {1e3 * t_n / T:7.1f} ns")


if __name__ == "__main__":
    main()
