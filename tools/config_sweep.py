#!/usr/bin/env python3
"""Times the hot-path ops at the shapes of every BASELINE.json config (HIP events on the launch stream) and prints one
table row per (config, dtype): scan_fwd = prep + apply, KPFF, achieved algorithmic GB/s.  Numbers go into DESIGN.md §7."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402


def ev_time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3            # us


def main():
    dev = torch.device("cuda")
    cfgs = [("cfg1 1x8x49", 1, 8, 7, 7), ("cfg2 16x32x49", 16, 32, 7, 7), ("cfg3 8x20x256", 8, 20, 16, 16),
            ("cfg5 2x512x256", 2, 512, 16, 16), ("cfg5-chunk 2x32x256", 2, 32, 16, 16)]
    Hh, Dk, Dv, Cp = 1, 64, 256, 256
    print(f"{'config':22s} {'dtype':5s} {'prep us':>9s} {'scan us':>9s} {'fwd us':>9s} {'alg MB':>8s} {'GB/s':>8s} {'frac':>7s} {'kpff us':>9s}")
    for name, B, T, h, w in cfgs:
        N = h * w
        for dt in (torch.bfloat16, torch.float32):
            g = torch.Generator(device=dev).manual_seed(1)
            q, k = (torch.randn(B, T, N, Hh, Dk, device=dev, generator=g).to(dt) for _ in range(2))
            v = torch.randn(B, T, N, Hh, Dv, device=dev, generator=g).to(dt)
            al = 2 + torch.randn(B, T, Hh, device=dev, generator=g); be = torch.randn(B, T, N, Hh, device=dev, generator=g)
            ws = torch.empty(ops.scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=dev)
            r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=dt); s = torch.empty(B, Hh, Dk, Dv, device=dev)
            tp = ev_time(lambda: ops.scan_prep(q, k, v, be, ws, flags=3))
            ta = ev_time(lambda: ops.scan_apply(q, al, ws, Dv, flags=3, out=r, state_out=s))
            tf = ev_time(lambda: ops.scan_fwd(q, k, v, al, be, flags=3, workspace=ws, out=r, state_out=s))
            es = q.element_size()
            alg = B * T * (es * N * (2 * Hh * Dk + 2 * Hh * Dv) + 4 * Hh * (1 + N)) + B * 2 * 4 * Hh * Dk * Dv
            L = k.reshape(B * T, N, Dk); P = torch.randn(B * T, N, Cp, device=dev, generator=g).to(dt)
            cin = Cp + Dk + Dv
            wa = torch.randn(2 * Cp, cin, device=dev, generator=g) / cin ** 0.5; ba = torch.zeros(2 * Cp, device=dev)
            wl = torch.randn(Cp, Dk, device=dev, generator=g) / 8; wg = torch.randn(Cp, Dv, device=dev, generator=g) / 16
            f = torch.empty(B * T, N, Cp, device=dev, dtype=dt)
            tk = ev_time(lambda: ops.kpff_fwd(L, r.reshape(B * T, N, Dv), P, wa, ba, wl, wg, h, w, out=f))
            gbs = alg / (tf * 1e-6) / 1e9
            print(f"{name:22s} {'bf16' if dt == torch.bfloat16 else 'f32':5s} {tp:9.1f} {ta:9.1f} {tf:9.1f} {alg / 1e6:8.1f} {gbs:8.1f} {gbs / 8000:7.4f} {tk:9.1f}")
            if T >= 256:                        # long clips: the opt-in time-segmented scan (not bit-identical to chunked calls)
                ts = ev_time(lambda: ops.scan_fwd_segmented(q, k, v, al, be, segments=16, flags=3), iters=5)
                gs = alg / (ts * 1e-6) / 1e9
                print(f"{name + ' /16 seg':22s} {'bf16' if dt == torch.bfloat16 else 'f32':5s} {'':>9s} {'':>9s} {ts:9.1f} {alg / 1e6:8.1f} {gs:8.1f} {gs / 8000:7.4f}")


if __name__ == "__main__":
    main()
