#!/usr/bin/env python3
"""Diagnostic only: builds ablation variants of the library (extra -D flags on the command line, -DGDKVM_ABL_RNOCVT | _RNOSTORE | _RNOLOAD | _RNOMMA)
into gpurun_out/ and times scan_prep / scan_apply with them.  Ablated builds compute wrong results by design; their
timings say which role of the serial kernel bounds a frame.  Never part of the product."""
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")


def main():
    flags = sys.argv[1:]
    so = os.path.join(ROOT, "gpurun_out", "libgdkvm_hip_abl.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    # only the scan kernel is rebuilt with the flags; every other object is the product build's (csrc/_obj travels with the tree)
    obj = os.path.join(ROOT, "gpurun_out", "gdr_scan_abl.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-c"] + flags +
                          ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, "gdr_scan.hip"), "-o", obj])
    others = [o for o in sorted(glob.glob(os.path.join(CSRC, "_obj", "*.o"))) if not o.endswith("gdr_scan.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    from gdkvm_amd import ops
    ops._SO = so
    from tools.config_sweep import ev_time
    dev = torch.device("cuda")
    out = []
    for (B, T, N) in [(16, 32, 49), (16, 128, 49)]:
        Hh, Dk, Dv, dt = 1, 64, 256, torch.bfloat16
        q, k = (torch.randn(B, T, N, Hh, Dk, device=dev).to(dt) for _ in range(2))
        v = torch.randn(B, T, N, Hh, Dv, device=dev).to(dt)
        al = 2 + torch.randn(B, T, Hh, device=dev); be = torch.randn(B, T, N, Hh, device=dev)
        ws = torch.empty(ops.scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=dev)
        r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=dt); s = torch.empty(B, Hh, Dk, Dv, device=dev)
        tp = ev_time(lambda: ops.scan_prep(q, k, v, be, ws, flags=3))
        ta = ev_time(lambda: ops.scan_apply(q, al, ws, Dv, flags=3, out=r, state_out=s))
        out.append(f"T={T}: prep {tp:.1f} scan {ta:.1f} us ({ta / T * 1e3:.0f} ns/frame)")
    print(f"{' '.join(flags) or 'baseline':60s} " + " | ".join(out))


if __name__ == "__main__":
    main()
