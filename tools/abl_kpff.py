#!/usr/bin/env python3
"""Diagnostic only: rebuilds csrc/kpff.hip with extra -D flags (e.g. -DKPFF_OT=2, -DKPFF_SKIP_POOL) into gpurun_out/, links it with
the product's other objects and times the bf16 KPFF kernel at the cfg2 shape.  Never part of the product."""
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
    so = os.path.join(ROOT, "gpurun_out", "libgdkvm_hip_ablk.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    obj = os.path.join(ROOT, "gpurun_out", "kpff_abl.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + flags +
                          ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, "kpff.hip"), "-o", obj])
    others = [o for o in sorted(glob.glob(os.path.join(CSRC, "_obj", "*.o"))) if not o.endswith("kpff.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    from gdkvm_amd import ops
    ops._SO = so
    from tools.config_sweep import ev_time
    dev = torch.device("cuda"); B, T, N, Dk, Dv, Cp = 16, 32, 49, 64, 256, 256
    g = torch.Generator(device=dev).manual_seed(1)
    L = torch.randn(B * T, N, Dk, device=dev, generator=g).bfloat16(); G = torch.randn(B * T, N, Dv, device=dev, generator=g).bfloat16()
    P = torch.randn(B * T, N, Cp, device=dev, generator=g).bfloat16()
    cin = Cp + Dk + Dv
    wa = torch.randn(2 * Cp, cin, device=dev, generator=g) / cin ** 0.5; ba = torch.zeros(2 * Cp, device=dev)
    wl = torch.randn(Cp, Dk, device=dev, generator=g) / 8; wg = torch.randn(Cp, Dv, device=dev, generator=g) / 16
    f = torch.empty(B * T, N, Cp, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(ops.load().gdkvm_kpff_workspace_bytes(Dk, Dv, Cp, 1), dtype=torch.uint8, device=dev)
    ops.kpff_fwd(L, G, P, wa, ba, wl, wg, 7, 7, out=f, workspace=ws)
    t = ev_time(lambda: ops.kpff_fwd(L, G, P, wa, ba, wl, wg, 7, 7, out=f, workspace=ws, packed=True), iters=50)
    print(f"{' '.join(flags) or 'baseline':50s} kpff bf16 cfg2: {t:.1f} us")


if __name__ == "__main__":
    main()
