#!/usr/bin/env python3
"""Diagnostic only: builds csrc/kpff.hip with -DKPFF_STAMPS (s_memtime stamps at the phase boundaries of kpff_bf16_kernel, lane 0 of
every wave of the first 64 workgroups, into a buffer of their own), runs the kernel at the cfg2 shape and prints the median time of
each phase in cycles and microseconds.  Never part of the product."""
import ctypes
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")


def proj(so):
    """--proj: the stamps of proj_gates_kernel's tile workgroups (stage | barrier | gate logits | MFMA tiles + stores | barrier | norms)."""
    from gdkvm_amd import ops
    dev = torch.device("cuda"); B, T, N, Hh, Dk, Dv, Cp = 16, 32, 49, 1, 64, 256, 256
    g = torch.Generator(device=dev).manual_seed(1)
    p = torch.randn(B * T, N, Cp, device=dev, generator=g).bfloat16()
    w = torch.randn(2 * Dk + Dv, Cp, device=dev, generator=g) / Cp ** 0.5
    b = torch.randn(2 * Dk + Dv, device=dev, generator=g)
    wg, bg = torch.randn(Hh, Cp, device=dev, generator=g) / Cp ** 0.5, torch.zeros(Hh, device=dev)
    wd, bd = torch.randn(Hh, Cp, device=dev, generator=g) / Cp ** 0.5, 2 + torch.zeros(Hh, device=dev)
    pack = ops.pack_rows_weight(w)
    for _ in range(20):
        ops.proj_gates(p, pack, b, wg, bg, wd, bd, Hh, Dk, Dv)
    buf = torch.zeros(64 * 8 * 16, dtype=torch.int64, device=dev)
    raw = ctypes.CDLL(so)
    raw.gdkvm_kpff_diag_set_buffer.argtypes = [ctypes.c_void_p]
    raw.gdkvm_kpff_diag_set_buffer(buf.data_ptr())
    ops.proj_gates(p, pack, b, wg, bg, wd, bd, Hh, Dk, Dv)
    torch.cuda.synchronize()
    st = buf.cpu().reshape(64, 8, 16).double()
    t0 = st[:, :, 0].min(1, keepdim=True).values
    names = {1: "token rows requested and written to LDS", 2: "barrier", 3: "write-gate logits", 4: "output tiles (MFMA, bias, stores, norm partials)",
             5: "barrier", 6: "inverse norms"}
    last = 0
    print("proj_gates_kernel<64>: median over 64 workgroups x 8 waves, cycles after the workgroup's start | phase cycles")
    for slot in range(1, 7):
        end = (st[:, :, slot] - t0).median().item()
        print(f"{slot:4d}  {names[slot]:55s} {end:12.0f} {end - last:12.0f}")
        last = end


def main():
    flags = ["-DKPFF_STAMPS"] + [a for a in sys.argv[1:] if a != "--proj"]
    so = os.path.join(ROOT, "gpurun_out", "libgdkvm_hip_stampk.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    obj = os.path.join(ROOT, "gpurun_out", "kpff_stamp.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + flags +
                          ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, "kpff.hip"), "-o", obj])
    others = [o for o in sorted(glob.glob(os.path.join(CSRC, "_obj", "*.o"))) if not o.endswith("kpff.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    from gdkvm_amd import ops
    ops._SO = so
    lib = ops.load()
    if "--proj" in sys.argv:
        return proj(so)
    dev = torch.device("cuda"); B, T, N, Dk, Dv, Cp = 16, 32, 49, 64, 256, 256
    g = torch.Generator(device=dev).manual_seed(1)
    L = torch.randn(B * T, N, Dk, device=dev, generator=g).bfloat16(); G = torch.randn(B * T, N, Dv, device=dev, generator=g).bfloat16()
    P = torch.randn(B * T, N, Cp, device=dev, generator=g).bfloat16()
    cin = Cp + Dk + Dv
    wa = torch.randn(2 * Cp, cin, device=dev, generator=g) / cin ** 0.5; ba = torch.zeros(2 * Cp, device=dev)
    wl = torch.randn(Cp, Dk, device=dev, generator=g) / 8; wg = torch.randn(Cp, Dv, device=dev, generator=g) / 16
    f = torch.empty(B * T, N, Cp, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(lib.gdkvm_kpff_workspace_bytes(Dk, Dv, Cp, 1), dtype=torch.uint8, device=dev)
    ops.kpff_fwd(L, G, P, wa, ba, wl, wg, 7, 7, out=f, workspace=ws)
    for _ in range(20):                                       # warm clocks and caches
        ops.kpff_fwd(L, G, P, wa, ba, wl, wg, 7, 7, out=f, workspace=ws, packed=True)
    buf = torch.zeros(64 * 8 * 16, dtype=torch.int64, device=dev)
    raw = ctypes.CDLL(so)
    raw.gdkvm_kpff_diag_set_buffer.argtypes = [ctypes.c_void_p]
    raw.gdkvm_kpff_diag_set_buffer(buf.data_ptr())
    ops.kpff_fwd(L, G, P, wa, ba, wl, wg, 7, 7, out=f, workspace=ws, packed=True)
    torch.cuda.synchronize()
    st = buf.cpu().reshape(64, 8, 16).double()
    t0 = st[:, :, 0].min(1, keepdim=True).values                # a workgroup's first stamp
    names = {1: "P / L rows staged (loads issued and written to LDS)", 2: "G pooled", 3: "barrier", 4: "gate mixes, channels 0-127",
             5: "L and G mixes", 6: "epilogue", 8: "gate mixes, channels 128-255", 9: "L and G mixes", 10: "epilogue"}
    prev = 0
    print("slot  phase                                                   median end (cycles after the workgroup's start)   phase cycles")
    last = None
    for slot in (1, 2, 3, 4, 5, 6, 8, 9, 10):
        end = (st[:, :, slot] - t0).median().item()
        print(f"{slot:4d}  {names[slot]:55s} {end:12.0f} {end - (last or 0):12.0f}")
        last = end
    print(f"spread of workgroup start times over the 64 stamped workgroups: {(t0.max() - t0.min()).item():.0f} cycles")


if __name__ == "__main__":
    main()
