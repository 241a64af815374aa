#!/usr/bin/env python3
"""Diagnostic only: times kpff_bf16_kernel at the cfg2 shape with one phase compiled out per variant
(-DKPFF_SKIP_GEMM / _EPI / _POOL), to see which phase the kernel's time sits in.  Outputs of the variants are wrong
by construction; only the timings matter (cdna_hip_programming.md §7 'Ablate')."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")


def run(flags):
    so = os.path.join(ROOT, "gpurun_out", "libkpff_diag_%s.so" % ("_".join(f[2:] for f in flags) or "base"))
    os.makedirs(os.path.dirname(so), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("kpff.hip", "gdkvm_api.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + flags +
                          ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", so] + srcs)
    lib = ctypes.CDLL(so)
    dev = torch.device("cuda")
    BT, N, Ck, Cv, Cp = 512, 49, 64, 256, 256
    g = torch.Generator(device=dev).manual_seed(0)
    L = torch.randn(BT, N, Ck, device=dev, generator=g).bfloat16()
    G = torch.randn(BT, N, Cv, device=dev, generator=g).bfloat16()
    P = torch.randn(BT, N, Cp, device=dev, generator=g).bfloat16()
    cin = Cp + Ck + Cv
    wa = torch.randn(2 * Cp, cin, device=dev, generator=g) / cin ** 0.5
    ba = torch.zeros(2 * Cp, device=dev)
    wl = torch.randn(Cp, Ck, device=dev, generator=g) / 8
    wg = torch.randn(Cp, Cv, device=dev, generator=g) / 16
    out = torch.empty(BT, N, Cp, device=dev, dtype=torch.bfloat16)
    lib.gdkvm_kpff_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(lib.gdkvm_kpff_workspace_bytes(Ck, Cv, Cp, 1), dtype=torch.uint8, device=dev)
    vp = ctypes.c_void_p
    lib.gdkvm_kpff_fwd.argtypes = [vp] * 9 + [ctypes.c_size_t] + [ctypes.c_int] * 7 + [vp]
    call = lambda: lib.gdkvm_kpff_fwd(L.data_ptr(), G.data_ptr(), P.data_ptr(), wa.data_ptr(), ba.data_ptr(), wl.data_ptr(),
                                      wg.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), BT, Ck, Cv, Cp, 7, 7, 1, None)
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record(); torch.cuda.synchronize()
    print(f"{' '.join(flags) or 'full kernel':60s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per call (incl. 5 us weight pack)")


if __name__ == "__main__":
    for fl in ([], ["-DKPFF_ABL_WSAME"], ["-DKPFF_SKIP_GEMM"], ["-DKPFF_SKIP_EPI"], ["-DKPFF_SKIP_POOL"], ["-DKPFF_SKIP_GEMM", "-DKPFF_SKIP_EPI"],
               ["-DKPFF_SKIP_GEMM", "-DKPFF_SKIP_EPI", "-DKPFF_SKIP_POOL"]):
        run(fl)
