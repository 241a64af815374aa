#!/usr/bin/env python3
"""Diagnostic only: libgdkvm_hip_diag.so (-DGDKVM_DIAG: s_memtime stamps at the phase boundaries of gdr_prepm_kernel) and where
one workgroup's time goes.  Shares of a diagnostic build, never a quoted run time (cdna_hip_programming.md §7 'In-kernel stamps').
  python tools/diag_scan.py build [-D...]   (here: cross-compiles into tools/_abl/)
  python tools/diag_scan.py run             (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")
SO = os.path.join(ROOT, "tools", "_abl", "libgdkvm_hip_diag.so")


def build(extra):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("gdr_prep.hip", "gdr_scan.hip", "gdr_train.hip", "gdr_scan_bwd.hip", "gdr_readout_train.hip", "gdkvm_api.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGDKVM_DIAG"] + extra + [
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", SO] + srcs)


def run():
    import torch
    lib = ctypes.CDLL(SO)
    B, T, N, Hh, Dk, Dv = 16, 32, 49, 1, 64, 256
    if len(sys.argv) > 4:                                  # run B T N  (frames of more than 64 tokens: GDKVM_PREP_FUSE=0/1 picks the path;
        B, T, N = (int(x) for x in sys.argv[2:5])          #  the fused walk re-stamps per chunk, so the last chunk's phases are shown)
    dev = torch.device("cuda")
    q, k = (torch.randn(B, T, N, Hh, Dk, device=dev).bfloat16() for _ in range(2))
    v = torch.randn(B, T, N, Hh, Dv, device=dev).bfloat16()
    al = 2 + torch.randn(B, T, Hh, device=dev); be = torch.randn(B, T, N, Hh, device=dev)
    lib.gdkvm_scan_workspace_bytes.restype = ctypes.c_size_t
    wsb = lib.gdkvm_scan_workspace_bytes(B, T, Hh, N, Dk, Dv)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=torch.bfloat16); s = torch.empty(B, Hh, Dk, Dv, device=dev)
    diag = torch.zeros((T + 1) * 8, dtype=torch.int64, device=dev)
    lib.gdkvm_diag_set_buffer(ctypes.c_void_p(diag.data_ptr()))
    vp = ctypes.c_void_p
    lib.gdkvm_scan_fwd.argtypes = [vp] * 10 + [ctypes.c_size_t] + [ctypes.c_int] * 9 + [vp]
    for _ in range(3):
        rc = lib.gdkvm_scan_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), al.data_ptr(), be.data_ptr(), None, r.data_ptr(),
                                s.data_ptr(), None, ws.data_ptr(), wsb, B, T, Hh, N, Dk, Dv, 1, 2, 3, None)
        assert rc == 0
        torch.cuda.synchronize()
    pr = diag.cpu().reshape(T + 1, 8)[T]
    print("gdr_prepm_kernel, block 0 wave 0, s_memtime ticks (100 MHz: 10 ns each):")
    for i, n in enumerate(["entry loads issued + norms/gates (phase 0)", "Gram blocks (phase 1)", "T_II forward substitution (phase 2)",
                           "back substitution on Kn tile (phase 3)", "P tiles (phase 4a)", "G tiles (phase 4b)"]):
        print(f"  {n:48s} {int(pr[i + 1] - pr[i]):8d}")
    print(f"  total {int(pr[6] - pr[0])}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run()
