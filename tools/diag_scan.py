#!/usr/bin/env python3
"""Diagnostic only: builds libgdkvm_hip_diag.so (-DGDKVM_DIAG: s_memtime stamps in gdr_affine_scan_kernel) and prints where
one workgroup's cycles go per frame.  Shares of a diagnostic build, never a quoted run time
(cdna_hip_programming.md §7 'In-kernel stamps')."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")
SO = os.path.join(ROOT, "gpurun_out", "libgdkvm_hip_diag.so")


def main():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("gdr_prep.hip", "gdr_scan.hip", "gdkvm_api.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGDKVM_DIAG"] + sys.argv[1:] + [
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", SO] + srcs)
    print("variant:", sys.argv[1:])
    lib = ctypes.CDLL(SO)
    B, T, N, Hh, Dk, Dv = 16, 32, 49, 1, 64, 256
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
    dall = diag.cpu().reshape(T + 1, 8)
    pr = dall[T]
    print("prep (gdr_prepm_kernel), block 0 wave 0, ticks:")
    for i, n in enumerate(["entry loads issued + norms/gates (phase 0)", "Gram blocks (phase 1)", "T_II forward substitution (phase 2)",
                           "back substitution on Kn tile (phase 3)", "P tiles (phase 4a)", "G tiles (phase 4b)"]):
        print(f"  {n:48s} {int(pr[i + 1] - pr[i]):8d}")
    print(f"  total {int(pr[6] - pr[0])}")
    d8 = dall[:T]
    d = d8[:, :4]
    seg = torch.stack([d8[:, 4] - d8[:, 0], d8[:, 5] - d8[:, 4], d8[:, 6] - d8[:, 5], d8[:, 1] - d8[:, 6], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]], 1).float()
    names = ["S term images: 6 ds_read_b128 (waited)", "next frame's P/G/a out of the ring (8 reads, waited)", "gate (sigmoid of alpha_t)",
             "12 bf16 MFMA + a*acc + G", "publish S (split3 + 3 ds_write_b64, waited)", "barrier"]
    print("s_memtime ticks per frame (median over frames 2..T-1), block 0 wave 0 (a state wave)")
    for i, n in enumerate(names):
        print(f"  {n:48s} {seg[2:, i].median().item():8.0f}  (min {seg[2:, i].min().item():.0f} max {seg[2:, i].max().item():.0f})")
    print("  per-frame totals:", [int(x) for x in (d[:, 3] - d[:, 0]).tolist()])
    print("  gaps between frames:", [int(x) for x in (d[1:, 0] - d[:-1, 3]).tolist()])
    print(f"  frame total {(d[2:, 3] - d[2:, 0]).float().median().item():.0f}; whole scan {(d[-1, 3] - d[0, 0]).item()} ticks")


if __name__ == "__main__":
    main()
