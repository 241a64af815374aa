#!/usr/bin/env python3
"""Diagnostic only: a library built with -DGDKVM_PIPE_STAMPS (wall-clock stamps inside the three kernels of the concurrent form of
gdkvm_scan_fwd, csrc/gdr_pipeline.hip) and the timeline they give: when each group of 12 frames was folded, scanned and read out.
  python tools/pipe_timeline.py build          (here: cross-compiles into tools/_abl/)
  python tools/pipe_timeline.py run B T N      (GPU box; GDKVM_SCAN_PIPE etc. as set by the caller)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")
SO = os.path.join(ROOT, "tools", "_abl", "libgdkvm_hip_pstamp.so")


def build(extra):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("gdr_prep.hip", "gdr_scan.hip", "gdr_pipeline.hip", "gdr_train.hip", "gdr_scan_bwd.hip", "gdr_readout_train.hip", "gdkvm_api.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-DGDKVM_PIPE_STAMPS"] + extra + [
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", SO] + srcs)


def run():
    import torch
    lib = ctypes.CDLL(SO)
    B, T, N = (int(x) for x in sys.argv[2:5])
    Hh, Dk, Dv, G = 1, 64, 256, 12
    dev = torch.device("cuda")
    q, k = (torch.randn(B, T, N, Hh, Dk, device=dev).bfloat16() for _ in range(2))
    v = torch.randn(B, T, N, Hh, Dv, device=dev).bfloat16()
    al = 2 + torch.randn(B, T, Hh, device=dev); be = torch.randn(B, T, N, Hh, device=dev)
    lib.gdkvm_scan_workspace_bytes.restype = ctypes.c_size_t
    wsb = lib.gdkvm_scan_workspace_bytes(B, T, Hh, N, Dk, Dv)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=torch.bfloat16); s = torch.empty(B, Hh, Dk, Dv, device=dev)
    FH, nsl, ngrp = B * T * Hh, Dv // 16, (T + G - 1) // G
    st_f = torch.zeros(FH * 2, dtype=torch.int64, device=dev)
    st_s = torch.zeros(B * Hh * nsl * 64, dtype=torch.int64, device=dev)
    st_r = torch.zeros(FH * 3, dtype=torch.int64, device=dev)
    vp = ctypes.c_void_p
    lib.gdkvm_pipe_set_stamps.argtypes = [vp, vp, vp]
    assert lib.gdkvm_pipe_set_stamps(st_f.data_ptr(), st_s.data_ptr(), st_r.data_ptr()) == 0
    lib.gdkvm_scan_fwd.argtypes = [vp] * 10 + [ctypes.c_size_t] + [ctypes.c_int] * 9 + [vp]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    # a kernel of another kind running beside the call on a second stream: "copy" (HBM traffic), "gemm" (MFMA, little traffic), "alu"
    contend = os.environ.get("PIPE_CONTEND", "")
    side = torch.cuda.Stream()
    big_a = torch.empty(64 << 20, dtype=torch.float32, device=dev); big_b = torch.empty_like(big_a)
    ga = torch.randn(4096, 4096, device=dev).bfloat16(); gb = torch.randn(4096, 4096, device=dev).bfloat16()
    small = torch.randn(1 << 20, device=dev)
    for it in range(3):
        if contend:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(8):
                    if contend == "copy":
                        big_b.copy_(big_a)
                    elif contend == "gemm":
                        torch.mm(ga, gb)
                    else:
                        for _ in range(8):
                            small = torch.sin(small)
        ev[0].record()
        rc = lib.gdkvm_scan_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), al.data_ptr(), be.data_ptr(), None, r.data_ptr(),
                                s.data_ptr(), None, ws.data_ptr(), wsb, B, T, Hh, N, Dk, Dv, 1, 2, 3, None)
        assert rc == 0
        ev[1].record()
        torch.cuda.synchronize()
    if os.environ.get("GDKVM_PIPE_STAGES"):
        sc = st_s.cpu().view(B, nsl, 64).double() / 100.0
        per = [(float(sc[:, :, 2 + g].max()) - float(sc[:, :, 1 + g].max())) for g in range(0, ngrp - 2)]
        print(f"{B}x{T}x{N} stages {os.environ['GDKVM_PIPE_STAGES']} dbg {os.environ.get('GDKVM_PIPE_DBG')} contend '{contend}': {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us by events; "
              f"recurrence {float(sc[..., 63].max()) - float(sc[..., 0].min()):.1f} us; us per group of 12 frames: median {sorted(per)[len(per) // 2]:.2f}, "
              f"min {min(per):.2f}, max {max(per):.2f}")
        return
    print(f"{B}x{T}x{N}: last call {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us by events (stamped build; GDKVM_SCAN_PIPE={os.environ.get('GDKVM_SCAN_PIPE')})")
    f = st_f.cpu().view(B, T, 2).double() / 100.0          # us
    sc = st_s.cpu().view(B, nsl, 64).double() / 100.0
    rd = st_r.cpu().view(B, T, 3).double() / 100.0
    t0 = min(float(f[..., 0][f[..., 0] > 0].min()) if (f[..., 0] > 0).any() else 1e30, float(sc[..., 0][sc[..., 0] > 0].min()))
    print(f"recurrence workgroups start {float(sc[..., 0].min()) - t0:.1f} .. {float(sc[..., 0].max()) - t0:.1f} us, end {float(sc[..., 63].max()) - t0:.1f} us;  "
          f"fold first start {float(f[..., 0].min()) - t0:.1f}, last end {float(f[..., 1].max()) - t0:.1f};  "
          f"read-out first start {float(rd[..., 0][rd[..., 0] > 0].min()) - t0:.1f}, last end {float(rd[..., 2].max()) - t0:.1f}")
    print("group   fold done   scan done (max over slices)   read-out: first start / counter seen (median) / last end")
    for g in range(ngrp):
        a, b = g * G, min(T, (g + 1) * G)
        fd = float(f[:, a:b, 1].max()) - t0
        sd = float(sc[:, :, 1 + min(g, 61)].max()) - t0 if g + 1 < ngrp else float(sc[:, :, 63].max()) - t0
        print(f"{g:5d} {fd:11.1f} {sd:11.1f}                    {float(rd[:, a:b, 0].min()) - t0:11.1f} {float(rd[:, a:b, 1].median()) - t0:11.1f} {float(rd[:, a:b, 2].max()) - t0:11.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run()
