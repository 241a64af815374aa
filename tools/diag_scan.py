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
    nwg = B * T * Hh * ((N + 63) // 64)
    diag = torch.zeros((T + 1) * 8 + 2 * nwg, dtype=torch.int64, device=dev)
    lib.gdkvm_diag_set_buffer(ctypes.c_void_p(diag.data_ptr()))
    vp = ctypes.c_void_p
    lib.gdkvm_scan_fwd.argtypes = [vp] * 10 + [ctypes.c_size_t] + [ctypes.c_int] * 9 + [vp]
    if "--inner" in sys.argv:                              # prep alone, with the norms handed in (the product's path behind gdkvm_proj_gates)
        lib.gdkvm_scan_prep_normed.argtypes = [vp] * 6 + [ctypes.c_size_t] + [ctypes.c_int] * 9 + [vp]
        norms = torch.stack([1.0 / torch.sqrt(k.float().pow(2).sum(-1) + 1e-6), 1.0 / torch.sqrt(q.float().pow(2).sum(-1) + 1e-6)], -1).contiguous()
    for _ in range(3):
        if "--inner" in sys.argv:
            rc = lib.gdkvm_scan_prep_normed(q.data_ptr(), k.data_ptr(), v.data_ptr(), be.data_ptr(), norms.data_ptr(), ws.data_ptr(), wsb,
                                            B, T, Hh, N, Dk, Dv, 1, 2, 3, None)
            assert rc == 0
            torch.cuda.synchronize()
            continue
        rc = lib.gdkvm_scan_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), al.data_ptr(), be.data_ptr(), None, r.data_ptr(),
                                s.data_ptr(), None, ws.data_ptr(), wsb, B, T, Hh, N, Dk, Dv, 1, 2, 3, None)
        assert rc == 0
        torch.cuda.synchronize()
    if "--span" in sys.argv:                               # built with -DGDKVM_DIAG_SPAN: entry / exit time of every workgroup of one prep launch
        lib.gdkvm_scan_prep.argtypes = [vp] * 5 + [ctypes.c_size_t] + [ctypes.c_int] * 9 + [vp]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(4):
            ev[0].record()
            rc = lib.gdkvm_scan_prep(q.data_ptr(), k.data_ptr(), v.data_ptr(), be.data_ptr(), ws.data_ptr(), wsb, B, T, Hh, N, Dk, Dv, 1, 2, 3, None)
            ev[1].record()
            assert rc == 0
            torch.cuda.synchronize()
        span = diag.cpu()[(T + 1) * 8:].reshape(nwg, 2).double()
        st, en = span[:, 0].clone(), span[:, 1].clone()
        # the counter is not chip-wide: workgroups whose entry times lie within 2^17 ticks of each other share a counter domain, and each
        # domain is taken against its own first entry
        order = torch.argsort(st)
        dom, d = torch.zeros(nwg, dtype=torch.long), 0
        for a_, b_ in zip(order[:-1], order[1:]):
            if st[b_] - st[a_] > 2 ** 17:
                d += 1
            dom[b_] = d
        ends = []
        for x in range(d + 1):
            m = dom == x
            t0 = span[m, 0].min()
            st[m] -= t0; en[m] -= t0
            ends.append((int(m.sum()), int(en[m].max())))
        q_ = lambda x, p: float(x.quantile(p))
        print(f"{nwg} workgroups; events around the call: {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us")
        print(f"entry ticks after the first: median {q_(st, .5):.0f}  p90 {q_(st, .9):.0f}  max {float(st.max()):.0f}")
        print(f"life ticks: min {float((en - st).min()):.0f} median {q_(en - st, .5):.0f} max {float((en - st).max()):.0f}")
        print(f"last exit: {float(en.max()):.0f} ticks after its domain's first entry; (workgroups, last exit) per domain: {ends}")
        return
    if "--twice" in sys.argv:                              # built with -DGDKVM_DIAG_TWICE: prep alone, first and second pass over the same frame
        lib.gdkvm_scan_prep.argtypes = [vp] * 5 + [ctypes.c_size_t] + [ctypes.c_int] * 9 + [vp]
        for _ in range(3):
            rc = lib.gdkvm_scan_prep(q.data_ptr(), k.data_ptr(), v.data_ptr(), be.data_ptr(), ws.data_ptr(), wsb, B, T, Hh, N, Dk, Dv, 1, 2, 3, None)
            assert rc == 0
            torch.cuda.synchronize()
        rows = diag.cpu().reshape(-1)[:(T + 1) * 8].reshape(T + 1, 8)
        names = ["phase 0", "Gram", "T_II", "back subst", "P tiles", "G tiles"]
        print("gdr_prepm_kernel block 0 wave 0, ticks per phase: first pass | second pass over the same frame")
        for i, n in enumerate(names):
            print(f"  {n:12s} {int(rows[T][i + 1] - rows[T][i]):8d} {int(rows[T - 1][i + 1] - rows[T - 1][i]):8d}")
        print(f"  total        {int(rows[T][6] - rows[T][0]):8d} {int(rows[T - 1][6] - rows[T - 1][0]):8d}")
        return
    pr = diag.cpu().reshape(-1)[:(T + 1) * 8].reshape(T + 1, 8)[T]
    print("gdr_prepm_kernel, block 0 wave 0, s_memtime ticks (100 MHz: 10 ns each):")
    for i, n in enumerate(["entry loads issued + norms/gates (phase 0)", "Gram blocks (phase 1)", "T_II forward substitution (phase 2)",
                           "back substitution on Kn tile (phase 3)", "P tiles (phase 4a)", "G tiles (phase 4b)"]):
        print(f"  {n:48s} {int(pr[i + 1] - pr[i]):8d}")
    print(f"  total {int(pr[6] - pr[0])}")
    p2 = diag.cpu().reshape(-1)[:(T + 1) * 8].reshape(T + 1, 8)[T - 1]
    if "--inner" in sys.argv:                              # (prep alone: the scan's own stamps do not overwrite row T - 1)
        print(f"  inside phase 0, ticks after entry: loads requested {int(p2[0] - pr[0])}, key rows staged {int(p2[1] - pr[0])}, "
              f"norms / gates {int(p2[2] - pr[0])}, barrier passed {int(pr[1] - pr[0])}, Kn^T images built {int(p2[3] - pr[0])}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run()
