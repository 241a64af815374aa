#!/usr/bin/env python3
"""Diagnostic only: ablation variants of the serial scan kernel (gdr_scan.hip compiled with -DGDKVM_ABL_* flags) and their
timings.  Ablated builds compute wrong results by design; their timings say which role of the kernel bounds a frame.
  python tools/abl_scan.py build            (here: cross-compiles every variant into tools/_abl/, which travels to the GPU box)
  python tools/abl_scan.py run              (GPU box: times scan_prep / scan_apply with each variant)
Never part of the product."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_abl")
VARIANTS = {
    "baseline": [],
    "r_honly": ["-DGDKVM_ABL_R_HONLY"],
    "r_nostore": ["-DGDKVM_ABL_R_NOSTORE"],
    "r_noprep": ["-DGDKVM_ABL_R_NOPREP"],
    "r_all": ["-DGDKVM_ABL_R_HONLY", "-DGDKVM_ABL_R_NOSTORE", "-DGDKVM_ABL_R_NOPREP"],
    # the deferred read-out of frames of more than 64 tokens (gdr_readout_kernel): `python tools/abl_scan.py ro`
    "ro_nostore": ["-DGDKVM_ABL_RO_NOSTORE"],
    "ro_noq": ["-DGDKVM_ABL_RO_NOQ"],
    "ro_nostore_noq": ["-DGDKVM_ABL_RO_NOSTORE", "-DGDKVM_ABL_RO_NOQ"],
    # the frame-parallel side (gdr_prep.hip is the file compiled with the flags): `python tools/abl_scan.py prep`
    "prep_vsame": ["-DGDKVM_ABL_PREP_VSAME"],
}


def run_prep(name):
    """scan_prep at cfg2 / cfg3 / cfg5"""
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, name + ".so")
    from tools.config_sweep import ev_time
    dev = torch.device("cuda")
    out = []
    torch.manual_seed(0)
    for (B, T, N) in [(16, 32, 49), (8, 20, 256), (2, 512, 256)]:
        Hh, Dk, Dv, dt = 1, 64, 256, torch.bfloat16
        q, k = (torch.randn(B, T, N, Hh, Dk, device=dev).to(dt) for _ in range(2))
        v = torch.randn(B, T, N, Hh, Dv, device=dev).to(dt)
        be = torch.randn(B, T, N, Hh, device=dev)
        ws = torch.empty(ops.scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=dev)
        tp = ev_time(lambda: ops.scan_prep(q, k, v, be, ws, flags=3))
        out.append(f"B={B} T={T} N={N}: prep {tp:.1f} us")
    print(f"{name:28s} " + " | ".join(out), flush=True)



def run_ro(name):
    """scan_apply with and without the read-out at the shapes whose frames exceed 64 tokens (cfg5, cfg3): the difference is the state-image dump
    + gdr_readout_kernel."""
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, name + ".so")
    from tools.config_sweep import ev_time
    dev = torch.device("cuda")
    out = []
    torch.manual_seed(0)
    for (B, T, N) in [(2, 512, 256), (8, 20, 256)]:
        Hh, Dk, Dv, dt = 1, 64, 256, torch.bfloat16
        q, k = (torch.randn(B, T, N, Hh, Dk, device=dev).to(dt) for _ in range(2))
        v = torch.randn(B, T, N, Hh, Dv, device=dev).to(dt)
        al = 2 + torch.randn(B, T, Hh, device=dev); be = torch.randn(B, T, N, Hh, device=dev)
        ws = torch.empty(ops.scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=dev)
        r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=dt); s = torch.empty(B, Hh, Dk, Dv, device=dev)
        ops.scan_prep(q, k, v, be, ws, flags=3)
        ta = ev_time(lambda: ops.scan_apply(q, al, ws, Dv, flags=3, out=r, state_out=s))
        tn = ev_time(lambda: ops.scan_apply(q, al, ws, Dv, flags=3, state_out=s, want_readout=False))
        out.append(f"B={B} T={T} N={N}: apply {ta:.1f} us, states only {tn:.1f}, read-out side {ta - tn:.1f}")
    print(f"{name:28s} " + " | ".join(out), flush=True)



def build(only=None):
    from gdkvm_amd.build import EXTRA_FLAGS
    os.makedirs(OUT, exist_ok=True)
    for name, flags in VARIANTS.items():
        if only and name not in only:
            continue
        src = "gdr_prep" if name.startswith("prep_") else "gdr_scan"
        others = [o for o in sorted(glob.glob(os.path.join(CSRC, "_obj", "*.o"))) if not o.endswith(src + ".o")]
        obj, so = os.path.join(OUT, name + ".o"), os.path.join(OUT, name + ".so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + EXTRA_FLAGS.get(src + ".hip", []) +
                              flags + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, src + ".hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
        os.remove(obj)
        print("built", so, flush=True)


def run_one(name):
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, name + ".so")
    from tools.config_sweep import ev_time
    dev = torch.device("cuda")
    out = []
    torch.manual_seed(0)
    ref_path = os.path.join(ROOT, "gpurun_out", "abl_ref.pt")
    for (B, T, N) in [(16, 32, 49), (16, 128, 49)]:
        Hh, Dk, Dv, dt = 1, 64, 256, torch.bfloat16
        q, k = (torch.randn(B, T, N, Hh, Dk, device=dev).to(dt) for _ in range(2))
        v = torch.randn(B, T, N, Hh, Dv, device=dev).to(dt)
        al = 2 + torch.randn(B, T, Hh, device=dev); be = torch.randn(B, T, N, Hh, device=dev)
        ws = torch.empty(ops.scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=dev)
        r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=dt); s = torch.empty(B, Hh, Dk, Dv, device=dev)
        ops.scan_prep(q, k, v, be, ws, flags=3)
        ops.scan_apply(q, al, ws, Dv, flags=3, out=r, state_out=s)
        if T == 32:                                   # same seeded inputs in every process: the baseline's result is the reference
            if name == "baseline":
                os.makedirs(os.path.dirname(ref_path), exist_ok=True)
                torch.save((r.cpu(), s.cpu()), ref_path)
            elif os.path.exists(ref_path):
                r0, s0 = torch.load(ref_path)
                out.append(f"[vs baseline: r {(r.cpu().float() - r0.float()).abs().max().item():.2e} s {(s.cpu() - s0).abs().max().item():.2e}]")
        tp = ev_time(lambda: ops.scan_prep(q, k, v, be, ws, flags=3))
        ta = ev_time(lambda: ops.scan_apply(q, al, ws, Dv, flags=3, out=r, state_out=s))
        tn = ev_time(lambda: ops.scan_apply(q, al, ws, Dv, flags=3, state_out=s, want_readout=False))
        out.append(f"T={T}: prep {tp:.1f} scan {ta:.1f} us ({ta / T * 1e3:.0f} ns/frame; states only {tn / T * 1e3:.0f})")
    print(f"{name:28s} " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "ro":
        for name in ["baseline"] + [v for v in VARIANTS if v.startswith("ro_")]:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one_ro", name])
    elif sys.argv[1] == "one_ro":
        run_ro(sys.argv[2])
    elif sys.argv[1] == "prep":
        for name in ["baseline"] + [v for v in VARIANTS if v.startswith("prep_")]:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one_prep", name])
    elif sys.argv[1] == "one_prep":
        run_prep(sys.argv[2])
    elif sys.argv[1] == "run":
        # one process per variant: the library is loaded once per process
        for name in (sys.argv[2:] or list(VARIANTS)):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", name])
    else:
        run_one(sys.argv[2])
