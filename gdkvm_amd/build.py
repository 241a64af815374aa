"""Builds gdkvm_amd/libgdkvm_hip.so (gfx950 only) in-tree with hipcc.  `python -m gdkvm_amd.build`."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
SO = os.path.join(PKG, "libgdkvm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
# per-file extras.  gdr_scan.hip: the chain wave's re-split runs beside MFMAs, where a packed fp32 instruction (v_pk_fma_f32 ...)
# costs more issue time than the two scalar ones it replaces (MI355X_MICROARCH.md, cycle constants) -- keep the SLP vectoriser off
EXTRA_FLAGS = {"gdr_scan.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


# the files that define the kernels of one gdkvm_scan_fwd (prep + compose + apply + read-out): a committed PMC summary of
# these kernels (profiles/*_pmc_*.csv) is only quoted by bench.py while this hash still matches the sources it was measured on
SCAN_SOURCES = ("gdr_prep.hip", "gdr_scan.hip", "gdr_device.hpp", "gdr_ws.hpp", "gdkvm_common.hpp")


def source_hash(names=SCAN_SOURCES) -> str:
    """sha256 (first 16 hex digits) over the named csrc/ files, in the given order."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not os.path.exists(SO):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(ROOT, "include", "gdkvm.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip under csrc/ into one shared library; objects are cached under csrc/_obj."""
    if not force and not _stale():
        return SO
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_m = max(os.path.getmtime(p) for p in glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(ROOT, "include", "gdkvm.h")])
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
