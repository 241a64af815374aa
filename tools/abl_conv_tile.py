#!/usr/bin/env python3
"""Diagnostic only: ablation variants of conv3x3_tile.hip (-DCT_ABL_*: wrong results by design) and their timings at two cfg2
layers.  Says which of LDS reads / weight loads / band DMA / MFMAs bounds a k-step.
  python tools/abl_conv_tile.py build      (here; into tools/_abl/)      python tools/abl_conv_tile.py run    (GPU box)"""
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
    "weights_one_address": ["-DCT_ABL_NOW"],
    "no_weight_loads": ["-DCT_ABL_NOWLOAD"],
    "no_lds_reads": ["-DCT_ABL_NOLDS"],
    "no_band_dma": ["-DCT_ABL_NODMA"],
    "no_mfma": ["-DCT_ABL_NOMFMA"],
    "no_lds_no_wload": ["-DCT_ABL_NOLDS", "-DCT_ABL_NOWLOAD"],
    "no_lds_no_wload_no_dma": ["-DCT_ABL_NOLDS", "-DCT_ABL_NOWLOAD", "-DCT_ABL_NODMA"],
    "no_epilogue_store": ["-DCT_ABL_NOEPI"],
    "stamps": ["-DCT_DIAG"],
    "pix144": ["-DCT_PIX_BYTES=144"],
    "pix176": ["-DCT_PIX_BYTES=176"],
    "pix208": ["-DCT_PIX_BYTES=208"],
}


def build(only):
    os.makedirs(OUT, exist_ok=True)
    others = [o for o in sorted(glob.glob(os.path.join(CSRC, "_obj", "*.o"))) if not o.endswith("conv3x3_tile.o")]
    for name, flags in VARIANTS.items():
        if only and name not in only:
            continue
        obj, so = os.path.join(OUT, "ct_" + name + ".o"), os.path.join(OUT, "ct_" + name + ".so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + flags +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, "conv3x3_tile.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
        os.remove(obj)
        print("built", so, flush=True)


def c64stamps():
    """conv3x3_c64.hip built with -DCV_DIAG: where a tile's time goes (workgroup 0, its first tiles)."""
    import ctypes
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, "c64_stamps.so")
    lib = ops.load()
    x = torch.randn(512, 64, 28, 28, device="cuda").relu().bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device="cuda") / 24.0).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(64, device="cuda")
    buf = torch.zeros(8 * 4 * 8, dtype=torch.int64, device="cuda")
    lib.gdkvm_cv_diag_buffer.argtypes = [ctypes.c_void_p]
    lib.gdkvm_cv_diag_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        ops.conv_bias_act(x, w, b, None, 1, 1, True, 4)
    torch.cuda.synchronize()
    t = buf.cpu().reshape(8, 4, 8)
    print(f"64->64@28: entry -> first tile {int(t[0, 0, 0] - t[0, 0, 6])} cycles; per tile and wave: MFMAs | barrier | fetch issue | epilogue | (tile)")
    for c in range(7):
        for wv in range(4):
            r = t[c, wv]
            print(f"  tile {c} wave {wv}: {int(r[1] - r[0]):6d} | {int(r[2] - r[1]):6d} | {int(r[3] - r[2]):6d} | {int(r[4] - r[3]):6d} | ({int(r[4] - r[0]):6d})")


def build_c64():
    os.makedirs(OUT, exist_ok=True)
    others = [o for o in sorted(glob.glob(os.path.join(CSRC, "_obj", "*.o"))) if not o.endswith("conv3x3_c64.o")]
    obj, so = os.path.join(OUT, "c64_stamps.o"), os.path.join(OUT, "c64_stamps.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-DCV_DIAG",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, "conv3x3_c64.hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    os.remove(obj)
    print("built", so, flush=True)


def stamps():
    import ctypes
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, "ct_stamps.so")
    lib = ops.load()
    C, H, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (128, 14, 128)
    x = torch.randn(512, C, H, H, device="cuda").relu().bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device="cuda") / (C * 9) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(K, device="cuda")
    buf = torch.zeros(8 * 8 * 8, dtype=torch.int64, device="cuda")
    lib.gdkvm_ct_diag_buffer.argtypes = [ctypes.c_void_p]
    lib.gdkvm_ct_diag_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        ops.conv_bias_act(x, w, b, None, 1, 1, True, 5, ops.conv3x3_pack_weights(w))
    torch.cuda.synchronize()
    t = buf.cpu().reshape(8, 8, 8)                      # [chunk][wave][slot]
    t0 = t[0, :, 0].min()
    print(f"{C}->{K}@{H}: s_memtime cycles, workgroup 0; per chunk and wave: arrive | vmcnt wait | barrier wait | setup | trip 0 | trips 1-2 | (chunk total)")
    print(f"  kernel entry -> first chunk: {int(t[0, 0, 0] - t[0, 0, 6])} cycles (wave 0), {int(t[0, 4, 0] - t[0, 4, 6])} (wave 4)")
    for c in range(8):
        if t[c, 0, 5] == 0:
            break
        if t[c, 0, 7]:
            print(f"  tile ends with chunk {c}: epilogue {int(t[c, 0, 7] - t[c, 0, 5])} cycles (wave 0), {int(t[c, 4, 7] - t[c, 4, 5])} (wave 4); since entry {int(t[c, 0, 7] - t[0, 0, 6])}")
        for wv in range(8):
            r = t[c, wv]
            print(f"  chunk {c} wave {wv}: at {int(r[0] - t0):7d} | {int(r[1] - r[0]):6d} | {int(r[2] - r[1]):6d} | {int(r[3] - r[2]):6d} | {int(r[4] - r[3]):6d} | {int(r[5] - r[4]):6d} | ({int(r[5] - r[0]):6d})")


def run_one(name):
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, "ct_" + name + ".so")
    from tools.conv_probe import ev
    out = []
    for (C, H, K) in [(128, 14, 128), (192, 28, 64), (384, 14, 128)]:
        x = torch.randn(512, C, H, H, device="cuda").relu().bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 3, 3, device="cuda") / (C * 9) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
        b = torch.randn(K, device="cuda")
        pk = ops.conv3x3_pack_weights(w)
        out.append(f"{C}->{K}@{H}: {ev(lambda: ops.conv_bias_act(x, w, b, None, 1, 1, True, 5, pk)):6.1f} us")
    print(f"{name:26s} " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "build_c64":
        build_c64()
    elif sys.argv[1] == "c64stamps":
        c64stamps()
    elif sys.argv[1] == "stamps":
        stamps()
    elif sys.argv[1] == "run":
        for name in (sys.argv[2:] or list(VARIANTS)):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", name])
    else:
        run_one(sys.argv[2])
