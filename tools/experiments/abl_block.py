#!/usr/bin/env python3
"""Diagnostic only: conv3x3_block64.hip built with -DCB_DIAG (s_memtime stamps of workgroup 0's first tiles): where a tile's time goes.
  python tools/abl_block.py build   (here; into tools/_abl/)        python tools/abl_block.py run   (GPU box)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_abl")


def build():
    os.makedirs(OUT, exist_ok=True)
    others = [o for o in sorted(glob.glob(os.path.join(CSRC, "_obj", "*.o"))) if not o.endswith("conv3x3_block64.o")]
    obj, so = os.path.join(OUT, "cb_stamps.o"), os.path.join(OUT, "cb_stamps.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-DCB_DIAG",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(ROOT, "tools", "experiments", "conv3x3_block64.hip"), "-o", obj] + sys.argv[2:])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    os.remove(obj)
    print("built", so, flush=True)


def run():
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, "cb_stamps.so")
    lib = ops.load()
    cl = torch.channels_last
    x = torch.randn(512, 64, 28, 28, device="cuda").bfloat16().contiguous(memory_format=cl)
    w1, w2 = ((torch.randn(64, 64, 3, 3, device="cuda") / 24).bfloat16().contiguous(memory_format=cl) for _ in range(2))
    b1, b2 = torch.randn(64, device="cuda"), torch.randn(64, device="cuda")
    p1, p2 = ops.conv3x3_pack_weights(w1), ops.conv3x3_pack_weights(w2)
    buf = torch.zeros(8 * 4 * 8, dtype=torch.int64, device="cuda")
    lib.gdkvm_cb_diag_buffer.argtypes = [ctypes.c_void_p]
    lib.gdkvm_cb_diag_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        ops.conv_block_bias_act(x, p1, b1, p2, b2)
    torch.cuda.synchronize()
    t = buf.cpu().reshape(8, 4, 8)
    print("per tile and wave (s_memtime ticks, 100 MHz => x ~20 for cycles): conv1 MFMAs | epilogue 1 | barrier A + fetch issue | conv2 MFMAs | barrier B | epilogue 2 | (tile)")
    for c in range(8):
        for wv in range(4):
            r = t[c, wv]
            print(f"  tile {c} wave {wv}: " + " | ".join(f"{int(r[i + 1] - r[i]):6d}" for i in range(6)) + f" | ({int(r[6] - r[0]):6d})")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
