#!/usr/bin/env python3
"""Diagnostic only: ablation builds of conv_igemm.hip (wrong results by design) and their timings on the strided layers.
  python tools/abl_igemm.py build   (here)      python tools/abl_igemm.py run   (GPU box)"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_abl")
VARIANTS = {"ig_base": [], "ig_samepix": ["-DIG_ABL_SAMEPIX"], "ig_samew": ["-DIG_ABL_SAMEW"], "ig_both": ["-DIG_ABL_SAMEPIX", "-DIG_ABL_SAMEW"],
            "ig_diag": ["-DIG_DIAG"], "ig_wn2": ["-DIG_WAVES_N=2"], "ig_wn1": ["-DIG_WAVES_N=1"]}

if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    others = [o for o in sorted(glob.glob(os.path.join(CSRC, "_obj", "*.o"))) if not o.endswith("conv_igemm.o")]
    for name, flags in VARIANTS.items():
        obj, so = os.path.join(OUT, name + ".o"), os.path.join(OUT, name + ".so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + flags +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, "conv_igemm.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
        os.remove(obj)
elif sys.argv[1] == "run":
    for name in VARIANTS:
        if name != "ig_diag":
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", name])
elif sys.argv[1] == "stamps":
    # s_memtime stamps of workgroup (0, 0), wave 0 (shares of a diagnostic build, never a quoted run time)
    import ctypes
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, "ig_diag.so")
    lib = ops.load()
    diag = torch.zeros(16 * 8, dtype=torch.int64, device="cuda")
    ctypes.CDLL(ops._SO).gdkvm_ig_diag_buffer(ctypes.c_void_p(diag.data_ptr()))
    cl = dict(memory_format=torch.channels_last)
    n, c, h, k, rs, st, pad = 512, 64, 28, 128, 3, 2, 1
    x = torch.randn(n, c, h, h, device="cuda").bfloat16().contiguous(**cl)
    w = (torch.randn(k, c, rs, rs, device="cuda") / (rs * rs * c) ** 0.5).bfloat16().contiguous(**cl)
    b = torch.randn(k, device="cuda")
    pk = ops.conv_igemm_pack_weights(w)
    for _ in range(3):
        ops.conv_bias_act(x, w, b, None, st, pad, True, ops.CONV_KERNEL_IGEMM, pk)
    torch.cuda.synchronize()
    d = diag.cpu().reshape(16, 8)
    print("prologue (gathers, weights, first put, barrier):", int(d[15, 1] - d[15, 0]), " all steps:", int(d[15, 2] - d[15, 1]), " epilogue:", int(d[15, 3] - d[15, 2]))
    print("step   reads+MFMAs   put(wait)   issue loads   barrier")
    for l in range(9):
        print(f"{l:4d} {int(d[l,1]-d[l,0]):12d} {int(d[l,2]-d[l,1]):11d} {int(d[l,3]-d[l,2]):13d} {int(d[l,4]-d[l,3]):9d}")
else:
    import torch
    from gdkvm_amd import ops
    ops._SO = os.path.join(OUT, sys.argv[2] + ".so")
    from tools.config_sweep import ev_time
    cl = dict(memory_format=torch.channels_last)
    out = []
    for n, c, h, k, rs, st, pad in ((512, 64, 28, 128, 3, 2, 1), (512, 128, 14, 256, 3, 2, 1), (512, 128, 14, 128, 3, 1, 1)):
        x = torch.randn(n, c, h, h, device="cuda").bfloat16().contiguous(**cl)
        w = (torch.randn(k, c, rs, rs, device="cuda") / (rs * rs * c) ** 0.5).bfloat16().contiguous(**cl)
        b = torch.randn(k, device="cuda")
        pk = ops.conv_igemm_pack_weights(w)
        out.append(f"{ev_time(lambda: ops.conv_bias_act(x, w, b, None, st, pad, True, ops.CONV_KERNEL_IGEMM, pk)):6.1f}")
    print(f"{sys.argv[2]:12s} " + " ".join(out) + " us", flush=True)
