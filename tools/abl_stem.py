#!/usr/bin/env python3
"""Diagnostic only: builds csrc/stem_conv_pool.hip alone with ablation flags (-DSTEM_ABL_NOMFMA | _NODMA | _NOPOOL, -DSTEM_GRID=n)
into gpurun_out/ and times it at the cfg2 shape (512 x 56 x 56 x 16).  Ablated builds compute wrong results by design ("a+b" =
both flags in one build).  Never part of the product."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STUB = r'''
#include <cstdarg>
int gdkvm_fail(int code, const char*, ...) { return code; }
int gdkvm_check_device(void) { return 0; }
'''


def main():
    from tools.conv_probe import ev  # noqa
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    stub = os.path.join(out_dir, "abl_stem_stub.hip")
    open(stub, "w").write(STUB)
    cl = dict(memory_format=torch.channels_last)
    xs = torch.randn(512, 16, 56, 56, device="cuda").bfloat16().contiguous(**cl)
    w = (torch.randn(64, 16, 4, 4, device="cuda") / 16).bfloat16().contiguous(**cl)
    b = torch.randn(64, device="cuda")
    y = torch.empty(512, 64, 28, 28, device="cuda", dtype=torch.bfloat16).contiguous(**cl)
    for flags in [[]] + [f.split("+") for f in sys.argv[1:]]:
        so = os.path.join(out_dir, "libabl_stem_" + "_".join(f.replace("-D", "").replace("=", "") for f in flags) + ".so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + flags +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gdkvm_amd", "csrc"), "-o", so,
                               os.path.join(ROOT, "gdkvm_amd", "csrc", "stem_conv_pool.hip"), stub])
        lib = ctypes.CDLL(so)
        lib.gdkvm_stem_conv_pool.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        s = torch.cuda.current_stream().cuda_stream
        t = ev(lambda: lib.gdkvm_stem_conv_pool(xs.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 512, 56, 56, 1, s))
        print(f"{' '.join(flags) or 'baseline':50s} {t:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
