#!/usr/bin/env python3
"""Diagnostic only: builds tools/experiments/conv3x3_c64_instrumented.hip -- the 64 -> 64 convolution kernel with ablation hooks
(-DCONV_ABL_NOMFMA | _NODMA | _NOSTORE | _NOLDS), a grid override (-DCONV_GRID=n) and s_memtime stamps (-DCONV_DIAG) -- into
gpurun_out/ and times it at the layer1 shape (512 x 28 x 28 x 64).  Ablated builds compute wrong results by design; the
timings say which part of a tile bounds the kernel ("a+b" on the command line = both flags in one build).  The instrumented
copy also carries the experiments that did NOT pay (contiguous tile ranges, constants in LDS, inline-asm operand reads with
hand-placed waits, three-stage operand prefetch): 42 us against the product kernel's 40 us.  Never part of the product."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STUB = r'''
#include <hip/hip_runtime.h>
#include <cstdarg>
int gdkvm_fail(int code, const char*, ...) { return code; }
int gdkvm_check_device(void) { return 0; }
int gdkvm_conv3x3_c64_launch(const void*, const void*, const float*, const void*, void*, int, int, int, int, hipStream_t);
extern "C" int run(const void* x, const void* w, const float* b, const void* r, void* y, int N, int H, int W, int relu, void* st)
{ return gdkvm_conv3x3_c64_launch(x, w, b, r, y, N, H, W, relu, static_cast<hipStream_t>(st)); }
'''


def main():
    from tools.conv_probe import ev  # noqa
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    stub = os.path.join(out_dir, "abl_conv_stub.hip")
    open(stub, "w").write(STUB)
    variants = [[]] + [f.split("+") for f in sys.argv[1:]]          # "a+b" = both flags in one build
    N, H = 512, 28
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, 64, H, H, device="cuda").relu().bfloat16().contiguous(**cl)
    w = (torch.randn(64, 64, 3, 3, device="cuda") / 24).bfloat16().contiguous(**cl)
    b = torch.randn(64, device="cuda")
    r = torch.randn(N, 64, H, H, device="cuda").bfloat16().contiguous(**cl)
    y = torch.empty_like(x)
    for flags in variants:
        so = os.path.join(out_dir, "libabl_conv_" + "_".join(f.replace("-D", "") for f in flags) + ".so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + flags +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gdkvm_amd", "csrc"), "-o", so,
                               os.path.join(ROOT, "tools", "experiments", "conv3x3_c64_instrumented.hip"), stub])
        lib = ctypes.CDLL(so)
        lib.run.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        s = torch.cuda.current_stream().cuda_stream
        t0 = ev(lambda: lib.run(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), N, H, H, 1, s))
        t1 = ev(lambda: lib.run(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), y.data_ptr(), N, H, H, 1, s))
        print(f"{' '.join(flags) or 'baseline':50s} {t0:6.1f} us   with residual {t1:6.1f} us", flush=True)
        if "-DCONV_DIAG" in flags:                       # stamps of workgroup 0 (s_memtime ticks, 100 MHz): per wave and tile
            lib.run(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), N, H, H, 1, s)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (4 * 64 * 8))()
            lib.conv_diag_read(buf)
            names = ["k-loop", "barrier", "fetch", "epilogue", "to next"]
            for wv in range(4):
                for t in range(7):
                    st = [buf[(wv * 8 + t) * 8 + k] for k in range(5)]
                    nxt = buf[(wv * 8 + t + 1) * 8] if t < 6 else st[4]
                    d = [st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], nxt - st[4]]
                    print(f"  wave {wv} tile {t}: " + "  ".join(f"{n} {v * 10:5d} ns" for n, v in zip(names, d)) + f"   (start {(st[0] - buf[0]) * 10} ns)")


if __name__ == "__main__":
    main()
