#!/usr/bin/env python3
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tools.conv_probe import LAYERS, ev  # noqa: E402

lib = ctypes.CDLL(os.path.join(HERE, "libcksweep.so"))
names = open(os.path.join(HERE, "built.txt")).read().split()
N = 512
cl = dict(memory_format=torch.channels_last)
for lname, C, H, K, R, st, with_res in LAYERS:
    if R != 3:
        continue
    pad = 1
    x = torch.randn(N, C, H, H, device="cuda").relu().bfloat16().contiguous(**cl)
    w = (torch.randn(K, C, R, R, device="cuda") / (C * 9) ** 0.5).bfloat16().contiguous(**cl)
    b = torch.randn(K, device="cuda")
    Ho = (H + 2 - 3) // st + 1
    y = torch.empty(N, K, Ho, Ho, device="cuda", dtype=torch.bfloat16).contiguous(**cl)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, st, pad).relu()
    s = torch.cuda.current_stream().cuda_stream
    line = f"{lname:24s}"
    for nm in names:
        fn = getattr(lib, nm)
        fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 10 + [ctypes.c_void_p]
        call = lambda: fn(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), N, C, H, H, K, 3, 3, st, pad, 1, s)
        rc = call()
        if rc:
            line += f" {nm.split('_')[0]}  n/a "
            continue
        torch.cuda.synchronize()
        err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
        line += f" {nm.split('_')[0]} {ev(call):6.1f}" + ("" if err < 1e-2 else "!")
    print(line, flush=True)
print(" ".join(names))
