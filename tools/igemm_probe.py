#!/usr/bin/env python3
"""Diagnostic: the four strided layers of an EchoNet forward (512 frames) on the implicit-GEMM kernel vs the framework convolution +
gdkvm_bias_act, and a stride-1 layer both hand-written kernels serve."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from tools.config_sweep import ev_time

dev = torch.device("cuda")
cl = dict(memory_format=torch.channels_last)
for name, n, c, h, k, rs, st, pad in (("3x3/2  64->128 @28", 512, 64, 28, 128, 3, 2, 1), ("3x3/2 128->256 @14", 512, 128, 14, 256, 3, 2, 1),
                                      ("1x1/2  64->128 @28", 512, 64, 28, 128, 1, 2, 0), ("1x1/2 128->256 @14", 512, 128, 14, 256, 1, 2, 0),
                                      ("3x3/1 128->128 @14", 512, 128, 14, 128, 3, 1, 1), ("3x3/1 256->256 @7", 512, 256, 7, 256, 3, 1, 1)):
    x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(**cl)
    w = (torch.randn(k, c, rs, rs, device=dev) / (rs * rs * c) ** 0.5).bfloat16().contiguous(**cl)
    b = torch.randn(k, device=dev)
    pk = ops.conv_igemm_pack_weights(w)
    t_ig = ev_time(lambda: ops.conv_bias_act(x, w, b, None, st, pad, True, ops.CONV_KERNEL_IGEMM, pk))
    def lib():
        y = F.conv2d(x, w, None, st, pad)
        return ops.bias_act_(y, b, None, True)
    t_lib = ev_time(lib)
    ho = (h + 2 * pad - rs) // st + 1
    gf = 2.0 * n * ho * ho * k * c * rs * rs / 1e9
    extra = ""
    if rs == 3 and st == 1:
        extra = f"   3x3/1/1 kernel {ev_time(lambda: ops.conv_bias_act(x, w, b, None, 1, 1, True, 5, ops.conv3x3_pack_weights(w))):6.1f} us"
    print(f"{name}: implicit GEMM {t_ig:6.1f} us ({gf / t_ig * 1e3:6.0f} TFLOP/s)   framework conv + bias_act {t_lib:6.1f} us{extra}", flush=True)
