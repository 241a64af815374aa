#!/usr/bin/env python3
"""Per-layer timing of the hand-written convolution weight gradient (gdkvm_conv3x3_wgrad) against the framework's, cfg4 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402
from tools.conv_probe import ev  # noqa: E402

torch.backends.cudnn.benchmark = True
for name, C, H, K in [("64->64 @28", 64, 28, 64), ("128->128 @14", 128, 14, 128), ("256->256 @7", 256, 7, 256), ("384->128 @14", 384, 14, 128),
                      ("192->64 @28", 192, 28, 64)]:
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(512, C, H, H, device="cuda").bfloat16().contiguous(**cl)
    dy = torch.randn(512, K, H, H, device="cuda").bfloat16().contiguous(**cl)
    w = torch.randn(K, C, 3, 3, device="cuda").bfloat16().contiguous(**cl)
    lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    print(f"{name:14s} framework {ev(lib):7.1f} us   hand-written {ev(lambda: ops.conv3x3_wgrad(x, dy)):7.1f} us", flush=True)
