#!/usr/bin/env python3
"""Times ops.conv_wgrad_strided on the four strided layers of the training step (512 frames): us per call.
    [GDKVM_CW_WGS=<target workgroups>] python tools/wgrad_strided_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gdkvm_amd import ops  # noqa: E402

ops.require_native()
cl = torch.channels_last
for (n, c, k, h, r, st, pad) in ((512, 64, 128, 28, 3, 2, 1), (512, 64, 128, 28, 1, 2, 0), (512, 128, 256, 14, 3, 2, 1), (512, 128, 256, 14, 1, 2, 0)):
    x = torch.randn(n, c, h, h, device="cuda").bfloat16().contiguous(memory_format=cl)
    ho = (h + 2 * pad - r) // st + 1
    dy = torch.randn(n, k, ho, ho, device="cuda").bfloat16().contiguous(memory_format=cl)
    like = torch.empty(k, c, r, r, device="cuda").contiguous(memory_format=cl)
    for _ in range(3):
        ops.conv_wgrad_strided(x, dy, like, st, pad)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); ops.conv_wgrad_strided(x, dy, like, st, pad); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{c:4d} -> {k:4d} {r}x{r}/{st} @ {h:2d}: {1e3 * ms[len(ms) // 2]:7.1f} us  (GDKVM_CW_WGS={os.environ.get('GDKVM_CW_WGS', 'default')})", flush=True)
