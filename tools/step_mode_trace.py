#!/usr/bin/env python3
"""The per-frame step mode alone (one hipGraph replay of 16 clips x 32 frames, ONE stream inside) for rocprofv3 --kernel-trace: which kernels a frame's
~143 us are.   python3 tools/step_mode_trace.py [streams]"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda")
torch.manual_seed(1)
m = GDKVM(dataclasses.replace(GDKVMConfig(), mask_feedback=True)).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
x = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()
g = m.graphed_segment(x, streams=streams)
for _ in range(3):
    g(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    g(x)
b.record(); torch.cuda.synchronize()
print(f"step mode, {streams} stream(s) inside: {a.elapsed_time(b) / 5:.3f} ms per 16 x 32 frames", flush=True)
