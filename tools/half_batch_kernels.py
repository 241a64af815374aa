#!/usr/bin/env python3
"""Runs the cfg2 forward on the whole batch and on half of it (eagerly, one stream) a few times: under rocprofv3 --kernel-trace --stats the two
sets of kernel durations show which kernels lose efficiency at half the batch (the two-stream forward runs half-batch kernels)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda")
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
frames = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
with torch.no_grad():
    for _ in range(12):
        model.segment(frames[:n])
torch.cuda.synchronize()
