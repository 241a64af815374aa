#!/usr/bin/env python3
"""Eager against one hipGraph replay per step for the cfg2 inference forward (fused build, bf16): is the step bound by the host's ~25 launch calls
on this box?   python3 tools/graph_forward_probe.py [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)
dev = torch.device("cuda")
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
frames = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()


def timed(f):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


with torch.no_grad():
    eager = timed(lambda: model.segment(frames))
    ref = model.segment(frames)[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            model.segment(frames)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model.segment(frames)[0]
    graph = timed(g.replay)
    print(f"eager {eager:.4f} ms   graph replay {graph:.4f} ms   masks equal: {bool(torch.equal(out, ref))}")
    eager2 = timed(lambda: model.segment(frames))
    print(f"eager again {eager2:.4f} ms")
