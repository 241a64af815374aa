#!/usr/bin/env python3
"""Diagnostic: GDKVM.segment() at cfg2 as eager launches vs one captured HIP graph replayed (torch.cuda.CUDAGraph)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig

dev = torch.device("cuda")
torch.manual_seed(1)
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
frames = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()


def run(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    eager = lambda: model.segment(frames)[0]
    for _ in range(3):
        ref = eager()
    print(f"eager  {run(eager):.3f} ms", flush=True)
    static_in = frames.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            model.segment(static_in)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = model.segment(static_in)[0]
    g.replay()
    torch.cuda.synchronize()
    print("graph == eager:", torch.equal(static_out, ref), flush=True)
    print(f"graph  {run(g.replay):.3f} ms", flush=True)
    static_in.copy_(torch.rand_like(frames))
    g.replay()
    want = model.segment(static_in)[0]
    print("new input, graph == eager:", torch.equal(static_out, want))
