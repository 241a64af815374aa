#!/usr/bin/env python3
"""Probe: does capturing one cfg2 forward (model.segment) in a HIP graph pay?  Prints eager vs graph-replay ms/step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.backends.cudnn.benchmark = True
from gdkvm_amd.model import GDKVM, GDKVMConfig

dev = torch.device("cuda")
torch.manual_seed(1)
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
frames = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()

def step():
    with torch.no_grad():
        return model.segment(frames)[0]

for _ in range(6):
    ref = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"eager  {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
g.replay()
torch.cuda.synchronize()
print("graph == eager:", torch.equal(out, ref))
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"graph  {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")
