#!/usr/bin/env python3
"""A/B of two builds of the library on ONE GPU box: times the cfg2 inference forward with each .so in turn (a fresh process per
library, alternating), because forward times differ by a few per cent between boxes.  usage: ab_forward.py a.so b.so [rounds=3]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(so):
    import torch
    from gdkvm_amd import ops
    ops._SO = so
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    torch.manual_seed(0)
    dev = torch.device("cuda")
    model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    frames = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()
    with torch.no_grad():
        for _ in range(10):
            model.segment(frames)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            model.segment(frames)
        b.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(so):28s} forward {a.elapsed_time(b) / 40:.4f} ms", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "one":
        one(sys.argv[2])
    else:
        for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
            for so in sys.argv[1:3]:
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", os.path.abspath(so)])
