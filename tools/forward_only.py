#!/usr/bin/env python3
"""Runs the cfg2 inference forward (fused build, bf16, 16 clips x 32 frames of 112x112) a few times and nothing else: the
target of the rocprofv3 --pmc passes whose per-kernel summaries are profiles/*_forward_cfg2_pmc_*.csv.
usage: python tools/forward_only.py [iterations=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402


def main():
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.manual_seed(0)
    dev = torch.device("cuda")
    model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    frames = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()
    with torch.no_grad():
        for _ in range(it):
            mask, _ = model.segment(frames)
    torch.cuda.synchronize()
    print("forward x", it, "mask sum", int(mask.sum()))


if __name__ == "__main__":
    main()
