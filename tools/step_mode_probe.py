#!/usr/bin/env python3
"""Per-frame step mode (GDKVMConfig(mask_feedback=True)) at the cfg2 shape: one hipGraph of the 32-frame loop, the batch cut into 1 / 2 / 4 / 8
groups of clips on as many streams inside the graph (clips never interact; in step mode every kernel works on ONE frame per clip, so a
group's kernels are small and latency-bound: more groups = more of them in flight).   python3 tools/step_mode_probe.py [clips=16]"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig, GraphedSegment  # noqa: E402


def main():
    torch.manual_seed(1)
    dev = torch.device("cuda")
    m = GDKVM(GDKVMConfig(mask_feedback=True)).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    fr = torch.rand(B, 32, 3, 112, 112, device=dev).bfloat16()
    ref = m.segment(fr)[0].clone()
    for streams in ((1, 2, 4, 8, 16) if B == 16 else (1, 2)):
        g = GraphedSegment(m, fr.clone(), streams=streams)
        assert torch.equal(g(fr)[0], ref)
        for _ in range(5):
            g(fr)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            g(fr)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"clips={B:3d} streams={streams:2d}: {ms:.3f} ms per {B} x 32 frames = {B * 32 / ms * 1e3:,.0f} frames/s", flush=True)
        del g


if __name__ == "__main__":
    main()
