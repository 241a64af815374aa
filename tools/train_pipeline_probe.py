#!/usr/bin/env python3
"""The graphed training step (configs[3] per-GPU shape) fed from pinned host batches through DevicePrefetcher: float32 frames + int64 labels
(128 MB per batch) against uint8 frames + uint8 labels (26 MB, cast on the copy stream), against resident inputs.  python3 tools/train_pipeline_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402
from gdkvm_amd.pipeline import DevicePrefetcher  # noqa: E402
from gdkvm_amd.train import GraphedTrainStep  # noqa: E402


def main():
    dev = torch.device("cuda")
    torch.manual_seed(3)
    model = GDKVM(GDKVMConfig()).train().to(dev).to(memory_format=torch.channels_last)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)
    u8 = (torch.rand(16, 32, 3, 112, 112) * 255).round().to(torch.uint8)
    frames = u8.to(dev).float().mul_(1 / 255)
    target = (torch.rand(16, 32, 112, 112) > 0.5).long().to(dev)
    g = GraphedTrainStep(model, opt, frames, target, torch.bfloat16, warmup=2)
    for _ in range(3):
        g(frames, target)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        g(frames, target)
    torch.cuda.synchronize()
    print(f"resident inputs: {1e2 * (time.perf_counter() - t0):.3f} ms per step", flush=True)
    feeds = {"float32 frames + int64 labels (128 MB)": [(frames.cpu().pin_memory(), target.cpu().pin_memory()) for _ in range(2)],
             "uint8 frames + uint8 labels (26 MB)": [(u8.clone().pin_memory(), target.cpu().to(torch.uint8).pin_memory()) for _ in range(2)]}
    for name, hostb in feeds.items():
        for slots in (2, 3, 4):
            for rep in range(2):
                n = 0
                for f, t in DevicePrefetcher((hostb[i % 2] for i in range(24)), dev, slots=slots, frames_dtype=torch.float32):
                    if n == 4:
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                    g(f, t)
                    n += 1
                torch.cuda.synchronize()
                print(f"{name}, {slots} slots: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per step", flush=True)


if __name__ == "__main__":
    main()
