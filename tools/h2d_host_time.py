#!/usr/bin/env python3
"""Host time of an asynchronous host-to-device copy from pinned memory (does the call return before the copy is done?).  python3 tools/h2d_host_time.py"""
import time
import torch

dev = torch.device("cuda")
torch.zeros(1, device=dev)
for mb in (1, 19, 19, 64):
    h = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True)
    h.fill_(3)
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    for name, st in (("current stream", torch.cuda.current_stream()), ("side stream", side)):
        with torch.cuda.stream(st):
            for _ in range(3):
                d.copy_(h, non_blocking=True)
            torch.cuda.synchronize()
            calls = []
            t0 = time.perf_counter()
            for _ in range(8):
                c0 = time.perf_counter()
                d.copy_(h, non_blocking=True)
                calls.append(1e3 * (time.perf_counter() - c0))
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"{mb:3d} MB pinned -> device x 8 on the {name:14s}: host per call " + " ".join(f"{c:.3f}" for c in calls) +
              f" ms; all done after {1e3 * (t2 - t0):.3f} ms ({8 * mb / 1024 / (t2 - t0):.1f} GB/s)", flush=True)
