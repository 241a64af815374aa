#!/usr/bin/env python3
"""Where a host-fed forward's time goes (bench.py `pipeline` leg): host time per iteration of DevicePrefetcher + SegmentRunner against the GPU's.
    python3 tools/pipeline_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402
from gdkvm_amd.pipeline import DevicePrefetcher, SegmentRunner  # noqa: E402


def main():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    host = [(torch.randint(0, 256, (16, 32, 3, 112, 112), dtype=torch.uint8).pin_memory(), torch.zeros(16, dtype=torch.uint8).pin_memory()) for _ in range(6)]
    for label, body, thr in (("prefetch only", lambda r, f: None, False), ("prefetch + replay", lambda r, f: r(f), False),
                             ("prefetch + replay (worker thread)", lambda r, f: r(f), True)):
        runner = SegmentRunner(model, min_repeats=1, in_flight=int(os.environ.get("PROBE_IN_FLIGHT", "1")))
        pre = DevicePrefetcher((host[i % 6] for i in range(46)), dev, slots=3, frames_dtype=torch.bfloat16, threaded=thr)
        n, host_s, next_s = 0, 0.0, 0.0
        it = iter(pre)
        while True:
            h0 = time.perf_counter()
            try:
                f, _ = next(it)
            except StopIteration:
                break
            h1 = time.perf_counter()
            if n == 6:
                torch.cuda.synchronize(); t0 = time.perf_counter(); host_s = next_s = 0.0; h0 = h1 = time.perf_counter()
            body(runner, f)
            h2 = time.perf_counter()
            next_s += h1 - h0
            host_s += h2 - h1
            n += 1
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(f"{label:34s}: wall {1e3 * wall / 40:.3f} ms per batch; host: next(prefetcher) {1e3 * next_s / 40:.3f} ms, runner {1e3 * host_s / 40:.3f} ms per batch", flush=True)
    g = runner._graphs[next(iter(runner._graphs))]
    gs = [v for k, v in g.items() if k != "failed"]
    fr = gs[0].frames
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(40):
        gs[i % len(gs)](gs[i % len(gs)].frames)
    b.record(); torch.cuda.synchronize()
    print(f"replays alone       : {a.elapsed_time(b) / 40:.3f} ms per batch")
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    for i in range(20):
        gs[i % len(gs)].graph.replay()
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host time of graph.replay(): {1e3 * (h1 - h0) / 20:.3f} ms per call (20 queued back to back)")
    s2 = torch.cuda.Stream()
    dst = torch.empty_like(host[0][0], device=dev)
    a.record()
    for i in range(40):
        with torch.cuda.stream(s2):
            dst.copy_(host[i % 6][0], non_blocking=True)
        gs[i % len(gs)](gs[i % len(gs)].frames)
    b.record(); s2.synchronize(); torch.cuda.synchronize()
    print(f"replays beside bare H2D copies on a second stream: {a.elapsed_time(b) / 40:.3f} ms per batch")


if __name__ == "__main__":
    main()
