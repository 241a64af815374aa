#!/usr/bin/env python3
"""Does running the two halves of the batch on two streams of ONE hipGraph help?  (Every kernel of the forward fills the chip and drains with a
tail; kernels of an independent half could fill the other's tails and overlap memory-bound kernels with MFMA-bound ones.)
    python tools/two_stream_probe.py [parts=2]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402

split = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 and "," in sys.argv[1] else None
parts = len(split) if split else (int(sys.argv[1]) if len(sys.argv) > 1 else 2)
torch.manual_seed(0)
dev = torch.device("cuda")
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
frames = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()


def timeit(fn, n=60):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


with torch.no_grad():
    for _ in range(3):
        ref = model.segment(frames)[0]
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        out1 = model.segment(frames)[0]
    print(f"one stream, one graph        {timeit(g1.replay):.4f} ms")

    chunks = list(frames.split(split, 0)) if split else list(frames.chunk(parts, 0))
    pr = os.environ.get("PROBE_PRIORITIES")                     # e.g. "-1,0": stream priorities (lower = higher priority)
    streams = [torch.cuda.Stream(priority=int(v)) for v in pr.split(",")] if pr else [torch.cuda.Stream() for _ in range(parts)]
    for c in chunks:                                           # warm every shape (packs, attributes) outside the capture
        model.segment(c)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    outs = [None] * parts
    with torch.cuda.graph(g2):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i, (s, c) in enumerate(zip(streams, chunks)):
            with torch.cuda.stream(s):
                if i and os.environ.get("PROBE_DELAY_US"):             # experiment: the later streams start late (phases out of step)
                    torch.cuda._sleep(int(float(os.environ["PROBE_DELAY_US"]) * 2100 * i))
                outs[i] = model.segment(c)[0]
        for s in streams:
            cur.wait_stream(s)
    print(f"{parts} streams in one graph {split or ''}      {timeit(g2.replay):.4f} ms")
    g2.replay(); torch.cuda.synchronize()
    print("masks equal:", bool(torch.equal(torch.cat(outs, 0), ref)))
    # the halves one after the other on ONE stream (what the split alone costs)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        o3 = [model.segment(c)[0] for c in chunks]
    print(f"{parts} parts, one stream          {timeit(g3.replay):.4f} ms")
