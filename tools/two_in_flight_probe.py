#!/usr/bin/env python3
"""Two forwards in flight: the cfg2 forward's captured graphs (two streams inside each) replayed alternately on TWO host streams, so that batch
i + 1 starts while batch i is still running (separate memory pools), against the same graphs replayed on one stream.
    python3 tools/two_in_flight_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig, GraphedSegment  # noqa: E402


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda")
    model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    frames = [torch.rand(clips, 32, 3, 112, 112, device=dev).bfloat16() for _ in range(4)]
    print(f"{clips} clips per forward", flush=True)
    for inner in ((2, 1) if clips >= 16 else (1,)):
        gs = [GraphedSegment(model, f, streams=inner) for f in frames]          # (own pools)
        ref = [g(f)[0].clone() for g, f in zip(gs, frames)]
        ss = [torch.cuda.Stream() for _ in range(4)]

        def run(k, n=96):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                if k > 1:
                    with torch.cuda.stream(ss[i % k]):
                        gs[i % 4].graph.replay()
                else:
                    gs[i % 4].graph.replay()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n
        for k in (1, 2, 3, 4):
            run(k, 24)
        print(f"streams inside each graph = {inner}: " + "   ".join(f"{k} in flight {run(k):.4f}" for k in (1, 2, 3, 4, 1, 2, 3, 4)) + "  ms/forward", flush=True)
        assert all(torch.equal(g.out[0], r) for g, r in zip(gs, ref))
        del gs


if __name__ == "__main__":
    main()
