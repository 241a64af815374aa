#!/usr/bin/env python3
"""configs[4] as a module workload (2 clips x 512 frames of 256 x 256, 16 chunks of 32 frames, state carried): segment_clip(graph=True) with the
next chunk's encoder beside the current chunk's memory path and decoder (PipelinedClip) against one whole-forward graph per chunk.
    python3 tools/clip_pipeline_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdkvm_amd.model as M  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
model = M.GDKVM(M.GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
for B, T, S, chunk in ((2, 512, 256, 32), (2, 512, 256, 64), (1, 256, 112, 32), (4, 256, 112, 32)):
    f = torch.rand(B, T, 3, S, S, device=dev).to(torch.bfloat16)
    res = {}
    for pipelined in (True, False, True, False):
        M._CLIP_PIPELINE = pipelined
        out = model.segment_clip(f, chunk, graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = model.segment_clip(f, chunk, graph=True)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 3
        res.setdefault(pipelined, []).append(ms)
        ref = res.setdefault("ref", out)
        assert torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2])
    print(f"{B} clips x {T} frames of {S} x {S}, chunks of {chunk}: pipelined " + " / ".join(f"{m:.2f}" for m in res[True]) +
          " ms   one graph per chunk " + " / ".join(f"{m:.2f}" for m in res[False]) + f" ms   ({B * T / min(res[True]) :.0f} vs {B * T / min(res[False]):.0f} frames/ms)", flush=True)
    del f
