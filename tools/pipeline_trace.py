#!/usr/bin/env python3
"""The host-fed forward loop alone (DevicePrefetcher + SegmentRunner, bench.py's `pipeline` leg), for
rocprofv3 --kernel-trace --memory-copy-trace: where the GPU idles between forwards.   python3 tools/pipeline_trace.py [batches]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig  # noqa: E402
from gdkvm_amd.pipeline import DevicePrefetcher, SegmentRunner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 36
dev = torch.device("cuda")
torch.manual_seed(0)
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
host = [(torch.randint(0, 256, (16, 32, 3, 112, 112), dtype=torch.uint8).pin_memory(), torch.zeros(16, dtype=torch.uint8).pin_memory()) for _ in range(6)]
runner = SegmentRunner(model, min_repeats=1, in_flight=int(os.environ.get("PROBE_IN_FLIGHT", "1")))
pre = DevicePrefetcher((host[i % 6] for i in range(n)), dev, slots=int(os.environ.get("PROBE_SLOTS", "3")), frames_dtype=torch.bfloat16)
pend, k = None, 0
for f, _ in pre:
    if k == 8:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    nxt = runner.submit(f)
    if pend is not None:
        pend.get()
    pend = nxt
    k += 1
pend.get()
torch.cuda.synchronize()
print(f"in_flight={runner.in_flight} slots={pre.slots}: {1e3 * (time.perf_counter() - t0) / (n - 8):.3f} ms per batch", flush=True)
