#!/usr/bin/env python3
"""Soak: many rounds of the forwards-in-flight loop (model.InFlightSegments) and of the host-fed loop (DevicePrefetcher + SegmentRunner, results one
batch behind), every result compared with the eager forward's.  python3 tools/soak_in_flight.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd.model import GDKVM, GDKVMConfig, InFlightSegments  # noqa: E402
from gdkvm_amd.pipeline import DevicePrefetcher, SegmentRunner  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
torch.manual_seed(5)
model = GDKVM(GDKVMConfig()).to(dev).eval().to(memory_format=torch.channels_last)
nb = 8
u8 = [torch.randint(0, 256, (16, 32, 3, 112, 112), dtype=torch.uint8) for _ in range(nb)]
batches = [torch.empty(b.shape, dtype=torch.bfloat16, device=dev) for b in u8]
for d, b in zip(batches, u8):
    torch.mul(b.to(dev), 1.0 / 255.0, out=d)
targets = [(torch.rand(16, 32, 112, 112, device=dev) > 0.5).to(torch.uint8) for _ in range(nb)]
with torch.no_grad():
    lg = model(batches[0].float(), _lowres=True)
    model.decoder.head.bias[1] += (lg[:, :, 0] - lg[:, :, 1]).median()
model = model.fuse_for_inference().to(torch.bfloat16)
want = [tuple(t.clone() for t in model.segment(b, target=t_)[:2]) for b, t_ in zip(batches, targets)]
assert not torch.equal(want[0][0], want[1][0])
ring = InFlightSegments(model, batches, targets, in_flight=2)
bad = 0
for r in range(rounds):
    outs = [ring.launch(i) for i in range(nb)]
    ring.synchronize()
    for i, o in enumerate(outs):
        if not (torch.equal(o[0], want[i][0]) and torch.equal(o[1], want[i][1])):
            bad += 1
print(f"resident batches: {rounds * nb} forwards in flight, {bad} differ from the eager forward", flush=True)
host = [(b.pin_memory(), t.cpu().pin_memory()) for b, t in zip(u8, targets)]
runner = SegmentRunner(model, min_repeats=1)
n = rounds * nb
bad2, k, pend = 0, 0, None


def check(res, idx):
    m, c = res
    return int(not (torch.equal(m, want[idx % nb][0]) and torch.equal(c, want[idx % nb][1])))


for f, t in DevicePrefetcher((host[i % nb] for i in range(n)), dev, slots=3, frames_dtype=torch.bfloat16, target_dtype=torch.uint8):
    nxt = runner.submit(f, t)
    if pend is not None:
        bad2 += check(pend.get(), k - 1)
    pend = nxt
    k += 1
bad2 += check(pend.get(), k - 1)
torch.cuda.synchronize()
print(f"host-fed batches: {n} forwards ({runner.replays} replays, {runner.eager_calls} eager), {bad2} differ from the eager forward", flush=True)
sys.exit(1 if bad or bad2 else 0)
