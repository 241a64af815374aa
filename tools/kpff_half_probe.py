#!/usr/bin/env python3
"""KPFF (bf16) at the full cfg2 batch (512 frames of 7x7 tokens) and at half of it: us per call.  [GDKVM_KPFF_PAIR_MIN=n: one frame per workgroup below n tiles]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gdkvm_amd import ops  # noqa: E402
from gdkvm_amd.model import KPFFParams  # noqa: E402

ops.require_native()
dev = torch.device("cuda")
torch.manual_seed(0)
kp = KPFFParams(64, 256, 256).to(dev)
for BT in (512, 256):
    L = torch.randn(BT, 49, 64, device=dev).bfloat16(); G = torch.randn(BT, 49, 256, device=dev).bfloat16(); P = torch.randn(BT, 49, 256, device=dev).bfloat16()
    ws = torch.empty(ops.load().gdkvm_kpff_workspace_bytes(64, 256, 256, 1), dtype=torch.uint8, device=dev)
    ops.kpff_fwd(L, G, P, kp.wa, kp.ba, kp.wl, kp.wg, 7, 7, workspace=ws)
    fn = lambda: ops.kpff_fwd(L, G, P, kp.wa, kp.ba, kp.wl, kp.wg, 7, 7, workspace=ws, packed=True)
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"frames {BT}: {1e3 * ms[len(ms) // 2]:6.1f} us   (GDKVM_KPFF_PAIR_MIN={os.environ.get('GDKVM_KPFF_PAIR_MIN', 'default')})", flush=True)
