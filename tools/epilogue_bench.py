#!/usr/bin/env python3
"""Times the conv-epilogue kernels (row n1) at their cfg2 shapes with HIP events and prints achieved HBM GB/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from tools.config_sweep import ev_time

dev = torch.device("cuda")
bf = torch.bfloat16
def cl(*shape):
    return torch.randn(*shape, device=dev).to(bf).contiguous(memory_format=torch.channels_last)

BT = 512
rows = []
for (c, hw, res) in [(64, 28, False), (64, 28, True), (128, 14, True), (256, 7, True), (128, 14, False), (64, 28, False)]:
    x = cl(BT, c, hw, hw); r = cl(BT, c, hw, hw) if res else None; b = torch.randn(c, device=dev)
    t = ev_time(lambda: ops.bias_act_(x, b, r, True))
    byts = x.numel() * 2 * (3 if res else 2)
    rows.append((f"bias_act C={c} {hw}x{hw} res={res}", t, byts))
x = cl(BT, 64, 56, 56); b = torch.randn(64, device=dev)
t = ev_time(lambda: ops.bias_relu_maxpool(x, b))
rows.append(("bias_relu_maxpool 64ch 56x56 -> 28x28", t, x.numel() * 2 + x.numel() // 4 * 2))
for (c1, hl, c2, H) in [(256, 7, 128, 14), (128, 14, 64, 28)]:
    lo = cl(BT, c1, hl, hl); sk = cl(BT, c2, H, H)
    t = ev_time(lambda: ops.upsample_cat(lo, sk))
    rows.append((f"upsample_cat {c1}x{hl}x{hl} + {c2}x{H}x{H}", t, lo.numel() * 2 + sk.numel() * 2 + BT * (c1 + c2) * H * H * 2))
for name, t, byts in rows:
    print(f"{name:48s} {t:8.1f} us  {byts / 1e6:8.1f} MB  {byts / t / 1e3:8.1f} GB/s")
