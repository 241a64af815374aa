#!/usr/bin/env python3
"""Probe: MIOpen's fused conv+bias+ReLU (torch.miopen_convolution_relu / _add_relu) vs conv (find mode) + gdkvm_bias_act,
at the encoder's layer shapes (512 frames, bf16, channels_last)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.backends.cudnn.benchmark = True
from gdkvm_amd import ops
from tools.config_sweep import ev_time
import torch.nn.functional as F

dev = torch.device("cuda"); bf = torch.bfloat16
for (cin, cout, hw, k, stride) in [(64, 64, 28, 3, 1), (64, 128, 28, 3, 2), (128, 128, 14, 3, 1), (256, 256, 7, 3, 1), (384, 128, 14, 3, 1), (192, 64, 28, 3, 1)]:
    x = torch.randn(512, cin, hw, hw, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).to(bf).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev)
    bb = b.to(bf)
    pad = k // 2
    def unfused():
        y = F.conv2d(x, w, None, stride, pad)
        return ops.bias_act_(y, b, None, True)
    def conv_only():
        return F.conv2d(x, w, None, stride, pad)
    def fused():
        return torch.miopen_convolution_relu(x, w, bb, [stride, stride], [pad, pad], [1, 1], 1)
    tc, tu = ev_time(conv_only), ev_time(unfused)            # plain conv first: its find result must not come from the fused call
    try:
        yf = fused(); yu = unfused()
        err = (yf.float() - yu.float()).abs().max().item()
        tf = ev_time(fused)
    except Exception as e:
        err, tf = float("nan"), float("nan"); print("fused failed:", str(e)[:100])
    print(f"cin {cin:3d} cout {cout:3d} {hw:2d}x{hw:<2d} s{stride}: conv {tc:7.1f} us  conv+epilogue {tu:7.1f} us  miopen fused {tf:7.1f} us  max|d| {err:.3f}")
