"""Time of the stem's weight gradient at the training shape (512 frames of 112 x 112, 3 channels): gdkvm_stem_wgrad_nchw against the
framework's convolution_backward on the same operands.   python3 tools/stem_wgrad_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdkvm_amd import ops

torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.randn(n, 3, 112, 112, device="cuda").bfloat16()
dy = torch.randn(n, 64, 56, 56, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 3, 7, 7, device="cuda")


def timed(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


ref = torch.ops.aten.convolution_backward(dy.float(), x.float(), w, None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1, (False, True, False))[1]
got = ops.stem_wgrad(x, dy, like=w)
print("max |hip - fp32 framework| / max |ref| = %.3g" % ((got - ref).abs().max().item() / ref.abs().max().item()))
wb = w.bfloat16()
print("gdkvm_stem_wgrad_nchw           %.1f us" % timed(lambda: ops.stem_wgrad(x, dy, like=w)))
print("framework convolution_backward  %.1f us" % timed(lambda: torch.ops.aten.convolution_backward(dy, x, wb, None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1, (False, True, False))))
