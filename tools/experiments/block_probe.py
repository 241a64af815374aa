#!/usr/bin/env python3
"""Times a stride-4 residual block (64 -> 64 -> 64 channels, 512 frames of 28 x 28) as ONE fused launch (gdkvm_conv_block_bias_act) against the
two launches of the 64 -> 64 kernel it replaces, alone and with a copy kernel on a second stream (what the two-stream forward does to it).
    python3 tools/block_probe.py [frames=512]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gdkvm_amd import ops  # noqa: E402


def ev(fn, it=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    cl = torch.channels_last
    torch.manual_seed(0)
    xs = [torch.randn(n, 64, 28, 28, device="cuda").bfloat16().contiguous(memory_format=cl) for _ in range(6)]
    w1, w2 = ((torch.randn(64, 64, 3, 3, device="cuda") / 24).bfloat16().contiguous(memory_format=cl) for _ in range(2))
    b1, b2 = torch.randn(64, device="cuda"), torch.randn(64, device="cuda")
    p1, p2 = ops.conv3x3_pack_weights(w1), ops.conv3x3_pack_weights(w2)
    i = [0]

    def fused():
        i[0] += 1
        return ops.conv_block_bias_act(xs[i[0] % 6], p1, b1, p2, b2)

    def two():
        i[0] += 1
        x = xs[i[0] % 6]
        return ops.conv_bias_act(ops.conv_bias_act(x, w1, b1, None, 1, 1, True, 4, p1), w2, b2, x, 1, 1, True, 4, p2)

    i[0] = 0
    yf = fused()
    i[0] = 0
    assert torch.equal(yf, two())
    print(f"frames {n}: fused block {ev(fused):.1f} us   two launches {ev(two):.1f} us")


if __name__ == "__main__":
    main()
