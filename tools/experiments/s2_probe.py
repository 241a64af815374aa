#!/usr/bin/env python3
"""The two strided blocks of the cfg2 forward (512 frames): the stride-2 halo-band kernel against the general implicit-GEMM kernel, us per call (both with
the 1x1 / stride-2 branch in the same launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gdkvm_amd import ops  # noqa: E402

ops.require_native()
cl = torch.channels_last


def ev(fn, it=30):
    for _ in range(5):
        fn()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in e)
    return 1e3 * ms[len(ms) // 2]


n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for c, k, h in ((64, 128, 28), (128, 256, 14)):
    x = torch.randn(n, c, h, h, device="cuda").relu().bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(k, c, 3, 3, device="cuda") / (9 * c) ** 0.5).bfloat16().contiguous(memory_format=cl)
    wd = (torch.randn(k, c, 1, 1, device="cuda") / c ** 0.5).bfloat16().contiguous(memory_format=cl)
    b = torch.randn(k, device="cuda")
    p2 = ops.conv3x3s2_pack_weights(w, wd)
    pi, pdi = ops.conv_igemm_pack_weights(w), ops.conv_igemm_pack_weights(wd)
    t_new = ev(lambda: ops.conv3x3s2_down_bias_act(x, p2, b, k, True, True))
    t_old = ev(lambda: ops.conv_down_bias_act(x, w, b, pi, wd, pdi, None, 2, True))
    flop = 2.0 * n * (h // 2) ** 2 * k * c * 10
    print(f"{c:4d} -> {k:4d} s2 @ {h:2d} ({n} frames): band kernel {t_new:6.1f} us ({flop / t_new / 1e6:6.0f} TFLOP/s)   general kernel {t_old:6.1f} us", flush=True)
