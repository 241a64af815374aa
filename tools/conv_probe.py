#!/usr/bin/env python3
"""Per-layer timing of the hand-written fused-epilogue convolutions (gdkvm_conv_bias_act: kernel 4 = 64 -> 64 channels, kernel 5 =
64-channel LDS chunks) against the MIOpen convolution + gdkvm_bias_act pass they replace, at the cfg2 shapes of the encoder /
decoder (512 frames).  Decides model.FusedConv's kernel choice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402

torch.backends.cudnn.benchmark = True


def ev(fn, it=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


LAYERS = [("layer1 64->64 @28", 64, 28, 64, 3, 1, True), ("layer2.0 64->128 s2", 64, 28, 128, 3, 2, False),
          ("layer2 128->128 @14", 128, 14, 128, 3, 1, True), ("layer3.0 128->256 s2", 128, 14, 256, 3, 2, False),
          ("layer3 256->256 @7", 256, 7, 256, 3, 1, True), ("up8.conv1 384->128 @14", 384, 14, 128, 3, 1, False),
          ("up4.conv1 192->64 @28", 192, 28, 64, 3, 1, False), ("down 64->128 1x1 s2", 64, 28, 128, 1, 2, False),
          ("down 128->256 1x1 s2", 128, 14, 256, 1, 2, False)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    cl = dict(memory_format=torch.channels_last)
    for name, C, H, K, R, st, with_res in LAYERS:
        pad = R // 2
        x = torch.randn(N, C, H, H, device="cuda").relu().bfloat16().contiguous(**cl)
        w = (torch.randn(K, C, R, R, device="cuda") / (C * R * R) ** 0.5).bfloat16().contiguous(**cl)
        b = torch.randn(K, device="cuda")
        Ho = (H + 2 * pad - R) // st + 1
        for res in ([None, "res"] if with_res else [None]):
            r = torch.randn(N, K, Ho, Ho, device="cuda").bfloat16().contiguous(**cl) if res else None
            ref = torch.nn.functional.conv2d(x.float(), w.float(), b, st, pad)
            ref = (ref + r.float() if res else ref).relu()

            def mi():
                z = torch.nn.functional.conv2d(x, w, None, st, pad)
                return ops.bias_act_(z, b, r, True)
            line = f"{name:24s} {'+res' if res else '    '} miopen+epilogue {ev(mi):6.1f} us (conv {ev(lambda: torch.nn.functional.conv2d(x, w, None, st, pad)):6.1f}) | hand-written:"
            for t in ():
                try:
                    y = ops.conv_bias_act(x, w, b, r, st, pad, True, t)
                    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
                    line += f" k{t} {ev(lambda: ops.conv_bias_act(x, w, b, r, st, pad, True, t)):6.1f}" + ("" if err < 1e-2 else f"(ERR {err:.1e})")
                except ops.GdkvmError:
                    line += f" k{t}   n/a "
            if R == 3 and st == 1 and C % 64 == 0:
                pk = ops.conv3x3_pack_weights(w)
                for t in ((4, 5) if C == 64 and K == 64 else (5, 7, 8, 10)):
                    if t in (7, 10) and K % 128:
                        continue
                    y = ops.conv_bias_act(x, w, b, r, st, pad, True, t, pk)
                    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
                    line += f" | k{t} packed {ev(lambda: ops.conv_bias_act(x, w, b, r, st, pad, True, t, pk)):6.1f}" + ("" if err < 1e-2 else f"(ERR {err:.1e})")
            print(line, flush=True)


if __name__ == "__main__":
    main()
