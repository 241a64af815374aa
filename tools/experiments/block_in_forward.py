#!/usr/bin/env python3
"""A/B of the fused residual block (tools/experiments/conv3x3_block64.hip, built into tools/_abl/cb.so by abl_block.py) INSIDE the cfg2 forward:
BasicBlock.forward is patched to send the two stride-4 blocks through gdkvm_conv_block_bias_act; the forward is captured as GraphedSegment with
one and with two streams, with and without the patch, masks compared, replays timed in alternation.   python tools/experiments/block_in_forward.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gdkvm_amd import model as M  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_abl", "cb.so"))
lib.gdkvm_conv_block_bias_act.restype = ctypes.c_int
lib.gdkvm_conv_block_bias_act.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
USE = [False]
orig = M.BasicBlock.forward


def patched(self, x):
    c1, c2 = self.conv1, self.conv2
    if (USE[0] and self.down is None and isinstance(c1, M.FusedConv) and x.is_cuda and x.dtype == torch.bfloat16 and c1.conv.in_channels == 64
            and c1.conv.out_channels == 64 and x.shape[-1] <= 28):
        x = x.contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(x)
        rc = lib.gdkvm_conv_block_bias_act(x.data_ptr(), c1._packed(x.device).data_ptr(), c1.epi.bias.data_ptr(), c2._packed(x.device).data_ptr(),
                                           c2.epi.bias.data_ptr(), y.data_ptr(), x.shape[0], 64, x.shape[2], x.shape[3], 1,
                                           torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return y
    return orig(self, x)


M.BasicBlock.forward = patched


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda")
    model = M.GDKVM(M.GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    frames = [torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16() for _ in range(4)]
    graphs = {}
    for streams in (1, 2):
        for use in (False, True):
            USE[0] = use
            graphs[(streams, use)] = [M.GraphedSegment(model, f, streams=streams) for f in frames]
    ref = graphs[(1, False)][0](frames[0])[0].clone()
    for k, g in graphs.items():
        assert torch.equal(g[0](frames[0])[0], ref), k
    print("masks equal across all four forms")

    def ev(gs, it=40):
        for i in range(10):
            gs[i % 4](frames[i % 4])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(it):
            gs[i % 4](frames[i % 4])
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it
    for rnd in range(3):
        print("  ".join(f"streams={s} fused={'yes' if u else 'no '}: {ev(graphs[(s, u)]):.4f} ms" for s in (1, 2) for u in (False, True)), flush=True)


if __name__ == "__main__":
    main()
