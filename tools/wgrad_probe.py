#!/usr/bin/env python3
"""Tall-K weight-gradient GEMMs of the training step (dW = dY^T X, K = B*T*N = 25088 tokens): library GEMM as one call vs
as a batched split-K call (partials [S, M, N] summed in fp32).  Numbers quoted in DESIGN.md §8 (n1, training)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ev_time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda")
    K = 25088
    for M, N in ((512, 256), (512, 64), (256, 256), (64, 256), (256, 64)):
        a = torch.randn(K, M, device=dev).bfloat16(); b = torch.randn(K, N, device=dev).bfloat16()
        ref = a.float().t() @ b.float()
        t0 = ev_time(lambda: a.t() @ b)
        line = f"M={M:4d} N={N:4d}  mm {t0:7.1f} us"
        for S in (16, 32, 64, 128):
            if K % S:
                continue
            a3, b3 = a.view(S, K // S, M), b.view(S, K // S, N)
            f = lambda: torch.bmm(a3.transpose(1, 2), b3).float().sum(0)
            t = ev_time(f)
            err = (f() - ref).abs().max().item() / ref.abs().max().item()
            line += f" | S={S}: {t:6.1f} us err {err:.1e}"
            try:
                g = lambda: torch.bmm(a3.transpose(1, 2), b3, out_dtype=torch.float32).sum(0)
                t2 = ev_time(g)
                err2 = (g() - ref).abs().max().item() / ref.abs().max().item()
                line += f" f32out {t2:6.1f} us err {err2:.1e}"
            except Exception as e:                       # noqa: BLE001
                line += f" f32out n/a ({type(e).__name__})"
        e0 = ((a.t() @ b).float() - ref).abs().max().item() / ref.abs().max().item()
        print(line + f" | mm err {e0:.1e}", flush=True)


if __name__ == "__main__":
    main()
