#!/usr/bin/env python3
"""Eager train_step under different AdamW implementations / library settings from the same seed on the same batch: the loss of every
step and, after the first step, what fraction of each parameter's elements moved and by how much.

    python tools/opt_step_diag.py [B T S] [--steps 6] [--lr 1e-4]
One JSON line per variant."""
import argparse
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", type=int, default=[16, 32, 112])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--lr", type=float, default=1e-4)
    args = ap.parse_args()
    B, T, S = args.shape
    from gdkvm_amd import ops
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from gdkvm_amd.train import train_step
    ops.require_native()
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    m0 = GDKVM(GDKVMConfig()).train().to(dev).to(memory_format=torch.channels_last)
    g = torch.Generator(device="cpu").manual_seed(3000)
    frames = torch.rand(B, T, 3, S, S, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    target = ((((yy - S / 2) / (S * 0.3)) ** 2 + ((xx - S / 2) / (S * 0.2)) ** 2) < 1).long().expand(B, T, S, S).contiguous().to(dev)
    variants = [("default", {}, False), ("fused", {"fused": True}, False), ("fused+capturable", {"fused": True, "capturable": True}, False),
                ("foreach+capturable", {"foreach": True, "capturable": True}, False), ("default, benchmark", {}, True)]
    grads0 = None
    for name, kw, bench in variants:
        torch.backends.cudnn.benchmark = bench
        m = copy.deepcopy(m0)
        opt = torch.optim.AdamW(m.parameters(), lr=args.lr, **kw)
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        losses = [float(train_step(m, opt, frames, target, torch.bfloat16))]
        gr = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        moved = {}
        for n, p in m.named_parameters():
            d = (p.detach() - before[n]).abs()
            moved[n] = (float((d > 0).float().mean()), float(d.max()), float(d.mean()), tuple(p.stride()) != tuple(p.grad.stride()))
        for _ in range(args.steps - 1):
            losses.append(float(train_step(m, opt, frames, target, torch.bfloat16)))
        low = sorted((v[2], n, v) for n, v in moved.items())[:8]
        out = {"variant": name, "losses": [round(x, 5) for x in losses],
               "mean_abs_move_all": round(sum(v[2] for v in moved.values()) / len(moved), 8),
               "least_moved": [{"param": n, "frac_moved": round(v[0], 4), "max": round(v[1], 8), "mean": round(v[2], 8), "grad_stride_differs": v[3]}
                               for _, n, v in low]}
        if grads0 is None:
            grads0 = gr
        else:
            out["max_grad_rel_diff_vs_first_variant"] = max(
                (gr[n].float() - grads0[n].float()).abs().max().item() / max(grads0[n].float().abs().max().item(), 1e-30) for n in gr)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
