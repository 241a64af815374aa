#!/usr/bin/env python3
"""Eager train_step against GraphedTrainStep from the same seed on the same batch: the loss of every step and, per replay, which
parameters moved and by how much (round 5: the graphed step's loss lagged the eager one's ~10x -- this is the tool that finds where).

    python tools/graph_step_diag.py [B T S] [--steps 12] [--lr 1e-4] [--warmup 3]
Prints one JSON line per phase to stdout (loss lists, the parameters whose movement differs most between the two runs)."""
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
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    B, T, S = args.shape
    from gdkvm_amd import ops
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from gdkvm_amd.train import GraphedTrainStep, train_step
    ops.require_native()
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(3)
    m_e = GDKVM(GDKVMConfig()).train().to(dev).to(memory_format=torch.channels_last)
    m_g = copy.deepcopy(m_e)
    g = torch.Generator(device="cpu").manual_seed(3000)
    frames = torch.rand(B, T, 3, S, S, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    target = ((((yy - S / 2) / (S * 0.3)) ** 2 + ((xx - S / 2) / (S * 0.2)) ** 2) < 1).long().expand(B, T, S, S).contiguous().to(dev)
    o_e = torch.optim.AdamW(m_e.parameters(), lr=args.lr, fused=True, capturable=True)
    o_g = torch.optim.AdamW(m_g.parameters(), lr=args.lr, fused=True, capturable=True)
    total = 1 + args.warmup + args.steps

    def snap(m):
        return {n: p.detach().clone() for n, p in m.named_parameters()}

    # eager: `total` steps, the weights after each
    le, we = [], []
    for _ in range(total):
        le.append(float(train_step(m_e, o_e, frames, target, torch.bfloat16)))
        we.append(snap(m_e))
    print(json.dumps({"eager_losses": [round(x, 5) for x in le]}), flush=True)

    # graphed: one set-up step, `warmup` eager steps inside the constructor, then replays
    lg = [float(train_step(m_g, o_g, frames, target, torch.bfloat16))]
    gs = GraphedTrainStep(m_g, o_g, frames, target, torch.bfloat16, warmup=args.warmup)
    lg += [float("nan")] * (args.warmup - 1) + [float(gs.loss)]
    wg = [None] * (1 + args.warmup - 1) + [snap(m_g)]
    for _ in range(args.steps):
        lg.append(float(gs(frames, target)))
        wg.append(snap(m_g))
    print(json.dumps({"graph_losses": [round(x, 5) for x in lg]}), flush=True)

    # per step after the capture: the parameters whose distance from the eager run's weights is largest, next to how far the eager step
    # itself moved them
    rows = []
    for s in range(args.warmup, total):
        if wg[s] is None:
            continue
        worst = []
        for n in wg[s]:
            d = (wg[s][n].float() - we[s][n].float()).abs().max().item()
            mv_e = (we[s][n].float() - we[s - 1][n].float()).abs().max().item()
            mv_g = (wg[s][n].float() - wg[s - 1][n].float()).abs().max().item() if wg[s - 1] is not None else float("nan")
            worst.append((d, n, mv_e, mv_g))
        worst.sort(reverse=True)
        rows.append({"step": s, "top": [{"param": n, "dist_to_eager": round(d, 7), "eager_moved": round(a, 7), "graph_moved": round(b, 7)}
                                         for d, n, a, b in worst[:6]],
                     "params_not_moving_in_graph": [n for d, n, a, b in worst if b == 0.0 and a > 0.0][:40]})
    for r in rows[:3] + rows[-1:]:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
