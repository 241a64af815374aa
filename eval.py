#!/usr/bin/env python3
"""Evaluation entry point: per-class Dice of the argmax masks over a dataset split, clips sharded over the GPUs of one
node (no data-path collective; only the integer Dice counts are summed at the end).

    python eval.py --config config/config_gdkvm_01.yaml --weights outputs/gdkvm_step3000.pth [key=value ...]"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "config", "config_gdkvm_01.yaml"))
    ap.add_argument("--weights", default="")
    ap.add_argument("--split", default="val")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)

    from gdkvm_amd import ops
    from gdkvm_amd.config import load_config
    from gdkvm_amd.data import build_dataset
    from gdkvm_amd.distributed import init_from_env, shard_range
    from gdkvm_amd.model import GDKVM, GDKVMConfig

    cfg = load_config(args.config, args.overrides)
    ops.require_native()
    rank, world, local = init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(cfg.seed)
    mcfg = GDKVMConfig(num_classes=cfg.data.num_classes, heads=cfg.model.heads, value_dim=cfg.model.value_dim, rule=cfg.model.rule,
                       scan_segments=cfg.model.scan_segments)
    model = GDKVM(mcfg).eval()
    if args.weights:
        model.load_state_dict(torch.load(args.weights, map_location="cpu")["model"])
    model = model.fuse_for_inference().to(dev)
    if cfg.precision == "bf16":
        model = model.to(torch.bfloat16)
    model = model.to(memory_format=torch.channels_last)

    ds = build_dataset(cfg, args.split)
    lo, hi = shard_range(len(ds), world, rank)
    counts = torch.zeros(cfg.data.num_classes, 3, dtype=torch.int64, device=dev)
    vis_left = cfg.eval_stage.num_vis if rank == 0 else 0
    for i in range(lo, hi, cfg.batch_size):
        items = [ds[j] for j in range(i, min(i + cfg.batch_size, hi))]
        frames = torch.stack([x for x, _ in items]).to(dev)
        target = torch.stack([y for _, y in items]).to(dev).to(torch.uint8)
        mask, c = model.segment(frames, target=target)
        # only frames that carry labels count (EchoNet-Dynamic: the two traced frames of a clip -- gdkvm_amd.data.IGNORE_LABEL everywhere
        # else, where a predicted pixel must not enter |A|): a labelled frame has a non-empty target in some class
        labelled = (c[..., 2].sum(-1, keepdim=True) > 0).unsqueeze(-1)
        counts += (c * labelled).sum((0, 1)).long()
        if vis_left > 0:
            from PIL import Image
            os.makedirs(os.path.join(cfg.run_dir, "vis"), exist_ok=True)
            scale = 255 // max(cfg.data.num_classes - 1, 1)
            Image.fromarray((mask[0, 0].cpu().numpy() * scale).astype("uint8")).save(os.path.join(cfg.run_dir, "vis", f"mask_{i:05d}.png"))
            vis_left -= 1
    if world > 1:
        torch.distributed.all_reduce(counts)                      # the only exchange: 3 integers per class
    if rank == 0:
        dice = ops.dice_from_counts(counts).tolist()
        print(json.dumps({"split": args.split, "clips": len(ds), "dice_per_class": [round(d, 5) for d in dice],
                          "mean_foreground_dice": round(sum(dice[1:]) / max(len(dice) - 1, 1), 5)}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
