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

    from gdkvm_amd.pipeline import DevicePrefetcher, SegmentRunner
    ds = build_dataset(cfg, args.split, as_uint8=True)            # bytes across PCIe; cast and scaled on the GPU
    lo, hi = shard_range(len(ds), world, rank)
    counts = torch.zeros(cfg.data.num_classes, 3, dtype=torch.int64, device=dev)
    vis_left = cfg.eval_stage.num_vis if rank == 0 else 0
    # this rank's shard through a prefetching loader (worker processes decode, pinned staging, host-to-device copies on a side stream) into
    # ONE captured forward per batch shape (SegmentRunner -> GraphedSegment: a hipGraph replay per batch; the short last batch runs eagerly)
    dl = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, range(lo, hi)), batch_size=cfg.batch_size, shuffle=False, num_workers=2, pin_memory=True)
    fdt = torch.bfloat16 if cfg.precision == "bf16" else torch.float32
    runner = SegmentRunner(model, graph=os.environ.get("GDKVM_FWD_GRAPH", "1") != "0", in_flight=int(os.environ.get("GDKVM_FWD_IN_FLIGHT", "2")))
    i = lo

    def batches():
        """(mask, counts) per batch, one batch behind the submissions: the host queues batch i + 1 (copy, cast, replay) before it reads
        batch i's result, and two forwards are in flight (SegmentRunner(in_flight=2); GDKVM_FWD_IN_FLIGHT=1: one at a time)"""
        pending = None
        for frames, target in DevicePrefetcher(dl, dev, slots=3, frames_dtype=fdt, target_dtype=torch.uint8):
            nxt = runner.submit(frames, target)
            if pending is not None:
                yield pending.get()
            pending = nxt
        if pending is not None:
            yield pending.get()

    for mask, c in batches():
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
        i += mask.shape[0]
    if world > 1:
        torch.distributed.all_reduce(counts)                      # the only exchange: 3 integers per class
    if rank == 0:
        dice = ops.dice_from_counts(counts).tolist()
        print(json.dumps({"split": args.split, "clips": len(ds), "forward": {"graph_replays": runner.replays, "eager_calls": runner.eager_calls},
                          "dice_per_class": [round(d, 5) for d in dice],
                          "mean_foreground_dice": round(sum(dice[1:]) / max(len(dice) - 1, 1), 5)}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
