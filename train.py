#!/usr/bin/env python3
"""Training entry point: one process per GPU under torchrun, ONE gradient all-reduce per step over RCCL.
The step runs in the form bench.py measures -- one hipGraph replay: forward, backward, the flat-bucket all-reduce as a node of the graph,
fused AdamW (gdkvm_amd.pipeline.make_train_step; GDKVM_TRAIN_GRAPH=0 or a failed capture: the eager DistributedDataParallel step, said
loudly) -- on batches prefetched through pinned memory on a side stream (gdkvm_amd.pipeline.DevicePrefetcher).

    python train.py --config config/config_gdkvm_01.yaml [key=value ...]                       # one GPU
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 train.py --config ... # one node

Mirrors the recipe the reference's guide shows (batch_size 8, learning_rate 1e-4, num_iterations 3000, offline metrics,
weights under outputs/): /root/reference/website/src/pages/[lang]/reprod/index.astro:238-269."""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "config", "config_gdkvm_01.yaml"))
    ap.add_argument("--resume", default="")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)

    from gdkvm_amd import ops
    from gdkvm_amd.config import load_config
    from gdkvm_amd.data import build_dataset
    from gdkvm_amd.distributed import init_from_env
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from gdkvm_amd.pipeline import DevicePrefetcher, make_train_step
    from gdkvm_amd.runlog import OfflineRun

    cfg = load_config(args.config, args.overrides)
    ops.require_native()
    rank, world, local = init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(cfg.seed)                                   # identical initial weights on every rank

    mcfg = GDKVMConfig(num_classes=cfg.data.num_classes, heads=cfg.model.heads, value_dim=cfg.model.value_dim, rule=cfg.model.rule)
    model = GDKVM(mcfg).train().to(dev).to(memory_format=torch.channels_last)
    step0, epoch0, opt_state = 0, None, None
    if args.resume:
        ck = torch.load(args.resume, map_location=dev)
        model.load_state_dict(ck["model"]); opt_state = ck["optimizer"]; step0 = ck["step"]
        epoch0 = ck.get("epoch")

    ds = build_dataset(cfg, "train", as_uint8=True)               # bytes across PCIe (frames AND labels); scaled to [0, 1] on the GPU
    sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True, seed=cfg.seed) if world > 1 else None
    dl = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, sampler=sampler, shuffle=sampler is None, drop_last=True,
                                     num_workers=2, persistent_workers=True, pin_memory=True)
    run = OfflineRun(cfg.run_dir, cfg.to_dict(), cfg.eval_stage.wandb_mode, enabled=rank == 0)
    amp = torch.bfloat16 if cfg.precision == "bf16" else None

    if len(dl) == 0:
        raise SystemExit(f"train.py: {len(ds)} clips give no full batch of {cfg.batch_size} on each of {world} rank(s) (drop_last): "
                         "lower batch_size or add data")
    # a resumed run continues the shuffle sequence where the checkpoint left it (older checkpoints: derived from the step count)
    epoch = epoch0 if epoch0 is not None else step0 // len(dl)
    step, t_log = step0, time.perf_counter()
    train_one = opt = None                                        # built on the first batch (the graph is captured for its shape)
    use_graph = os.environ.get("GDKVM_TRAIN_GRAPH", "1") != "0"
    adamw = lambda params, fused, capturable: torch.optim.AdamW(params, lr=cfg.learning_rate, **({"fused": True, "capturable": capturable} if fused else {}))
    while step < cfg.num_iterations:
        if sampler is not None:
            sampler.set_epoch(epoch)
        # host batches staged through pinned memory and copied on a side stream while the previous step computes (DevicePrefetcher)
        for frames, target in DevicePrefetcher(dl, dev, slots=3, frames_dtype=torch.float32):
            if train_one is None:
                # the step in the form bench.py measures: one hipGraph replay (forward + backward + gradient all-reduce node + fused AdamW);
                # a capture that fails falls back to the eager DistributedDataParallel step, loudly
                train_one, opt, how = make_train_step(model, adamw, frames, target, amp, world, dev, graph=use_graph, opt_state=opt_state)
                if rank == 0:
                    print(f"train step: {how['launch']}; gradient exchange: {how['grad_sync']}; optimiser: {how['optimizer']}", flush=True)
                step += how.get("warmup_steps", 0)                # (the capture's warm-up steps are real optimiser steps on this batch)
            loss = train_one(frames, target)
            step += 1
            if step % cfg.log_every == 0 and rank == 0:
                torch.cuda.synchronize()
                dt = time.perf_counter() - t_log; t_log = time.perf_counter()
                fps = world * cfg.batch_size * cfg.data.frames * cfg.log_every / dt
                run.log(step, loss=float(loss), frames_per_s=fps)
                print(f"step {step:6d}  loss {float(loss):.4f}  {fps:9.0f} frames/s", flush=True)
            if (step % cfg.save_every == 0 or step >= cfg.num_iterations) and rank == 0:
                os.makedirs(cfg.run_dir, exist_ok=True)
                torch.save({"model": model.state_dict(), "optimizer": opt.state_dict(), "step": step, "epoch": epoch, "config": cfg.to_dict()},
                           os.path.join(cfg.run_dir, f"gdkvm_step{min(step, cfg.num_iterations)}.pth"))
            if step >= cfg.num_iterations:
                break
        epoch += 1
    run.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
