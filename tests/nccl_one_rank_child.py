#!/usr/bin/env python3
"""Child process of tests/test_zz_nccl_gpu.py: BASELINE.json configs[3] (DDP training over RCCL) as far as ONE GPU allows.

A one-rank ``nccl`` process group is legal on one GPU and runs the real RCCL code path: communicator set-up, DDP's reducer
(bucket views, autograd hooks on the HIP autograd Functions, the all-reduce launch) and all_gather of device tensors.  The
group is initialised before any other GPU call of this process, which is why this is a child and not a pytest function.

    python tests/nccl_one_rank_child.py ddp [B T S]     three train steps, DDP-wrapped (forced) against the bare model
    python tests/nccl_one_rank_child.py graph [B T S]   the step AND its gradient all-reduce captured in one hipGraph, against the eager step
    python tests/nccl_one_rank_child.py infer           the inference graph (two streams) captured and replayed between collectives
    python tests/nccl_one_rank_child.py cp              context_parallel_scan's exchange branch against gdkvm_scan_fwd
Prints ONE JSON line; the parent asserts on it.  (reprod/index.astro:238-249 of the reference's website is the recipe: a
torch.distributed launcher, one process per GPU.)"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def init_group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    return dev


def ddp(dev, B, T, S):
    from gdkvm_amd import ops
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from gdkvm_amd.train import train_step, wrap_ddp
    ops.require_native()
    torch.manual_seed(3)
    bare = GDKVM(GDKVMConfig()).train().to(dev).to(memory_format=torch.channels_last)
    twin = copy.deepcopy(bare)
    again = copy.deepcopy(bare)                               # a second BARE copy: the run-to-run spread of the backward itself
    wrapped = wrap_ddp(twin, dev, force=True)
    assert isinstance(wrapped, torch.nn.parallel.DistributedDataParallel), type(wrapped)
    g = torch.Generator(device="cpu").manual_seed(3000)
    frames = torch.rand(B, T, 3, S, S, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    target = ((((yy - S / 2) / (S * 0.3)) ** 2 + ((xx - S / 2) / (S * 0.2)) ** 2) < 1).long().expand(B, T, S, S).contiguous().to(dev)
    opt_b = torch.optim.AdamW(bare.parameters(), lr=1.0e-4)
    opt_w = torch.optim.AdamW(twin.parameters(), lr=1.0e-4)
    opt_a = torch.optim.AdamW(again.parameters(), lr=1.0e-4)
    losses_b, losses_w, rel, nograd, rel_name, spread, spread_name = [], [], 0.0, [], "", 0.0, ""
    for it in range(3):
        lb = train_step(bare, opt_b, frames, target, torch.bfloat16)
        lw = train_step(wrapped, opt_w, frames, target, torch.bfloat16)       # "unused parameter" would raise here on step 2
        losses_b.append(float(lb)); losses_w.append(float(lw))
        if it == 0:                                                            # same weights, same batch: same gradients
            train_step(again, opt_a, frames, target, torch.bfloat16)
            for (n, pb), pw, pa in zip(bare.named_parameters(), twin.parameters(), again.parameters()):
                if pb.grad is None or pw.grad is None:
                    nograd.append(n)
                    continue
                den = max(pb.grad.float().abs().max().item(), 1e-30)
                d = (pb.grad.float() - pw.grad.float()).abs().max().item() / den
                if d > rel:
                    rel, rel_name = d, n
                d = (pb.grad.float() - pa.grad.float()).abs().max().item() / den
                if d > spread:
                    spread, spread_name = d, n
    torch.cuda.synchronize()
    wdiff = max((pb.detach().float() - pw.detach().float()).abs().max().item() for pb, pw in zip(bare.parameters(), twin.parameters()))
    return {"mode": "ddp", "backend": dist.get_backend(), "world": dist.get_world_size(), "shape": [B, T, S, S],
            "loss_bare": losses_b, "loss_ddp": losses_w, "grad_rel_diff_step1": rel, "grad_rel_diff_param": rel_name,
            "bare_vs_bare_rel_diff_step1": spread, "bare_vs_bare_param": spread_name, "params_without_grad": nograd,
            "weight_abs_diff_after_3_steps": wdiff}


def graph(dev, B, T, S):
    """The training step with its gradient exchange as ONE hipGraph (GraphedTrainStep + FlatGradSync) on the one-rank RCCL group, against the
    eager bare step from the same start: every gradient kernel is deterministic and the mean over one rank is the identity, so the losses
    and the weights must be EQUAL."""
    from gdkvm_amd import ops
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from gdkvm_amd.train import FlatGradSync, GraphedTrainStep, train_step
    ops.require_native()
    torch.manual_seed(3)
    bare = GDKVM(GDKVMConfig()).train().to(dev).to(memory_format=torch.channels_last)
    twin = copy.deepcopy(bare)
    g = torch.Generator(device="cpu").manual_seed(3000)
    frames = [torch.rand(B, T, 3, S, S, generator=g).to(dev) for _ in range(4)]
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    target = ((((yy - S / 2) / (S * 0.3)) ** 2 + ((xx - S / 2) / (S * 0.2)) ** 2) < 1).long().expand(B, T, S, S).contiguous().to(dev)
    opt_b = torch.optim.AdamW(bare.parameters(), lr=1.0e-3, fused=True, capturable=True)
    opt_t = torch.optim.AdamW(twin.parameters(), lr=1.0e-3, fused=True, capturable=True)
    sync = FlatGradSync(twin)
    sync.broadcast_parameters()
    warm = 2
    gstep = GraphedTrainStep(twin, opt_t, frames[0], target, torch.bfloat16, warmup=warm, grad_sync=sync)
    for _ in range(warm):
        train_step(bare, opt_b, frames[0], target, torch.bfloat16)
    lb, lg = [], []
    for i in range(1, 4):
        lb.append(float(train_step(bare, opt_b, frames[i], target, torch.bfloat16)))
        lg.append(float(gstep(frames[i], target)))
    torch.cuda.synchronize()
    wdiff = max((pb.detach().float() - pw.detach().float()).abs().max().item() for pb, pw in zip(bare.parameters(), twin.parameters()))
    return {"mode": "graph", "backend": dist.get_backend(), "world": dist.get_world_size(), "shape": [B, T, S, S], "loss_bare": lb,
            "loss_graph": lg, "weight_abs_diff": wdiff, "bucket_elems": sync.flat.numel(),
            "grads_are_bucket_views": all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))}


def infer(dev):
    """What `bench.py --gpus N` does on every rank of an N > 1 run, on the one-rank group: collectives (barrier, all-reduce) around a
    forward that is captured into a hipGraph -- two groups of clips on two streams -- while the process group's watchdog thread is alive."""
    from gdkvm_amd import ops
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    ops.require_native()
    torch.manual_seed(5)
    model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    frames = torch.rand(8, 4, 3, 112, 112, device=dev).bfloat16()
    t = torch.ones(4, device=dev)
    for _ in range(3):
        dist.all_reduce(t)                                   # outstanding collective work right in front of the capture
    dist.barrier()
    with torch.no_grad():
        want = model.segment(frames)[0].clone()
        dist.all_reduce(t)
        g = model.graphed_segment(frames)
        same = bool(torch.equal(g(frames)[0], want))
        dist.barrier()
        for _ in range(5):
            g(frames)
        dist.all_reduce(t)
        same = same and bool(torch.equal(g(frames)[0], want))
    torch.cuda.synchronize()
    return {"mode": "infer", "backend": dist.get_backend(), "world": dist.get_world_size(), "streams": g.streams, "masks_equal": same}


def cp(dev):
    from gdkvm_amd import ops
    from gdkvm_amd.distributed import context_parallel_scan
    ops.require_native()
    out = {"mode": "cp", "backend": dist.get_backend(), "world": dist.get_world_size(), "cases": []}
    gq = torch.Generator(device=dev).manual_seed(7)
    for (B, T, N, Dv, dt, with_state) in ((2, 8, 49, 256, torch.bfloat16, False), (2, 8, 49, 256, torch.bfloat16, True),
                                          (1, 4, 256, 64, torch.bfloat16, False), (2, 6, 49, 64, torch.float32, True)):
        Hh, Dk = 1, 64
        q, k = (torch.randn(B, T, N, Hh, Dk, device=dev, generator=gq).to(dt) for _ in range(2))
        v = torch.randn(B, T, N, Hh, Dv, device=dev, generator=gq).to(dt)
        al = 2 + torch.randn(B, T, Hh, device=dev, generator=gq)
        be = torch.randn(B, T, N, Hh, device=dev, generator=gq)
        st = 0.3 * torch.randn(B, Hh, Dk, Dv, device=dev, generator=gq) if with_state else None
        r0, s0 = ops.scan_fwd(q, k, v, al, be, st, flags=3)
        r1, s1 = context_parallel_scan(q, k, v, al, be, st, flags=3, force_exchange=True)
        out["cases"].append({"shape": [B, T, N, Dv], "dtype": str(dt), "state": with_state,
                             "readout_bit_equal": bool(torch.equal(r0, r1)),
                             "state_bit_equal": bool(torch.equal(s0, s1)),
                             "state_abs_diff": (s0 - s1).abs().max().item()})
    return out


def main():
    mode = sys.argv[1]
    dev = init_group()
    try:
        if mode == "ddp":
            B, T, S = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (16, 32, 112)
            res = ddp(dev, B, T, S)
        elif mode == "graph":
            B, T, S = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (4, 8, 112)
            res = graph(dev, B, T, S)
        elif mode == "infer":
            res = infer(dev)
        elif mode == "cp":
            res = cp(dev)
        else:
            raise SystemExit(f"unknown mode {mode}")
        print(json.dumps(res), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
