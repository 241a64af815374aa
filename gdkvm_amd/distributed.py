"""One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU in tests).

The memory path shards over CLIPS: the recurrent state is per clip and nothing in LKVA / GDR / KPFF mixes clips
(SURVEY.md §8e), so inference needs no data-path collective -- each rank runs its contiguous shard of the batch and
results are only gathered when the caller wants them in one place.  Training adds exactly one exchange per step,
the DDP gradient all-reduce (see gdkvm_amd.train once the backward kernels exist)."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank); a single process needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of n clips owned by `rank`; the first n % world ranks get one extra (ragged is fine)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_clips(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather per-rank clip shards (dim 0, possibly ragged) back into batch order on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_range(n_total, world, r) for r in range(world)]
    most = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)


def context_parallel_scan(q, k, v, alpha, beta, state=None, rule: int = 2, flags: int = 0, backend=None, force_exchange: bool = False):
    """Sequence-parallel scan of clips whose TIME axis is sharded over the ranks of the default process group (rank r holds
    frames [r*T_local, (r+1)*T_local) of every clip) -- SURVEY.md §8f row n3.

    Each rank computes, for its frames only, the state-transition matrix Phi_r and the zero-start end state S_loc_r
    (gdkvm_scan_prep + gdkvm_scan_transition + gdkvm_scan_apply without read-out), ONE all-gather exchanges them
    ([B,Hh,Dk,Dk+Dv] fp32 per rank: 80 KB per clip-head at Dk=64, Dv=256), every rank folds the maps of the ranks before it
    into its true start state, S_start_r = Phi_{r-1}(...Phi_0 S_0 + S_loc_0...) + S_loc_{r-1}, and scans its frames from
    there, reusing its prepared workspace.  Returns (R_local, S_final) with S_final identical on every rank.
    ``backend`` defaults to gdkvm_amd.ops (HIP); tests inject a CPU implementation of the same three calls.
    ``force_exchange`` takes the exchange branch on a one-rank group too (transition matrix, all-gather of device tensors, fold,
    second pass): the one-GPU RCCL test of this path."""
    if backend is None:
        from . import ops as backend
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    ws = backend.new_workspace(B, T, Hh, N, Dk, Dv, q.device)
    if rule == 1:                                       # delta_parallel: full-range operands in every stage (include/gdkvm.h)
        flags |= 8                                      # GDKVM_FLAG_WIDE_RANGE
    backend.scan_prep(q, k, v, beta, ws, rule=rule, flags=flags)
    cur = state if state is not None else torch.zeros((B, Hh, Dk, Dv), dtype=torch.float32, device=q.device)
    if world > 1 or (force_exchange and dist.is_initialized()):
        phi = backend.scan_transition(q, alpha, ws, Dv, flags=flags)
        _, s_loc = backend.scan_apply(q, alpha, ws, Dv, flags=flags, want_readout=False)
        mine = torch.cat([phi, s_loc], -1).contiguous()                      # [B,Hh,Dk,Dk+Dv]
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                                         # the only exchange of the whole scan
        stitch = getattr(backend, "scan_stitch", None)
        if stitch is not None:
            # the ranks' maps folded by ONE launch (gdkvm_scan_stitch: the serial stitch of the time-segmented scan, exact fp32 MFMA):
            # starts[:, r] is the state rank r's frames begin with, `end` the state after the last rank's -- identical on every rank
            both = torch.stack(every, 1)                                     # [B, world, Hh, Dk, Dk+Dv]
            starts, end = stitch(both[..., :Dk].contiguous(), both[..., Dk:].contiguous(), cur if state is not None else None)
            start, cur = starts[:, rank], end
        else:                                                                # (a test backend without the kernel: the same fold, rank by rank)
            start = None
            for r in range(world):
                if r == rank:
                    start = cur
                cur = torch.matmul(every[r][..., :Dk], cur) + every[r][..., Dk:]
        r_local, _ = backend.scan_apply(q, alpha, ws, Dv, state=start.contiguous(), flags=flags)
        return r_local, cur
    return backend.scan_apply(q, alpha, ws, Dv, state=cur if state is not None else None, flags=flags)
