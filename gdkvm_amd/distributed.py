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
