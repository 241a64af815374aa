"""N>1 path on CPU: two gloo ranks each segment their shard of the clips (with the CPU reference module -- tests
may use the oracle) and gather; the result must equal the single-process run bit for bit.  Covers shard_range
(ragged batches), gather_clips and the 'no data-path collective' claim of DESIGN.md."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from gdkvm_amd.distributed import gather_clips, shard_range


def test_shard_range_covers_batch_exactly():
    for n in (0, 1, 5, 16, 17):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, frames, state, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from gdkvm_amd.distributed import init_from_env
    from gdkvm_amd.model import GDKVMConfig
    from oracle.model_ref import GDKVMRef
    init_from_env("gloo")
    model = GDKVMRef(GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)).eval()
    model.load_state_dict(state)
    lo, hi = shard_range(frames.shape[0], world, rank)
    with torch.no_grad():
        mask, _ = model.segment(frames[lo:hi])
    full = gather_clips(mask, frames.shape[0])
    if rank == 0:
        out.put(full.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    from gdkvm_amd.model import GDKVMConfig
    from oracle.model_ref import GDKVMRef
    torch.manual_seed(0)
    cfg = GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)
    model = GDKVMRef(cfg).eval()
    with torch.no_grad():
        model.decoder.head.bias[1] += 0.05          # random init is one class everywhere; tilt it so masks are mixed
    frames = torch.rand(3, 4, 3, 64, 64)            # 3 clips over 2 ranks: ragged shards
    with torch.no_grad():
        want, _ = model.segment(frames)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, frames, model.state_dict(), out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert (got == want.numpy()).all()
