"""N>1 path on CPU: two gloo ranks each segment their shard of the clips (with the CPU reference module -- tests
may use the oracle) and gather; the result must equal the single-process run bit for bit.  Covers shard_range
(ragged batches), gather_clips and the 'no data-path collective' claim of DESIGN.md."""
import os
import socket

import numpy as np
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


def test_two_rank_ddp_gradients_match_full_batch():
    """Training shards clips too; the only exchange is DDP's gradient all-reduce.  With equal shards the averaged
    gradients must equal the single-process gradients of the mean loss over the whole batch (BatchNorm in eval-style
    statistics would differ per shard, so the tiny model is built without running-stat dependence: momentum-free check
    is done on all non-BatchNorm-sensitive parameters via frozen BN)."""
    from gdkvm_amd.model import GDKVMConfig
    from gdkvm_amd.train import segmentation_loss
    from oracle.model_ref import GDKVMRef
    torch.manual_seed(1)
    cfg = GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)
    model = GDKVMRef(cfg).train()
    for m in model.modules():                       # batch statistics depend on the shard: freeze BN for this identity
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    frames = torch.rand(2, 2, 3, 32, 32)
    target = (torch.rand(2, 2, 32, 32) > 0.5).long()
    # per-shard mean losses averaged == DDP semantics
    g_sum = None
    for lo in (0, 1):
        model.zero_grad()
        segmentation_loss(model(frames[lo:lo + 1]), target[lo:lo + 1]).backward()
        g = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        g_sum = g if g_sum is None else {n: g_sum[n] + g[n] for n in g}
    want = {n: v / 2 for n, v in g_sum.items()}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    state = model.state_dict()
    procs = [ctx.Process(target=_ddp_worker_frozen, args=(r, 2, port, frames, target, state, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for n in want:
        assert torch.allclose(torch.from_numpy(got[n]), want[n], atol=1e-6, rtol=1e-4), n


def _ddp_worker_frozen(rank, world, port, frames, target, state, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from gdkvm_amd.distributed import init_from_env
    from gdkvm_amd.model import GDKVMConfig
    from gdkvm_amd.train import segmentation_loss, wrap_ddp
    from oracle.model_ref import GDKVMRef
    init_from_env("gloo")
    model = GDKVMRef(GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)).train()
    model.load_state_dict(state)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    ddp = wrap_ddp(model)
    lo, hi = shard_range(frames.shape[0], world, rank)
    segmentation_loss(ddp(frames[lo:hi]), target[lo:hi]).backward()
    if rank == 0:
        out.put({n: p.grad.numpy().copy() for n, p in model.named_parameters() if p.grad is not None})
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _flat_sync_worker(rank, world, port, frames, target, state, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from gdkvm_amd.distributed import init_from_env
    from gdkvm_amd.model import GDKVMConfig
    from gdkvm_amd.train import FlatGradSync, train_step
    from oracle.model_ref import GDKVMRef
    init_from_env("gloo")
    torch.manual_seed(100 + rank)                    # every rank builds DIFFERENT weights: broadcast_parameters must make them rank 0's
    model = GDKVMRef(GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)).train().to(memory_format=torch.channels_last)
    if rank == 0:
        model.load_state_dict(state)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    sync = FlatGradSync(model)
    sync.broadcast_parameters()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    lo, hi = shard_range(frames.shape[0], world, rank)
    losses = [float(train_step(model, opt, frames[lo:hi], target[lo:hi], None, grad_sync=sync)) for _ in range(2)]
    grads_are_views = all(p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == p.stride() for p, v in zip(sync.params, sync.views))
    out.put((rank, {n: p.detach().numpy().copy() for n, p in model.named_parameters()}, losses, grads_are_views))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_flat_gradient_sync_matches_full_batch_steps():
    """gdkvm_amd.train.FlatGradSync (the graph-capturable form of the step's one exchange: one flat bucket, one all-reduce, gradients as
    views of it) on two gloo ranks: after broadcast_parameters and two SGD steps on their shards both ranks hold the SAME weights, equal to
    a single process stepping on the mean of the two shard gradients; channels_last parameters keep their strides in the bucket views."""
    from gdkvm_amd.model import GDKVMConfig
    from gdkvm_amd.train import segmentation_loss
    from oracle.model_ref import GDKVMRef
    torch.manual_seed(1)
    cfg = GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)
    model = GDKVMRef(cfg).train().to(memory_format=torch.channels_last)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    frames = torch.rand(2, 2, 3, 32, 32)
    target = (torch.rand(2, 2, 32, 32) > 0.5).long()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(2):                               # the reference run: mean of the per-shard gradients, one step
        g_sum = None
        for lo in (0, 1):
            model.zero_grad()
            segmentation_loss(model(frames[lo:lo + 1]), target[lo:lo + 1]).backward()
            g = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
            g_sum = g if g_sum is None else [a if b is None else (b if a is None else a + b) for a, b in zip(g_sum, g)]
        for p, gs in zip(model.parameters(), g_sum):
            p.grad = None if gs is None else gs / 2
        opt.step()
    want = {n: p.detach().clone() for n, p in model.named_parameters()}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_flat_sync_worker, args=(r, 2, port, frames, target, state, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((rk, (w, l, v)) for rk, w, l, v in (out.get(timeout=300) for _ in range(2)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][2] and got[1][2]
    for n in want:
        assert (got[0][0][n] == got[1][0][n]).all(), n                                   # the ranks agree exactly
        assert torch.allclose(torch.from_numpy(got[0][0][n]), want[n], atol=2e-6, rtol=1e-4), n


class _OracleBackend:
    """CPU stand-in for gdkvm_amd.ops in context_parallel_scan: the same three calls, computed with the numpy oracle
    (the transition matrix as the scan of an identity state with zero values)."""

    @staticmethod
    def new_workspace(B, T, Hh, N, Dk, Dv, device):
        return {}

    @staticmethod
    def scan_prep(q, k, v, beta, ws, rule=2, flags=0):
        ws.update(k=k.numpy(), v=v.numpy(), beta=beta.numpy(), rule=rule)

    @staticmethod
    def scan_transition(q, alpha, ws, Dv, flags=0):
        from oracle import gdkvm_oracle as O
        Dk = q.shape[-1]
        eye = np.broadcast_to(np.eye(Dk), (q.shape[0], q.shape[3], Dk, Dk)).copy()
        _, phi = O.scan(q.numpy(), ws["k"], np.zeros(ws["v"].shape[:-1] + (Dk,)), alpha.numpy(), ws["beta"], s0=eye, rule=ws["rule"], flags=flags)
        return torch.from_numpy(phi).float()

    @staticmethod
    def scan_apply(q, alpha, ws, Dv, state=None, flags=0, want_readout=True):
        from oracle import gdkvm_oracle as O
        R, S = O.scan(q.numpy(), ws["k"], ws["v"], alpha.numpy(), ws["beta"], s0=None if state is None else state.numpy(),
                      rule=ws["rule"], flags=flags)
        return (torch.from_numpy(R).float() if want_readout else None), torch.from_numpy(S).float()


def _cp_worker(rank, world, port, arrs, s0, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from gdkvm_amd.distributed import context_parallel_scan, init_from_env
    init_from_env("gloo")
    T = arrs[0].shape[1]
    lo, hi = shard_range(T, world, rank)
    q, k, v, a, b = (torch.from_numpy(np.ascontiguousarray(x[:, lo:hi])) for x in arrs)
    r, s = context_parallel_scan(q, k, v, a, b, torch.from_numpy(s0), rule=2, flags=3, backend=_OracleBackend)
    out.put((rank, r.numpy(), s.numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_context_parallel_scan_two_ranks():
    """Time-sharded clips over 2 gloo ranks: read-outs of both halves and the final state equal the single-process scan."""
    from oracle import gdkvm_oracle as O
    from tests.util import make_scan_inputs
    arrs = make_scan_inputs(2, 6, 9, 2, 8, 5, seed=70, normalized=False, logits=True)
    s0 = np.random.default_rng(71).standard_normal((2, 2, 8, 5)).astype(np.float32)
    R, S = O.scan(*arrs, s0=s0, rule=2, flags=3)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_cp_worker, args=(r, 2, port, arrs, s0, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((rk, (r, s)) for rk, r, s in (out.get(timeout=240) for _ in range(2)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_allclose(np.concatenate([got[0][0], got[1][0]], 1), R, atol=1e-5)
    np.testing.assert_allclose(got[0][1], S, atol=1e-5); np.testing.assert_allclose(got[1][1], S, atol=1e-5)


# ------------------------------------------------------------------------------------------- bench.py --gpus N launch path
def _bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


def test_bench_self_launches_two_gloo_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts torch.distributed.run itself (as a child process) and
    relays rank 0's ONE JSON line; exercised on CPU ranks with --selftest-launcher (gloo, no kernels)."""
    import json
    p = _bench("--gpus", "2", "--steps", "3", "--selftest-launcher")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["ranks_seen"] == 2


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus 2 on a box with fewer GPUs, or under a launcher with another world size, exits non-zero instead of silently
    measuring one rank (this container has no GPU at all)."""
    p = _bench("--gpus", "2", "--steps", "1")
    assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout)
    p = _bench("--gpus", "2", "--steps", "1", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
