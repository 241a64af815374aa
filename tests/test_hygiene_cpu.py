"""CPU checks of round 6's host-side rules: pack / graph invalidation is per model and only on real changes, a convolution that leaves the
hand-written path says so (and raises under GDKVM_STRICT=1), the launcher honours the reference guide's CUDA_VISIBLE_DEVICES, the
flat-gradient exchange refuses a parameter without a gradient, stream groups of a captured segment must keep 16-byte output slabs."""
import os
import subprocess
import types
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eval_on_an_eval_model_keeps_graphs_valid_and_epochs_are_per_model():
    from gdkvm_amd import model as M
    a = M.GDKVM(M.GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)).eval()
    b = M.GDKVM(M.GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)).eval()
    ea, eb = a.__dict__["_pack_epoch"], b.__dict__["_pack_epoch"]
    a.eval()                                                     # a defensive eval(): no mode change, nothing invalidated
    assert a.__dict__["_pack_epoch"] == ea
    a.train()
    assert a.__dict__["_pack_epoch"] == ea + 1                  # a real mode change drops the packs
    a.eval()
    ka, kb = M._epoch_of(a), M._epoch_of(b)
    M.weights_changed(a)                                         # an optimiser step on `a` ...
    assert M._epoch_of(a) != ka and M._epoch_of(b) == kb         # ... leaves a frozen model `b` (teacher / EMA copy) alone
    assert b.__dict__["_pack_epoch"] == eb
    M.weights_changed()                                          # the process-wide form tells everyone
    assert M._epoch_of(b) != kb
    # the sub-modules that cache packs share their model's cell, before and after folding
    assert a.decoder.__dict__["_epoch_cell"] is a.__dict__["_epoch_cell"]
    a.fuse_for_inference()
    convs = [m for m in a.modules() if isinstance(m, M.FusedConv)]
    assert convs and all(m.__dict__["_epoch_cell"] is a.__dict__["_epoch_cell"] for m in convs)
    wrapped = types.SimpleNamespace(module=a)                    # DistributedDataParallel-style wrapper
    k2 = M._epoch_of(a)
    M.weights_changed(wrapped)
    assert M._epoch_of(a) != k2


def test_library_fallbacks_are_loud(monkeypatch):
    from gdkvm_amd import model as M
    gpu_like = types.SimpleNamespace(is_cuda=True)
    cpu_like = types.SimpleNamespace(is_cuda=False)
    M._FALLBACKS_SEEN.clear()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        M._library_fallback(cpu_like, "layer", "why")            # the CPU reference module is plain torch by design: silent
        assert not rec
        M._library_fallback(gpu_like, "encoder.layer9.conv", "odd width")
        M._library_fallback(gpu_like, "encoder.layer9.conv", "odd width")    # once per layer and reason
        assert len(rec) == 1 and "encoder.layer9.conv" in str(rec[0].message) and issubclass(rec[0].category, RuntimeWarning)
    monkeypatch.setattr(M, "_STRICT", True)
    with pytest.raises(RuntimeError, match="GDKVM_STRICT"):
        M._library_fallback(gpu_like, "decoder.up4.conv", "fp32 inference")
    M._library_fallback(cpu_like, "decoder.up4.conv", "fp32 inference")      # still silent on the CPU


def test_train_sh_honours_cuda_visible_devices():
    """/root/reference/website/src/pages/[lang]/reprod/index.astro:238 sets CUDA_VISIBLE_DEVICES: 0,1 -- the launcher takes it (HIP_VISIBLE_DEVICES wins)."""
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "MASTER_PORT")}
    run = lambda **kw: subprocess.run(["bash", os.path.join(ROOT, "train.sh")], env=dict(env, GDKVM_TRAIN_SH_DRY_RUN="1", **kw),
                                      capture_output=True, text=True, check=True).stdout.strip()
    assert run(CUDA_VISIBLE_DEVICES="0,1") == "HIP_VISIBLE_DEVICES=0,1 NGPU=2 MASTER_PORT=29500"
    assert run(CUDA_VISIBLE_DEVICES="0,1", HIP_VISIBLE_DEVICES="3", MASTER_PORT="29511") == "HIP_VISIBLE_DEVICES=3 NGPU=1 MASTER_PORT=29511"
    assert run() == "HIP_VISIBLE_DEVICES=0 NGPU=1 MASTER_PORT=29500"


def test_flat_grad_sync_refuses_a_parameter_without_gradient():
    import torch.distributed as dist
    from gdkvm_amd.train import FlatGradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
        unused = torch.nn.Linear(3, 3)
        net.add_module("unused", unused)
        sync = FlatGradSync(net)
        net[1](net[0](torch.randn(5, 4))).sum().backward()
        with pytest.raises(RuntimeError, match="unused"):
            sync()
        lenient = FlatGradSync(net, allow_unused=True)
        lenient()
        assert unused.weight.grad is not None and float(unused.weight.grad.abs().max()) == 0.0
        lenient.broadcast_buffers()
    finally:
        dist.destroy_process_group()


def test_segment_stream_groups_need_aligned_output_slabs():
    """GraphedSegment(streams=n) hands each group a slice of ONE mask / counts result and the mask kernel wants 16-byte aligned outputs:
    10 clips x 3 frames x 2 classes = 360-byte count slabs per group of 5 -- the automatic choice falls back to one stream, an explicit
    streams=2 says why it cannot be served.  (The check sits in front of anything that needs a device.)"""
    from gdkvm_amd import model as M
    m = M.GDKVM(M.GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=32)).eval()
    fake = types.SimpleNamespace(is_cuda=True, shape=(10, 3, 3, 20, 20), device="cuda")
    tgt = object()
    with pytest.raises(ValueError, match="16-byte"):
        M.GraphedSegment(m, fake, tgt, streams=2)
