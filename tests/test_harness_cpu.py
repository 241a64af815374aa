"""Host logic of the harness (row n2): configuration, datasets, offline logging."""
import json
import os

import numpy as np
import pytest
import torch

from gdkvm_amd.config import load_config
from gdkvm_amd.data import NpzClips, SyntheticEchoClips, build_dataset
from gdkvm_amd.runlog import OfflineRun

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_keys_of_the_reference_guide():
    cfg = load_config(os.path.join(ROOT, "config", "config_gdkvm_01.yaml"))
    assert (cfg.batch_size, cfg.learning_rate, cfg.num_iterations) == (8, 1.0e-4, 3000)       # reprod/index.astro:246-249
    assert cfg.eval_stage.num_vis == 0 and cfg.eval_stage.wandb_mode == "offline"              # :250-252
    cfg = load_config(os.path.join(ROOT, "config", "config_gdkvm_01.yaml"), ["batch_size=2", "eval_stage.num_vis=3", "data.size=112"])
    assert cfg.batch_size == 2 and cfg.eval_stage.num_vis == 3 and cfg.data.size == 112
    with pytest.raises(KeyError):
        load_config(None, ["no_such_key=1"])
    with pytest.raises(ValueError):
        load_config(None, ["broken"])


def test_synthetic_dataset_is_deterministic_and_labelled():
    ds = SyntheticEchoClips(4, 6, 64, num_classes=4, seed=1)
    x, y = ds[2]
    x2, y2 = ds[2]
    assert x.shape == (6, 3, 64, 64) and y.shape == (6, 64, 64) and torch.equal(x, x2) and torch.equal(y, y2)
    assert 0.0 <= x.min() and x.max() <= 1.0 and set(torch.unique(y).tolist()) == {0, 1, 2, 3}
    assert not torch.equal(y[0], y[3])                         # the cavity pulsates


def test_npz_dataset_roundtrip(tmp_path):
    d = tmp_path / "train"; d.mkdir()
    fr = np.random.default_rng(0).integers(0, 255, (5, 32, 32), dtype=np.uint8)
    mk = (fr > 128).astype(np.uint8)
    np.savez(d / "clip0.npz", frames=fr, masks=mk)
    x, y = NpzClips(str(tmp_path), "train", 4)[0]
    assert x.shape == (4, 3, 32, 32) and y.shape == (4, 32, 32) and torch.equal(y, torch.from_numpy(mk[:4].astype(np.int64)))
    cfg = load_config(None, [f"data_path={tmp_path}", "data.kind=npy_clips", "data.frames=4"])
    assert len(build_dataset(cfg, "train")) == 1


def test_offline_run_log(tmp_path):
    run = OfflineRun(str(tmp_path), {"a": 1}, "offline")
    run.log(1, loss=0.5); run.log(2, loss=0.25); run.close()
    lines = open(os.path.join(run.dir, "metrics.jsonl")).read().strip().split("\n")
    assert [json.loads(l)["loss"] for l in lines] == [0.5, 0.25] and "offline-run-" in run.dir
    assert not OfflineRun(str(tmp_path), {}, "disabled").enabled


def test_graph_wrappers_and_new_switches_fail_loudly_on_the_cpu():
    """The graph-capture wrappers (model.GraphedSegment, train.GraphedTrainStep) are device-only and say so; the product has no CPU route
    for the hand-written operators they replay (ops raise without a device tensor); the configuration carries the long-clip switch."""
    import pytest
    import torch
    from gdkvm_amd import ops
    from gdkvm_amd.model import GDKVM, GDKVMConfig, GraphedSegment
    from gdkvm_amd.train import GraphedTrainStep
    cfg = GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=64)
    assert cfg.scan_segments == 1                          # serial scan (chunk bit-identity) unless asked otherwise
    model = GDKVM(cfg).eval()
    frames = torch.rand(1, 2, 3, 32, 32)
    with pytest.raises(RuntimeError, match="device"):
        GraphedSegment(model, frames)
    with pytest.raises(RuntimeError, match="device"):
        GraphedTrainStep(model, None, frames, torch.zeros(1, 2, 32, 32, dtype=torch.long))
    x = torch.randn(8, 64)
    lin = torch.nn.Linear(64, 16)
    with pytest.raises(Exception):                          # token-major products exist on the device only (no silent CPU fallback)
        ops.token_projections(x, (lin,))
    assert not ops.bn_relu_pool_served(torch.zeros(1, 8, 4, 4)) and not ops.head_served(torch.zeros(1, 64, 4, 4), torch.nn.Conv2d(64, 2, 1))
