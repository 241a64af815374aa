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


def test_polygon_fill_and_clip_indices():
    from gdkvm_amd.data import clip_indices, polygon_mask
    m = polygon_mask([2, 9, 9, 2], [3, 3, 7, 7], 12, 12)                       # an axis-aligned rectangle: columns 2..9, rows 3..6 (half-open in y)
    assert m.sum() == 8 * 4 and m[3:7, 2:10].all() and not m[7].any() and not m[:, 10].any()
    tri = polygon_mask([0, 10, 0], [0, 0, 10], 11, 11)                        # a right triangle: row r holds columns 0 .. 10 - r
    assert all(tri[r].sum() == 11 - r for r in range(10)) and tri.sum() == sum(11 - r for r in range(10))
    assert polygon_mask([1, 2], [1, 2], 4, 4).sum() == 0                       # degenerate
    for n, lab, T in ((120, (34, 51), 32), (40, (3, 38), 16), (20, (5, 9), 32), (200, (10, 150), 8)):
        idx = clip_indices(n, lab, T)
        assert len(idx) == T and set(lab) <= set(idx.tolist()) and (np.diff(idx) >= 0).all() and idx.min() >= 0 and idx.max() < n
        assert np.array_equal(idx, clip_indices(n, lab, T))                   # deterministic


def test_echonet_layout_converter_and_loader(tmp_path):
    """A tiny tree in EchoNet-Dynamic's published layout (FileList.csv, VolumeTracings.csv, Videos/) -> tools/convert_echonet.py ->
    EchoNetNpz: the traced frames carry the polygon's mask, every other frame of the clip is IGNORE_LABEL, which the loss leaves out."""
    import csv
    from tools.convert_echonet import convert
    from gdkvm_amd.data import IGNORE_LABEL, EchoNetNpz
    from gdkvm_amd.train import segmentation_loss
    src, dst = tmp_path / "echonet", tmp_path / "out"
    (src / "Videos").mkdir(parents=True)
    rng = np.random.default_rng(0)
    names = {"0XAAA": ("TRAIN", 40, (7, 22)), "0XBBB": ("VAL", 30, (4, 19)), "0XCCC": ("TRAIN", 12, (2, 9))}
    with open(src / "FileList.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["FileName", "EF", "ESV", "EDV", "FrameHeight", "FrameWidth", "FPS", "NumberOfFrames", "Split"])
        for n, (sp, nf, _) in names.items():
            w.writerow([n, 55.0, 40.0, 90.0, 112, 112, 50, nf, sp])
        w.writerow(["0XMISSING", 55.0, 40.0, 90.0, 112, 112, 50, 10, "TEST"])     # listed, not traced: skipped, not fatal
    with open(src / "VolumeTracings.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["FileName", "X1", "Y1", "X2", "Y2", "Frame"])
        for n, (_, nf, traced) in names.items():
            np.save(src / "Videos" / f"{n}.npy", rng.integers(0, 255, (nf, 112, 112), dtype=np.uint8))
            for k, fr in enumerate(traced):
                half = 20 - 6 * k                                              # systole: a narrower ventricle
                w.writerow([n + ".avi", 56, 20, 56, 90, fr])                   # long axis
                for y in np.linspace(22, 88, 20):                              # 20 chords across it
                    w.writerow([n + ".avi", 56 - half, y, 56 + half, y, fr])
    res = convert(str(src), str(dst), log=lambda *a: None)
    assert res["converted"] == {"train": 2, "val": 1} and [s[0] for s in res["skipped"]] == ["0XMISSING"]
    ds = EchoNetNpz(str(dst), "train", 16)
    x, y = ds[0]
    assert x.shape == (16, 3, 112, 112) and y.shape == (16, 112, 112) and 0.0 <= x.min() and x.max() <= 1.0
    labelled = [t for t in range(16) if (y[t] != IGNORE_LABEL).any()]
    assert len(labelled) == 2 and all(set(torch.unique(y[t]).tolist()) == {0, 1} for t in labelled)
    a0, a1 = (int((y[t] == 1).sum()) for t in labelled)
    assert abs(a0 - 41 * 67) < 150 and abs(a1 - 29 * 67) < 150 and a0 > a1   # the two tracings' areas (2 half + 1 wide, ~67 rows tall)
    assert all((y[t] == IGNORE_LABEL).all() for t in range(16) if t not in labelled)
    x2, y2 = EchoNetNpz(str(dst), "train", 16)[1]                            # a 12-frame video: the clip repeats frames, both labels present
    assert sum(bool((y2[t] != IGNORE_LABEL).any()) for t in range(16)) == 2
    cfg = load_config(None, [f"data_path={dst}", "data.kind=echonet_npz", "data.frames=8", "data.size=112", "data.num_classes=2"])
    assert len(build_dataset(cfg, "val")) == 1
    # unlabelled frames carry no loss: the value equals the loss over the two labelled frames alone
    logits = torch.randn(1, 16, 2, 112, 112)
    full = segmentation_loss(logits, y[None])
    two = segmentation_loss(logits[:, labelled], y[None][:, labelled])
    assert torch.allclose(full, two, atol=1e-5)


def test_camus_layout_converter_and_loader(tmp_path):
    """A tiny tree in the CAMUS NIfTI release's layout -> tools/convert_camus.py -> CamusPng: both chamber views, four classes, 10 frames."""
    from tools.convert_camus import convert, read_nifti, write_nifti
    from gdkvm_amd.data import IGNORE_LABEL, CamusPng
    src, dst = tmp_path / "camus", tmp_path / "png"
    rng = np.random.default_rng(1)
    for pid, split in (("patient0001", "training"), ("patient0002", "validation")):
        (src / pid).mkdir(parents=True)
        for view, nf in (("2CH", 14), ("4CH", 23)):
            img = rng.integers(0, 255, (70, 96, nf), dtype=np.uint8)          # stored [x, y, frame]
            lab = np.zeros((70, 96, nf), np.uint8)
            lab[20:50, 20:70] = 2; lab[28:42, 28:60] = 1; lab[25:45, 72:90] = 3
            write_nifti(str(src / pid / f"{pid}_{view}_half_sequence.nii.gz"), img)
            write_nifti(str(src / pid / f"{pid}_{view}_half_sequence_gt.nii.gz"), lab)
        with open(src / f"subgroup_{split}.txt", "w") as f:
            f.write(pid + "\n")
    back = read_nifti(str(src / "patient0001" / "patient0001_2CH_half_sequence_gt.nii.gz"))
    assert back.shape == (70, 96, 14) and back.dtype == np.uint8 and set(np.unique(back)) == {0, 1, 2, 3}
    assert convert(str(src), str(dst), size=64, frames=10, log=lambda *a: None) == {"sequences": 4}
    ds = CamusPng(str(dst), "train", 10)
    assert len(ds) == 2 and sorted(ds.view(i) for i in range(2)) == ["2CH", "4CH"]
    x, y = ds[0]
    assert x.shape == (10, 3, 64, 64) and y.shape == (10, 64, 64) and set(torch.unique(y).tolist()) == {0, 1, 2, 3}
    assert (y[0] == 3).float().mean() > 0.02 and y[0, 10, 10] == 0            # the atrium sits below the ventricle after the [x, y] -> rows = y turn
    os.remove(dst / "train" / "patient0001" / "2CH" / "mask_004.png")        # a frame without a mask file is unlabelled
    _, y = ds[0]
    assert (y[4] == IGNORE_LABEL).all() and (y[3] != IGNORE_LABEL).all()
    assert len(CamusPng(str(dst), "val", 20)) == 2 and CamusPng(str(dst), "val", 20)[0][0].shape[0] == 20      # shorter sequences repeat the last frame
    assert len(CamusPng(str(dst), "train", 10, views=("4CH",))) == 1
