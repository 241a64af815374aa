"""Clip datasets for the harness.  No dataset ships with this repository and there is no network,
so the default is a seeded synthetic echo-like clip generator; two on-disk formats are supported for real data:

  npy_clips   <root>/<split>/<name>.npz with  frames [T,H,W] or [T,H,W,3] uint8  and  masks [T,H,W] uint8
              (how EchoNet-Dynamic AVIs can be pre-extracted -- video decoding needs cv2, which this image lacks)
  camus_png   <root>/<split>/<patient>/<view>/frame_XXX.png + mask_XXX.png  (the reference's processed
              "camus_png256x256_10f" layout is not documented beyond its name -- website reprod/index.astro:217,246 -- so
              this is the builder's own convention)
"""
from __future__ import annotations

import glob
import os
from typing import Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


class SyntheticEchoClips(Dataset):
    """Speckled clips with a pulsating ellipse ("ventricle") and, for >2 classes, concentric wall / atrium regions.
    Deterministic per index, so ranks and epochs are reproducible."""

    def __init__(self, n_clips: int, frames: int, size: int, num_classes: int = 2, seed: int = 0):
        self.n, self.T, self.S, self.C, self.seed = n_clips, frames, size, num_classes, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i) -> Tuple[torch.Tensor, torch.Tensor]:
        rng = np.random.default_rng(self.seed * 1_000_003 + i)
        S, T = self.S, self.T
        yy, xx = np.mgrid[0:S, 0:S].astype(np.float32)
        cy, cx = S * (0.45 + 0.1 * rng.random()), S * (0.45 + 0.1 * rng.random())
        ry0, rx0 = S * (0.22 + 0.06 * rng.random()), S * (0.15 + 0.05 * rng.random())
        phase, rate = rng.random() * 2 * np.pi, 2 * np.pi / max(T, 2)
        frames = np.empty((T, 3, S, S), np.float32)
        masks = np.zeros((T, S, S), np.uint8)
        for t in range(T):
            k = 1.0 + 0.18 * np.sin(phase + rate * t)
            d = ((yy - cy) / (ry0 * k)) ** 2 + ((xx - cx) / (rx0 * k)) ** 2
            m = np.zeros((S, S), np.uint8)
            if self.C > 2:
                m[d < 1.45] = 2                                   # myocardium ring
            m[d < 1.0] = 1                                        # cavity
            if self.C > 3:
                m[(((yy - cy - 1.6 * ry0) / (0.6 * ry0)) ** 2 + ((xx - cx) / (1.1 * rx0)) ** 2) < 1.0] = 3   # atrium
            tissue = 0.55 - 0.4 * (m == 1) + 0.15 * (m == 2)
            speckle = np.sqrt(-2.0 * np.log(np.clip(rng.random((S, S)), 1e-7, 1.0))) * 0.25      # Rayleigh
            img = np.clip(tissue * speckle * 1.6, 0, 1).astype(np.float32)
            frames[t] = img[None]
            masks[t] = m
        return torch.from_numpy(frames), torch.from_numpy(masks.astype(np.int64))


class NpzClips(Dataset):
    def __init__(self, root: str, split: str, frames: int):
        self.files = sorted(glob.glob(os.path.join(root, split, "*.npz")))
        if not self.files:
            raise FileNotFoundError(f"no .npz clips under {os.path.join(root, split)}")
        self.T = frames

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        z = np.load(self.files[i])
        fr, mk = z["frames"][: self.T], z["masks"][: self.T]
        if fr.ndim == 3:
            fr = np.repeat(fr[..., None], 3, -1)
        x = torch.from_numpy(fr.astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous()
        return x, torch.from_numpy(mk.astype(np.int64))


class CamusPng(Dataset):
    def __init__(self, root: str, split: str, frames: int):
        from PIL import Image                                     # noqa: F401  (fail early if Pillow is missing)
        self.seqs = sorted(d for d in glob.glob(os.path.join(root, split, "*", "*")) if os.path.isdir(d))
        if not self.seqs:
            raise FileNotFoundError(f"no <patient>/<view> folders under {os.path.join(root, split)}")
        self.T = frames

    def __len__(self):
        return len(self.seqs)

    def __getitem__(self, i):
        from PIL import Image
        fr = sorted(glob.glob(os.path.join(self.seqs[i], "frame_*.png")))[: self.T]
        mk = [f.replace("frame_", "mask_") for f in fr]
        x = np.stack([np.asarray(Image.open(f).convert("RGB"), np.float32) / 255.0 for f in fr])
        y = np.stack([np.asarray(Image.open(f), np.int64) for f in mk])
        return torch.from_numpy(x).permute(0, 3, 1, 2).contiguous(), torch.from_numpy(y)


def build_dataset(cfg, split: str = "train", n_synthetic: int = 256) -> Dataset:
    d = cfg.data
    if d.kind == "synthetic" or not cfg.data_path:
        return SyntheticEchoClips(n_synthetic if split == "train" else max(8, n_synthetic // 8), d.frames, d.size,
                                  d.num_classes, seed=cfg.seed + (0 if split == "train" else 7919))
    if d.kind == "npy_clips":
        return NpzClips(cfg.data_path, split, d.frames)
    if d.kind == "camus_png":
        return CamusPng(cfg.data_path, split, d.frames)
    raise ValueError(f"unknown data.kind {d.kind!r}")
