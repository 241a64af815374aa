"""Clip datasets for the harness.  No dataset ships with this repository and there is no network,
so the default is a seeded synthetic echo-like clip generator; two on-disk formats are supported for real data:

  npy_clips    <root>/<split>/<name>.npz with  frames [T,H,W] or [T,H,W,3] uint8  and  masks [T,H,W] uint8
  echonet_npz  what tools/convert_echonet.py writes from an EchoNet-Dynamic download (Videos/*.avi, FileList.csv, VolumeTracings.csv --
               the "raw data" link of the reference's guide, website reprod/index.astro:222): one .npz per video with the WHOLE video and
               the two traced frames' masks; EchoNetNpz cuts a T-frame clip around the traced frames and marks every other frame
               UNLABELLED (label 255, which the loss leaves out: gdkvm_amd.train.segmentation_loss, gdkvm_seg_loss_fwd)
  camus_png    <root>/<split>/<patient>/<view>/frame_XXX.png + mask_XXX.png, view in {2CH, 4CH}, masks 0 bg / 1 LV / 2 myocardium / 3 LA --
               what tools/convert_camus.py writes from the CAMUS NIfTI release (guide: index.astro:221); the reference's own processed
               "camus_png256x256_10f" tree is not documented beyond its name (index.astro:217,246), so this layout is the builder's
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

    def __init__(self, n_clips: int, frames: int, size: int, num_classes: int = 2, seed: int = 0, as_uint8: bool = False):
        self.n, self.T, self.S, self.C, self.seed, self.as_uint8 = n_clips, frames, size, num_classes, seed, as_uint8

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
        if self.as_uint8:                                         # bytes, as a video decoder would hand them over (as_uint8: see build_dataset)
            return torch.from_numpy(np.round(frames * 255.0).astype(np.uint8)), torch.from_numpy(masks)
        return torch.from_numpy(frames), torch.from_numpy(masks.astype(np.int64))


IGNORE_LABEL = 255             # an unlabelled pixel / frame: outside [0, num_classes), so the loss and the Dice counts skip it


def polygon_mask(xs, ys, height: int, width: int) -> np.ndarray:
    """uint8 [height, width] mask of the closed polygon (xs[i], ys[i]) -- even-odd scanline fill at pixel centres, vertices in pixel
    coordinates (x to the right, y down), as the EchoNet tracings are given.  Pure numpy: the converter runs without skimage / cv2."""
    xs, ys = np.asarray(xs, np.float64), np.asarray(ys, np.float64)
    mask = np.zeros((height, width), np.uint8)
    n = len(xs)
    if n < 3:
        return mask
    x0, y0, x1, y1 = xs, ys, np.roll(xs, -1), np.roll(ys, -1)
    cols = np.arange(width) + 0.0
    for r in range(max(0, int(np.floor(ys.min()))), min(height, int(np.ceil(ys.max())) + 1)):
        yc = r + 0.0
        cross = (y0 <= yc) != (y1 <= yc)                     # edges that straddle the scanline (half-open: a vertex counts once)
        if not cross.any():
            continue
        xi = np.sort(x0[cross] + (yc - y0[cross]) * (x1[cross] - x0[cross]) / (y1[cross] - y0[cross]))
        inside = np.zeros(width, bool)
        for a, b in zip(xi[0::2], xi[1::2]):
            inside |= (cols >= a) & (cols <= b)
        mask[r] = inside
    return mask


def echonet_tracing_polygon(rows):
    """The polygon of one traced frame from its VolumeTracings.csv rows [(X1, Y1, X2, Y2), ...]: row 0 is the long axis, every
    following row one chord across the ventricle; the outline runs down the chords' first end points and back up their second ones
    (the construction EchoNet-Dynamic's own loader uses [LIT])."""
    r = np.asarray(rows, np.float64)
    x = np.concatenate([r[1:, 0], r[1:, 2][::-1]])
    y = np.concatenate([r[1:, 1], r[1:, 3][::-1]])
    return x, y


def clip_indices(n_frames: int, labelled, T: int, pad: int = 2) -> np.ndarray:
    """T frame indices of a clip that CONTAINS every labelled frame (EchoNet: ED and ES), evenly spread from a little before the first to
    a little after the last (deterministic); videos shorter than T repeat their last frame."""
    lab = sorted(int(i) for i in labelled)
    lo, hi = max(0, lab[0] - pad), min(n_frames - 1, lab[-1] + pad)
    if hi - lo + 1 < T:                                     # widen the window to T frames where the video allows
        grow = T - (hi - lo + 1)
        lo = max(0, lo - grow // 2)
        hi = min(n_frames - 1, lo + T - 1)
        lo = max(0, hi - T + 1)
    idx = np.round(np.linspace(lo, hi, T)).astype(np.int64)
    taken = set()
    for f in lab:                                           # every labelled frame sits in the clip, each on its own position
        order = np.argsort(np.abs(idx - f), kind="stable")
        j = next(int(o) for o in order if int(o) not in taken)
        idx[j] = f
        taken.add(j)
    return np.sort(idx)


class EchoNetNpz(Dataset):
    """EchoNet-Dynamic as tools/convert_echonet.py leaves it: <root>/<split>/<FileName>.npz with `video` [F,H,W] uint8 (grey),
    `traced` [2] frame indices and `masks` [2,H,W] uint8 (0 background, 1 left ventricle).  A sample is a T-frame clip around the two
    traced frames: frames [T,3,H,W] float in [0,1], labels [T,H,W] int64 with IGNORE_LABEL on every untraced frame."""

    def __init__(self, root: str, split: str, frames: int, as_uint8: bool = False):
        self.files = sorted(glob.glob(os.path.join(root, split, "*.npz")))
        if not self.files:
            raise FileNotFoundError(f"no converted EchoNet videos (.npz) under {os.path.join(root, split)}: run tools/convert_echonet.py")
        self.T, self.as_uint8 = frames, as_uint8

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        z = np.load(self.files[i])
        video, traced, masks = z["video"], z["traced"], z["masks"]
        idx = clip_indices(video.shape[0], traced, self.T)
        clip = np.ascontiguousarray(video[idx])
        x = torch.from_numpy(clip if self.as_uint8 else clip.astype(np.float32) / 255.0).unsqueeze(1).expand(-1, 3, -1, -1).contiguous()
        y = torch.full((self.T,) + video.shape[1:], IGNORE_LABEL, dtype=torch.uint8 if self.as_uint8 else torch.int64)
        for f, m in zip(traced, masks):
            y[int(np.nonzero(idx == int(f))[0][0])] = torch.from_numpy(m.astype(np.uint8 if self.as_uint8 else np.int64))
        return x, y


class NpzClips(Dataset):
    def __init__(self, root: str, split: str, frames: int, as_uint8: bool = False):
        self.files = sorted(glob.glob(os.path.join(root, split, "*.npz")))
        if not self.files:
            raise FileNotFoundError(f"no .npz clips under {os.path.join(root, split)}")
        self.T, self.as_uint8 = frames, as_uint8

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        z = np.load(self.files[i])
        fr, mk = z["frames"][: self.T], z["masks"][: self.T]
        if fr.ndim == 3:
            fr = np.repeat(fr[..., None], 3, -1)
        if self.as_uint8:
            return torch.from_numpy(np.ascontiguousarray(fr, dtype=np.uint8)).permute(0, 3, 1, 2).contiguous(), torch.from_numpy(mk.astype(np.uint8))
        x = torch.from_numpy(fr.astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous()
        return x, torch.from_numpy(mk.astype(np.int64))


class CamusPng(Dataset):
    """CAMUS sequences as PNG trees (tools/convert_camus.py): <root>/<split>/<patient>/<view>/frame_XXX.png + mask_XXX.png with view in
    {2CH, 4CH} (both chamber views are samples of their own, BASELINE.json configs[2]); masks hold class indices 0..3; a frame without a
    mask file is unlabelled (IGNORE_LABEL).  Sequences longer than `frames` are subsampled evenly, shorter ones repeat their last frame."""

    VIEWS = ("2CH", "4CH")

    def __init__(self, root: str, split: str, frames: int, views=VIEWS, as_uint8: bool = False):
        from PIL import Image                                     # noqa: F401  (fail early if Pillow is missing)
        self.as_uint8 = as_uint8
        self.seqs = sorted(d for d in glob.glob(os.path.join(root, split, "*", "*")) if os.path.isdir(d) and os.path.basename(d) in views)
        if not self.seqs:
            raise FileNotFoundError(f"no <patient>/<{'|'.join(views)}> folders under {os.path.join(root, split)}")
        self.T = frames

    def __len__(self):
        return len(self.seqs)

    def view(self, i) -> str:
        return os.path.basename(self.seqs[i])

    def __getitem__(self, i):
        from PIL import Image
        fr = sorted(glob.glob(os.path.join(self.seqs[i], "frame_*.png")))
        if not fr:
            raise FileNotFoundError(f"{self.seqs[i]} holds no frame_*.png")
        pick = np.round(np.linspace(0, len(fr) - 1, self.T)).astype(int) if len(fr) >= self.T else \
            np.minimum(np.arange(self.T), len(fr) - 1)
        xs, ys = [], []
        for j in pick:
            img = np.asarray(Image.open(fr[j]).convert("RGB"), np.uint8) if self.as_uint8 else np.asarray(Image.open(fr[j]).convert("RGB"), np.float32) / 255.0
            mk = fr[j].replace("frame_", "mask_")
            ldt = np.uint8 if self.as_uint8 else np.int64
            xs.append(img)
            ys.append(np.asarray(Image.open(mk), ldt) if os.path.exists(mk) else np.full(img.shape[:2], IGNORE_LABEL, ldt))
        return torch.from_numpy(np.stack(xs)).permute(0, 3, 1, 2).contiguous(), torch.from_numpy(np.stack(ys))


def build_dataset(cfg, split: str = "train", n_synthetic: int = 256, as_uint8: bool = False) -> Dataset:
    """as_uint8: frames as uint8 [T,3,H,W] (0..255: what the files hold) and labels as uint8 (IGNORE_LABEL = 255 fits) instead of float32 in
    [0,1] and int64 -- a fifth of the bytes per batch across PCIe (EchoNet shape, 16 clips: 26 MB instead of 128 MB); the entry points ask for
    it and scale on the GPU (gdkvm_amd.pipeline.DevicePrefetcher).  The HIP loss and mask kernels take uint8 labels as they are."""
    d = cfg.data
    if d.kind == "synthetic" or not cfg.data_path:
        return SyntheticEchoClips(n_synthetic if split == "train" else max(8, n_synthetic // 8), d.frames, d.size,
                                  d.num_classes, seed=cfg.seed + (0 if split == "train" else 7919), as_uint8=as_uint8)
    if d.kind == "npy_clips":
        return NpzClips(cfg.data_path, split, d.frames, as_uint8=as_uint8)
    if d.kind == "echonet_npz":
        return EchoNetNpz(cfg.data_path, split, d.frames, as_uint8=as_uint8)
    if d.kind == "camus_png":
        return CamusPng(cfg.data_path, split, d.frames, as_uint8=as_uint8)
    raise ValueError(f"unknown data.kind {d.kind!r}")
