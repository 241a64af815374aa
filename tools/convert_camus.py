#!/usr/bin/env python3
"""CAMUS NIfTI release -> the PNG tree gdkvm_amd.data.CamusPng reads (BASELINE.json configs[2]).

The dataset as published (the "raw data" link of the reference's guide, /root/reference/website/src/pages/[lang]/reprod/index.astro:221):
    <src>/patientXXXX/patientXXXX_<2CH|4CH>_half_sequence.nii.gz        the ED -> ES image sequence  [W, H, F]
    <src>/patientXXXX/patientXXXX_<2CH|4CH>_half_sequence_gt.nii.gz     its labels: 0 background, 1 LV, 2 myocardium, 3 left atrium
    (optionally <src>/subgroup_{training,validation,testing}.txt: one patient id per line; without them every patient goes to --split)
Written (what the reference calls "camus_png256x256_10f": 256 x 256, 10 frames per sequence -- index.astro:217,246):
    <dst>/<split>/patientXXXX/<2CH|4CH>/frame_000.png ... mask_000.png ...   `--frames` frames evenly spread over the sequence, resized
    to `--size` (images bilinear, labels nearest)

NIfTI-1 is read here directly (a 348-byte header + a raw array, gzip around it): no nibabel in this image.  Nothing touches a GPU.

    python tools/convert_camus.py <src> <dst> [--size 256] [--frames 10] [--split train]
"""
import argparse
import glob
import gzip
import os
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}


def read_nifti(path: str) -> np.ndarray:
    """The voxel array of a NIfTI-1 file (.nii or .nii.gz), indexed [x, y, z(, t)] as stored (Fortran order), scaled by scl_slope/inter."""
    raw = (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")).read()
    if len(raw) < 352:
        raise ValueError(f"{path}: too short for a NIfTI-1 header")
    end = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[:4])[0] != 348 or raw[344:347] not in (b"n+1", b"ni1"):
        raise ValueError(f"{path}: not a NIfTI-1 file")
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype = struct.unpack(end + "h", raw[70:72])[0]
    vox_offset = int(struct.unpack(end + "f", raw[108:112])[0])
    slope, inter = struct.unpack(end + "2f", raw[112:120])
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: NIfTI datatype {datatype} is not handled")
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    arr = np.frombuffer(raw, dt, int(np.prod(shape)), vox_offset).reshape(shape, order="F")
    if slope not in (0.0, 1.0) or inter != 0.0:
        arr = arr.astype(np.float32) * (slope if slope != 0.0 else 1.0) + inter
    return arr


def write_nifti(path: str, arr: np.ndarray) -> None:
    """A minimal NIfTI-1 writer (tests build tiny CAMUS-shaped trees with it)."""
    code = {np.dtype(v): k for k, v in _DTYPES.items()}[arr.dtype]
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, arr.ndim, *(list(arr.shape) + [1] * (7 - arr.ndim)))
    struct.pack_into("<h", hdr, 70, code)
    struct.pack_into("<h", hdr, 72, arr.dtype.itemsize * 8)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    hdr[344:348] = b"n+1\0"
    data = bytes(hdr) + np.asfortranarray(arr).tobytes(order="F")
    (gzip.open(path, "wb") if path.endswith(".gz") else open(path, "wb")).write(data)


def resize(img: np.ndarray, size: int, nearest: bool) -> np.ndarray:
    """[H, W] -> [size, size]; half-pixel-centre sampling; bilinear for images, nearest for label maps."""
    h, w = img.shape
    ys = (np.arange(size) + 0.5) * h / size - 0.5
    xs = (np.arange(size) + 0.5) * w / size - 0.5
    if nearest:
        yi = np.clip(np.round(ys).astype(int), 0, h - 1)
        xi = np.clip(np.round(xs).astype(int), 0, w - 1)
        return img[yi][:, xi]
    y0 = np.clip(np.floor(ys).astype(int), 0, h - 1); y1 = np.clip(y0 + 1, 0, h - 1); fy = np.clip(ys - y0, 0, 1)[:, None]
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1); x1 = np.clip(x0 + 1, 0, w - 1); fx = np.clip(xs - x0, 0, 1)[None, :]
    f = img.astype(np.float32)
    top = f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx
    bot = f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def _splits(src: str, default: str) -> dict:
    out = {}
    for name, split in (("training", "train"), ("validation", "val"), ("testing", "test")):
        p = os.path.join(src, f"subgroup_{name}.txt")
        if os.path.exists(p):
            for line in open(p):
                if line.strip():
                    out[line.strip()] = split
    return out if out else defaultdict_const(default)


class defaultdict_const(dict):
    def __init__(self, v):
        super().__init__()
        self.v = v

    def get(self, k, d=None):
        return self.v


def convert(src: str, dst: str, size: int = 256, frames: int = 10, split: str = "train", log=print) -> dict:
    from PIL import Image
    splits = _splits(src, split)
    done = 0
    for pdir in sorted(glob.glob(os.path.join(src, "patient*"))):
        pid = os.path.basename(pdir)
        for view in ("2CH", "4CH"):
            seq = [p for e in (".nii.gz", ".nii") for p in glob.glob(os.path.join(pdir, f"{pid}_{view}_half_sequence{e}"))]
            gt = [p for e in (".nii.gz", ".nii") for p in glob.glob(os.path.join(pdir, f"{pid}_{view}_half_sequence_gt{e}"))]
            if not seq or not gt:
                continue
            img, lab = read_nifti(seq[0]), read_nifti(gt[0])                  # [W, H, F] each
            if img.shape != lab.shape or img.ndim != 3:
                log(f"  skipped {pid} {view}: image {img.shape} against labels {lab.shape}")
                continue
            out = os.path.join(dst, splits.get(pid, split), pid, view)
            os.makedirs(out, exist_ok=True)
            pick = np.round(np.linspace(0, img.shape[2] - 1, frames)).astype(int) if img.shape[2] >= frames else np.arange(img.shape[2])
            for j, f in enumerate(pick):
                a = resize(np.asarray(img[:, :, f]).T, size, False)           # stored [x, y] -> rows = y
                m = resize(np.asarray(lab[:, :, f]).T.astype(np.int64), size, True)
                hi = float(a.max()) or 1.0
                Image.fromarray(np.clip(a * (255.0 / hi if hi > 255.0 else 1.0), 0, 255).astype(np.uint8)).save(os.path.join(out, f"frame_{j:03d}.png"))
                Image.fromarray(m.astype(np.uint8)).save(os.path.join(out, f"mask_{j:03d}.png"))
            done += 1
    log(f"converted {done} sequences")
    return {"sequences": done}


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--split", default="train")
    a = ap.parse_args()
    convert(a.src, a.dst, a.size, a.frames, a.split)
