#!/usr/bin/env python3
"""EchoNet-Dynamic download -> the .npz tree gdkvm_amd.data.EchoNetNpz reads.

The dataset as published (the "raw data" link of the reference's guide, /root/reference/website/src/pages/[lang]/reprod/index.astro:222):
    <src>/Videos/<FileName>.avi          112 x 112 grey videos
    <src>/FileList.csv                   FileName, EF, ESV, EDV, FrameHeight, FrameWidth, FPS, NumberOfFrames, Split (TRAIN / VAL / TEST)
    <src>/VolumeTracings.csv             FileName, X1, Y1, X2, Y2, Frame -- 21 rows per traced frame (long axis + 20 chords), two traced
                                         frames per video (end-diastole, end-systole)
Written:
    <dst>/<train|val|test>/<FileName>.npz   video [F,H,W] uint8, traced [2] int (frame indices, video order), masks [2,H,W] uint8 (0 / 1)

Decoding an .avi needs OpenCV (cv2); this image has none, so run the converter where the dataset was downloaded.  For videos already
decoded by other means, <src>/Videos/<FileName>.npy ([F,H,W] or [F,H,W,3] uint8) is taken instead of the .avi -- which is also how the
CPU tests exercise this script.  Nothing here touches a GPU.

    python tools/convert_echonet.py <src> <dst> [--limit N]
"""
import argparse
import csv
import os
import sys
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_video(path_no_ext: str) -> np.ndarray:
    """[F,H,W] uint8 grey frames of Videos/<name>: a pre-decoded .npy if present, else the .avi through OpenCV."""
    if os.path.exists(path_no_ext + ".npy"):
        v = np.load(path_no_ext + ".npy")
        return (v if v.ndim == 3 else v[..., 0]).astype(np.uint8)
    avi = path_no_ext + ".avi"
    if not os.path.exists(avi):
        raise FileNotFoundError(f"neither {avi} nor {path_no_ext}.npy exists")
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError("decoding .avi needs OpenCV (pip install opencv-python-headless) -- or decode to Videos/<FileName>.npy yourself") from e
    cap = cv2.VideoCapture(avi)
    frames = []
    while True:
        ok, fr = cap.read()
        if not ok:
            break
        frames.append(cv2.cvtColor(fr, cv2.COLOR_BGR2GRAY))
    cap.release()
    if not frames:
        raise RuntimeError(f"{avi}: no frame decoded")
    return np.stack(frames).astype(np.uint8)


def read_tracings(path: str):
    """{FileName without extension: {frame index: [(X1, Y1, X2, Y2), ...] in file order}}"""
    out = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = os.path.splitext(row["FileName"])[0]
            out[name][int(float(row["Frame"]))].append(tuple(float(row[k]) for k in ("X1", "Y1", "X2", "Y2")))
    return out


def convert(src: str, dst: str, limit: int = 0, log=print) -> dict:
    from gdkvm_amd.data import echonet_tracing_polygon, polygon_mask
    tracings = read_tracings(os.path.join(src, "VolumeTracings.csv"))
    done, skipped = defaultdict(int), []
    with open(os.path.join(src, "FileList.csv"), newline="") as f:
        rows = list(csv.DictReader(f))
    for row in rows[: limit or None]:
        name = os.path.splitext(row["FileName"])[0]
        split = row["Split"].strip().lower()
        tr = tracings.get(name)
        if not tr or len(tr) < 2:
            skipped.append((name, "fewer than two traced frames"))
            continue
        try:
            video = read_video(os.path.join(src, "Videos", name))
        except (FileNotFoundError, RuntimeError) as e:
            skipped.append((name, str(e)))
            continue
        traced = sorted(tr)[:2] if len(tr) == 2 else sorted(tr, key=lambda k: -len(tr[k]))[:2]
        traced = sorted(t for t in traced)
        if traced[-1] >= video.shape[0]:
            skipped.append((name, f"traced frame {traced[-1]} beyond the video's {video.shape[0]} frames"))
            continue
        h, w = video.shape[1:]
        masks = np.stack([polygon_mask(*echonet_tracing_polygon(tr[t]), h, w) for t in traced])
        os.makedirs(os.path.join(dst, split), exist_ok=True)
        np.savez_compressed(os.path.join(dst, split, name + ".npz"), video=video, traced=np.asarray(traced, np.int64), masks=masks)
        done[split] += 1
    log(f"converted {dict(done)}; skipped {len(skipped)}")
    for n, why in skipped[:20]:
        log(f"  skipped {n}: {why}")
    return {"converted": dict(done), "skipped": skipped}


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--limit", type=int, default=0)
    a = ap.parse_args()
    convert(a.src, a.dst, a.limit)
