"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the fp64 numpy oracle):
CPU side checks both oracle restatements against them; GPU side checks the HIP kernels (through the C ABI)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import gdkvm_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCAN = sorted(glob.glob(os.path.join(G, "scan_*.npz")))
KPFF = sorted(glob.glob(os.path.join(G, "kpff_*.npz")))


def test_golden_files_present():
    assert len(SCAN) == 4 and len(KPFF) == 2 and os.path.exists(os.path.join(G, "argmax_dice.npz"))


@pytest.mark.parametrize("path", SCAN, ids=os.path.basename)
def test_oracles_reproduce_golden_scan(path):
    z = np.load(path)
    s0 = z["s0"] if "s0" in z else None
    R, S = O.scan(z["q"], z["k"], z["v"], z["alpha"], z["beta"], s0=s0, rule=int(z["rule"]), flags=int(z["flags"]))
    np.testing.assert_allclose(R, z["R"], atol=1e-6); np.testing.assert_allclose(S, z["S"], atol=1e-6)
    for math, tol in (("f64", 2e-6), ("f32", 1e-4)):
        Rc, Sc = c_oracle.scan(z["q"], z["k"], z["v"], z["alpha"], z["beta"], s0, int(z["rule"]), int(z["flags"]), math=math)
        assert np.abs(Rc - z["R"]).max() <= tol and np.abs(Sc - z["S"]).max() <= tol


@pytest.mark.parametrize("path", KPFF, ids=os.path.basename)
def test_oracles_reproduce_golden_kpff(path):
    z = np.load(path)
    Fc = c_oracle.kpff(z["L"], z["G"], z["P"], z["Wa"], z["ba"], z["Wl"], z["Wg"], int(z["h"]), int(z["w"]))
    assert np.abs(Fc - z["F"]).max() <= 2e-6


def test_oracles_reproduce_golden_argmax():
    z = np.load(os.path.join(G, "argmax_dice.npz"))
    m, c = c_oracle.argmax_dice(z["logits"], z["target"])
    assert np.array_equal(m, z["mask"]) and np.array_equal(c, z["counts"])


def _dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t.to(dtype) if dtype is not None else t


@pytest.mark.gpu
@pytest.mark.parametrize("path", SCAN, ids=os.path.basename)
def test_hip_reproduces_golden_scan(hip, path):
    z = np.load(path)
    s0 = _dev(z["s0"]) if "s0" in z else None
    R, S = hip.scan_fwd(_dev(z["q"]), _dev(z["k"]), _dev(z["v"]), _dev(z["alpha"]), _dev(z["beta"]), s0,
                        rule=int(z["rule"]), flags=int(z["flags"]))
    assert np.abs(R.cpu().numpy() - z["R"]).max() <= 1e-4 and np.abs(S.cpu().numpy() - z["S"]).max() <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("path", KPFF, ids=os.path.basename)
def test_hip_reproduces_golden_kpff(hip, path):
    z = np.load(path)
    F = hip.kpff_fwd(*(_dev(z[n]) for n in ("L", "G", "P", "Wa", "ba", "Wl", "Wg")), int(z["h"]), int(z["w"]))
    assert np.abs(F.cpu().numpy() - z["F"]).max() <= 1e-4


@pytest.mark.gpu
def test_hip_reproduces_golden_argmax(hip):
    z = np.load(os.path.join(G, "argmax_dice.npz"))
    for dt in (torch.float32, torch.bfloat16):
        m, c = hip.argmax_dice(_dev(z["logits"], dt), _dev(z["target"]))
        assert np.array_equal(m.cpu().numpy(), z["mask"]) and np.array_equal(c.cpu().numpy(), z["counts"])
