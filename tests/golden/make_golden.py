#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded input/expected-output vectors for the memory path.

The reference snapshot holds no implementation or fixture for this path ("parity unpinned"), so
these vectors come from the repo's own fp64 oracle (oracle/gdkvm_oracle.py, numpy), which is pinned by the
analytic known-answer tests in tests/test_oracle_kat.py.  They freeze the SPEC-v0 semantics: any later change to
the oracle, the C restatement or the HIP kernels that moves a result shows up against these files.
    python tests/golden/make_golden.py          (rewrites the .npz files in place)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gdkvm_oracle as O          # noqa: E402
from tests.util import make_kpff_inputs, make_scan_inputs  # noqa: E402


def main():
    # scan: every rule, raw inputs + prologue flags, a carried-in state, N off the 16-token tile edge
    for rule in (0, 1, 2):
        q, k, v, a, b = make_scan_inputs(2, 3, 21, 1, 64, 32, seed=100 + rule, normalized=False, logits=True, corr=0.6)
        s0 = (0.2 * np.random.default_rng(7).standard_normal((2, 1, 64, 32))).astype(np.float32)
        R, S = O.scan(q, k, v, a, b, s0=s0, rule=rule, flags=3)
        np.savez_compressed(os.path.join(HERE, f"scan_rule{rule}.npz"), q=q, k=k, v=v, alpha=a, beta=b, s0=s0,
                            rule=rule, flags=3, R=R.astype(np.float32), S=S.astype(np.float32))
    # scan: cfg1-shaped frame count (49 tokens, two heads), pre-normalised inputs, no flags
    q, k, v, a, b = make_scan_inputs(1, 4, 49, 2, 64, 16, seed=104)
    R, S = O.scan(q, k, v, a, b)
    np.savez_compressed(os.path.join(HERE, "scan_n49_h2.npz"), q=q, k=k, v=v, alpha=a, beta=b, rule=2, flags=0,
                        R=R.astype(np.float32), S=S.astype(np.float32))
    # KPFF on the 7x7 grid of a 112x112 frame and on an 8x8 grid split... (two pooling regimes)
    for name, (h, w) in {"kpff_7x7": (7, 7), "kpff_10x6": (10, 6)}.items():
        L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(2, h, w, 32, 64, 32, seed=h * 10 + w)
        F = O.kpff(L, G, P, Wa, ba, Wl, Wg, h, w)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), L=L, G=G, P=P, Wa=Wa, ba=ba, Wl=Wl, Wg=Wg, h=h, w=w,
                            F=F.astype(np.float32))
    # argmax + Dice with exact ties and an out-of-range label
    rng = np.random.default_rng(5)
    logits = O.to_bf16_f32(np.round(rng.standard_normal((3, 4, 12, 12)) * 2) / 2)
    target = rng.integers(0, 5, (3, 12, 12)).astype(np.uint8)
    mask = O.argmax_mask(logits)
    i, p, t = O.dice_counts(mask, target, 4)
    np.savez_compressed(os.path.join(HERE, "argmax_dice.npz"), logits=logits, target=target, mask=mask,
                        counts=np.stack([i, p, t], -1).astype(np.int32))


if __name__ == "__main__":
    main()
