#!/usr/bin/env python3
"""Randomized parity sweep of gdkvm_scan_fwd against the C oracle (fp64 math): shapes, rules, flags, dtypes, carried state.
    python tools/stress_scan.py [cases=150] [seed=0]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops
from oracle import c_oracle, gdkvm_oracle as O
from tests.util import make_scan_inputs

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"S": 0.0, "R": 0.0}
bad = 0
for i in range(cases):
    B = int(rng.integers(1, 4)); T = int(rng.integers(1, 12)); Hh = int(rng.integers(1, 3))
    N = int(rng.choice([1, 7, 16, 17, 49, 63, 64, 65, 100, 128, 130, 196, 256])); Dv = int(rng.choice([16, 32, 48, 64, 256]))
    if N > 64 and Dv == 256: Dv = 64                       # keep the scalar oracle quick
    rule = int(rng.integers(0, 3)); flags = int(rng.choice([0, 3])); bf = bool(rng.integers(0, 2)); with_state = bool(rng.integers(0, 2))
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=int(rng.integers(1 << 30)), normalized=not flags, logits=bool(flags),
                                     corr=float(rng.uniform(0, 0.9)))
    if bf: q, k, v = (O.to_bf16_f32(x) for x in (q, k, v))
    s0 = (rng.standard_normal((B, Hh, 64, Dv)) * 0.3).astype(np.float32) if with_state else None
    dt = torch.bfloat16 if bf else torch.float32
    t = [torch.from_numpy(x).cuda().to(dt) for x in (q, k, v)] + [torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()]
    R, S = ops.scan_fwd(*t, None if s0 is None else torch.from_numpy(s0).cuda(), rule=rule, flags=flags)
    Ro, So = c_oracle.scan(q, k, v, a, b, s0, rule, flags, math="f64")
    # delta_parallel is not contractive (I - sum b k k^T can have eigenvalues << -1): the state may grow by orders of
    # magnitude per frame, so errors are judged relative to the magnitude the oracle reaches
    scale = max(1.0, float(np.abs(So).max()), float(np.abs(Ro).max()) if Ro.size else 0.0)
    eS = float(np.abs(S.cpu().numpy() - So).max()) / scale
    dR = np.abs(R.float().cpu().numpy() - Ro) - (np.abs(Ro) * 2.0 ** -8 if bf else 0)
    eR = (float(dR.max()) if dR.size else 0.0) / scale
    worst["S"] = max(worst["S"], eS); worst["R"] = max(worst["R"], eR)
    if eS > 1e-4 or eR > 1e-4:
        bad += 1
        print(f"FAIL case {i}: B={B} T={T} N={N} Hh={Hh} Dv={Dv} rule={rule} flags={flags} bf16={bf} state={with_state}: dS={eS:.2e} dR={eR:.2e} (relative to {scale:.2e})")
print(f"{cases} cases, {bad} failures; worst state error {worst['S']:.2e}, worst read-out excess {worst['R']:.2e} (tolerance 1e-4, relative to max(1, |S|, |R|))")
sys.exit(1 if bad else 0)
