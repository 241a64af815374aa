"""Seeded synthetic inputs for the memory path."""
import numpy as np


def make_scan_inputs(B, T, N, Hh, Dk, Dv, seed=0, normalized=True, logits=False, corr=0.0):
    """q,k ~ N(0,1) (L2-normalised unless the kernel is asked to do it), v ~ N(0,1),
    alpha = sigmoid(N(2,1)), beta = sigmoid(N(0,1)).  ``corr`` blends neighbouring tokens' keys
    (CNN feature maps are spatially correlated, which stresses the triangular solve)."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((B, T, N, Hh, Dk)).astype(np.float32)
    k = rng.standard_normal((B, T, N, Hh, Dk)).astype(np.float32)
    if corr > 0 and N > 1:
        base = rng.standard_normal((B, T, 1, Hh, Dk)).astype(np.float32)
        k = ((1 - corr) * k + corr * base).astype(np.float32)
    v = rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32)
    a = rng.normal(2.0, 1.0, (B, T, Hh)).astype(np.float32)
    b = rng.normal(0.0, 1.0, (B, T, N, Hh)).astype(np.float32)
    if normalized:
        q = q / np.sqrt((q.astype(np.float64) ** 2).sum(-1, keepdims=True) + 1e-12).astype(np.float32)
        k = k / np.sqrt((k.astype(np.float64) ** 2).sum(-1, keepdims=True) + 1e-12).astype(np.float32)
        q = q.astype(np.float32); k = k.astype(np.float32)
    if not logits:
        a = (1 / (1 + np.exp(-a))).astype(np.float32)
        b = (1 / (1 + np.exp(-b))).astype(np.float32)
    return q, k, v, a, b


def make_kpff_inputs(BT, h, w, Ck, Cv, Cp, seed=0):
    rng = np.random.default_rng(seed)
    N = h * w
    L = rng.standard_normal((BT, N, Ck)).astype(np.float32)
    G = rng.standard_normal((BT, N, Cv)).astype(np.float32)
    P = rng.standard_normal((BT, N, Cp)).astype(np.float32)
    Cin = Cp + Ck + Cv
    Wa = (rng.standard_normal((2 * Cp, Cin)) / np.sqrt(Cin)).astype(np.float32)
    ba = (0.1 * rng.standard_normal((2 * Cp,))).astype(np.float32)
    Wl = (rng.standard_normal((Cp, Ck)) / np.sqrt(Ck)).astype(np.float32)
    Wg = (rng.standard_normal((Cp, Cv)) / np.sqrt(Cv)).astype(np.float32)
    return L, G, P, Wa, ba, Wl, Wg
