"""Known-answer tests that pin the oracle.  The reference has no fixtures for
this path (parity unpinned), so these analytic cases are what anchors both oracle restatements."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import gdkvm_oracle as O
from tests.util import make_kpff_inputs, make_scan_inputs

RULES = [O.RULE_GATED_LINEAR, O.RULE_DELTA_PARALLEL, O.RULE_DELTA_SEQUENTIAL]


def _unit(d, i):
    e = np.zeros(d); e[i] = 1.0
    return e


def _one(q, k, v, a, b, **kw):
    """Single clip/head convenience: q,k [T,N,Dk] v [T,N,Dv] a [T] b [T,N]."""
    R, S = O.scan(q[None, :, :, None], k[None, :, :, None], v[None, :, :, None], a[None, :, None],
                  b[None, :, :, None], **kw)
    return R[0, :, :, 0], S[0, 0]


def test_k1_recall():
    Dk, Dv = 8, 5
    k = _unit(Dk, 3)[None, None].repeat(2, 0)            # T=2, N=1
    v = np.arange(1, Dv + 1, dtype=float)[None, None].repeat(2, 0)
    R, S = _one(k, k, v, np.ones(2), np.ones((2, 1)))
    assert np.array_equal(R[0, 0], np.zeros(Dv))         # frame 0 reads the empty state
    assert np.array_equal(R[1, 0], v[0, 0])              # frame 1 recalls exactly what frame 0 wrote


def test_k2_overwrite_vs_accumulate():
    Dk, Dv = 4, 3
    k = _unit(Dk, 1)[None, None].repeat(3, 0)
    v = np.stack([np.full(Dv, 1.0), np.full(Dv, 5.0), np.zeros(Dv)])[:, None]
    a, b = np.ones(3), np.ones((3, 1))
    R, _ = _one(k, k, v, a, b, rule=O.RULE_DELTA_SEQUENTIAL)
    assert np.array_equal(R[2, 0], v[1, 0])              # delta rule: second write replaces the first
    R, _ = _one(k, k, v, a, b, rule=O.RULE_GATED_LINEAR)
    assert np.array_equal(R[2, 0], v[0, 0] + v[1, 0])    # purely additive rule accumulates


def test_k3_orthonormal_keys():
    rng = np.random.default_rng(0)
    Dk, Dv, n = 16, 7, 16
    Qm, _ = np.linalg.qr(rng.standard_normal((Dk, Dk)))
    k = Qm[:n][None]                                      # T=1 frame of n orthonormal keys
    v = rng.standard_normal((1, n, Dv))
    k2 = np.concatenate([k, k]); v2 = np.concatenate([v, v])
    for rule in RULES:
        R, _ = _one(k2, k2, v2, np.ones(2), np.ones((2, n)), rule=rule)
        np.testing.assert_allclose(R[1], v[0], atol=1e-12)


def test_k4_alpha_zero_resets():
    q, k, v, a, b = make_scan_inputs(1, 3, 6, 1, 8, 4, seed=1)
    a = a.copy(); a[0, 2, 0] = 0.0
    _, S = O.scan(q, k, v, a, b)
    _, S_last = O.scan(q[:, 2:], k[:, 2:], v[:, 2:], a[:, 2:], b[:, 2:])
    np.testing.assert_allclose(S, S_last, atol=1e-14)


def test_k5_beta_zero_is_pure_decay():
    q, k, v, a, b = make_scan_inputs(2, 2, 5, 2, 8, 4, seed=2)
    s0 = np.random.default_rng(3).standard_normal((2, 2, 8, 4))
    for rule in RULES:
        _, S = O.scan(q, k, v, a, np.zeros_like(b), s0=s0, rule=rule)
        a64 = a.astype(np.float64)
        np.testing.assert_allclose(S, s0 * a64[:, 0, :, None, None] * a64[:, 1, :, None, None], rtol=1e-14)


@pytest.mark.parametrize("rule", RULES)
def test_k6_chunk_carry_bit_identity(rule):
    q, k, v, a, b = make_scan_inputs(2, 6, 7, 2, 8, 5, seed=4)
    R, S = O.scan(q, k, v, a, b, rule=rule, dtype=np.float32)
    R1, S1 = O.scan(q[:, :2], k[:, :2], v[:, :2], a[:, :2], b[:, :2], rule=rule, dtype=np.float32)
    R2, S2 = O.scan(q[:, 2:], k[:, 2:], v[:, 2:], a[:, 2:], b[:, 2:], s0=S1, rule=rule, dtype=np.float32)
    assert np.array_equal(np.concatenate([R1, R2], 1), R) and np.array_equal(S2, S)


@pytest.mark.parametrize("rule", RULES)
@pytest.mark.parametrize("corr", [0.0, 0.8])
def test_k7_wy_equals_sequential(rule, corr):
    q, k, v, a, b = make_scan_inputs(2, 4, 49, 1, 64, 32, seed=5, corr=corr)
    k = O.l2_normalize(k)
    Rs, Ss = O.scan(q, k, v, a, b, rule=rule, form="sequential")
    Rc, Sc = O.scan(q, k, v, a, b, rule=rule, form="chunk")
    np.testing.assert_allclose(Rc, Rs, atol=1e-9); np.testing.assert_allclose(Sc, Ss, atol=1e-9)
    Rc32, Sc32 = O.scan(q, k, v, a, b, rule=rule, form="chunk", dtype=np.float32)
    np.testing.assert_allclose(Rc32, Rs, atol=1e-4); np.testing.assert_allclose(Sc32, Ss, atol=1e-4)


def test_k8_dv_slice_independence_and_batch_equivariance():
    q, k, v, a, b = make_scan_inputs(3, 3, 9, 2, 8, 12, seed=6)
    R, S = O.scan(q, k, v, a, b)
    Rh, Sh = O.scan(q, k, v[..., 4:8], a, b)
    assert np.array_equal(Rh, R[..., 4:8]) and np.array_equal(Sh, S[..., 4:8])
    perm = [2, 0, 1]
    Rp, Sp = O.scan(q[perm], k[perm], v[perm], a[perm], b[perm])
    assert np.array_equal(Rp, R[perm]) and np.array_equal(Sp, S[perm])


def test_k9_argmax_ties_lowest_index():
    logits = np.zeros((1, 4, 2, 3), dtype=np.float32)
    logits[0, 2, 0, 0] = 1.0; logits[0, 3, 0, 0] = 1.0      # tie between 2 and 3 -> 2
    logits[0, 1, 1, 1] = -0.0                                # -0.0 == 0.0 -> still class 0
    m = O.argmax_mask(logits)
    assert m[0, 0, 0] == 2 and m[0, 1, 1] == 0 and m.sum() == 2
    mc, _ = c_oracle.argmax_dice(logits)
    assert np.array_equal(mc, m)


def test_k10_dice_edge_cases():
    z = np.zeros((1, 4, 4), dtype=np.uint8)
    one = np.ones((1, 4, 4), dtype=np.uint8)
    d = O.dice_from_counts(*O.dice_counts(z, z, 2))
    assert d[0, 0] == pytest.approx(1.0) and d[0, 1] == 1.0           # class 1: both empty -> 1
    d = O.dice_from_counts(*O.dice_counts(z, one, 2))
    assert d[0, 0] < 1e-6 and d[0, 1] < 1e-6                          # disjoint -> 0
    half = z.copy(); half[0, :2] = 1
    d = O.dice_from_counts(*O.dice_counts(half, half, 2))
    np.testing.assert_allclose(d, 1.0)


def test_k11_fp64_vs_fp32_cfg1():
    q, k, v, a, b = make_scan_inputs(1, 8, 49, 1, 64, 256, seed=0)
    R64, S64 = O.scan(q, k, v, a, b)
    R32, S32 = O.scan(q, k, v, a, b, dtype=np.float32)
    assert np.abs(R32 - R64).max() <= 1e-5 and np.abs(S32 - S64).max() <= 1e-5


# ------------------------------------------------------------- C restatement vs numpy restatement
@pytest.mark.parametrize("rule", RULES)
@pytest.mark.parametrize("flags", [0, 3])
def test_c_oracle_matches_numpy_scan(rule, flags):
    q, k, v, a, b = make_scan_inputs(2, 3, 10, 2, 16, 8, seed=7, normalized=not flags, logits=bool(flags))
    s0 = np.random.default_rng(8).standard_normal((2, 2, 16, 8)).astype(np.float32)
    R, S = O.scan(q, k, v, a, b, s0=s0, rule=rule, flags=flags)
    Rc, Sc = c_oracle.scan(q, k, v, a, b, s0=s0, rule=rule, flags=flags, math="f64")
    np.testing.assert_allclose(Rc, R, atol=2e-6); np.testing.assert_allclose(Sc, S, atol=2e-6)
    Rc, Sc = c_oracle.scan(q, k, v, a, b, s0=s0, rule=rule, flags=flags, math="f32")
    np.testing.assert_allclose(Rc, R, atol=1e-4); np.testing.assert_allclose(Sc, S, atol=1e-4)


def test_c_oracle_empty_inputs():
    for shape in [(0, 2, 4), (2, 0, 4), (2, 2, 0)]:
        B, T, N = shape
        q, k, v, a, b = make_scan_inputs(B, T, N, 1, 8, 4)
        s0 = np.ones((B, 1, 8, 4), dtype=np.float32)
        R, S = c_oracle.scan(q, k, v, a, b, s0=s0)
        assert R.shape == (B, T, N, 1, 4)
        Rn, Sn = O.scan(q, k, v, a, b, s0=s0)
        np.testing.assert_allclose(S, Sn, atol=1e-6)


@pytest.mark.parametrize("hw", [(7, 7), (4, 4), (5, 3), (1, 1)])
def test_c_oracle_matches_numpy_kpff(hw):
    h, w = hw
    L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(3, h, w, 8, 12, 16, seed=9)
    F = O.kpff(L, G, P, Wa, ba, Wl, Wg, h, w)
    Fc = c_oracle.kpff(L, G, P, Wa, ba, Wl, Wg, h, w)
    np.testing.assert_allclose(Fc, F, atol=2e-6)


def test_kpff_known_answers():
    # zero mixing weights -> F == P ; constant G -> multi-scale pooling is the identity
    L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(2, 7, 7, 8, 12, 16, seed=10)
    F = O.kpff(L, G, P, Wa, ba, 0 * Wl, 0 * Wg, 7, 7)
    assert np.array_equal(F, P.astype(np.float64))
    Gc = np.broadcast_to(G[:, :1], G.shape)
    np.testing.assert_allclose(O.multiscale_pool(Gc.astype(np.float64), 7, 7), Gc, atol=1e-12)
    # pooling preserves the frame mean only when cells tile the grid exactly (4x4 with s in {1,2,4})
    G4 = np.random.default_rng(0).standard_normal((1, 16, 3))
    np.testing.assert_allclose(O.multiscale_pool(G4, 4, 4).mean(1), G4.mean(1), atol=1e-12)


def test_argmax_dice_c_matches_numpy():
    rng = np.random.default_rng(11)
    logits = rng.standard_normal((5, 4, 9, 11)).astype(np.float32)
    logits[:, :, ::3] = np.round(logits[:, :, ::3])          # provoke exact ties
    target = rng.integers(0, 4, (5, 9, 11)).astype(np.uint8)
    m, counts = c_oracle.argmax_dice(logits, target)
    assert np.array_equal(m, O.argmax_mask(logits))
    i, p, t = O.dice_counts(m, target, 4)
    assert np.array_equal(counts[..., 0], i) and np.array_equal(counts[..., 1], p) and np.array_equal(counts[..., 2], t)


def test_bf16_rounding_helper():
    import torch
    x = np.random.default_rng(12).standard_normal(4096).astype(np.float32) * 100
    ref = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(O.to_bf16_f32(x), ref)


# ------------------------------------------------ the differentiable torch restatement (gradient oracle)
@pytest.mark.parametrize("rule", RULES)
def test_torch_ref_matches_numpy_oracle(rule):
    import torch
    from oracle import torch_ref as TR
    q, k, v, a, b = make_scan_inputs(2, 3, 9, 2, 16, 8, seed=21, normalized=False, logits=True)
    s0 = np.random.default_rng(22).standard_normal((2, 2, 16, 8))
    R, S = O.scan(q, k, v, a, b, s0=s0, rule=rule, flags=3)
    Rt, St = TR.scan(*(torch.from_numpy(x).double() for x in (q, k, v, a, b)), torch.from_numpy(s0), rule, 3)
    np.testing.assert_allclose(Rt.numpy(), R, atol=1e-12); np.testing.assert_allclose(St.numpy(), S, atol=1e-12)
    L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(2, 5, 6, 8, 12, 16, seed=23)
    F = O.kpff(L, G, P, Wa, ba, Wl, Wg, 5, 6)
    Ft = TR.kpff(*(torch.from_numpy(x).double() for x in (L, G, P, Wa, ba, Wl, Wg)), 5, 6)
    np.testing.assert_allclose(Ft.numpy(), F, atol=1e-12)


@pytest.mark.parametrize("rule", RULES)
@pytest.mark.parametrize("flags", [0, 3])
def test_backward_algorithm_matches_autograd(rule, flags):
    """The reverse recurrence + per-frame assembly (oracle/bwd_ref.py, the algorithm of the HIP backward) against
    autograd through the token-sequential definition."""
    import torch
    from oracle import bwd_ref
    from oracle import torch_ref as TR
    q, k, v, a, b = make_scan_inputs(2, 3, 7, 2, 8, 6, seed=31 + rule, normalized=not flags, logits=bool(flags), corr=0.5)
    rng = np.random.default_rng(32)
    s0 = rng.standard_normal((2, 2, 8, 6)); dR = rng.standard_normal((2, 3, 7, 2, 6)); dST = rng.standard_normal((2, 2, 8, 6))
    ts = [torch.from_numpy(np.asarray(x, np.float64)).requires_grad_() for x in (q, k, v, a, b, s0)]
    R, S = TR.scan(*ts, rule, flags)
    (R * torch.from_numpy(dR)).sum().add((S * torch.from_numpy(dST)).sum()).backward()
    got = bwd_ref.scan_backward(q, k, v, a, b, s0, dR, dST, rule, flags)
    for g, t in zip(got, ts):
        np.testing.assert_allclose(g, t.grad.numpy(), atol=1e-9)


def test_upsample_argmax_oracle_matches_torch_interpolate():
    import torch
    rng = np.random.default_rng(61)
    logits = rng.standard_normal((2, 3, 28, 28)).astype(np.float32)
    m, _ = c_oracle.upsample_argmax_dice(logits, 112, 112)
    up = torch.nn.functional.interpolate(torch.from_numpy(logits), size=(112, 112), mode="bilinear", align_corners=False)
    assert (up.argmax(1).numpy() != m).mean() <= 1e-4          # same formula; only FMA-contraction-level near-ties may differ
    same, _ = c_oracle.upsample_argmax_dice(logits, 28, 28)     # identity scale == plain argmax
    assert np.array_equal(same, O.argmax_mask(logits))


# ---------------------------------------------------------------------------------------- the `normalizer` flag of SURVEY A.1 (round 6)
def test_k12_normalizer_known_answers():
    """K12: z carries "the same recurrence on v == 1", the read-out is R / (|q . z| + eps).  Orthonormal keys, alpha = beta = 1:
    z = the sum of the written keys; reading with one of them leaves the value unchanged (q . z = 1), reading with the normalised sum
    of two of them returns the MEAN of their values (the un-normalised read-out is (v1 + v2) / sqrt 2, q . z = sqrt 2): the normaliser
    turns the read into a convex combination.  Both restatements (numpy: one more value channel; C: z carried explicitly)."""
    Dk, Dv = 8, 4
    k0 = np.stack([_unit(Dk, 1), _unit(Dk, 4)])                       # frame 0 writes e1 -> v1, e4 -> v2
    v0 = np.stack([np.arange(1.0, Dv + 1), 10 * np.arange(1.0, Dv + 1)])
    q1 = np.stack([_unit(Dk, 1), (_unit(Dk, 1) + _unit(Dk, 4)) / np.sqrt(2)])
    k = np.stack([k0, np.stack([_unit(Dk, 6), _unit(Dk, 7)])])
    q = np.stack([np.zeros((2, Dk)), q1])
    v = np.stack([v0, np.zeros((2, Dv))])
    args = (q[None, :, :, None], k[None, :, :, None], v[None, :, :, None], np.ones((1, 2, 1)), np.ones((1, 2, 2, 1)))
    for rule in (O.RULE_GATED_LINEAR, O.RULE_DELTA_SEQUENTIAL):
        R, S, z = O.scan_normalizer(*args, rule=rule, eps=1e-6)
        np.testing.assert_allclose(R[0, 1, 0, 0], v0[0] / (1 + 1e-6), rtol=1e-12)
        np.testing.assert_allclose(R[0, 1, 1, 0], 0.5 * (v0[0] + v0[1]), rtol=1e-6)
        assert np.array_equal(R[0, 0], np.zeros((2, 1, Dv)))              # empty memory: 0 / (0 + eps)
        assert np.array_equal(z[0, 0], _unit(Dk, 1) + _unit(Dk, 4) + _unit(Dk, 6) + _unit(Dk, 7))
        Rc, Sc, zc = c_oracle.scan_normalizer(*args, None, None, rule, 0, 1e-6, math="f64")
        np.testing.assert_allclose(Rc, R, atol=1e-6)
        np.testing.assert_allclose(zc, z, atol=0)
        np.testing.assert_allclose(Sc, S, atol=1e-6)


@pytest.mark.parametrize("rule", RULES)
def test_normalizer_restatements_agree_and_carry(rule):
    """numpy (augmented value channel) == C (explicit z) on random positive keys / queries (|q . z| away from 0), fp64 and fp32 arithmetic;
    chunked calls with (S, z) carried equal one call; the plain scan is untouched by the refactoring (S equals the normalizer-free S)."""
    rng = np.random.default_rng(7 + rule)
    B, T, N, Hh, Dk, Dv = 2, 5, 6, 2, 8, 5
    q = np.abs(rng.standard_normal((B, T, N, Hh, Dk))).astype(np.float32)
    k = np.abs(rng.standard_normal((B, T, N, Hh, Dk))).astype(np.float32)
    v = rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32)
    a = rng.normal(2, 1, (B, T, Hh)).astype(np.float32)
    b = rng.standard_normal((B, T, N, Hh)).astype(np.float32)
    R, S, z = O.scan_normalizer(q, k, v, a, b, None, None, rule, 3)
    Rc, Sc, zc = c_oracle.scan_normalizer(q, k, v, a, b, None, None, rule, 3, math="f64")
    assert np.abs(R - Rc).max() <= 1e-5 * max(1.0, np.abs(R).max()) and np.abs(S - Sc).max() <= 1e-6 and np.abs(z - zc).max() <= 1e-6
    Rf, Sf, zf = c_oracle.scan_normalizer(q, k, v, a, b, None, None, rule, 3, math="f32")
    assert np.abs(Rf - Rc).max() <= 1e-3 * max(1.0, np.abs(Rc).max()) and np.abs(Sf - Sc).max() <= 1e-4
    _, S_plain = c_oracle.scan(q, k, v, a, b, None, rule, 3, math="f64")
    assert np.array_equal(S_plain, Sc)
    R1, S1, z1 = c_oracle.scan_normalizer(q[:, :2], k[:, :2], v[:, :2], a[:, :2], b[:, :2], None, None, rule, 3)
    R2, S2, z2 = c_oracle.scan_normalizer(q[:, 2:], k[:, 2:], v[:, 2:], a[:, 2:], b[:, 2:], S1, z1, rule, 3)
    # (the carried S and z are rounded to fp32 on the way: where |q . z| is small the quotient amplifies that rounding)
    assert np.abs(np.concatenate([R1, R2], 1) - Rc).max() <= 1e-4 * max(1.0, np.abs(Rc).max()) and np.abs(S2 - Sc).max() <= 1e-6 and np.abs(z2 - zc).max() <= 1e-6


def test_mask_cell_mean_is_adaptive_avg_pool():
    """oracle.mask_cell_mean (the pooled foreground indicator the module's mask embedding multiplies) == torch's adaptive_avg_pool2d, on
    sizes that divide and sizes that do not."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    for (H, W, h, w) in [(112, 112, 7, 7), (256, 256, 16, 16), (30, 58, 4, 7), (15, 13, 2, 3)]:
        m = rng.random((3, H, W)) > 0.6
        got = O.mask_cell_mean(m, h, w)
        want = F.adaptive_avg_pool2d(torch.from_numpy(m.astype(np.float64))[:, None], (h, w)).reshape(3, h * w).numpy()
        np.testing.assert_allclose(got, want, atol=1e-15)


def test_c_oracle_is_clean_under_address_and_ub_sanitizers():
    """make -C oracle SAN=1 builds the scalar C oracle with -fsanitize=address,undefined together with oracle_selftest.c, which walks every
    entry point over ragged, exactly sized heap buffers (zero frames, zero tokens, carried states, every rule, the normalizer) and checks
    K1 / K12: a clean exit means no out-of-bounds access, no use of uninitialised scratch the sanitizers can see, no undefined arithmetic.
    (CPU only: GPU sanitizers are not available on this pool)"""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-C", here, "-s", "SAN=1"])
    out = subprocess.run([os.path.join(here, "_build", "oracle_selftest_san")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "oracle_selftest: ok" in out.stdout, out.stdout + out.stderr
