"""GPU parity: gdkvm_scan_fwd (HIP, through the C ABI) vs the CPU oracle on the same seeded inputs.
Tolerance: 1e-4 absolute on fp32 (BASELINE.json north_star); bf16 I/O is compared against the oracle fed
the same bf16-rounded inputs, with the bf16 output quantisation (2^-8 relative) added."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import gdkvm_oracle as O
from tests.util import make_scan_inputs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t.to(dtype) if dtype is not None else t


def _run(hip, q, k, v, a, b, s0=None, rule=2, flags=0, dtype=torch.float32):
    r, s = hip.scan_fwd(_dev(q, dtype), _dev(k, dtype), _dev(v, dtype), _dev(a), _dev(b),
                        None if s0 is None else _dev(s0), rule=rule, flags=flags)
    torch.cuda.synchronize()
    return r.float().cpu().numpy(), s.cpu().numpy()


@pytest.mark.parametrize("rule", [0, 1, 2])
@pytest.mark.parametrize("flags", [0, 3])
def test_scan_fp32_cfg1_shape(hip, rule, flags):
    """cfg1 shape (B=1,T=8,N=49,Dk=64,Dv=256) for every rule, with and without the a5 prologue."""
    q, k, v, a, b = make_scan_inputs(1, 8, 49, 1, 64, 256, seed=rule * 7 + flags, normalized=not flags,
                                     logits=bool(flags), corr=0.5)
    s0 = np.random.default_rng(1).standard_normal((1, 1, 64, 256)).astype(np.float32) * 0.1
    Rg, Sg = _run(hip, q, k, v, a, b, s0, rule, flags)
    Ro, So = c_oracle.scan(q, k, v, a, b, s0, rule, flags, math="f64")
    assert np.abs(Rg - Ro).max() <= TOL and np.abs(Sg - So).max() <= TOL


@pytest.mark.parametrize("shape", [
    (2, 3, 1, 1, 16), (2, 3, 16, 2, 32), (3, 2, 64, 1, 48), (2, 2, 65, 2, 16), (1, 2, 100, 1, 64),
    (1, 2, 128, 1, 32), (1, 2, 129, 1, 16), (1, 3, 256, 1, 64), (8, 2, 49, 1, 32), (16, 2, 7, 1, 16)])
def test_scan_fp32_ragged_shapes(hip, shape):
    """token counts on and off the 16-token tile edges, several heads, max N, B%8==0 (XCD mapping)."""
    B, T, N, Hh, Dv = shape
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=sum(shape), corr=0.7)
    Rg, Sg = _run(hip, q, k, v, a, b)
    Ro, So = c_oracle.scan(q, k, v, a, b, None, 2, 0, math="f64")
    assert Rg.shape == Ro.shape
    assert np.abs(Rg - Ro).max() <= TOL and np.abs(Sg - So).max() <= TOL


@pytest.mark.parametrize("case", [(8, 3, 49, 1, 320), (3, 4, 49, 1, 512), (8, 2, 33, 2, 320), (16, 2, 64, 1, 64), (8, 2, 17, 1, 272)])
def test_scan_bf16_wide_values_and_launch_forms(hip, case):
    """bf16 I/O, frames of at most 64 tokens, MORE than sixteen 16-column value tiles: the frame-parallel kernel's register-rich build
    requests a wave's first four V tiles at entry and refills the four register pairs for the tiles beyond (Dv 272 / 320 / 512) -- on both
    launch forms of that kernel (clips a multiple of 8 with one head go out as a 3-D grid whose x is the XCD, everything else 1-D), with a
    carried state, and at 64 tokens (no padding token: every k-step of the last block is kept)."""
    B, T, N, Hh, Dv = case
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=sum(case), normalized=False, logits=True, corr=0.5)
    t = [_dev(x, torch.bfloat16) for x in (q, k, v)] + [_dev(a), _dev(b)]
    s0 = _dev((0.3 * np.random.default_rng(5).standard_normal((B, Hh, 64, Dv))).astype(np.float32))
    R, S = hip.scan_fwd(*t, s0, rule=2, flags=3)
    Ro, So = c_oracle.scan(*(O.to_bf16_f32(x) for x in (q, k, v)), a, b, s0.cpu().numpy(), 2, 3)
    assert np.abs(S.cpu().numpy() - So).max() <= 1e-4
    assert np.abs(R.float().cpu().numpy() - Ro).max() <= 2.0 ** -7 * max(1.0, float(np.abs(Ro).max()))      # bf16 read-out


@pytest.mark.parametrize("case", [(2, 3, 65, 1, 64), (1, 4, 100, 2, 128), (3, 2, 130, 1, 192), (2, 2, 250, 1, 256), (1, 3, 256, 1, 256), (1, 2, 1024, 1, 64),
                                  (8, 2, 130, 1, 64), (16, 17, 70, 1, 128)])      # (8 / 16 clips: the XCD-aware frame order; 272 frames: the default choice)
@pytest.mark.parametrize("rule", [0, 2])
def test_fused_chunk_walk_is_the_chunk_parallel_path_bit_for_bit(hip, case, rule, monkeypatch):
    """Frames of more than 64 tokens, bf16: gdr_prepm_kernel walking a frame's chunks in ONE workgroup with the running map in
    registers (chosen when there are at least as many frames as CUs; forced here with GDKVM_PREP_FUSE=1) computes the same sums in
    the same order as chunk-parallel workgroups + gdr_compose_kernel (=0): read-outs and states are bit-identical, and right
    against the oracle.  Shapes the walk does not serve (fp32 I/O, Dv not a multiple of 64, delta_parallel) keep the other path."""
    B, T, N, Hh, Dv = case
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=sum(case) + rule, normalized=False, logits=True, corr=0.5)
    t = [_dev(x, torch.bfloat16) for x in (q, k, v)] + [_dev(a), _dev(b)]
    s0 = _dev((0.3 * np.random.default_rng(3).standard_normal((B, Hh, 64, Dv))).astype(np.float32))
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GDKVM_PREP_FUSE", mode)
        out[mode] = hip.scan_fwd(*t, s0, rule=rule, flags=3)
    assert torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][1], out["1"][1])
    if N <= 256:
        Ro, So = c_oracle.scan(*(O.to_bf16_f32(x) for x in (q, k, v)), a, b, s0.cpu().numpy(), rule, 3)
        assert np.abs(out["1"][1].cpu().numpy() - So).max() <= 1e-4
        assert np.all(np.abs(out["1"][0].float().cpu().numpy() - Ro) <= 1e-4 + np.abs(Ro) * 2.0 ** -7)
    monkeypatch.setenv("GDKVM_PREP_FUSE", "1")
    for alt in ([x.float() for x in t[:3]] + t[3:], [t[0], t[1], t[2][..., :48].contiguous(), t[3], t[4]]):     # fp32 I/O; Dv = 48
        monkeypatch.setenv("GDKVM_PREP_FUSE", "1")
        r1, s1 = hip.scan_fwd(*alt, rule=rule, flags=3)
        monkeypatch.setenv("GDKVM_PREP_FUSE", "0")
        r0, s0_ = hip.scan_fwd(*alt, rule=rule, flags=3)
        assert torch.equal(r0, r1) and torch.equal(s0_, s1)


@pytest.mark.parametrize("rule", [0, 1, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_chunked_frames_every_rule(hip, rule, dtype):
    """Frames of more than 64 tokens are folded per 64-token chunk and the chunks' affine maps composed
    (gdr_compose_kernel): 130 tokens = 64 + 64 + 2, every rule, with the a5 prologue and a carried state."""
    q, k, v, a, b = make_scan_inputs(2, 3, 130, 2, 64, 48, seed=40 + rule, normalized=False, logits=True, corr=0.6)
    s0 = np.random.default_rng(2).standard_normal((2, 2, 64, 48)).astype(np.float32) * 0.1
    if dtype == torch.bfloat16:
        q, k, v = (O.to_bf16_f32(x) for x in (q, k, v))
    Rg, Sg = _run(hip, q, k, v, a, b, s0, rule, 3, dtype=dtype)
    Ro, So = c_oracle.scan(q, k, v, a, b, s0, rule, 3, math="f64")
    assert np.abs(Sg - So).max() <= TOL
    assert np.all(np.abs(Rg - Ro) <= TOL + (np.abs(Ro) * 2.0 ** -8 if dtype == torch.bfloat16 else 0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_every_clip_length_1_to_13(hip, dtype):
    """The serial kernel unrolls its frame loop (10 frames in the bf16 arm, 4 in the fp32 arm, ring of 5 / 4 slots, read
    waves by 4): every residue of T must take the right peeled tail, ring slot and operand buffer."""
    for T in range(1, 14):
        q, k, v, a, b = make_scan_inputs(1, T, 20, 1, 64, 32, seed=100 + T, normalized=False, logits=True, corr=0.5)
        if dtype == torch.bfloat16:
            q, k, v = (O.to_bf16_f32(x) for x in (q, k, v))
        s0 = np.random.default_rng(T).standard_normal((1, 1, 64, 32)).astype(np.float32) * 0.2
        Rg, Sg = _run(hip, q, k, v, a, b, s0, 2, 3, dtype=dtype)
        Ro, So = c_oracle.scan(q, k, v, a, b, s0, 2, 3, math="f64")
        assert np.abs(Sg - So).max() <= TOL, T
        assert np.all(np.abs(Rg - Ro) <= TOL + (np.abs(Ro) * 2.0 ** -8 if dtype == torch.bfloat16 else 0)), T


def test_scan_matches_numpy_oracle_too(hip):
    q, k, v, a, b = make_scan_inputs(1, 3, 20, 2, 64, 16, seed=5)
    Rg, Sg = _run(hip, q, k, v, a, b)
    Ro, So = O.scan(q, k, v, a, b)
    assert np.abs(Rg - Ro).max() <= TOL and np.abs(Sg - So).max() <= TOL


def test_scan_known_answers_on_gpu(hip):
    """K1/K2/K3 of SURVEY.md A.6 straight on the kernel."""
    Dk, Dv = 64, 16
    k = np.zeros((1, 3, 1, 1, Dk), np.float32); k[..., 5] = 1
    v = np.stack([np.full(Dv, 1.0), np.full(Dv, 5.0), np.zeros(Dv)]).astype(np.float32)[None, :, None, None]
    a = np.ones((1, 3, 1), np.float32); b = np.ones((1, 3, 1, 1), np.float32)
    R, _ = _run(hip, k, k, v, a, b, rule=2)
    assert np.array_equal(R[0, 0, 0, 0], np.zeros(Dv)) and np.array_equal(R[0, 1, 0, 0], v[0, 0, 0, 0])
    assert np.array_equal(R[0, 2, 0, 0], v[0, 1, 0, 0])                   # overwrite
    R, _ = _run(hip, k, k, v, a, b, rule=0)
    assert np.array_equal(R[0, 2, 0, 0], v[0, 0, 0, 0] + v[0, 1, 0, 0])  # accumulate
    Qm, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((Dk, Dk)))
    kk = np.repeat(Qm[None, None, :49, None, :].astype(np.float32), 2, 1)
    vv = np.repeat(np.random.default_rng(1).standard_normal((1, 1, 49, 1, Dv)).astype(np.float32), 2, 1)
    R, _ = _run(hip, kk, kk, vv, np.ones((1, 2, 1), np.float32), np.ones((1, 2, 49, 1), np.float32))
    np.testing.assert_allclose(R[0, 1], vv[0, 0], atol=2e-5)              # orthonormal recall


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_chunk_carry_bit_identity(hip, dtype):
    """K6 / cfg5 property: chunked calls with the state carried == one call, bit for bit."""
    q, k, v, a, b = make_scan_inputs(2, 12, 49, 1, 64, 64, seed=11)
    tq, tk, tv = (_dev(x, dtype) for x in (q, k, v)); ta, tb = _dev(a), _dev(b)
    R, S = hip.scan_fwd(tq, tk, tv, ta, tb)
    s = None; parts = []
    for lo, hi in [(0, 1), (1, 5), (5, 12)]:
        r, s = hip.scan_fwd(tq[:, lo:hi].contiguous(), tk[:, lo:hi].contiguous(), tv[:, lo:hi].contiguous(),
                            ta[:, lo:hi].contiguous(), tb[:, lo:hi].contiguous(), s)
        parts.append(r)
    assert torch.equal(torch.cat(parts, 1), R) and torch.equal(s, S)


def test_scan_deterministic_and_slice_independent(hip):
    q, k, v, a, b = make_scan_inputs(2, 4, 49, 1, 64, 64, seed=12)
    R1, S1 = _run(hip, q, k, v, a, b); R2, S2 = _run(hip, q, k, v, a, b)
    assert np.array_equal(R1, R2) and np.array_equal(S1, S2)
    Rh, Sh = _run(hip, q, k, np.ascontiguousarray(v[..., 16:48]), a, b)      # K8: Dv columns never interact
    assert np.array_equal(Rh, R1[..., 16:48]) and np.array_equal(Sh, S1[..., 16:48])


@pytest.mark.parametrize("rule", [0, 2])
def test_scan_bf16_io(hip, rule):
    q, k, v, a, b = make_scan_inputs(2, 6, 49, 1, 64, 64, seed=13, normalized=False, logits=True)
    qb, kb, vb = (O.to_bf16_f32(x) for x in (q, k, v))
    Rg, Sg = _run(hip, q, k, v, a, b, None, rule, 3, dtype=torch.bfloat16)
    Ro, So = c_oracle.scan(qb, kb, vb, a, b, None, rule, 3, math="f64")
    assert np.abs(Sg - So).max() <= TOL                        # the state is fp32 end to end
    assert np.all(np.abs(Rg - Ro) <= TOL + np.abs(Ro) * 2.0 ** -8)


def test_scan_empty_and_degenerate(hip):
    for B, T, N in [(0, 2, 4), (2, 0, 4), (2, 2, 0)]:
        q, k, v, a, b = make_scan_inputs(B, T, N, 1, 64, 16)
        s0 = np.full((B, 1, 64, 16), 0.5, np.float32)
        R, S = _run(hip, q, k, v, a, b, s0)
        assert R.shape == (B, T, N, 1, 16)
        _, So = c_oracle.scan(q, k, v, a, b, s0)
        if B:
            np.testing.assert_allclose(S, So, atol=1e-6)       # N=0 / T=0: pure decay / identity


def test_scan_error_codes(hip):
    q, k, v, a, b = make_scan_inputs(1, 1, 4, 1, 264, 16)          # (Dk below 64 through zero channels, 72 .. 256 on the general kernel; beyond: refused)
    with pytest.raises(hip.GdkvmError, match="Dk"):
        _run(hip, q, k, v, a, b)
    q, k, v, a, b = make_scan_inputs(1, 1, 4, 1, 64, 24)
    with pytest.raises(hip.GdkvmError, match="Dv"):
        _run(hip, q, k, v, a, b)
    q, k, v, a, b = make_scan_inputs(1, 1, 4, 1, 64, 16)
    with pytest.raises(hip.GdkvmError, match="device"):
        hip.scan_fwd(*(torch.from_numpy(x) for x in (q, k, v, a, b)))
    with pytest.raises(hip.GdkvmError, match="workspace"):
        hip.scan_fwd(_dev(q), _dev(k), _dev(v), _dev(a), _dev(b), workspace=torch.empty(64, dtype=torch.uint8, device="cuda"))


def test_scan_cfg2_full_size_properties(hip):
    """BASELINE cfg2 size (B=16,T=32,N=49,Dv=256, bf16): too big for the scalar oracle to be quick on every
    clip, so check size-independent properties: clips are independent (a sub-batch reproduces its rows bit for
    bit), and two clips are checked against the oracle outright."""
    q, k, v, a, b = make_scan_inputs(16, 32, 49, 1, 64, 256, seed=1, normalized=False, logits=True)
    t = [_dev(x, torch.bfloat16) for x in (q, k, v)] + [_dev(a), _dev(b)]
    R, S = hip.scan_fwd(*t, flags=3)
    Rs, Ss = hip.scan_fwd(*(x[8:16].contiguous() for x in t), flags=3)
    assert torch.equal(Rs, R[8:16]) and torch.equal(Ss, S[8:16])
    for c in (0, 15):
        Ro, So = c_oracle.scan(*(O.to_bf16_f32(x[c:c + 1]) for x in (q, k, v)), a[c:c + 1], b[c:c + 1], None, 2, 3)
        assert np.abs(S[c:c + 1].cpu().numpy() - So).max() <= TOL
        Rg = R[c:c + 1].float().cpu().numpy()
        assert np.all(np.abs(Rg - Ro) <= TOL + np.abs(Ro) * 2.0 ** -8)


def test_scan_randomized_sweep(hip):
    """60 random (shape, rule, flags, dtype, carried state) cases against the fp64 oracle -- the short form of
    tools/stress_scan.py.  delta_parallel is not contractive, so errors are judged relative to the magnitude reached."""
    rng = np.random.default_rng(11)
    for i in range(60):
        B, T, Hh = int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(1, 3))
        N = int(rng.choice([1, 7, 16, 17, 49, 63, 64, 65, 100, 128, 130, 196, 256]))
        Dv = int(rng.choice([16, 32, 48, 64])) if N > 64 else int(rng.choice([16, 32, 48, 64, 256]))
        rule, flags, bf, with_state = int(rng.integers(0, 3)), int(rng.choice([0, 3])), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=int(rng.integers(1 << 30)), normalized=not flags,
                                         logits=bool(flags), corr=float(rng.uniform(0, 0.9)))
        if bf:
            q, k, v = (O.to_bf16_f32(x) for x in (q, k, v))
        s0 = (rng.standard_normal((B, Hh, 64, Dv)) * 0.3).astype(np.float32) if with_state else None
        Rg, Sg = _run(hip, q, k, v, a, b, s0, rule, flags, dtype=torch.bfloat16 if bf else torch.float32)
        Ro, So = c_oracle.scan(q, k, v, a, b, s0, rule, flags, math="f64")
        scale = max(1.0, float(np.abs(So).max()), float(np.abs(Ro).max()))
        case = (i, B, T, N, Hh, Dv, rule, flags, bf, with_state)
        assert np.abs(Sg - So).max() <= TOL * scale, case
        assert np.all(np.abs(Rg - Ro) <= TOL * scale + (np.abs(Ro) * 2.0 ** -8 if bf else 0)), case


@pytest.mark.parametrize("N", [49, 130])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_state_history(hip, N, dtype):
    """s_hist (the training forward's by-product) holds the state BEFORE every frame -- also for frames of more than 64 tokens,
    where the read-out runs as its own kernel: entry t equals the end state of a call on the first t frames, and the read-out
    is the one of the call without history (up to fp32 re-association: with a history the frame-parallel side runs the
    training kernels, WY factors + fold, instead of the direct M-form)."""
    B, T, Hh, Dv = 2, 5, 1, 32
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=77, normalized=False, logits=True, corr=0.5)
    t = [_dev(x, dtype) for x in (q, k, v)] + [_dev(a), _dev(b)]
    hist = torch.empty(B, T, Hh, 64, Dv, device="cuda")
    R, S = hip.scan_fwd(*t, flags=3, state_hist=hist)
    R0, S0 = hip.scan_fwd(*t, flags=3)
    tol = dict(rtol=0, atol=2e-5)
    assert torch.allclose(S, S0, **tol) and torch.allclose(R.float(), R0.float(), rtol=2.0 ** -7, atol=2e-5)
    assert torch.count_nonzero(hist[:, 0]) == 0
    for n in range(1, T):
        _, Sn = hip.scan_fwd(*(x[:, :n].contiguous() for x in t), flags=3)
        assert torch.allclose(hist[:, n], Sn, **tol), n


@pytest.mark.parametrize("N,dtype", [(300, torch.float32), (1024, torch.bfloat16), (1000, torch.float32)])
def test_scan_frames_of_more_than_256_tokens(hip, N, dtype):
    """512x512 inputs give N = 1024 tokens per frame at stride 16: the per-64-token fold + composition carries any chunk count
    (the round-1 ABI stopped at 256)."""
    q, k, v, a, b = make_scan_inputs(1, 3, N, 2, 64, 32, seed=N, normalized=False, logits=True, corr=0.5)
    s0 = np.random.default_rng(3).standard_normal((1, 2, 64, 32)).astype(np.float32) * 0.1
    if dtype == torch.bfloat16:
        q, k, v = (O.to_bf16_f32(x) for x in (q, k, v))
    Rg, Sg = _run(hip, q, k, v, a, b, s0, 2, 3, dtype=dtype)
    Ro, So = c_oracle.scan(q, k, v, a, b, s0, 2, 3, math="f64")
    assert np.abs(Sg - So).max() <= TOL
    assert np.all(np.abs(Rg - Ro) <= TOL + (np.abs(Ro) * 2.0 ** -8 if dtype == torch.bfloat16 else 0))


def test_scan_rejects_what_the_kernels_are_not_built_for(hip):
    """The limits that remain fail loudly with a message (never a silent fallback): Dk above 256 or no multiple of 8 at the C ABI, Dv
    not a multiple of 16, more than 4096 tokens per frame, unknown flags."""
    from gdkvm_amd import ops
    def call(N=8, Dk=64, Dv=16, flags=0):
        lib = ops.load()
        z = torch.zeros(1 * 1 * max(N, 1) * 1 * max(Dk, Dv, 64) * 4 + 1024, device="cuda")      # (also the [Dk, Dv] state out)
        ws = torch.empty(max(ops.scan_workspace_bytes(1, 1, 1, min(N, 4096), 64, 16), 4096), dtype=torch.uint8, device="cuda")
        return lib.gdkvm_scan_fwd(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), None, z.data_ptr(), z.data_ptr(), None,
                                  ws.data_ptr(), ws.numel(), 1, 1, 1, N, Dk, Dv, 0, 2, flags, None)
    assert call() == 0
    for kw, msg in [(dict(Dk=36), "Dk=36"), (dict(Dk=264), "Dk=264"), (dict(Dk=132), "Dk=132"), (dict(Dv=24), "Dv=24"), (dict(N=4097), "N=4097"), (dict(flags=64), "flags")]:
        assert call(**kw) == -1 and msg in ops.load().gdkvm_last_error().decode(), kw


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(32, 49, 32), (8, 7, 16), (56, 130, 64)])
def test_scan_narrow_keys_at_the_c_abi(hip, case, dtype):
    """gdkvm_scan_fwd with 8 <= Dk < 64 (multiples of 8): the call zero-extends q, k and the state inside its workspace -- the same
    bits as the host-side padding (ops._pad_keys) onto the Dk = 64 kernels, and right against the oracle run at the narrow width;
    s_hist (training) at a narrow width is refused."""
    from gdkvm_amd import ops
    Dk, N, Dv = case
    q, k, v, a, b = make_scan_inputs(2, 3, N, 1, Dk, Dv, seed=Dk + N, normalized=False, logits=True, corr=0.4)
    s0 = (0.2 * np.random.default_rng(Dk).standard_normal((2, 1, Dk, Dv))).astype(np.float32)
    t = [_dev(x, dtype) for x in (q, k, v)] + [_dev(a), _dev(b)]
    st = _dev(s0)
    assert ops.scan_workspace_bytes(2, 3, 1, N, Dk, Dv) > ops.scan_workspace_bytes(2, 3, 1, N, 64, Dv)
    R, S = hip.scan_fwd(*t, st, flags=3)
    qp, kp, sp = ops._pad_keys(t[0], t[1], st)
    R2, S2 = hip.scan_fwd(qp, kp, t[2], t[3], t[4], sp, flags=3)
    assert S.shape == (2, 1, Dk, Dv) and torch.equal(R, R2) and torch.equal(S, S2[:, :, :Dk]) and not S2[:, :, Dk:].any()
    inp = [O.to_bf16_f32(x) for x in (q, k, v)] if dtype == torch.bfloat16 else [q, k, v]
    Ro, So = c_oracle.scan(*inp, a, b, s0, 2, 3, math="f64")
    assert np.abs(S.cpu().numpy() - So).max() <= TOL
    assert np.all(np.abs(R.float().cpu().numpy() - Ro) <= TOL + (np.abs(Ro) * 2.0 ** -8 if dtype == torch.bfloat16 else 0))
    R3, S3 = hip.scan_fwd(*t, None, flags=3)              # no state in
    Ro3, So3 = c_oracle.scan(*inp, a, b, None, 2, 3, math="f64")
    assert np.abs(S3.cpu().numpy() - So3).max() <= TOL
    hist = torch.empty(2, 3, 1, Dk, Dv, device="cuda")
    with pytest.raises(hip.GdkvmError, match="s_hist"):
        hip.scan_fwd(*t, st, flags=3, state_hist=hist)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rule", [0, 1, 2])
@pytest.mark.parametrize("case", [(128, 49, 32, 2), (72, 7, 16, 1), (256, 70, 48, 1)])
def test_scan_wide_keys_on_the_general_kernel(hip, case, rule, dtype):
    """gdkvm_scan_fwd with 64 < Dk <= 256 (multiples of 8): the definitional recurrence on the device (csrc/gdr_general.hip, fp32 FMAs in
    a fixed order) against the oracle at that width -- every rule, both I/O types, several heads, ragged token counts, a state carried in,
    gates as logits and in-kernel normalisation; a clip cut into calls is bit-identical to one call; without a read-out only the state
    comes back; s_hist (training) is refused; gdkvm_scan_status has nothing to report."""
    from gdkvm_amd import ops
    Dk, N, Dv, Hh = case
    B, T = 2, 4
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, Dk, Dv, seed=Dk + N + rule, normalized=False, logits=True, corr=0.4)
    s0 = (0.2 * np.random.default_rng(Dk).standard_normal((B, Hh, Dk, Dv))).astype(np.float32)
    t = [_dev(x, dtype) for x in (q, k, v)] + [_dev(a), _dev(b)]
    st = _dev(s0)
    assert ops.scan_workspace_bytes(B, T, Hh, N, Dk, Dv) <= 4096
    R, S = hip.scan_fwd(*t, st, rule=rule, flags=3, check=True)
    inp = [O.to_bf16_f32(x) for x in (q, k, v)] if dtype == torch.bfloat16 else [q, k, v]
    Ro, So = c_oracle.scan(*inp, a, b, s0, rule, 3, math="f64")
    scale = max(1.0, float(np.abs(So).max()))
    assert np.abs(S.cpu().numpy() - So).max() <= TOL * scale
    assert np.all(np.abs(R.float().cpu().numpy() - Ro) <= TOL * scale + (np.abs(Ro) * 2.0 ** -8 if dtype == torch.bfloat16 else 0))
    # two calls with the state carried == one call, bit for bit
    Ra, Sa = hip.scan_fwd(*(x[:, :3].contiguous() for x in t), st, rule=rule, flags=3)
    Rb, Sb = hip.scan_fwd(*(x[:, 3:].contiguous() for x in t), Sa, rule=rule, flags=3)
    assert torch.equal(torch.cat([Ra, Rb], 1), R) and torch.equal(Sb, S)
    # no read-out: the state alone; no state in: zeros
    Rn, Sn = hip.scan_fwd(*t, st, rule=rule, flags=3, readout=False)
    assert Rn is None and torch.equal(Sn, S)
    _, S3 = hip.scan_fwd(*t, None, rule=rule, flags=3)
    _, So3 = c_oracle.scan(*inp, a, b, None, rule, 3, math="f64")
    assert np.abs(S3.cpu().numpy() - So3).max() <= TOL * max(1.0, float(np.abs(So3).max()))
    with pytest.raises(hip.GdkvmError, match="s_hist"):
        hip.scan_fwd(*t, st, rule=rule, flags=3, state_hist=torch.empty(B, T, Hh, Dk, Dv, device="cuda"))


@pytest.mark.parametrize("Dk", [32, 16, 48, 20])
def test_scan_narrower_key_dims_through_zero_channels(hip, Dk):
    """Dk below 64 (multiples of 8 inside gdkvm_scan_fwd, other widths padded at the host seam in gdkvm_amd/ops.py): exact against
    the oracle run at the narrow Dk, state carried in and out at [B,Hh,Dk,Dv]."""
    q, k, v, a, b = make_scan_inputs(2, 4, 49, 1, Dk, 32, seed=Dk, normalized=False, logits=True, corr=0.4)
    s0 = np.random.default_rng(Dk).standard_normal((2, 1, Dk, 32)).astype(np.float32) * 0.2
    Rg, Sg = _run(hip, q, k, v, a, b, s0, 2, 3)
    Ro, So = c_oracle.scan(q, k, v, a, b, s0, 2, 3, math="f64")
    assert Sg.shape == So.shape and np.abs(Sg - So).max() <= TOL and np.abs(Rg - Ro).max() <= TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rule", [0, 2])
@pytest.mark.parametrize("log2_scale", [12, 20, 40])
def test_scan_large_values_and_carried_state(hip, rule, log2_scale, dtype):
    """The default forward carries the state as fp16 pairs at 2^-e.  With V and the carried state scaled by 2^12 ... 2^40 the
    kernel must size e by the call's own bound (8 (max|S_0| + sum_t max|G_t|), gdr_scan.hip) instead of saturating at |S| ~ 1e6:
    results against the oracle within the usual tolerance RELATIVE to the scale, for the delta rule and for gated_linear (whose
    state grows with T N |v|), both I/O types, through the C ABI with default flags -- never rc 0 with saturated values."""
    B, T, N, Hh, Dv = 2, 6, 49, 1, 64
    scale = 2.0 ** log2_scale
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=40 + rule + log2_scale, normalized=False, logits=True, corr=0.5)
    v = (v * scale).astype(np.float32)
    s0 = (np.random.default_rng(2).standard_normal((B, Hh, 64, Dv)) * scale).astype(np.float32)
    if dtype == torch.bfloat16:
        q, k, v = (O.to_bf16_f32(x) for x in (q, k, v))
    Rg, Sg = _run(hip, q, k, v, a, b, s0, rule, 3, dtype)
    Ro, So = c_oracle.scan(q, k, v, a, b, s0, rule, 3, math="f64")
    assert np.isfinite(Rg).all() and np.isfinite(Sg).all()
    hip.scan_fwd(_dev(q, dtype), _dev(k, dtype), _dev(v, dtype), _dev(a), _dev(b), _dev(s0), rule=rule, flags=3, check=True)    # (in range: no exception)
    assert np.abs(Sg - So).max() <= TOL * scale, (np.abs(Sg - So).max() / scale)
    out_q = 2.0 ** -8 if dtype == torch.bfloat16 else 0.0
    assert np.all(np.abs(Rg - Ro) <= TOL * scale + np.abs(Ro) * out_q), (np.abs(Rg - Ro).max() / scale)
    # the same clip cut into two calls with the state carried: the exponent of each call is sized by that call's own bound
    h = T // 2
    r1, s1 = _run(hip, q[:, :h], k[:, :h], v[:, :h], a[:, :h], b[:, :h], s0, rule, 3, dtype)
    r2, s2 = _run(hip, q[:, h:], k[:, h:], v[:, h:], a[:, h:], b[:, h:], s1, rule, 3, dtype)
    assert np.abs(s2 - So).max() <= TOL * scale
    assert np.all(np.abs(np.concatenate([r1, r2], 1) - Ro) <= TOL * scale + np.abs(Ro) * out_q)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_large_values_in_frames_of_more_than_64_tokens(hip, dtype, monkeypatch):
    """Frames of 130 tokens (chunk composition, deferred read-out).  2^12: right, relative to the scale.  2^22: the composition's
    own fp16-pair re-split of the running map (fixed exponent 4) overflows -- that must come back as NaNs, never as plausible
    saturated numbers; GDKVM_FLAG_WIDE_RANGE (three bf16 terms, the whole fp32 range) serves such inputs."""
    B, T, N, Hh, Dv = 1, 3, 130, 1, 64
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=77, normalized=False, logits=True, corr=0.5)
    if dtype == torch.bfloat16:
        q, k, v = (O.to_bf16_f32(x) for x in (q, k, v))
    for fuse in ("0", "1"):
        monkeypatch.setenv("GDKVM_PREP_FUSE", fuse)
        v12 = (v * 4096.0).astype(np.float32)
        Rg, Sg = _run(hip, q, k, v12, a, b, None, 2, 3, dtype)
        Ro, So = c_oracle.scan(q, k, v12, a, b, None, 2, 3, math="f64")
        assert np.abs(Sg - So).max() <= TOL * 4096 and np.all(np.abs(Rg - Ro) <= TOL * 4096 + np.abs(Ro) * 2.0 ** -8)
        v22 = (v * 2.0 ** 22).astype(np.float32)
        Rg, Sg = _run(hip, q, k, v22, a, b, None, 2, 3, dtype)
        assert np.isnan(Sg).all() and np.isnan(Rg[:, 1:]).all()             # (frame 0 reads the zero start state)
        # the call itself returned 0 (asynchronous: no kernel had seen the data); gdkvm_scan_status reports it as GDKVM_ERR_RANGE, the
        # ordinary call and the full-range one as fine, and check=True turns it into an exception
        ws = hip.new_workspace(B, T, Hh, N, 64, Dv, torch.device("cuda"))
        t22 = [_dev(x, dtype) for x in (q, k, v22)] + [_dev(a), _dev(b)]
        hip.scan_fwd(*t22, flags=3, workspace=ws)
        with pytest.raises(hip.GdkvmError, match=r"\(-7\).*GDKVM_FLAG_WIDE_RANGE"):
            hip.scan_status(ws, B, T, Hh, N, 64, Dv, 3)
        with pytest.raises(hip.GdkvmError, match="range"):
            hip.scan_fwd(*t22, flags=3, check=True)
        hip.scan_fwd(*t22, flags=3 | 8, workspace=ws, check=True)
        hip.scan_fwd(_dev(q, dtype), _dev(k, dtype), _dev(v12, dtype), _dev(a), _dev(b), flags=3, workspace=ws, check=True)
        Rw, Sw = _run(hip, q, k, v22, a, b, None, 2, 3 | 8, dtype)
        Ro, So = c_oracle.scan(q, k, v22, a, b, None, 2, 3, math="f64")
        assert np.abs(Sw - So).max() <= TOL * 2.0 ** 22 and np.all(np.abs(Rw - Ro) <= TOL * 2.0 ** 22 + np.abs(Ro) * 2.0 ** -8)
