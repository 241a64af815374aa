"""GPU parity of the backward kernels: HIP gradients vs CPU autograd through the fp64 torch
restatement (oracle/torch_ref.py) on the same seeded inputs.  Tolerance 1e-4 absolute, scaled by the gradient's
magnitude for the larger shapes (fp32 accumulation over T frames and N tokens)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as TR
from tests.util import make_scan_inputs

pytestmark = pytest.mark.gpu


def _ref_grads(q, k, v, a, b, s0, dR, dS, rule, flags):
    ts = [torch.from_numpy(np.asarray(x, np.float64)).requires_grad_() for x in (q, k, v, a, b, s0)]
    R, S = TR.scan(*ts, rule, flags)
    ((R * torch.from_numpy(dR).double()).sum() + (S * torch.from_numpy(dS).double()).sum()).backward()
    return [t.grad.numpy() for t in ts]


def _hip_grads(hip, q, k, v, a, b, s0, dR, dS, rule, flags, dtype=torch.float32):
    dev = lambda x, dt=None: (torch.from_numpy(np.ascontiguousarray(x)).cuda().to(dt) if dt else torch.from_numpy(np.ascontiguousarray(x)).cuda())
    tq, tk, tv = (dev(x, dtype).requires_grad_() for x in (q, k, v))
    ta, tb, ts0 = (dev(x).requires_grad_() for x in (a, b, s0))
    R, S = hip.scan(tq, tk, tv, ta, tb, ts0, rule, flags)
    torch.autograd.backward([R, S], [dev(dR, dtype), dev(dS)])
    return [t.grad.float().cpu().numpy() for t in (tq, tk, tv, ta, tb, ts0)]


@pytest.mark.parametrize("rule", [0, 1, 2])
@pytest.mark.parametrize("flags", [0, 3])
def test_scan_backward_fp32(hip, rule, flags):
    B, T, N, Hh, Dk, Dv = 2, 4, 49, 1, 64, 80
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, Dk, Dv, seed=40 + rule, normalized=not flags, logits=bool(flags), corr=0.5)
    rng = np.random.default_rng(41)
    s0 = (0.3 * rng.standard_normal((B, Hh, Dk, Dv))).astype(np.float32)
    dR = rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32)
    dS = rng.standard_normal((B, Hh, Dk, Dv)).astype(np.float32)
    ref = _ref_grads(q, k, v, a, b, s0, dR, dS, rule, flags)
    got = _hip_grads(hip, q, k, v, a, b, s0, dR, dS, rule, flags)
    for name, g, r in zip("q k v alpha beta s0".split(), got, ref):
        tol = 1e-4 * max(1.0, np.abs(r).max())
        assert np.abs(g - r).max() <= tol, (name, np.abs(g - r).max(), np.abs(r).max())


@pytest.mark.parametrize("shape", [(1, 2, 1, 1, 16), (2, 3, 17, 2, 32), (1, 2, 64, 1, 48), (3, 5, 7, 1, 256)])
def test_scan_backward_shapes(hip, shape):
    B, T, N, Hh, Dv = shape
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=sum(shape), normalized=False, logits=True, corr=0.7)
    rng = np.random.default_rng(43)
    s0 = (0.3 * rng.standard_normal((B, Hh, 64, Dv))).astype(np.float32)
    dR = rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32)
    dS = rng.standard_normal((B, Hh, 64, Dv)).astype(np.float32)
    ref = _ref_grads(q, k, v, a, b, s0, dR, dS, 2, 3)
    got = _hip_grads(hip, q, k, v, a, b, s0, dR, dS, 2, 3)
    for name, g, r in zip("q k v alpha beta s0".split(), got, ref):
        assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), (name, np.abs(g - r).max())


def test_scan_backward_randomized_sweep(hip):
    """30 random (shape, rule, flags) cases, T = 1..9 (every tail of the reverse-mode serial kernel's unrolled loop), N <= 64."""
    rng = np.random.default_rng(5)
    for i in range(30):
        B, T, Hh = int(rng.integers(1, 3)), int(rng.integers(1, 10)), int(rng.integers(1, 3))
        N, Dv = int(rng.choice([1, 5, 16, 17, 33, 49, 64])), int(rng.choice([16, 32, 48, 80]))
        rule, flags = int(rng.choice([0, 2])), int(rng.choice([0, 3]))
        q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=int(rng.integers(1 << 30)), normalized=not flags,
                                         logits=bool(flags), corr=float(rng.uniform(0, 0.8)))
        s0 = (0.3 * rng.standard_normal((B, Hh, 64, Dv))).astype(np.float32)
        dR = rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32)
        dS = rng.standard_normal((B, Hh, 64, Dv)).astype(np.float32)
        ref = _ref_grads(q, k, v, a, b, s0, dR, dS, rule, flags)
        got = _hip_grads(hip, q, k, v, a, b, s0, dR, dS, rule, flags)
        for name, g, r in zip("q k v alpha beta s0".split(), got, ref):
            assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), (i, B, T, N, Hh, Dv, rule, flags, name, np.abs(g - r).max())


@pytest.mark.parametrize("case", [(2, 3, 100, 1, 32, 2, 3), (1, 2, 130, 2, 48, 2, 3), (1, 3, 256, 1, 64, 2, 0), (2, 2, 65, 1, 16, 0, 3)])
def test_scan_backward_more_than_64_tokens(hip, case):
    """Frames of more than 64 tokens train through 64-token pseudo-frames (gdkvm_scan_train_fwd / gdkvm_scan_train_bwd: two C calls
    and one workspace, no framework op in between): gradients of the whole op -- incl. the read-out's path back into the state
    recurrence (gdkvm_scan_state_bwd's d_hist) -- against fp64 autograd."""
    B, T, N, Hh, Dv, rule, flags = case
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, 64, Dv, seed=sum(case), normalized=not flags, logits=bool(flags), corr=0.5)
    rng = np.random.default_rng(47)
    s0 = (0.3 * rng.standard_normal((B, Hh, 64, Dv))).astype(np.float32)
    dR = rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32)
    dS = rng.standard_normal((B, Hh, 64, Dv)).astype(np.float32)
    ref = _ref_grads(q, k, v, a, b, s0, dR, dS, rule, flags)
    got = _hip_grads(hip, q, k, v, a, b, s0, dR, dS, rule, flags)
    for name, g, r in zip("q k v alpha beta s0".split(), got, ref):
        assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), (name, np.abs(g - r).max(), np.abs(r).max())


def test_scan_forward_values_through_the_chunked_training_path(hip):
    """... and its forward values equal the inference path's."""
    q, k, v, a, b = make_scan_inputs(2, 3, 130, 1, 64, 32, seed=9, normalized=False, logits=True, corr=0.5)
    t = [torch.from_numpy(x).cuda() for x in (q, k, v, a, b)]
    R0, S0 = hip.scan_fwd(*t, flags=3)
    R1, S1 = hip.scan(*(x.clone().requires_grad_() for x in t), None, 2, 3)
    assert torch.allclose(R1, R0, atol=2e-5) and torch.allclose(S1, S0, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_train_entries_at_64_tokens_or_fewer_are_the_direct_pair(hip, dtype):
    """N <= 64: gdkvm_scan_train_fwd / _bwd == gdkvm_scan_fwd (history inside the workspace) + gdkvm_scan_bwd, bit for bit; and the
    limits that remain fail loudly (delta_parallel with more than 64 tokens; a workspace that is too small)."""
    from gdkvm_amd import ops
    q, k, v, a, b = make_scan_inputs(2, 4, 49, 1, 64, 32, seed=12, normalized=False, logits=True, corr=0.5)
    dev = lambda x, dt=None: torch.from_numpy(x).cuda().to(dt or torch.float32)
    mk = lambda: [dev(q, dtype).requires_grad_(), dev(k, dtype).requires_grad_(), dev(v, dtype).requires_grad_(), dev(a).requires_grad_(), dev(b).requires_grad_()]
    g = torch.Generator(device="cuda").manual_seed(1)
    dR = torch.randn(2, 4, 49, 1, 32, device="cuda", generator=g).to(dtype)
    outs = []
    for fn in (ops._ScanFunction, ops._ScanTrainFunction):
        t = mk()
        R, S = fn.apply(*t, None, 2, 3)
        (R.float() * dR.float()).sum().backward()
        outs.append([R, S] + [x.grad for x in t])
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    big = make_scan_inputs(1, 2, 100, 1, 64, 16, seed=13, normalized=False, logits=True)
    with pytest.raises(hip.GdkvmError, match="delta_parallel"):
        ops._ScanTrainFunction.apply(*(dev(x) for x in big), None, 1, 3)
    lib = hip.load()
    need = int(lib.gdkvm_scan_train_workspace_bytes(1, 2, 1, 100, 64, 16, 0))
    assert need > int(lib.gdkvm_scan_workspace_bytes(1, 2, 1, 100, 64, 16))
    t = [dev(x) for x in big]
    ws = torch.empty(need // 2, dtype=torch.uint8, device="cuda")
    r = torch.empty(1, 2, 100, 1, 16, device="cuda")
    rc = lib.gdkvm_scan_train_fwd(*(x.data_ptr() for x in t), None, r.data_ptr(), None, ws.data_ptr(), ws.numel(), 1, 2, 1, 100, 64, 16, 0, 2, 3, None)
    assert rc == -5 and "workspace" in lib.gdkvm_last_error().decode()          # GDKVM_ERR_WORKSPACE


def test_scan_backward_bf16_io(hip):
    """bf16 tensors: the kernels differentiate the exact-fp32 function of the bf16-rounded inputs; the returned
    gradients are rounded to bf16 (2^-8 relative)."""
    from oracle import gdkvm_oracle as O
    B, T, N, Hh, Dk, Dv = 2, 3, 49, 1, 64, 64
    q, k, v, a, b = make_scan_inputs(B, T, N, Hh, Dk, Dv, seed=44, normalized=False, logits=True)
    rng = np.random.default_rng(45)
    s0 = np.zeros((B, Hh, Dk, Dv), np.float32)
    dR = O.to_bf16_f32(rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32))
    dS = rng.standard_normal((B, Hh, Dk, Dv)).astype(np.float32)
    ref = _ref_grads(*(O.to_bf16_f32(x) for x in (q, k, v)), a, b, s0, dR, dS, 2, 3)
    got = _hip_grads(hip, q, k, v, a, b, s0, dR, dS, 2, 3, dtype=torch.bfloat16)
    for name, g, r in zip("q k v alpha beta s0".split(), got, ref):
        lim = 1e-4 * max(1.0, np.abs(r).max()) + (np.abs(r) * 2.0 ** -7 if name in "qkv" else 0)
        assert np.all(np.abs(g - r) <= lim), (name, np.abs(g - r).max())


def test_scan_backward_no_state_input(hip):
    q, k, v, a, b = make_scan_inputs(1, 2, 9, 1, 64, 16, seed=46)
    dev = lambda x: torch.from_numpy(x).cuda()
    tq = dev(q).requires_grad_()
    R, S = hip.scan(tq, dev(k), dev(v), dev(a), dev(b))
    R.sum().backward()
    assert tq.grad is not None and torch.isfinite(tq.grad).all()


@pytest.mark.parametrize("case", [(3, 7, 7, 64, 256, 256), (2, 16, 16, 32, 64, 64), (2, 5, 3, 16, 16, 32)])
def test_kpff_backward_fp32(hip, case):
    from tests.util import make_kpff_inputs
    BT, h, w, Ck, Cv, Cp = case
    arrs = make_kpff_inputs(BT, h, w, Ck, Cv, Cp, seed=sum(case))
    dF = np.random.default_rng(50).standard_normal((BT, h * w, Cp)).astype(np.float32)
    ts = [torch.from_numpy(x).double().requires_grad_() for x in arrs]
    (TR.kpff(*ts, h, w) * torch.from_numpy(dF).double()).sum().backward()
    ref = [t.grad.numpy() for t in ts]
    gs = [torch.from_numpy(x).cuda().requires_grad_() for x in arrs]
    F = hip.kpff(*gs, h, w)
    F.backward(torch.from_numpy(dF).cuda())
    for name, g, r in zip("L G P Wa ba Wl Wg".split(), gs, ref):
        err = np.abs(g.grad.cpu().numpy() - r).max()
        assert err <= 2e-4 * max(1.0, np.abs(r).max()), (name, err, np.abs(r).max())


def test_kpff_backward_bf16(hip):
    from oracle import gdkvm_oracle as O
    from tests.util import make_kpff_inputs
    BT, h, w, Ck, Cv, Cp = 3, 7, 7, 64, 256, 256
    arrs = make_kpff_inputs(BT, h, w, Ck, Cv, Cp, seed=51)
    dF = O.to_bf16_f32(np.random.default_rng(52).standard_normal((BT, h * w, Cp)).astype(np.float32))
    rnd = [O.to_bf16_f32(x) if i != 4 else x for i, x in enumerate(arrs)]      # features and weights as the bf16 arm sees them
    ts = [torch.from_numpy(x).double().requires_grad_() for x in rnd]
    (TR.kpff(*ts, h, w) * torch.from_numpy(dF).double()).sum().backward()
    ref = [t.grad.numpy() for t in ts]
    gs = [torch.from_numpy(x).cuda().to(torch.bfloat16 if i < 3 else torch.float32).requires_grad_() for i, x in enumerate(arrs)]
    hip.kpff(*gs, h, w).backward(torch.from_numpy(dF).cuda().bfloat16())
    for name, g, r in zip("L G P Wa ba Wl Wg".split(), gs, ref):
        err = np.abs(g.grad.float().cpu().numpy() - r)
        scale = np.abs(r).max()
        assert err.max() <= 0.03 * scale + 1e-2 and err.mean() <= 4e-3 * scale + 1e-3, (name, err.max(), err.mean(), scale)


@pytest.mark.parametrize("N,Dv,flags,dtype", [(65, 48, 1, torch.float32), (130, 64, 0, torch.float32), (256, 256, 1, torch.float32),
                                             (196, 32, 1, torch.bfloat16), (7, 16, 1, torch.float32)])
def test_readout_from_state_history_and_its_backward(hip, N, Dv, flags, dtype):
    """gdkvm_readout_fwd / gdkvm_readout_bwd (the training read-out of frames of more than 64 tokens) against fp64 numpy:
    R = diag(qinv) Q S,  dQ through the L2 normalisation,  dS = Qn^T dR written at the frames' slots of the history."""
    from gdkvm_amd import ops
    from oracle import gdkvm_oracle as O
    B, T, Hh, Dk, C = 2, 3, 2, 64, 3                                   # C pseudo-frames per frame: frame t reads hist[:, t*C]
    rng = np.random.default_rng(N + Dv)
    q = rng.standard_normal((B, T, N, Hh, Dk)).astype(np.float32)
    hist = rng.standard_normal((B, T * C, Hh, Dk, Dv)).astype(np.float32)
    dR = rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32)
    if dtype == torch.bfloat16:
        q, dR = O.to_bf16_f32(q), O.to_bf16_f32(dR)
    tq = torch.from_numpy(q).cuda().to(dtype).requires_grad_()
    th = torch.from_numpy(hist).cuda().requires_grad_()
    r = ops._ReadoutFunction.apply(tq, th, C, flags)
    r.backward(torch.from_numpy(dR).cuda().to(dtype))
    q64, S = q.astype(np.float64), hist.astype(np.float64)[:, ::C]      # [B,T,Hh,Dk,Dv]
    qi = 1.0 / np.sqrt((q64 ** 2).sum(-1, keepdims=True) + 1e-12) if flags & 1 else np.ones_like(q64[..., :1])
    qn = q64 * qi
    R = np.einsum("btnhd,bthde->btnhe", qn, S)
    dqn = np.einsum("btnhe,bthde->btnhd", dR.astype(np.float64), S)
    dq = qi * (dqn - qn * (qn * dqn).sum(-1, keepdims=True)) if flags & 1 else dqn
    dS = np.einsum("btnhd,btnhe->bthde", qn, dR.astype(np.float64))
    tol = (lambda ref: 1e-4 * max(1.0, np.abs(ref).max()) + (np.abs(ref) * 2.0 ** -7 if dtype == torch.bfloat16 else 0))
    assert np.all(np.abs(r.detach().float().cpu().numpy() - R) <= tol(R))
    assert np.all(np.abs(tq.grad.float().cpu().numpy() - dq) <= tol(dq))
    gh = th.grad.cpu().numpy()
    assert np.abs(gh[:, ::C] - dS).max() <= 1e-4 * max(1.0, np.abs(dS).max())
    mask = np.ones(T * C, bool); mask[::C] = False
    assert not gh[:, mask].any()                                       # the other slots of the history get no gradient here
