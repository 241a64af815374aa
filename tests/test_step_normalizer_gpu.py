"""GPU parity for round 6's two SPEC-v0 flags, through the C ABI (ops -> libgdkvm_hip.so) against the CPU oracle:
  * the `normalizer` flag of the LKVA read: gdkvm_scan_fwd_normalizer, GDKVMConfig(normalizer=True);
  * the per-frame step mode with mask feedback: gdkvm_lkva_read, gdkvm_mask_embed_add,
    GDKVMConfig(mask_feedback=True).
Tolerances: fp32 I/O 1e-4 (relative to the data's scale where a quotient is involved); bf16 I/O adds the stored read-outs' 2^-8."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import gdkvm_oracle as O

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t.to(dtype) if dtype is not None else t


def _positive_inputs(B, T, N, Hh, Dk, Dv, seed):
    """keys / queries with positive entries: |q . z| stays away from 0, so the normalised read-out is well conditioned"""
    rng = np.random.default_rng(seed)
    q = np.abs(rng.standard_normal((B, T, N, Hh, Dk))).astype(np.float32)
    k = np.abs(rng.standard_normal((B, T, N, Hh, Dk))).astype(np.float32)
    v = rng.standard_normal((B, T, N, Hh, Dv)).astype(np.float32)
    a = rng.normal(2, 1, (B, T, Hh)).astype(np.float32)
    b = rng.standard_normal((B, T, N, Hh)).astype(np.float32)
    return q, k, v, a, b


@pytest.mark.parametrize("rule", [0, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 6, 49, 1, 256), (3, 4, 20, 2, 64), (1, 3, 100, 1, 128)])
def test_scan_normalizer_matches_the_oracle_and_carries_z(hip, rule, dtype, shape):
    """R / (|q . z| + eps), S_T and z_T against the fp64 C oracle (which carries z explicitly -- the kernel rides it as a value channel);
    frames of fewer and of more than 64 tokens, two heads; a clip in two calls with (S, z) carried is bit-identical to one call."""
    B, T, N, Hh, Dv = shape
    q, k, v, a, b = _positive_inputs(B, T, N, Hh, 64, Dv, seed=sum(shape) + rule)
    rng = np.random.default_rng(1)
    s0 = (0.2 * rng.standard_normal((B, Hh, 64, Dv))).astype(np.float32)
    z0 = (0.2 * np.abs(rng.standard_normal((B, Hh, 64)))).astype(np.float32)
    t = [_dev(x, dtype) for x in (q, k, v)] + [_dev(a), _dev(b)]
    R, S, Z = hip.scan_fwd_normalizer(*t, _dev(s0), _dev(z0), rule=rule, flags=3, eps=1e-6)
    torch.cuda.synchronize()
    rq = (lambda x: O.to_bf16_f32(x)) if dtype == torch.bfloat16 else (lambda x: x)
    Ro, So, Zo = c_oracle.scan_normalizer(rq(q), rq(k), rq(v), a, b, s0, z0, rule, 3, 1e-6, math="f64")
    assert np.abs(S.cpu().numpy() - So).max() <= 1e-4 and np.abs(Z.cpu().numpy() - Zo).max() <= 1e-4
    scale = max(1.0, float(np.abs(Ro).max()))
    tol = 1e-4 * scale if dtype == torch.float32 else 3 * 2.0 ** -8 * scale
    assert np.abs(R.float().cpu().numpy() - Ro).max() <= tol
    # chunked == one call, bit for bit
    c = T // 2
    R1, S1, Z1 = hip.scan_fwd_normalizer(*(x[:, :c].clone() for x in t), _dev(s0), _dev(z0), rule=rule, flags=3, eps=1e-6)
    R2, S2, Z2 = hip.scan_fwd_normalizer(*(x[:, c:].clone() for x in t), S1, Z1, rule=rule, flags=3, eps=1e-6)
    assert torch.equal(torch.cat([R1, R2], 1), R) and torch.equal(S2, S) and torch.equal(Z2, Z)
    # no carried state: z starts at zero, the first frame's read-out is 0 / eps = 0
    R0, _, _ = hip.scan_fwd_normalizer(*t, None, None, rule=rule, flags=3, eps=1e-6)
    assert float(R0[:, 0].abs().max()) == 0.0


def test_scan_normalizer_argument_errors(hip):
    q, k, v, a, b = (_dev(x) for x in _positive_inputs(1, 2, 8, 1, 64, 32, 0))
    with pytest.raises(hip.GdkvmError, match="eps"):
        hip.scan_fwd_normalizer(q, k, v, a, b, eps=0.0)
    with pytest.raises(hip.GdkvmError):
        hip.scan_fwd_normalizer(q, k, v, a, b, z=torch.zeros(1, 1, 32, device="cuda"))            # z must be [B,Hh,Dk]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(16, 49, 1, 256), (3, 130, 2, 48), (1, 256, 1, 256), (2, 7, 1, 16)])
def test_lkva_read_is_q_times_state(hip, dtype, shape):
    """gdkvm_lkva_read: R = Qn S for one frame per clip -- against numpy fp64 on the same (bf16-rounded) rows; with the norms computed in
    the kernel and with gdkvm_proj_gates-style norms handed in; a second call gives the same bits."""
    B, N, Hh, Dv = shape
    rng = np.random.default_rng(sum(shape))
    q = rng.standard_normal((B, N, Hh, 64)).astype(np.float32)
    s = rng.standard_normal((B, Hh, 64, Dv)).astype(np.float32)
    qd = _dev(q, dtype)
    qr = qd.float().cpu().numpy().astype(np.float64)
    inv = 1.0 / np.sqrt((qr ** 2).sum(-1, keepdims=True) + 1e-12)
    want = np.einsum("bnhd,bhdc->bnhc", qr * inv, s.astype(np.float64))
    got = hip.lkva_read(qd, _dev(s), flags=1)
    tol = 1e-4 if dtype == torch.float32 else 2.0 ** -8 * max(1.0, np.abs(want).max())
    assert np.abs(got.float().cpu().numpy() - want).max() <= tol
    norms = torch.stack([torch.zeros(B * N, Hh, device="cuda"), _dev(inv.reshape(B * N, Hh).astype(np.float32))], -1).contiguous()
    got2 = hip.lkva_read(qd, _dev(s), flags=1, norms=norms)
    assert np.abs(got2.float().cpu().numpy() - want).max() <= tol
    assert torch.equal(hip.lkva_read(qd, _dev(s), flags=1), got)
    raw = hip.lkva_read(qd, _dev(s), flags=0)                                                     # no normalisation
    want_raw = np.einsum("bnhd,bhdc->bnhc", qr, s.astype(np.float64))
    assert np.abs(raw.float().cpu().numpy() - want_raw).max() <= (1e-4 if dtype == torch.float32 else 2.0 ** -8) * max(1.0, np.abs(want_raw).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(112, 112, 7, 7), (256, 256, 16, 16), (30, 58, 4, 7), (15, 13, 2, 3)])
def test_mask_embed_add_pools_like_adaptive_avg_pool(hip, dtype, geom):
    """gdkvm_mask_embed_add: v += w_embed * mean over the token's adaptive-average-pool cell of (mask != 0), 255 = unlabelled = background."""
    H, W, h, w = geom
    rng = np.random.default_rng(H + w)
    F_, C = 5, 48
    mask = rng.integers(0, 3, (F_, H, W)).astype(np.uint8)
    mask[rng.random((F_, H, W)) < 0.05] = 255
    v = rng.standard_normal((F_, h * w, C)).astype(np.float32)
    we = rng.standard_normal(C).astype(np.float32)
    vd = _dev(v, dtype)
    base = vd.float().cpu().numpy().astype(np.float64)
    hip.mask_embed_add_(vd, _dev(mask), _dev(we), h, w)
    want = base + O.mask_cell_mean((mask != 0) & (mask != 255), h, w)[:, :, None] * we.astype(np.float64)[None, None, :]
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
    assert np.abs(vd.float().cpu().numpy() - want).max() <= tol * max(1.0, np.abs(want).max())


def _mixed_head(ref, frames, **kw):
    """shift the head bias so that the reference masks are mixed (a random-init head puts one class on every pixel)"""
    with torch.no_grad():
        lr = ref(frames, **kw)
        gap = (lr[:, :, 0] - lr[:, :, 1]).median()
        ref.decoder.head.bias[1] += gap
    return ref


def test_module_normalizer_flag_matches_the_plain_restatement(hip):
    """GDKVMConfig(normalizer=True): the fp32 module on the GPU against oracle/model_plain.plain_forward (fp64, nothing imported from the
    product) -- logits within 1e-3, the [S | z] state within 1e-4; the state carried over two calls equals one call."""
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from oracle.model_plain import plain_forward
    torch.manual_seed(11)
    cfg = GDKVMConfig(normalizer=True)
    model = GDKVM(cfg).eval()
    with torch.no_grad():
        for p in (model.query_proj.weight, model.key_proj.weight):       # positive keys / queries: a well-conditioned quotient
            p.abs_()
    frames = torch.rand(2, 4, 3, 112, 112)
    lp, sp = plain_forward(model.state_dict(), frames, normalizer=True)
    gm = model.cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():
        lg, sg = gm(frames.cuda(), return_state=True)
        assert tuple(sg.shape) == (2, 1, 64, 257)
        l1, s1 = gm(frames[:, :2].cuda(), return_state=True)
        l2, s2 = gm(frames[:, 2:].cuda(), state=s1, return_state=True)
    scale = max(1.0, float(lp.abs().max()))
    assert (lg.double().cpu() - lp).abs().max().item() <= 1e-3 * scale
    assert (sg.double().cpu() - sp).abs().max().item() <= 1e-4 * max(1.0, float(sp.abs().max()))
    assert (torch.cat([l1, l2], 1) - lg).abs().max().item() <= 1e-4 * scale and (s2 - sg).abs().max().item() <= 1e-5
    with pytest.raises(NotImplementedError):
        gm.train()(frames.cuda())                                         # an inference flag


def test_module_mask_feedback_matches_the_plain_restatement(hip):
    """GDKVMConfig(mask_feedback=True), fp32 module on the GPU (HIP read / KPFF / mask / embed / write per frame) against the independent
    fp64 restatement: stride-4 logits within 1e-3, the final state within 1e-3, masks bit-equal wherever the reference margin between the
    two largest logits exceeds the logit tolerance (a flipped pixel at zero margin is not an error; it then also feeds back, which is
    why the bounds are those of a recurrent chain); Dice counts consistent with the masks; with and without a first-frame mask."""
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from oracle.model_plain import plain_forward
    from oracle.model_ref import GDKVMRef
    torch.manual_seed(5)
    cfg = GDKVMConfig(mask_feedback=True)
    ref = GDKVMRef(cfg).eval()
    ref.math = "f64"
    with torch.no_grad():
        ref.mask_embed.weight.mul_(4.0)                                   # make the fed-back mask matter
    frames = torch.rand(2, 5, 3, 112, 112)
    m0 = torch.zeros(2, 1, 112, 112)
    m0[:, :, 30:80, 40:90] = 1.0
    _mixed_head(ref, frames, mask0=m0)
    model = GDKVM(cfg).eval()
    model.load_state_dict(ref.state_dict())
    gm = model.cuda().to(memory_format=torch.channels_last)
    tgt = (torch.rand(2, 5, 112, 112) > 0.5).to(torch.uint8)
    for mask0 in (m0, None):
        taps = {}
        lp, sp = plain_forward(ref.state_dict(), frames, mask0=mask0, mask_feedback=True, lowres=True, taps=taps)
        with torch.no_grad():
            lg, sg = gm(frames.cuda(), mask0=None if mask0 is None else mask0.cuda(), return_state=True, _lowres=True)
            mk, counts = gm.segment(frames.cuda(), tgt.cuda(), **({} if mask0 is None else {"mask0": mask0.cuda()}))
        assert (lg.double().cpu() - lp).abs().max().item() <= 1e-3
        assert (sg.double().cpu() - sp).abs().max().item() <= 1e-3
        pm, margin = taps["mask"], taps["margin"]
        fg = (pm != 0).float().mean().item()
        assert 0.05 < fg < 0.95, f"degenerate reference masks (foreground {fg})"
        decided = margin > 2e-3
        assert decided.float().mean().item() > 0.5                           # (random-init logits: a median margin of a few 1e-3)
        assert torch.equal(mk.cpu()[decided], pm[decided])
        mc = mk.cpu()
        for c in range(cfg.num_classes):                                  # integer counts are those of the masks the kernel produced
            assert torch.equal(counts[..., c, 1].cpu().long(), (mc == c).sum((-1, -2)))
            assert torch.equal(counts[..., c, 0].cpu().long(), ((mc == c) & (tgt == c)).sum((-1, -2)))


def test_feedback_mode_in_the_fused_build_and_as_one_graph(hip):
    """The inference build (bf16, BatchNorm folded, every kernel hand-written) in step mode: the 4-frame loop captured as ONE hipGraph
    (GraphedSegment, one and two streams) replays the eager masks bit for bit; masks agree with the fp32 CPU reference on all but the
    pixels at bf16-size margins; the scan mode of the same weights gives different masks (the flag is live)."""
    import dataclasses
    from gdkvm_amd.model import GDKVM, GDKVMConfig, GraphedSegment
    from oracle.model_ref import GDKVMRef
    torch.manual_seed(9)
    cfg = GDKVMConfig(mask_feedback=True)
    ref = GDKVMRef(cfg).eval()
    with torch.no_grad():
        ref.mask_embed.weight.mul_(4.0)
    frames = torch.rand(8, 4, 3, 112, 112)
    _mixed_head(ref, frames)
    model = GDKVM(cfg).eval()
    model.load_state_dict(ref.state_dict())
    fm = model.cuda().fuse_for_inference().to(torch.bfloat16).to(memory_format=torch.channels_last)
    fr = frames.cuda().bfloat16()
    tgt = (torch.rand(8, 4, 112, 112, device="cuda") > 0.5).to(torch.uint8)
    mk, counts = fm.segment(fr, tgt)
    mr, _ = ref.segment(frames)
    agree = (mk.cpu() == mr).float().mean().item()
    assert agree >= 0.93, agree                                            # (random-init logits have a median margin of ~3e-3: tests/stage_error.py)
    for streams in (1, 2):
        g = GraphedSegment(fm, fr.clone(), tgt.clone(), streams=streams)
        m2, c2 = g(fr, tgt)
        assert torch.equal(m2, mk) and torch.equal(c2, counts)
    plain = GDKVM(dataclasses.replace(cfg, mask_feedback=False)).eval()
    plain.load_state_dict(ref.state_dict())
    pm = plain.cuda().fuse_for_inference().to(torch.bfloat16).to(memory_format=torch.channels_last)
    assert not torch.equal(pm.segment(fr)[0], mk)


def test_step_mode_without_feedback_weight_is_the_scan_mode(hip):
    """A property that ties the two time loops together: with a ZERO mask-embedding weight nothing is fed back, and the per-frame step mode
    (gdkvm_lkva_read + T = 1 writes, frame by frame) must compute what the scan mode (one gdkvm_scan_fwd over all frames) computes -- other
    kernels, other operation order, same function: fp32 module logits within 1e-3, final state within 1e-4; at the full cfg2 shape in bf16
    the masks agree on all but the pixels at bf16-size margins."""
    import dataclasses
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    torch.manual_seed(17)
    cfg = GDKVMConfig()
    scan_m = GDKVM(cfg).eval()
    with torch.no_grad():
        scan_m.mask_embed.weight.zero_()
    step_m = GDKVM(dataclasses.replace(cfg, mask_feedback=True)).eval()
    step_m.load_state_dict(scan_m.state_dict())
    frames = torch.rand(2, 6, 3, 112, 112, device="cuda")
    scan_g, step_g = scan_m.cuda().to(memory_format=torch.channels_last), step_m.cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():
        l0, s0 = scan_g(frames, return_state=True, _lowres=True)
        l1, s1 = step_g(frames, return_state=True, _lowres=True)
    assert (l0 - l1).abs().max().item() <= 1e-3 and (s0 - s1).abs().max().item() <= 1e-4
    # full cfg2 shape, fused bf16 build, head balanced so that the masks are mixed
    with torch.no_grad():
        big = torch.rand(16, 32, 3, 112, 112, device="cuda")
        lg = scan_g(big[:2, :8], _lowres=True)
        gap = (lg[:, :, 0] - lg[:, :, 1]).median()
        scan_g.decoder.head.bias[1] += gap
        step_g.decoder.head.bias[1] += gap
    fs = scan_g.fuse_for_inference().to(torch.bfloat16)
    ft = step_g.fuse_for_inference().to(torch.bfloat16)
    m0, m1 = fs.segment(big.bfloat16())[0], ft.segment(big.bfloat16())[0]
    fg = (m0 != 0).float().mean().item()
    assert 0.1 < fg < 0.9, fg
    assert (m0 == m1).float().mean().item() >= 0.97


def test_normalizer_chunk_identity_at_the_full_cfg2_shape(hip):
    """gdkvm_scan_fwd_normalizer at BASELINE configs[1]'s size (16 clips x 32 frames x 49 tokens, Dv = 256, bf16): the clip as four calls of
    eight frames with (S, z) carried is bit-identical to one call -- the size-independent property of the scan, inherited by the normaliser."""
    q, k, v, a, b = _positive_inputs(16, 32, 49, 1, 64, 256, seed=11)
    t = [_dev(x, torch.bfloat16) for x in (q, k, v)] + [_dev(a), _dev(b)]
    R, S, Z = hip.scan_fwd_normalizer(*t, rule=2, flags=3)
    parts, s, z = [], None, None
    for c in range(0, 32, 8):
        r, s, z = hip.scan_fwd_normalizer(*(x[:, c:c + 8].clone() for x in t), s, z, rule=2, flags=3)
        parts.append(r)
    assert torch.equal(torch.cat(parts, 1), R) and torch.equal(s, S) and torch.equal(z, Z)
    assert torch.isfinite(R.float()).all() and float(Z.min()) > 0.0          # positive keys: z stays positive
