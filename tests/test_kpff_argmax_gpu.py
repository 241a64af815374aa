"""GPU parity: gdkvm_kpff_fwd and gdkvm_argmax_dice vs the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_oracle
from oracle import gdkvm_oracle as O
from tests.util import make_kpff_inputs

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t.to(dtype) if dtype is not None else t


@pytest.mark.parametrize("case", [(3, 7, 7, 64, 256, 256), (2, 16, 16, 64, 256, 256), (2, 4, 4, 16, 32, 64),
                                  (1, 5, 3, 32, 16, 16), (2, 14, 14, 64, 64, 128), (1, 1, 1, 16, 16, 16),
                                  (1, 6, 16, 16, 48, 80), (2, 8, 32, 16, 32, 48), (1, 5, 21, 16, 16, 32), (1, 32, 32, 16, 16, 16)])
def test_kpff_fp32(hip, case):
    BT, h, w, Ck, Cv, Cp = case
    L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(BT, h, w, Ck, Cv, Cp, seed=sum(case))
    F = hip.kpff_fwd(_dev(L), _dev(G), _dev(P), _dev(Wa), _dev(ba), _dev(Wl), _dev(Wg), h, w).cpu().numpy()
    Fo = c_oracle.kpff(L, G, P, Wa, ba, Wl, Wg, h, w)
    assert np.abs(F - Fo).max() <= 1e-4
    # without a workspace: the exact fp32-MFMA arm (the default above is the arm on bf16 splits when all channel counts are
    # multiples of 32, and this same kernel otherwise)
    Fe = hip.kpff_fwd(_dev(L), _dev(G), _dev(P), _dev(Wa), _dev(ba), _dev(Wl), _dev(Wg), h, w, exact=True).cpu().numpy()
    assert np.abs(Fe - Fo).max() <= 2e-5
    if Ck % 32 or Cv % 32 or Cp % 32:
        assert np.array_equal(F, Fe)


@pytest.mark.parametrize("case", [(4, 7, 7, 64, 256, 256), (2, 16, 16, 64, 256, 256), (3, 4, 4, 32, 32, 64), (2, 8, 32, 32, 64, 64),
                                  (3, 6, 19, 32, 32, 64), (1, 32, 32, 64, 256, 256), (5, 1, 1, 32, 32, 32)])
def test_kpff_fp32_on_bf16_splits(hip, case):
    """fp32 I/O, channels % 32 == 0, workspace given -> every product is x_h w_h + x_l w_h + x_h w_l on the bf16 MFMA (16
    significant bits per operand), fp32 accumulation, fp32 pooling / residual / epilogue.  Against the fp64-accumulating
    oracle: 3e-5 absolute at unit-variance features (the exact arm: 2e-5 of accumulation-order noise alone), inputs scaled by 2^20
    and 2^-20 keep the same RELATIVE error (bf16 terms carry the whole fp32 exponent range), and weights that are exact in bf16
    with bf16-exact features reproduce the exact arm to its accumulation order."""
    BT, h, w, Ck, Cv, Cp = case
    L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(BT, h, w, Ck, Cv, Cp, seed=sum(case) + 1)
    run = lambda *t, **kw: hip.kpff_fwd(*(_dev(x) for x in t), h, w, **kw).cpu().numpy()
    F = run(L, G, P, Wa, ba, Wl, Wg)
    Fo = c_oracle.kpff(L, G, P, Wa, ba, Wl, Wg, h, w)
    assert np.abs(F - Fo).max() <= 3e-5, np.abs(F - Fo).max()
    # a cached pack (gdkvm_kpff_fwd_packed) computes the same bits
    ws = torch.empty(int(hip.load().gdkvm_kpff_workspace_bytes(Ck, Cv, Cp, 0)), dtype=torch.uint8, device="cuda")
    F1 = hip.kpff_fwd(*(_dev(x) for x in (L, G, P, Wa, ba, Wl, Wg)), h, w, workspace=ws).cpu().numpy()
    F2 = hip.kpff_fwd(*(_dev(x) for x in (L, G, P, Wa, ba, Wl, Wg)), h, w, workspace=ws, packed=True).cpu().numpy()
    assert np.array_equal(F, F1) and np.array_equal(F, F2)
    # exponent range: the mixes scale with the features (gates saturate identically under the same pre-activations: scale
    # the features up and the gate weights down by the same power of two)
    for sc in (2.0 ** 20, 2.0 ** -20):
        Fs = run(L * sc, G * sc, P * sc, Wa / sc, ba, Wl, Wg)
        assert np.abs(Fs / sc - Fo).max() <= 3e-5
    # operands with at most 8 significant bits: their low terms vanish
    Lb, Gb, Pb, Wab, Wlb, Wgb = (O.to_bf16_f32(x) for x in (L, G, P, Wa, Wl, Wg))
    Fb, Fbe = run(Lb, Gb, Pb, Wab, ba, Wlb, Wgb), run(Lb, Gb, Pb, Wab, ba, Wlb, Wgb, exact=True)
    assert np.abs(Fb - Fbe).max() <= 2e-5                     # (the pooled feature is a mean: it keeps its low term)


@pytest.mark.parametrize("case", [(4, 7, 7, 64, 256, 256), (2, 16, 16, 64, 256, 256), (3, 4, 4, 32, 32, 64), (2, 14, 14, 64, 64, 128),
                                  (2, 8, 32, 32, 64, 64), (3, 6, 19, 32, 32, 64), (1, 32, 32, 64, 256, 256)])
def test_kpff_bf16_mfma_arm(hip, case):
    """bf16 I/O, channels % 32 == 0 -> bf16 MFMA arm: weights and the pooled feature are rounded to bf16 (bf16
    autocast accuracy).  Checked against the oracle fed the same bf16-rounded inputs AND weights; what is left is
    the bf16 rounding of the pooled feature and of the output: 4e-3 absolute + half an output ulp (|F| 2^-8)."""
    BT, h, w, Ck, Cv, Cp = case
    L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(BT, h, w, Ck, Cv, Cp, seed=sum(case))
    F = hip.kpff_fwd(_dev(L, torch.bfloat16), _dev(G, torch.bfloat16), _dev(P, torch.bfloat16),
                     _dev(Wa), _dev(ba), _dev(Wl), _dev(Wg), h, w).float().cpu().numpy()
    Fo = c_oracle.kpff(O.to_bf16_f32(L), O.to_bf16_f32(G), O.to_bf16_f32(P), O.to_bf16_f32(Wa), ba, O.to_bf16_f32(Wl),
                       O.to_bf16_f32(Wg), h, w)
    err = np.abs(F - Fo)
    # Measured over these cases (round 4): the largest error beyond the output's own bf16 rounding (half an ulp <= |F| 2^-8) is 3.1e-3
    # -- the bf16 rounding of the pooled feature through the 256-deep global mix -- and the mean error 1.34e-3 .. 1.41e-3; the bounds are
    # those figures plus a third (the round-1 bound, 1e-2 + |F| 2^-7 and a mean of 2e-3, would have passed a kernel twice as wrong)
    assert np.all(err <= 4e-3 + np.abs(Fo) * 2.0 ** -8), (err - np.abs(Fo) * 2.0 ** -8).max()
    assert err.mean() <= 1.8e-3, err.mean()


def test_kpff_bf16_io_exact_arm(hip):
    """bf16 I/O with channel counts that are not multiples of 32 keeps the exact fp32-MFMA arm."""
    BT, h, w, Ck, Cv, Cp = 4, 7, 7, 16, 48, 80
    L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(BT, h, w, Ck, Cv, Cp, seed=3)
    F = hip.kpff_fwd(_dev(L, torch.bfloat16), _dev(G, torch.bfloat16), _dev(P, torch.bfloat16),
                     _dev(Wa), _dev(ba), _dev(Wl), _dev(Wg), h, w).float().cpu().numpy()
    Fo = c_oracle.kpff(O.to_bf16_f32(L), O.to_bf16_f32(G), O.to_bf16_f32(P), Wa, ba, Wl, Wg, h, w)
    assert np.all(np.abs(F - Fo) <= 1e-4 + np.abs(Fo) * 2.0 ** -8)


def test_kpff_known_answers(hip):
    L, G, P, Wa, ba, Wl, Wg = make_kpff_inputs(2, 7, 7, 16, 32, 32, seed=4)
    F = hip.kpff_fwd(_dev(L), _dev(G), _dev(P), _dev(Wa), _dev(ba), _dev(0 * Wl), _dev(0 * Wg), 7, 7).cpu().numpy()
    assert np.array_equal(F, P)                                   # zero mixes -> identity on the pixel feature


@pytest.mark.parametrize("case", [(5, 4, 9, 11), (3, 2, 112, 112), (2, 4, 256, 256), (1, 1, 8, 8), (2, 7, 5, 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_argmax_dice_bit_exact(hip, case, dtype):
    BT, ncls, H, W = case
    rng = np.random.default_rng(sum(case))
    logits = rng.standard_normal(case).astype(np.float32)
    logits[:, :, ::3] = np.round(logits[:, :, ::3])              # exact ties -> lowest index must win
    logits = O.to_bf16_f32(logits)                               # representable in both io dtypes
    target = rng.integers(0, ncls + 1, (BT, H, W)).astype(np.uint8)   # includes an out-of-range label
    m, c = hip.argmax_dice(_dev(logits, dtype), _dev(target))
    mo, co = c_oracle.argmax_dice(logits, target)
    assert np.array_equal(m.cpu().numpy(), mo) and np.array_equal(c.cpu().numpy(), co)
    m2, c2 = hip.argmax_dice(_dev(logits, dtype))
    assert c2 is None and np.array_equal(m2.cpu().numpy(), mo)
    d = hip.dice_from_counts(c).cpu().numpy()
    np.testing.assert_allclose(d, O.dice_from_counts(co[..., 0], co[..., 1], co[..., 2]), rtol=1e-12)


@pytest.mark.parametrize("case", [(3, 2, 28, 28, 112, 112), (2, 4, 64, 64, 256, 256), (2, 3, 7, 5, 30, 17), (1, 2, 28, 28, 28, 28), (2, 6, 9, 8, 36, 32),
                                  (1, 5, 40, 40, 160, 160)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_argmax_dice_bit_exact(hip, case, dtype):
    """Fused bilinear upsample + argmax + Dice against the scalar oracle (same un-fused fp32 formula): bit-exact."""
    BT, ncls, hl, wl, H, W = case
    rng = np.random.default_rng(sum(case))
    logits = O.to_bf16_f32(rng.standard_normal((BT, ncls, hl, wl)).astype(np.float32))
    logits[:, :, ::4] = np.round(logits[:, :, ::4])
    target = rng.integers(0, ncls + 1, (BT, H, W)).astype(np.uint8)
    m, c = hip.upsample_argmax_dice(_dev(logits, dtype), H, W, _dev(target))
    mo, co = c_oracle.upsample_argmax_dice(logits, H, W, target)
    assert np.array_equal(m.cpu().numpy(), mo) and np.array_equal(c.cpu().numpy(), co)
    # and against torch's own upsample + argmax: identical up to fused-multiply-add rounding at near-ties
    ref = torch.nn.functional.interpolate(torch.from_numpy(logits), size=(H, W), mode="bilinear", align_corners=False).argmax(1)
    assert (ref.numpy() != mo).mean() <= 2e-3


@pytest.mark.parametrize("case", [(5, 64, 2, 28, 28, 112, 112), (2, 32, 4, 64, 64, 256, 256), (3, 64, 3, 7, 5, 30, 17), (2, 128, 6, 9, 8, 36, 32),
                                  (600, 64, 2, 28, 28, 112, 112)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_head_folded_into_the_argmax_kernel(hip, case, dtype):
    """gdkvm_head_upsample_argmax_dice == gdkvm_head_logits then gdkvm_upsample_argmax_dice, bit for bit (masks and counts): every
    block computes the class planes of the low-resolution rows it needs, with the head kernel's sums and its rounding."""
    BT, C, ncls, hl, wl, H, W = case
    g = torch.Generator().manual_seed(sum(case[1:]))
    x = torch.randn(min(BT, 8), C, hl, wl, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    if BT > 8:
        x = x.repeat(BT // 8, 1, 1, 1).contiguous(memory_format=torch.channels_last)
    BT = x.shape[0]
    w = (torch.randn(ncls, C, generator=g) / C ** 0.5).cuda()
    b = torch.randn(ncls, generator=g).cuda()
    target = torch.randint(0, ncls + 1, (BT, H, W), generator=g, dtype=torch.uint8).cuda()
    lo = hip.head_logits(x, w, b)
    m0, c0 = hip.upsample_argmax_dice(lo, H, W, target)
    m1, c1 = hip.head_upsample_argmax_dice(x, w, b, H, W, target)
    assert torch.equal(m0, m1) and torch.equal(c0, c1)
    m2, c2 = hip.head_upsample_argmax_dice(x, w, b, H, W)
    assert c2 is None and torch.equal(m2, m0)
    assert len(torch.unique(m0)) >= 2                              # (a mixed mask: the comparison is not vacuous)
    with pytest.raises(hip.GdkvmError):
        hip.head_upsample_argmax_dice(x[:, :C - 8].contiguous(memory_format=torch.channels_last) if C > 8 else x, w, b, H, W)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 64, 28, 28), (2, 8, 5, 7), (1, 256, 7, 7)])
def test_bias_act_epilogue(hip, dtype, shape):
    """Fused bias + residual + ReLU over NHWC against the same arithmetic in torch (fp32 math, one rounding)."""
    torch.manual_seed(sum(shape))
    x = torch.randn(shape, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    r = torch.randn(shape, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    b = torch.randn(shape[1], device="cuda")
    for res, relu in ((None, True), (r, True), (r, False), (None, False)):
        want = (x.float() if res is None else x.float() + res.float()) + b.reshape(1, -1, 1, 1)     # the kernel's order
        want = (want.clamp_min(0) if relu else want).to(dtype)
        got = hip.bias_act_(x.clone(memory_format=torch.channels_last), b, res, relu)
        assert torch.equal(got, want)


@pytest.mark.parametrize("case", [(4, 256, 7, 7, 128, 14, 14), (2, 128, 14, 14, 64, 28, 28), (1, 16, 5, 3, 8, 9, 7)])
def test_upsample_cat(hip, case):
    n, c1, hl, wl, c2, H, W = case
    torch.manual_seed(sum(case))
    lo = torch.randn(n, c1, hl, wl, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    sk = torch.randn(n, c2, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    got = hip.upsample_cat(lo, sk)
    up = torch.nn.functional.interpolate(lo.float(), size=(H, W), mode="bilinear", align_corners=False)
    want = torch.cat([up, sk.float()], 1)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got[:, c1:], sk)
    assert (got[:, :c1].float() - up).abs().max() <= 2.0 ** -7 * up.abs().max()     # one bf16 rounding of an fp32 blend
    # exactly 2x runs on the row-pair kernel (four tap loads per 2 x 2 outputs); the row-at-a-time kernel gives the same bits
    import os
    os.environ["GDKVM_UPSAMPLE_ROW_PAIRS"] = "0"
    try:
        assert torch.equal(hip.upsample_cat(lo, sk), got)
        assert torch.equal(hip.upsample_bilinear(lo, (H, W)), got[:, :c1])
    finally:
        del os.environ["GDKVM_UPSAMPLE_ROW_PAIRS"]
    assert torch.equal(hip.upsample_bilinear(lo, (H, W)), got[:, :c1])


def test_upsample_cat_more_than_2_20_output_rows(hip):
    """The kernel's float-reciprocal row split is exact below 2^20 rows; a larger batch (16400 frames x 64 rows) goes as several
    launches over image ranges and must equal the small-batch result frame for frame."""
    n, c1, hl, wl, c2, H, W = 16400, 8, 32, 2, 8, 64, 4
    torch.manual_seed(9)
    lo = torch.randn(n, c1, hl, wl, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    sk = torch.randn(n, c2, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    got = hip.upsample_cat(lo, sk)
    for sl in (slice(0, 3), slice(16380, 16384), slice(16384, 16400)):            # either side of the launch boundary
        part = hip.upsample_cat(lo[sl].contiguous(memory_format=torch.channels_last), sk[sl].contiguous(memory_format=torch.channels_last))
        assert torch.equal(got[sl], part)
    assert torch.equal(got[:, c1:], sk)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 16, 9, 11), (2, 64, 56, 56), (1, 8, 1, 1), (2, 24, 4, 7)])
def test_bias_relu_maxpool(hip, dtype, shape):
    """Stem tail (row n1): one pass == max_pool2d(relu(x + b), 3, 2, 1), bit for bit (max commutes with the monotonic x + b)."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, h, w, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(c, generator=g).cuda()
    got = hip.bias_relu_maxpool(x, b)
    want = F.max_pool2d(F.relu(x.float() + b.reshape(1, -1, 1, 1)).to(dtype).float(), 3, 2, 1).to(dtype)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


def test_bias_relu_maxpool_on_a_cropped_view(hip):
    """The stem's space-to-depth convolution returns one row and column too many: the pool reads the top-left corner."""
    g = torch.Generator().manual_seed(3)
    big = torch.randn(2, 16, 9, 11, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(16, generator=g).cuda()
    x = big[:, :, :8, :10]
    got = hip.bias_relu_maxpool(x, b)
    want = F.max_pool2d(F.relu(x.float() + b.reshape(1, -1, 1, 1)).bfloat16().float(), 3, 2, 1).bfloat16()
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_s2d_and_its_convolution(hip, dtype):
    """gdkvm_stem_s2d == pixel_unshuffle (+ zero channels), and the 4x4 space-to-depth kernel FusedConvPool builds reproduces the
    7x7 / stride 2 / pad 3 convolution."""
    from gdkvm_amd.model import FusedConv, FusedConvPool
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 20, 24, generator=g).to(dtype).cuda()
    xs = hip.stem_s2d(x, 16)
    want = torch.zeros(3, 16, 10, 12, dtype=dtype, device="cuda")
    want[:, :12] = F.pixel_unshuffle(x, 2)
    assert xs.is_contiguous(memory_format=torch.channels_last) and torch.equal(xs, want)
    conv = torch.nn.Conv2d(3, 8, 7, 2, 3, bias=True)
    torch.nn.init.normal_(conv.weight, std=0.1, generator=g)
    fc = FusedConvPool.__new__(FusedConvPool)
    torch.nn.Module.__init__(fc)
    base = FusedConv(conv, True)
    fc.conv, fc.epi, fc.relu = base.conv, base.epi, True
    fc = fc.enable_s2d().cuda()
    y7 = F.conv2d(x.float(), conv.weight.cuda().float(), None, 2, 3)
    y4 = F.conv2d(xs.float(), fc.w_s2d.float(), None, 1, 2)[:, :, :10, :12]
    assert (y7 - y4).abs().max().item() <= 1e-4 * max(1.0, y7.abs().max().item())


@pytest.mark.parametrize("kernel", [0, 4, 5])
@pytest.mark.parametrize("case", [(3, 64, 9, 11, 64, 1, True), (2, 16, 28, 28, 24, 2, False), (1, 128, 7, 7, 256, 1, True), (5, 8, 5, 4, 8, 1, False),
                                  (2, 192, 12, 12, 64, 1, False), (5, 64, 28, 28, 64, 1, True), (2, 64, 30, 41, 64, 1, False),
                                  (1, 64, 64, 64, 64, 1, True), (3, 64, 1, 1, 64, 1, False), (700, 64, 6, 5, 64, 1, True),
                                  (9, 128, 14, 14, 128, 1, True), (7, 256, 7, 7, 256, 1, True), (3, 384, 14, 14, 128, 1, False),
                                  (2, 192, 28, 28, 64, 1, False), (5, 128, 16, 10, 48, 1, True), (2, 64, 9, 70, 64, 1, False),
                                  (20, 256, 4, 4, 256, 1, True), (3, 128, 64, 64, 64, 1, False), (33, 64, 2, 3, 128, 1, False), (4, 64, 14, 14, 160, 1, True),
                                  (2, 128, 7, 9, 96, 1, False)])
def test_conv_with_fused_epilogue(hip, kernel, case):
    """gdkvm_conv_bias_act (both hand-written kernels, and the choice by shape) == act(conv2d + bias (+ residual)) computed in fp32
    and rounded once; shapes neither kernel covers -- strided layers, channel counts that are no multiple of 64 / 16, rows wider
    than 64 pixels for the chunked kernel -- fail loudly (the model keeps them on the framework convolution + epilogue pass)."""
    n, c, h, w, k, stride, with_res = case
    torch.manual_seed(sum(case[:6]) + kernel)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(n, c, h, w, device="cuda").bfloat16().contiguous(**cl)
    wt = (torch.randn(k, c, 3, 3, device="cuda") / (9 * c) ** 0.5).bfloat16().contiguous(**cl)
    b = torch.randn(k, device="cuda")
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    r = torch.randn(n, k, ho, wo, device="cuda").bfloat16().contiguous(**cl) if with_res else None
    c64 = c == 64 and k == 64
    served = stride == 1 and ((kernel in (0, 4) and c64) or (kernel == 5 and c % 64 == 0 and k % 16 == 0 and w <= 64)
                              or (kernel == 0 and not c64 and c % 64 == 0 and k % 16 == 0 and w <= 64))
    if not served:
        with pytest.raises(hip.GdkvmError):
            hip.conv_bias_act(x, wt, b, r, stride, 1, True, kernel)
        return
    for relu in (True, False):
        got = hip.conv_bias_act(x, wt, b, r, stride, 1, relu, kernel)
        want = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), stride, 1)
        want = want + r.double() if with_res else want
        want = want.relu() if relu else want
        assert got.shape == want.shape and got.is_contiguous(**cl) and got.dtype == torch.bfloat16
        assert (got.double() - want).abs().max() <= 2.0 ** -7 * max(1.0, want.abs().max().item())
    with pytest.raises(hip.GdkvmError):
        hip.conv_bias_act(x.float(), wt, b, r, stride, 1, True, kernel)
    if kernel == 5:
        # the wave-grid variants (6..8) and the fragment-ordered weights compute the same sums in the same order: bit-identical
        want = hip.conv_bias_act(x, wt, b, r, stride, 1, True, 5)
        pk = hip.conv3x3_pack_weights(wt)
        for variant in (5, 6, 7, 8, 11) + ((10,) if k % 128 == 0 else ()):  # (10, 11: four waves on half-size tiles, two workgroups per CU)
            assert torch.equal(hip.conv_bias_act(x, wt, b, r, stride, 1, True, variant), want)
            assert torch.equal(hip.conv_bias_act(x, wt, b, r, stride, 1, True, variant, pk), want)
        if c64:
            assert torch.equal(hip.conv_bias_act(x, wt, b, r, stride, 1, True, 4, pk), hip.conv_bias_act(x, wt, b, r, stride, 1, True, 4))


@pytest.mark.parametrize("case", [(9, 64, 28, 28, 128, 3, 2, 1, False), (5, 128, 14, 14, 256, 3, 2, 1, True), (7, 64, 28, 28, 128, 1, 2, 0, False),
                                  (3, 128, 14, 14, 256, 1, 2, 0, True), (2, 64, 20, 80, 128, 3, 1, 1, True), (1, 32, 9, 11, 128, 5, 2, 2, False),
                                  (3, 96, 7, 7, 128, 3, 1, 1, True), (1, 64, 70, 70, 256, 3, 2, 1, False), (2, 64, 5, 4, 128, 1, 1, 0, False)])
def test_conv_implicit_gemm_kernel(hip, case):
    """gdkvm_conv_bias_act kernel 9 (the general implicit-GEMM kernel: strided layers, 1x1 layers, rows wider than 64 pixels) ==
    act(conv2d + bias (+ residual)) computed in fp64 and rounded once; where the 3x3 / 1 / 1 kernel also serves the shape the two
    agree to a rounding of the sums' order; unpacked weights and channel counts it does not take fail loudly."""
    n, c, h, w, k, rs, stride, pad, with_res = case
    torch.manual_seed(sum(case[:8]))
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(n, c, h, w, device="cuda").bfloat16().contiguous(**cl)
    wt = (torch.randn(k, c, rs, rs, device="cuda") / (rs * rs * c) ** 0.5).bfloat16().contiguous(**cl)
    b = torch.randn(k, device="cuda")
    ho, wo = (h + 2 * pad - rs) // stride + 1, (w + 2 * pad - rs) // stride + 1
    r = torch.randn(n, k, ho, wo, device="cuda").bfloat16().contiguous(**cl) if with_res else None
    pk = hip.conv_igemm_pack_weights(wt)
    for relu in (True, False):
        got = hip.conv_bias_act(x, wt, b, r, stride, pad, relu, hip.CONV_KERNEL_IGEMM, pk)
        want = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), stride, pad)
        want = want + r.double() if with_res else want
        want = want.relu() if relu else want
        assert got.shape == want.shape and got.is_contiguous(**cl) and got.dtype == torch.bfloat16
        assert (got.double() - want).abs().max() <= 2.0 ** -7 * max(1.0, want.abs().max().item())
    if rs == 3 and stride == 1 and pad == 1 and c % 64 == 0 and w <= 64:
        other = hip.conv_bias_act(x, wt, b, r, 1, 1, True, 5)
        assert (got.float() * 0 + hip.conv_bias_act(x, wt, b, r, 1, 1, True, hip.CONV_KERNEL_IGEMM, pk).float() - other.float()).abs().max() \
            <= 2.0 ** -7 * max(1.0, other.float().abs().max().item())
    with pytest.raises(hip.GdkvmError, match="pack"):
        hip.conv_bias_act(x, wt, b, r, stride, pad, True, hip.CONV_KERNEL_IGEMM)
    with pytest.raises(hip.GdkvmError):
        hip.conv_bias_act(x, wt[:64].contiguous(**cl), b[:64], None, stride, pad, True, hip.CONV_KERNEL_IGEMM, hip.conv_igemm_pack_weights(wt[:64].contiguous(**cl)))


@pytest.mark.parametrize("case", [(9, 64, 28, 28, 128, 3, 2), (5, 128, 14, 14, 256, 3, 2), (2, 32, 9, 11, 128, 5, 2), (3, 96, 8, 8, 128, 3, 1), (1, 64, 31, 17, 128, 3, 2)])
def test_conv_with_its_downsample_branch_in_one_launch(hip, case):
    """gdkvm_conv_down_bias_act == (kernel 9 on the R x S layer, kernel 9 on the 1x1 layer of the same stride), bit for bit, and both
    right against fp64; with and without a bias on the branch."""
    n, c, h, w, k, rs, stride = case
    torch.manual_seed(sum(case))
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(n, c, h, w, device="cuda").bfloat16().contiguous(**cl)
    wt = (torch.randn(k, c, rs, rs, device="cuda") / (rs * rs * c) ** 0.5).bfloat16().contiguous(**cl)
    wd = (torch.randn(k, c, 1, 1, device="cuda") / c ** 0.5).bfloat16().contiguous(**cl)
    b, bd = torch.randn(k, device="cuda"), torch.randn(k, device="cuda")
    pk, pkd = hip.conv_igemm_pack_weights(wt), hip.conv_igemm_pack_weights(wd)
    zero = torch.zeros(k, device="cuda")
    for dbias in (None, bd):
        y, yd = hip.conv_down_bias_act(x, wt, b, pk, wd, pkd, dbias, stride, True)
        y0 = hip.conv_bias_act(x, wt, b, None, stride, rs // 2, True, hip.CONV_KERNEL_IGEMM, pk)
        yd0 = hip.conv_bias_act(x, wd, zero if dbias is None else dbias, None, stride, 0, False, hip.CONV_KERNEL_IGEMM, pkd)
        assert torch.equal(y, y0) and torch.equal(yd, yd0)
        want = torch.nn.functional.conv2d(x.double(), wd.double(), None if dbias is None else dbias.double(), stride, 0)
        assert (yd.double() - want).abs().max() <= 2.0 ** -7 * max(1.0, want.abs().max().item())
    with pytest.raises(hip.GdkvmError):
        hip.conv_down_bias_act(x, wt[:64].contiguous(**cl), b[:64], hip.conv_igemm_pack_weights(wt[:64].contiguous(**cl)), wd[:64].contiguous(**cl),
                               hip.conv_igemm_pack_weights(wd[:64].contiguous(**cl)))


@pytest.mark.parametrize("case", [(5, 256, 128, 14, 14, 128), (3, 128, 64, 28, 28, 64), (9, 64, 64, 7, 5, 48), (2, 192, 64, 10, 33, 160)])
def test_conv_over_a_concatenation_that_is_never_built(hip, case):
    """gdkvm_conv_cat_bias_act([x1 ; x2]) == gdkvm_conv_bias_act(cat(x1, x2)) bit for bit (plain and packed weights), and the
    enlargement alone (gdkvm_upsample_cat without a skip tensor) == the first C1 channels of upsample_cat."""
    n, c1, c2, h, w, k = case
    torch.manual_seed(sum(case))
    cl = dict(memory_format=torch.channels_last)
    lo = torch.randn(n, c1, (h + 1) // 2, (w + 1) // 2, device="cuda").bfloat16().contiguous(**cl)
    x2 = torch.randn(n, c2, h, w, device="cuda").bfloat16().contiguous(**cl)
    both = hip.upsample_cat(lo, x2)
    x1 = hip.upsample_bilinear(lo, (h, w))
    assert x1.is_contiguous(**cl) and torch.equal(x1, both[:, :c1]) and torch.equal(both[:, c1:], x2)
    wt = (torch.randn(k, c1 + c2, 3, 3, device="cuda") / (9 * (c1 + c2)) ** 0.5).bfloat16().contiguous(**cl)
    b = torch.randn(k, device="cuda")
    r = torch.randn(n, k, h, w, device="cuda").bfloat16().contiguous(**cl)
    pk = hip.conv3x3_pack_weights(wt)
    for res in (None, r):
        want = hip.conv_bias_act(both, wt, b, res, 1, 1, True, 5)
        assert torch.equal(hip.conv_cat_bias_act(x1, x2, wt, b, res, True), want)
        assert torch.equal(hip.conv_cat_bias_act(x1, x2, wt, b, res, True, 0, pk), want)
    with pytest.raises(hip.GdkvmError):
        hip.conv_cat_bias_act(x1[:, :c1 - 8].contiguous(**cl), x2, wt[:, :c1 + c2 - 8].contiguous(**cl), b)     # (chunks of 64 channels)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(6, 49, 256, 1), (3, 256, 256, 2), (2, 7, 64, 1), (5, 1, 32, 3)])
def test_gate_logits_one_pass(hip, dtype, case):
    """gdkvm_gate_logits == gate projection per token and decay projection of the token mean, in fp32 from the same inputs."""
    fr, n, cp, hh = case
    torch.manual_seed(sum(case))
    p = torch.randn(fr, n, cp, device="cuda").to(dtype)
    wg, bg = torch.randn(hh, cp, device="cuda") / cp ** 0.5, torch.randn(hh, device="cuda")
    wd, bd = torch.randn(hh, cp, device="cuda") / cp ** 0.5, torch.randn(hh, device="cuda")
    beta, alpha = hip.gate_logits(p, wg, bg, wd, bd)
    p64 = p.double()
    beta_ref = p64 @ wg.double().t() + bg.double()
    alpha_ref = p64.mean(1) @ wd.double().t() + bd.double()
    assert beta.dtype == torch.float32 and beta.shape == (fr, n, hh) and alpha.shape == (fr, hh)
    assert (beta.double() - beta_ref).abs().max() <= 1e-5 * max(1.0, beta_ref.abs().max().item())
    assert (alpha.double() - alpha_ref).abs().max() <= 1e-5 * max(1.0, alpha_ref.abs().max().item())


@pytest.mark.parametrize("case", [(3, 56, 56), (2, 128, 128), (1, 8, 8), (2, 30, 44), (5, 2, 2), (1, 57, 23)])
def test_stem_conv_pool_in_one_kernel(hip, case):
    """gdkvm_stem_conv_pool == max_pool(relu(conv4x4 pad 2 cropped + bias)) computed in fp64 from the same bf16 inputs, rounded once."""
    n, hs, ws = case
    torch.manual_seed(sum(case))
    cl = dict(memory_format=torch.channels_last)
    xs = torch.randn(n, 16, hs, ws, device="cuda").bfloat16().contiguous(**cl)
    w = (torch.randn(64, 16, 4, 4, device="cuda") / 16).bfloat16().contiguous(**cl)
    b = torch.randn(64, device="cuda")
    got = hip.stem_conv_pool(xs, w, b)
    conv = torch.nn.functional.conv2d(xs.double(), w.double(), b.double(), 1, 2)[:, :, :hs, :ws]
    want = torch.nn.functional.max_pool2d(conv.relu().bfloat16().double(), 3, 2, 1)      # the kernel rounds before pooling (max commutes)
    assert got.shape == want.shape and got.is_contiguous(**cl)
    assert (got.double() - want).abs().max() <= 2.0 ** -7 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("case", [(3, 3, 112, 112), (2, 3, 256, 256), (1, 1, 16, 16), (2, 4, 60, 88), (5, 3, 4, 4), (1, 2, 114, 46), (17, 3, 112, 112)])
def test_stem_reads_the_nchw_frames_itself(hip, case):
    """gdkvm_stem_conv_pool_nchw (the workgroup builds its band of the space-to-depth image from the NCHW frames on the way into LDS) is
    gdkvm_stem_s2d followed by gdkvm_stem_conv_pool, bit for bit: 1 .. 4 input channels, ragged sizes, more tiles than workgroups."""
    n, c, hh, ww = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, hh, ww, device="cuda").bfloat16()
    w = (torch.randn(64, 16, 4, 4, device="cuda") / 16).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(64, device="cuda")
    two = hip.stem_conv_pool(hip.stem_s2d(x, 16), w, b)
    one = hip.stem_conv_pool_nchw(x, w, b)
    torch.cuda.synchronize()
    assert one.shape == two.shape and torch.equal(one, two)


@pytest.mark.parametrize("case", [(25088, 256, (64, 64, 256)), (1000, 256, (64, 64, 256)), (77, 64, (16, 32)), (5, 512, (48,)), (129, 32, (16, 16, 16))])
def test_key_query_value_projections_in_one_pass(hip, case):
    """gdkvm_proj_rows == x W^T + b per projection, fp32 accumulation over the bf16-rounded weights, one rounding."""
    rows, k, widths = case
    torch.manual_seed(rows + k)
    x = torch.randn(rows, k, device="cuda").bfloat16()
    w = torch.randn(sum(widths), k, device="cuda") / k ** 0.5
    b = torch.randn(sum(widths), device="cuda")
    outs = hip.proj_rows(x, hip.pack_rows_weight(w), b, widths)
    ref = x.double() @ w.bfloat16().double().t() + b.double()
    assert len(outs) == len(widths)
    c0 = 0
    for o, wd in zip(outs, widths):
        assert o.shape == (rows, wd) and o.is_contiguous() and o.dtype == torch.bfloat16
        want = ref[:, c0:c0 + wd]
        assert (o.double() - want).abs().max() <= 2.0 ** -7 * max(1.0, want.abs().max().item())
        c0 += wd
    with pytest.raises(hip.GdkvmError):
        hip.proj_rows(x.float(), hip.pack_rows_weight(w), b, widths)


@pytest.mark.parametrize("case", [(512, 49, 256, 1, 64, 256), (40, 256, 256, 1, 64, 64), (7, 49, 256, 2, 64, 32), (3, 5, 64, 1, 64, 16),
                                  (1400, 49, 128, 1, 64, 128)])       # (the last: 68600 rows -> 128-token tiles)
def test_projections_gates_and_norms_in_one_launch(hip, case):
    """gdkvm_proj_gates == gdkvm_proj_rows + gdkvm_gate_logits + the inverse norms of the stored key / query rows.
    Projections bit-identical to gdkvm_proj_rows (same tile machinery), gate logits against fp64, norms against the rows as stored;
    and gdkvm_scan_fwd_normed fed those norms agrees with gdkvm_scan_fwd computing its own."""
    fr, n, cp, hh, dk, dv = case
    torch.manual_seed(sum(case))
    p = torch.randn(fr, n, cp, device="cuda").bfloat16()
    wk, wv = hh * dk, hh * dv
    w = torch.randn(2 * wk + wv, cp, device="cuda") / cp ** 0.5
    b = torch.randn(2 * wk + wv, device="cuda")
    wg, bg = torch.randn(hh, cp, device="cuda") / cp ** 0.5, torch.randn(hh, device="cuda")
    wd, bd = torch.randn(hh, cp, device="cuda") / cp ** 0.5, torch.randn(hh, device="cuda")
    pack = hip.pack_rows_weight(w)
    (k, q, v), (beta, alpha), norms = hip.proj_gates(p, pack, b, wg, bg, wd, bd, hh, dk, dv)
    k0, q0, v0 = hip.proj_rows(p.reshape(fr * n, cp), pack, b, (wk, wk, wv))
    assert torch.equal(k, k0) and torch.equal(q, q0) and torch.equal(v, v0)
    p64 = p.double()
    beta_ref = p64 @ wg.double().t() + bg.double()
    alpha_ref = p64.mean(1) @ wd.double().t() + bd.double()
    assert beta.shape == (fr, n, hh) and alpha.shape == (fr, hh) and norms.shape == (fr * n, hh, 2)
    assert (beta.double() - beta_ref).abs().max() <= 1e-5 * max(1.0, beta_ref.abs().max().item())
    assert (alpha.double() - alpha_ref).abs().max() <= 1e-5 * max(1.0, alpha_ref.abs().max().item())
    kn = 1.0 / (k.double().reshape(fr * n, hh, dk).pow(2).sum(-1) + 1e-12).sqrt()
    qn = 1.0 / (q.double().reshape(fr * n, hh, dk).pow(2).sum(-1) + 1e-12).sqrt()
    assert ((norms[..., 0].double() - kn).abs() <= 1e-6 * kn).all() and ((norms[..., 1].double() - qn).abs() <= 1e-6 * qn).all()
    assert torch.equal(norms, hip.proj_gates(p, pack, b, wg, bg, wd, bd, hh, dk, dv)[2])          # deterministic
    # the scan with the norms given against the scan computing them (B clips x T frames out of the rows)
    B = 1 if fr % 2 else 2
    T = fr // B
    if T <= 64:
        sh = lambda t, c: t.reshape(B, T, n, hh, c)
        args = (sh(q, dk), sh(k, dk), sh(v, dv), alpha.reshape(B, T, hh), beta.reshape(B, T, n, hh))
        r0, s0 = hip.scan_fwd(*args, flags=3)
        r1, s1 = hip.scan_fwd(*args, flags=3, norms=norms)
        assert (s1 - s0).abs().max() <= 1e-5 * max(1.0, s0.abs().max().item())
        assert (r1.float() - r0.float()).abs().max() <= 2.0 ** -7 * max(1.0, r0.float().abs().max().item())
        with pytest.raises(hip.GdkvmError):
            hip.scan_fwd(*args, flags=2, norms=norms)                  # norms go with FLAG_NORMALIZE_QK


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(5, 64, 28, 28, 2), (2, 64, 64, 64, 4), (3, 32, 5, 7, 3), (1, 256, 3, 3, 8)])
def test_head_logits_planes(hip, dtype, case):
    """gdkvm_head_logits == 1x1 convolution + bias, written as contiguous NCHW planes."""
    n, c, h, w, ncls = case
    if dtype == torch.float32 and ncls > c // 4 or dtype == torch.bfloat16 and ncls > c // 8:
        pytest.skip("more classes than lanes per pixel")
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    wt, b = torch.randn(ncls, c, device="cuda") / c ** 0.5, torch.randn(ncls, device="cuda")
    got = hip.head_logits(x, wt, b)
    want = torch.nn.functional.conv2d(x.double(), wt.double().reshape(ncls, c, 1, 1), b.double())
    assert got.shape == want.shape and got.is_contiguous() and got.dtype == dtype
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert (got.double() - want).abs().max() <= tol * max(1.0, want.abs().max().item())
