"""Row n1, training side: the fused BatchNorm (+ residual) (+ ReLU) passes (csrc/bn.hip), the decoder glue's backward
(gdkvm_upsample_cat_bwd), the stem max-pool, the split-K weight gradients and the fused objective (csrc/loss.hip), each
against torch autograd (fp64 on the CPU where rounding matters), same inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(8, 64, 28, 28), (3, 128, 7, 5), (2, 256, 7, 7), (5, 24, 9, 4), (2, 8, 1, 2), (16, 64, 56, 56)]


def _reference(x, w, b, res, relu, dy, mask, eps=1e-5):
    x64 = x.double().cpu().requires_grad_(True)
    w64, b64 = w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    r64 = None if res is None else res.double().cpu().requires_grad_(True)
    rm, rv = torch.zeros(x.shape[1], dtype=torch.float64), torch.ones(x.shape[1], dtype=torch.float64)
    y = F.batch_norm(x64, rm, rv, w64, b64, True, 0.1, eps)
    if r64 is not None:
        y = y + r64
    if relu:                                    # the mask of the run under test: an output within one rounding of zero may
        y = y * mask.double().cpu()             # legitimately fall on either side
    y.backward(dy.double().cpu())
    return y.detach(), x64.grad, w64.grad, b64.grad, (None if r64 is None else r64.grad), rm, rv


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", ["relu", "plain", "res_relu", "res"])
def test_bn_act_forward_backward(hip, dtype, shape, mode):
    if dtype == torch.float32 and shape[1] % 4:
        pytest.skip("fp32 needs C % 4 == 0")
    torch.manual_seed(sum(shape) + len(mode))
    relu, has_res = "relu" in mode, "res" in mode
    cl = dict(memory_format=torch.channels_last)
    x = (3.0 + 2.0 * torch.randn(shape, device="cuda")).to(dtype).contiguous(**cl)
    res = torch.randn(shape, device="cuda").to(dtype).contiguous(**cl) if has_res else None
    w = torch.rand(shape[1], device="cuda") + 0.5
    b = 0.3 * torch.randn(shape[1], device="cuda")
    dy = torch.randn(shape, device="cuda").to(dtype).contiguous(**cl)
    rm, rv = torch.zeros(shape[1], device="cuda"), torch.ones(shape[1], device="cuda")
    xg = x.clone(**cl).requires_grad_(True)
    wg, bg = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rg = None if res is None else res.clone(**cl).requires_grad_(True)
    y = hip.bn_act(xg, wg, bg, rm, rv, rg, 0.1, 1e-5, relu)
    assert y.dtype == dtype and y.is_contiguous(**cl)
    y.backward(dy)
    y_ref, dx_ref, dw_ref, db_ref, dr_ref, rm_ref, rv_ref = _reference(x, w, b, res, relu, dy, y.detach() > 0)
    # forward: fp32 arithmetic on the same inputs, one rounding to the io dtype
    eps_io = 2.0 ** -8 if dtype == torch.bfloat16 else 1e-5
    assert (y.double().cpu() - y_ref).abs().max() <= eps_io * max(1.0, y_ref.abs().max().item())
    assert (rm.double().cpu() - rm_ref).abs().max() <= 1e-5 and (rv.double().cpu() - rv_ref).abs().max() <= 1e-4
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-4
    scale = max(dx_ref.abs().max().item(), 1e-6)
    assert (xg.grad.double().cpu() - dx_ref).abs().max() <= tol * scale
    assert (wg.grad.double().cpu() - dw_ref).abs().max() <= tol * max(dw_ref.abs().max().item(), 1.0)
    assert (bg.grad.double().cpu() - db_ref).abs().max() <= tol * max(db_ref.abs().max().item(), 1.0)
    if has_res:
        assert (rg.grad.double().cpu() - dr_ref).abs().max() <= tol * max(dr_ref.abs().max().item(), 1.0)


def test_bn_variance_of_a_far_from_zero_signal(hip):
    """|mean| = 1000 std: a plain sum-of-squares variance in fp32 would lose every digit; the shifted sums do not."""
    torch.manual_seed(0)
    x = (500.0 + 0.5 * torch.randn(4, 32, 16, 16, device="cuda")).contiguous(memory_format=torch.channels_last)
    w, b = torch.ones(32, device="cuda"), torch.zeros(32, device="cuda")
    y, stats = hip.bn_act_fwd(x, w, b, relu=False)
    mean, rstd = stats[0], stats[1]
    x64 = x.double().cpu()
    var = x64.var((0, 2, 3), unbiased=False)
    assert (mean.double().cpu() - x64.mean((0, 2, 3))).abs().max() < 1e-3
    assert ((1.0 / rstd.double().cpu() ** 2 - 1e-5) / var - 1).abs().max() < 1e-3


def test_bn_is_deterministic(hip):
    torch.manual_seed(1)
    x = torch.randn(32, 64, 28, 28, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w, b = torch.rand(64, device="cuda"), torch.rand(64, device="cuda")
    dy = torch.randn_like(x)
    outs = []
    for _ in range(2):
        y, stats = hip.bn_act_fwd(x, w, b, relu=True)
        outs.append((y, stats) + tuple(hip.bn_act_bwd(x, y, dy, w, stats, True, True)))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    # the ReLU mask recomputed from x (no y read) is the mask of the saved output
    dx2, _, dg2, db2 = hip.bn_act_bwd(x, None, dy, w, stats, True)
    assert torch.equal(dx2, outs[0][2]) and torch.equal(dg2, outs[0][4]) and torch.equal(db2, outs[0][5])


def test_bn_module_path_matches_torch_batchnorm(hip):
    """model._bn_act on a BatchNorm2d in training mode: same output, same running statistics, same parameter gradients as
    torch's own BatchNorm2d -> ReLU (fp32)."""
    from gdkvm_amd.model import _bn_act
    torch.manual_seed(2)
    bn_a, bn_b = torch.nn.BatchNorm2d(32).cuda().train(), torch.nn.BatchNorm2d(32).cuda().train()
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5); bn_a.bias.normal_()
    bn_b.load_state_dict(bn_a.state_dict())
    x = torch.randn(6, 32, 10, 10, device="cuda").contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = _bn_act(bn_a, xa, True)
    yb = F.relu(bn_b(xb))
    assert torch.allclose(ya, yb, atol=1e-5, rtol=1e-5)
    (ya * ya).sum().backward(); (yb * yb).sum().backward()
    assert torch.allclose(xa.grad, xb.grad, atol=1e-4, rtol=1e-4)
    assert torch.allclose(bn_a.weight.grad, bn_b.weight.grad, atol=1e-3, rtol=1e-4)
    assert torch.allclose(bn_a.bias.grad, bn_b.bias.grad, atol=1e-3, rtol=1e-4)
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, atol=1e-6)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, atol=1e-6)
    assert int(bn_a.num_batches_tracked) == 1


def test_bn_rejects_bad_arguments(hip):
    x = torch.randn(2, 12, 4, 4, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.ones(12, device="cuda")
    with pytest.raises(hip.GdkvmError):
        hip.bn_act_fwd(x, w, w)                                   # C = 12 is not a multiple of 8 in bf16
    with pytest.raises(hip.GdkvmError):
        hip.bn_act_fwd(x.cpu(), w.cpu(), w.cpu())                 # no CPU path


@pytest.mark.parametrize("case", [(4, 256, 7, 7, 128, 14, 14), (2, 128, 14, 14, 64, 28, 28), (1, 16, 5, 3, 8, 9, 7),
                                  (2, 8, 6, 6, 8, 13, 11), (1, 8, 1, 1, 8, 2, 2), (2, 8, 9, 8, 16, 4, 5)])
def test_upsample_cat_backward(hip, case):
    n, c1, hl, wl, c2, H, W = case
    torch.manual_seed(sum(case))
    cl = dict(memory_format=torch.channels_last)
    lo = torch.randn(n, c1, hl, wl, device="cuda").bfloat16().contiguous(**cl).requires_grad_(True)
    sk = torch.randn(n, c2, H, W, device="cuda").bfloat16().contiguous(**cl).requires_grad_(True)
    dout = torch.randn(n, c1 + c2, H, W, device="cuda").bfloat16().contiguous(**cl)
    out = hip.upsample_cat(lo, sk)
    out.backward(dout)
    lo64 = lo.detach().double().cpu().requires_grad_(True)
    up = F.interpolate(lo64, size=(H, W), mode="bilinear", align_corners=False)
    up.backward(dout[:, :c1].double().cpu())
    assert torch.equal(sk.grad, dout[:, c1:])
    assert (lo.grad.double().cpu() - lo64.grad).abs().max() <= 2.0 ** -7 * max(lo64.grad.abs().max().item(), 1.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 16, 9, 11), (2, 64, 56, 56), (1, 8, 1, 1), (2, 24, 4, 7), (2, 8, 2, 5)])
def test_maxpool_forward_and_gather_backward(hip, dtype, shape):
    """Values bit-equal to torch's max_pool2d; gradients equal to torch's backward, ties included (coarse values force many)."""
    if dtype == torch.float32 and shape[1] % 4 or dtype == torch.bfloat16 and shape[1] % 8:
        pytest.skip("channel multiple")
    torch.manual_seed(sum(shape))
    cl = dict(memory_format=torch.channels_last)
    x = (torch.randn(shape, device="cuda") * 2).round().div(2).clamp_min(0).to(dtype).contiguous(**cl)     # ReLU-like, many ties
    dy_shape = (shape[0], shape[1], (shape[2] - 1) // 2 + 1, (shape[3] - 1) // 2 + 1)
    dy = torch.randn(dy_shape, device="cuda").to(dtype).contiguous(**cl)
    xa, xb = x.clone(**cl).requires_grad_(True), x.clone(**cl).float().requires_grad_(True)
    ya = hip.maxpool3x3s2(xa)
    yb = F.max_pool2d(xb, 3, 2, 1)
    assert torch.equal(ya.float(), yb)
    ya.backward(dy); yb.backward(dy.float())
    assert (xa.grad.float() - xb.grad).abs().max() <= (2.0 ** -7 if dtype == torch.bfloat16 else 1e-6) * max(1.0, xb.grad.abs().max().item())


@pytest.mark.parametrize("shape", [(4, 64, 56, 56), (3, 16, 9, 11), (2, 8, 1, 1), (2, 24, 4, 7), (5, 64, 13, 30)])
def test_batchnorm_relu_maxpool_as_one_op(hip, shape):
    """ops.bn_relu_pool (gdkvm_bn_pool_fwd_train / gdkvm_bn_pool_bwd: the training stem's tail without the full-resolution activation or
    its gradient) against ops.maxpool3x3s2(ops.bn_act(x, relu=True)): pooled values and running statistics bit for bit, gradients to fp32
    rounding (bit for bit on the gather form), on inputs with many ties (coarse values) and odd sizes."""
    torch.manual_seed(sum(shape))
    n, c, hh, ww = shape
    cl = dict(memory_format=torch.channels_last)
    x = (torch.randn(shape, device="cuda") * 4).round().div(4).bfloat16().contiguous(**cl)            # coarse values: ties in the windows
    g, b = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.2
    dy = torch.randn(n, c, (hh - 1) // 2 + 1, (ww - 1) // 2 + 1, device="cuda").bfloat16().contiguous(**cl)

    def run(fused):
        xa, ga, ba = x.clone(**cl).requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        if fused:
            y = hip.bn_relu_pool(xa, ga, ba, rm, rv, 0.1, 1e-5)
        else:
            y = hip.maxpool3x3s2(hip.bn_act(xa, ga, ba, rm, rv, None, 0.1, 1e-5, True))
        y.backward(dy)
        return y.detach(), xa.grad, ga.grad, ba.grad, rm, rv

    ya, dxa, dga, dba, rma, rva = run(False)
    yb, dxb, dgb, dbb, rmb, rvb = run(True)
    assert torch.equal(ya, yb) and torch.equal(rma, rmb) and torch.equal(rva, rvb)          # forward: the same bits
    # backward: the same gradient per element; the two sums over the pixels are added block-major on the 2 x 2 form (channel-group counts
    # that divide 256), so dgamma / dbeta agree to fp32 rounding and dx to a bf16 ulp on the rare element that rounding moves
    for a_, b_ in ((dga, dgb), (dba, dbb)):
        assert (a_ - b_).abs().max() <= 1e-5 * max(1.0, a_.abs().max().item())
    d = (dxa.float() - dxb.float()).abs()
    assert d.max() <= 2.0 ** -7 * max(1.0, dxa.float().abs().max().item()) and (d > 0).float().mean() <= 1e-2
    if 256 % (c // 8):
        assert torch.equal(dxa, dxb) and torch.equal(dga, dgb)                               # (the gather form: bit for bit)
    assert hip.bn_relu_pool_served(x) and not hip.bn_relu_pool_served(x.float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_token_linear_and_split_k_weight_gradient(hip, dtype):
    torch.manual_seed(5)
    K, cin, cout = 16 * 49 * 8, 64, 40
    x = torch.randn(K, cin, device="cuda").to(dtype)
    w = (torch.randn(cout, cin, device="cuda") / 8).requires_grad_(True)
    b = torch.randn(cout, device="cuda").requires_grad_(True)
    dy = torch.randn(K, cout, device="cuda").to(dtype)
    xg = x.clone().requires_grad_(True)
    y = hip.token_linear(xg, w, b)
    y.backward(dy)
    x64, w64, b64 = x.double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    if dtype == torch.bfloat16:
        w_used = w.detach().bfloat16().double().cpu()
        y64 = x64 @ w_used.t() + b.detach().bfloat16().double().cpu()
    else:
        y64 = F.linear(x64, w64, b64)
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-4
    assert (y.double().cpu() - y64.detach()).abs().max() <= tol * y64.abs().max().item()
    dw_ref = dy.double().cpu().t() @ x.double().cpu()
    assert w.grad.dtype == torch.float32
    assert (w.grad.double().cpu() - dw_ref).abs().max() <= 1e-5 * dw_ref.abs().max().item()        # fp32 partials: no bf16 rounding
    assert (b.grad.double().cpu() - dy.double().cpu().sum(0)).abs().max() <= 1e-4 * K ** 0.5
    dx_ref = dy.double().cpu() @ (w.detach().to(dtype).double().cpu())
    assert (xg.grad.double().cpu() - dx_ref).abs().max() <= tol * dx_ref.abs().max().item()
    # token counts that are no multiple of the row split or of the 32-row LDS slab
    assert torch.allclose(hip.wgrad(dy[:1001], x[:1001]), (dy[:1001].float().t() @ x[:1001].float()), rtol=1e-4, atol=1e-3)
    # the bias gradient from the same launch (gdkvm_gemm_tn_colsum): column sums of dy, fp32 accumulation, the same bits on every run
    dw2, db2 = hip.wgrad(dy[:1001], x[:1001], colsum=True)
    assert torch.equal(dw2, hip.wgrad(dy[:1001], x[:1001]))
    assert (db2.double().cpu() - dy[:1001].double().cpu().sum(0)).abs().max() <= 1e-4 * 1001 ** 0.5
    assert torch.equal(db2, hip.wgrad(dy[:1001], x[:1001], colsum=True)[1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", [(3, 64, 28, 28, 2), (2, 32, 9, 5, 4), (1, 64, 7, 7, 8)])
def test_head_forward_and_one_pass_backward(hip, case, dtype):
    """ops.head (gdkvm_head_logits forward, gdkvm_head_bwd backward) against conv2d in fp64 on the same operands: logits, dx, dW, db;
    the backward is the same bits on every run."""
    n, c, hh, ww, ncls = case
    if dtype == torch.float32 and ncls > c // 4:
        pytest.skip("classes <= C/4 lanes per pixel in fp32")
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, hh, ww, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    conv = torch.nn.Conv2d(c, ncls, 1).cuda()
    gz = torch.randn(n, ncls, hh, ww, device="cuda").to(dtype)
    z = hip.head(x, conv.weight, conv.bias)
    assert z.is_contiguous() and z.dtype == dtype
    z.backward(gz)
    x64 = x.detach().double().requires_grad_(True)
    w64, b64 = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    z64 = F.conv2d(x64, w64, b64)
    z64.backward(gz.double())
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert (z.double() - z64).abs().max() <= tol * max(1.0, z64.abs().max().item())
    assert (x.grad.double() - x64.grad).abs().max() <= tol * max(1.0, x64.grad.abs().max().item())
    assert conv.weight.grad.dtype == torch.float32 and (conv.weight.grad.double() - w64.grad).abs().max() <= 1e-5 * max(1.0, w64.grad.abs().max().item())
    assert (conv.bias.grad.double() - b64.grad).abs().max() <= 1e-5 * max(1.0, b64.grad.abs().max().item())
    g1 = (x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    x.grad = None; conv.zero_grad(set_to_none=True)
    hip.head(x, conv.weight, conv.bias).backward(gz)
    assert all(torch.equal(a, b) for a, b in zip(g1, (x.grad, conv.weight.grad, conv.bias.grad)))


def test_stacked_token_projections(hip):
    """ops.token_projections (key / query / value / gate as ONE stacked product forward, two backward) against the four separate
    ops.token_linear calls it replaces: the same outputs bit for bit (the same kernel on the same rows of the stacked weight), gradients
    equal up to the fp32 summation order of the data gradient (one product over the stacked width instead of four added up)."""
    torch.manual_seed(9)
    rows, cin = 2 * 3 * 49, 256
    x = torch.randn(rows, cin, device="cuda").bfloat16()
    convs = [torch.nn.Conv2d(cin, n, 1).cuda() for n in (64, 64, 256, 1)]
    gys = [torch.randn(rows, n, device="cuda").bfloat16() for n in (64, 64, 256, 1)]

    def run(stacked):
        xg = x.clone().requires_grad_(True)
        for c in convs:
            c.zero_grad(set_to_none=True)
        if stacked:
            ys = hip.token_projections(xg, convs)
        else:
            ys = []
            for c in convs:
                w2 = c.weight.reshape(c.out_channels, -1)
                if c.out_channels >= 8:
                    ys.append(hip.token_linear(xg, w2, c.bias))
                else:
                    pad = 16 - c.out_channels
                    ys.append(hip.token_linear(xg, F.pad(w2, (0, 0, 0, pad)), F.pad(c.bias, (0, pad)))[:, :c.out_channels])
        torch.autograd.backward(ys, gys)
        return [y.detach() for y in ys], xg.grad, [c.weight.grad.clone() for c in convs], [c.bias.grad.clone() for c in convs]

    ya, dxa, dwa, dba = run(False)
    yb, dxb, dwb, dbb = run(True)
    for a, b in zip(ya, yb):
        assert b.is_contiguous() and torch.equal(a, b)
    assert (dxa.float() - dxb.float()).abs().max() <= 2.0 ** -6 * dxa.float().abs().max()
    for a, b in zip(dwa + dba, dwb + dbb):
        assert a.shape == b.shape and b.dtype == torch.float32
        assert (a - b).abs().max() <= 1e-5 * max(1.0, a.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(25088, 576, 512), (1001, 64, 256), (130, 40, 32), (1, 8, 64), (4097, 256, 256)])
def test_hand_written_products_of_the_training_backward(hip, dtype, shape):
    """gdkvm_gemm_nt (C = A B^T + bias) and gdkvm_gemm_tn (C = A^T B, split over the rows, deterministic) against fp64."""
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    bt = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    c = hip.gemm_nt(a, bt, bias)
    ref = a.double() @ bt.double().t() + bias.double()
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert c.dtype == dtype and ((c.double() - ref).abs() <= tol * (1.0 + ref.abs())).all()
    b2 = torch.randn(M, N, device="cuda", generator=g).to(dtype)
    a2 = a[:, :(K // 8) * 8]
    d = hip.wgrad(a2, b2)                                               # [K, N] fp32 = a^T b2 over the M rows
    ref2 = a2.double().t() @ b2.double()
    assert d.dtype == torch.float32 and ((d.double() - ref2).abs() <= 1e-5 * (M ** 0.5) * (1.0 + ref2.abs() / M ** 0.5)).all()
    assert torch.equal(d, hip.wgrad(a2, b2))                            # fixed summation order


@pytest.mark.parametrize("case", [(6, 2, 28, 28, 112, 112, torch.int64), (3, 4, 64, 64, 256, 256, torch.uint8), (2, 3, 7, 5, 30, 17, torch.int64),
                                  (1, 8, 4, 4, 4, 4, torch.int64), (2, 2, 9, 9, 5, 6, torch.uint8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_objective_matches_the_torch_loss(hip, case, dtype):
    """ops.seg_loss(lowres, target) == segmentation_loss(interpolate(lowres), target), value and gradient."""
    from gdkvm_amd.train import segmentation_loss
    ni, c, h, w, H, W, tdt = case
    torch.manual_seed(sum(case[:6]))
    z = (2.0 * torch.randn(ni, c, h, w, device="cuda")).to(dtype)
    tgt = torch.randint(0, c, (ni, H, W), device="cuda").to(tdt)
    za = z.clone().requires_grad_(True)
    loss = hip.seg_loss(za, tgt, 0.7, 1.0)
    (3.0 * loss).backward()
    zb = z.double().cpu().requires_grad_(True)
    up = F.interpolate(zb, size=(H, W), mode="bilinear", align_corners=False)
    ref = segmentation_loss(up.reshape(1, ni, c, H, W), tgt.long().cpu().reshape(1, ni, H, W), 0.7, 1.0)
    # segmentation_loss computes in fp32 internally (.float()): restate in fp64 for the check of the gradient
    lg = up
    ce = F.cross_entropy(lg, tgt.long().cpu())
    p = lg.softmax(1)
    oh = F.one_hot(tgt.long().cpu(), c).permute(0, 3, 1, 2).double()
    dice = 1.0 - ((2 * (p * oh).sum((0, 2, 3)) + 1.0) / (p.sum((0, 2, 3)) + oh.sum((0, 2, 3)) + 1.0)).mean()
    ref64 = ce + 0.7 * dice
    (3.0 * ref64).backward()
    assert abs(loss.item() - ref64.item()) <= 1e-5 * max(1.0, abs(ref64.item()))
    assert abs(ref.item() - ref64.item()) <= 1e-4
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-4
    assert (za.grad.double().cpu() - zb.grad).abs().max() <= tol * zb.grad.abs().max().item()


@pytest.mark.parametrize("case", [(6, 64, 64, 28, 28), (3, 128, 128, 14, 14), (5, 256, 256, 7, 7), (2, 384, 128, 14, 14), (2, 192, 64, 28, 28),
                                  (3, 64, 192, 9, 13), (1, 128, 64, 64, 64)])
def test_training_convolution_forward_and_data_gradient(hip, case):
    """ops.conv3x3 (hand-written forward and data gradient, framework weight gradient) against conv2d autograd in fp64 on the
    same bf16-rounded operands; the data gradient is the SAME kernel on the flipped, transposed weights."""
    n, c, k, h, w = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(k, c, 3, 3, device="cuda") / (9 * c) ** 0.5).requires_grad_(True)          # fp32 master weights
    dy = torch.randn(n, k, h, w, device="cuda").bfloat16()
    assert hip.conv3x3_train_served(x, wt, (1, 1), (1, 1), (1, 1), 1)
    y = hip.conv3x3(x, wt)
    y.backward(dy)
    x64 = x.detach().double().requires_grad_(True)
    w64 = wt.detach().bfloat16().double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(dy.double())
    tol = 2.0 ** -7
    assert y.dtype == torch.bfloat16 and (y.double() - y64).abs().max() <= tol * max(1.0, y64.abs().max().item())
    assert x.grad.dtype == torch.bfloat16 and (x.grad.double() - x64.grad).abs().max() <= tol * max(1.0, x64.grad.abs().max().item())
    assert wt.grad.dtype == torch.float32 and (wt.grad.double() - w64.grad).abs().max() <= 2 * tol * max(1.0, w64.grad.abs().max().item())
    assert not hip.conv3x3_train_served(x, wt[:, :, :1, :1], (1, 1), (0, 0), (1, 1), 1)
    assert not hip.conv3x3_train_served(x, wt, (2, 2), (1, 1), (1, 1), 1)


@pytest.mark.parametrize("case", [(6, 64, 64, 28, 28), (3, 128, 128, 14, 14), (9, 256, 256, 7, 7), (2, 384, 128, 14, 14), (2, 192, 64, 28, 28),
                                  (3, 64, 192, 9, 13), (1, 128, 64, 64, 64), (5, 64, 64, 3, 2)])
def test_convolution_weight_gradient(hip, case):
    """gdkvm_conv3x3_wgrad == the weight gradient of conv2d(x, w, padding=1) in fp64 on the same bf16 operands; deterministic."""
    n, c, k, h, w = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, k, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    dw = hip.conv3x3_wgrad(x, dy)
    w64 = torch.zeros(k, c, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w64, None, 1, 1).backward(dy.double())
    assert dw.dtype == torch.float32 and dw.shape == w64.shape and dw.is_contiguous()
    assert (dw.double() - w64.grad).abs().max() <= 1e-5 * (n * h * w) ** 0.5 * max(1.0, w64.grad.abs().max().item() / (n * h * w) ** 0.5)
    assert torch.equal(dw, hip.conv3x3_wgrad(x, dy))
    # the same numbers in the memory order of a channels_last parameter (no re-layout copy when they become its .grad)
    dwc = hip.conv3x3_wgrad(x, dy, channels_last=True)
    assert dwc.is_contiguous(memory_format=torch.channels_last) and torch.equal(dwc, dw)


def test_fused_objective_ignores_unlabelled_pixels(hip):
    """Labels outside [0, C) (255 in uint8 annotation masks) carry no class: the cross-entropy averages over the labelled pixels
    only, as F.cross_entropy(ignore_index=...) does, and NO Dice sum sees them (not the prediction mass either: a clip with two
    traced frames out of 32 must not be told to predict nothing on the other 30).  The host-side loss agrees."""
    from gdkvm_amd.train import segmentation_loss
    ni, c, h, w, H, W = 3, 2, 28, 28, 112, 112
    torch.manual_seed(5)
    z = 2.0 * torch.randn(ni, c, h, w, device="cuda")
    tgt = torch.randint(0, c, (ni, H, W), device="cuda")
    tgt[torch.rand(ni, H, W, device="cuda") < 0.3] = 255
    for tdt in (torch.uint8, torch.int64):
        za = z.clone().requires_grad_(True)
        loss = hip.seg_loss(za, tgt.to(tdt), 0.7, 1.0)
        loss.backward()
        zb = z.double().cpu().requires_grad_(True)
        up = F.interpolate(zb, size=(H, W), mode="bilinear", align_corners=False)
        t = tgt.cpu()
        ce = F.cross_entropy(up, t, ignore_index=255)
        p = up.softmax(1) * (t != 255).unsqueeze(1)
        oh = torch.stack([(t == k) for k in range(c)], 1).double()
        dice = 1.0 - ((2 * (p * oh).sum((0, 2, 3)) + 1.0) / (p.sum((0, 2, 3)) + oh.sum((0, 2, 3)) + 1.0)).mean()
        ref = ce + 0.7 * dice
        ref.backward()
        assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
        assert (za.grad.double().cpu() - zb.grad).abs().max() <= 1e-4 * zb.grad.abs().max().item()
        host = segmentation_loss(up.detach().float().reshape(1, ni, c, H, W), t.reshape(1, ni, H, W), 0.7, 1.0)
        assert abs(host.item() - ref.item()) <= 1e-4


def test_fused_objective_with_no_labelled_pixel(hip):
    """A batch whose every label is 255: the cross-entropy is a mean over no pixel -- 0 in the HIP loss and in the host loss
    (F.cross_entropy would return NaN and hand it to AdamW); the Dice term sees empty targets.  Finite loss, finite gradient."""
    from gdkvm_amd.train import segmentation_loss
    ni, c, h, w, H, W = 2, 3, 16, 16, 64, 64
    torch.manual_seed(6)
    z = torch.randn(ni, c, h, w, device="cuda")
    tgt = torch.full((ni, H, W), 255, dtype=torch.uint8, device="cuda")
    za = z.clone().requires_grad_(True)
    loss = hip.seg_loss(za, tgt, 1.0, 1.0)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(za.grad).all()
    up = F.interpolate(z.cpu(), size=(H, W), mode="bilinear", align_corners=False).requires_grad_(True)
    host = segmentation_loss(up.reshape(1, ni, c, H, W), tgt.cpu().reshape(1, ni, H, W), 1.0, 1.0)
    host.backward()
    assert torch.isfinite(host) and torch.isfinite(up.grad).all()
    assert abs(loss.item() - host.item()) <= 1e-5


def test_train_step_uses_the_fused_objective(hip):
    """train_step on the GPU (stride-4 logits + HIP loss) and the plain route (full-resolution logits + torch loss) give the
    same loss and the same parameter gradients (fp32, no autocast)."""
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from gdkvm_amd.train import segmentation_loss, segmentation_loss_lowres
    torch.manual_seed(11)
    cfg = GDKVMConfig(widths=(16, 32, 64), pixel_dim=64, value_dim=64)
    model = GDKVM(cfg).cuda().train().to(memory_format=torch.channels_last)
    frames = torch.rand(2, 3, 3, 64, 64, device="cuda")
    target = (torch.rand(2, 3, 64, 64, device="cuda") > 0.5).long()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    la = segmentation_loss_lowres(model(frames, _lowres=True), target)
    la.backward()
    ga = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    model.load_state_dict(state)                       # same BatchNorm running statistics as before the first pass
    lb = segmentation_loss(model(frames), target)
    lb.backward()
    assert abs(la.item() - lb.item()) <= 1e-5 * max(1.0, abs(lb.item()))
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        scale = max(p.grad.abs().max().item(), 1e-6)
        assert (ga[n] - p.grad).abs().max().item() <= 2e-3 * scale, n


def _ellipse_batches(n, B, T, S, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    frames, target = [], []
    for i in range(n):
        cy, cx = S * (0.45 + 0.02 * i), S * (0.5 - 0.01 * i)
        m = ((((yy - cy) / (S * 0.3)) ** 2 + ((xx - cx) / (S * 0.2)) ** 2) < 1)
        tg = m.long().expand(B, T, S, S).contiguous()
        fr = torch.rand(B, T, 3, S, S, generator=g) * 0.5 + 0.5 * m.float()            # the label is visible in the frames: a learnable batch
        frames.append(fr.cuda()); target.append(tg.cuda())
    return frames, target


def test_training_step_captured_in_a_graph(hip):
    """GraphedTrainStep (warm-up steps, one capture, then replays) against the eager train_step from the same start, on a batch the model can
    learn (the label's ellipse is visible in the frames) at lr 1e-3: the loss of EVERY step agrees, the loss falls by a real margin, the
    weights end where the eager run's end -- and all of it also against the eager step under the DEFAULT (non-fused) AdamW, whose step bumps
    the parameters' version counters where the fused one does not (round 5: weight packs keyed on the counter stayed at the initial weights
    under the fused optimiser and the loss fell ~10x slower; with random labels and a bound wider than seven steps' movement the old form of
    this test could not see it).  Fresh batches go through the graph's input buffers."""
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from gdkvm_amd.train import GraphedTrainStep, train_step
    cfg = GDKVMConfig()
    steps, lr = 8, 1e-3
    frames, target = _ellipse_batches(steps + 1, 4, 4, 112, 21)

    def run(kind):
        torch.manual_seed(22)
        model = GDKVM(cfg).cuda().train().to(memory_format=torch.channels_last)
        w0 = {n: p.detach().clone() for n, p in model.named_parameters()}
        kw = {} if kind == "default" else {"fused": True, "capturable": True}
        opt = torch.optim.AdamW(model.parameters(), lr=lr, **kw)
        losses = []
        if kind == "graph":
            # the warm-up steps and the captured one all see batch 0; replays then take batches 1 ..
            step = GraphedTrainStep(model, opt, frames[0], target[0], torch.bfloat16, warmup=2)
            for i in range(1, steps + 1):
                losses.append(step(frames[i], target[i]))
            assert len({l.data_ptr() for l in losses}) == len(losses)        # (each call hands back its own loss tensor, not the graph's buffer)
            losses = [l.item() for l in losses]
        else:
            for _ in range(2):
                train_step(model, opt, frames[0], target[0], torch.bfloat16)
            for i in range(1, steps + 1):
                losses.append(train_step(model, opt, frames[i], target[i], torch.bfloat16).item())
        return losses, w0, {n: p.detach().clone() for n, p in model.named_parameters()}

    ld, w0, wd = run("default")
    lf, _, wf = run("fused")
    lg, _, wg = run("graph")
    print("default", ld, "fused", lf, "graph", lg)
    assert ld[-1] < ld[0] - 0.15, ld                               # it learns: a real margin, not noise
    # The step is deterministic since round 5 (no library convolution, no atomics): the graph replays the fused-eager step's kernels in
    # the same order on the same data, so losses and weights are EQUAL, bit for bit -- a stale weight pack or a re-ordered launch that moved
    # results by a few percent would pass any looser bound.
    assert lf == lg, (lf, lg)
    for n in wf:
        assert torch.equal(wf[n], wg[n]), n
    # Against the DEFAULT (non-fused) AdamW the arithmetic of the update differs (foreach vs fused kernels): every step's loss agrees to a
    # small fraction of the whole descent (packs stuck at the initial weights left the loss ~0.15 behind by the last step)
    tol = 0.08 * max(ld[0] - ld[-1], 0.15) + 3e-3
    for a, b, c in zip(ld, lf, lg):
        assert abs(a - b) <= tol and abs(a - c) <= tol, (ld, lf, lg)
    for n in wd:
        moved = (wd[n] - w0[n]).abs().max().item()
        if moved < 2e-3:
            continue                                               # (mask_embed: no gradient to speak of)
        # ten AdamW steps of lr 1e-3 took this weight from w0 to wd; the other two runs end near it, measured against that journey in the L2
        # norm (single elements are chaotic under Adam: a tiny gradient that changes sign moves its weight by a full step the other way)
        journey = (wd[n] - w0[n]).norm().item()
        assert (wd[n] - wf[n]).norm().item() <= 0.6 * journey, n
        assert (wd[n] - wg[n]).norm().item() <= 0.6 * journey, n
    with pytest.raises(RuntimeError):
        GraphedTrainStep(GDKVM(cfg).train(), None, torch.rand(1, 2, 3, 112, 112), torch.zeros(1, 2, 112, 112, dtype=torch.long))


def test_graphs_survive_or_refuse_a_weight_change(hip):
    """What a captured graph reads by address stays alive and current: a GraphedSegment replayed after the weights changed raises (it would
    segment with the packs of capture time); segment_clip(graph=True) captures anew; a GraphedTrainStep replayed after an eval() / train()
    toggle (which drops the module's pack caches) still trains like the eager step, because it holds the packs it re-packs into."""
    from gdkvm_amd.model import GDKVM, GDKVMConfig, weights_changed
    from gdkvm_amd.train import GraphedTrainStep, train_step
    torch.manual_seed(5)
    model = GDKVM(GDKVMConfig()).cuda().eval().to(torch.bfloat16).to(memory_format=torch.channels_last).fuse_for_inference()
    fr = torch.rand(1, 4, 3, 112, 112, device="cuda").bfloat16()
    gseg = model.graphed_segment(fr)
    m0 = gseg(fr)[0].clone()
    assert torch.equal(m0, model.segment(fr)[0])
    mc0 = model.segment_clip(fr, 2, graph=True)[0]
    with torch.no_grad():
        model.decoder.head.bias.mul_(-1.0)                         # (a plain in-place write: the version counter shows it)
        model.decoder.head.weight.data.mul_(-1.0)                  # (a write through .data: nothing shows it -- say so); logits -> -logits
    weights_changed()
    with pytest.raises(RuntimeError, match="changed since the capture"):
        gseg(fr)
    m1 = model.segment(fr)[0]
    mc1 = model.segment_clip(fr, 2, graph=True)[0]                 # captured anew under the new weights
    assert torch.equal(mc1, m1) and torch.equal(mc0, m0) and not torch.equal(m0, m1)
    model.eval()                                                   # invalidate_packed_weights(): the old graph still refuses
    with pytest.raises(RuntimeError, match="changed since the capture"):
        gseg(fr)

    frames, target = _ellipse_batches(5, 2, 4, 112, 9)
    def run(graphed):
        torch.manual_seed(23)
        tm = GDKVM(GDKVMConfig()).cuda().train().to(memory_format=torch.channels_last)
        opt = torch.optim.AdamW(tm.parameters(), lr=1e-3, fused=True, capturable=True)
        step = GraphedTrainStep(tm, opt, frames[0], target[0], torch.bfloat16, warmup=2) if graphed else None
        if not graphed:
            for _ in range(2):
                train_step(tm, opt, frames[0], target[0], torch.bfloat16)
        out = []
        for i in range(1, 5):
            if i == 3:
                tm.eval(); tm.train()                              # a validation pass between steps: every pack cache of the module is dropped
                junk = [torch.empty(1 << 22, device="cuda").normal_() for _ in range(8)]       # (and the allocator hands its blocks out again)
                del junk
            out.append((step(frames[i], target[i]) if graphed else train_step(tm, opt, frames[i], target[i], torch.bfloat16)).item())
        return out
    le, lg = run(False), run(True)
    print(le, lg)
    assert all(abs(a - b) <= 0.08 * max(abs(le[0] - le[-1]), 0.15) + 3e-3 for a, b in zip(le, lg)), (le, lg)


def test_training_weight_packs_in_one_launch(hip):
    """ops.conv3x3_train_packs: the forward and the data-gradient pack of several layers from their fp32 master weights in ONE launch
    (gdkvm_conv3x3_pack_weights_train) are, bit for bit, what the per-layer sequence produces -- cast to bf16, gdkvm_conv3x3_pack_weights,
    gdkvm_conv3x3_pack_weights_dgrad -- for contiguous and channels_last weights; ops.conv3x3 with the packs in place gives the same output
    and the same gradients as without them; EVERY call re-packs (a write that bumps no version counter -- a fused optimiser's step, a write
    through .data -- is picked up), and the packs are used only inside the scope of the call that made them."""
    from gdkvm_amd import ops
    lib = ops.load()
    torch.manual_seed(3)
    shapes = [(64, 64), (128, 64), (64, 192), (256, 256), (128, 384)]
    ws = [torch.randn(k, c, 3, 3, device="cuda") / (c * 9) ** 0.5 for k, c in shapes]
    ws[1] = ws[1].contiguous(memory_format=torch.channels_last)
    ws[3] = ws[3].contiguous(memory_format=torch.channels_last)
    ws = [torch.nn.Parameter(w) for w in ws]

    def reference_packs(w):
        k, c = w.shape[:2]
        wb = w.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        f0 = torch.empty(k * 9 * c, dtype=torch.bfloat16, device="cuda"); d0 = torch.empty_like(f0)
        assert lib.gdkvm_conv3x3_pack_weights(wb.data_ptr(), f0.data_ptr(), k, c, ops.BF16, None) == 0
        assert lib.gdkvm_conv3x3_pack_weights_dgrad(wb.data_ptr(), d0.data_ptr(), k, c, ops.BF16, None) == 0
        return f0, d0

    assert ops.conv3x3_train_packs(ws) == len(ws) and ops.conv3x3_train_packs(ws) == len(ws)
    for w in ws:
        f0, d0 = reference_packs(w)
        f1, d1 = ops._train_packs_of(w)
        torch.cuda.synchronize()
        assert torch.equal(f0.view(torch.int16), f1.view(torch.int16)) and torch.equal(d0.view(torch.int16), d1.view(torch.int16)), tuple(w.shape)
    ops.end_train_packs()
    assert all(ops._train_packs_of(w) is None for w in ws)         # outside the scope nothing is trusted
    # the differentiable op with and without the packs
    w = ws[1]
    x = torch.randn(4, 64, 14, 14, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    gy = torch.randn(4, 128, 14, 14, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    with ops.train_packs(ws):
        assert ops._train_packs_of(w) is not None
        y1 = ops.conv3x3(x, w); g1 = torch.autograd.grad(y1, (x, w), gy)
    y0 = ops.conv3x3(x, w); g0 = torch.autograd.grad(y0, (x, w), gy)
    assert torch.equal(y0, y1) and torch.equal(g0[0], g1[0]) and torch.equal(g0[1], g1[1])
    # a write no version counter records (what torch.optim.AdamW(fused=True) does to every parameter): the next call packs the new values
    v0 = ws[0]._version
    ws[0].data.mul_(0.5)
    assert ws[0]._version == v0
    with ops.train_packs(ws):
        f0, d0 = reference_packs(ws[0])
        f1, d1 = ops._train_packs_of(ws[0])
        torch.cuda.synchronize()
        assert torch.equal(f0.view(torch.int16), f1.view(torch.int16)) and torch.equal(d0.view(torch.int16), d1.view(torch.int16))
    ops.drop_train_packs(ws)
    assert all("_gdkvm_train_packs" not in w.__dict__ for w in ws)


@pytest.mark.parametrize("case", [(3, 3, 112, 112), (2, 1, 64, 48), (1, 4, 130, 118), (9, 3, 20, 256)])
def test_training_stem_convolution_on_the_stem_kernel(hip, case):
    """ops.stem_conv (gdkvm_stem_pack_s2d + gdkvm_stem_conv_nchw: the inference stem kernel in its convolution-only form) == conv2d(x, w,
    stride 2, padding 3) in fp64 on the bf16-rounded operands, rounded once to bf16; the weight gradient (the framework's) matches fp64
    autograd; ragged sizes (tiles of 9 x 57 outputs), 1 .. 4 input channels."""
    from gdkvm_amd import ops
    n, c, hh, ww = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, hh, ww, device="cuda").bfloat16()
    w = torch.nn.Parameter(torch.randn(64, c, 7, 7, device="cuda") / (49 * c) ** 0.5)
    y = ops.stem_conv(x, w)
    w64 = w.detach().bfloat16().double().requires_grad_()
    ref = F.conv2d(x.double(), w64, None, 2, 3)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    assert (y.double() - ref).abs().max() <= 2.0 ** -8 * max(1.0, ref.abs().max().item())
    gy = torch.randn_like(y)
    (dw,) = torch.autograd.grad(y, w, gy)
    (dref,) = torch.autograd.grad(ref, w64, gy.double())
    # the weight gradient is gdkvm_stem_wgrad_nchw: fp32 accumulation over the exact bf16 products, fixed order -> a few fp32 roundings of
    # the sum's magnitude away from fp64, and the same bits on every run
    assert dw.dtype == torch.float32 and dw.shape == w.shape
    scale = max(1.0, dref.abs().max().item())
    assert (dw.double() - dref).abs().max() <= 2e-5 * scale * max(1.0, (n * hh * ww / 4) ** 0.5 / 64)
    (dw2,) = torch.autograd.grad(ops.stem_conv(x, w), w, gy)
    assert torch.equal(dw, dw2)
    wcl = torch.nn.Parameter(w.detach().clone().contiguous(memory_format=torch.channels_last))      # the gradient takes the parameter's memory format
    (dw3,) = torch.autograd.grad(ops.stem_conv(x, wcl), wcl, gy)
    assert dw3.stride() == wcl.stride() and torch.equal(dw3, dw)


@pytest.mark.parametrize("case", [(3, 64, 128, 28, 28), (2, 128, 256, 14, 14), (2, 64, 128, 15, 13), (1, 128, 128, 7, 9), (5, 64, 256, 2, 2)])
def test_strided_block_convolutions_forward_and_backward(hip, case):
    """ops.conv_s2_block (csrc/conv_s2_train.hip): a residual block's 3x3 / stride-2 / pad-1 convolution and its 1x1 / stride-2 branch -- both
    outputs, the gradient of the input (both branches in one accumulation, four parity classes) and both weight gradients against fp64 autograd
    on the bf16-rounded operands; even and odd sizes; contiguous and channels_last parameters; the same bits on a second run."""
    from gdkvm_amd import ops
    n, c, k, hh, ww = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, hh, ww, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    w = torch.nn.Parameter((torch.randn(k, c, 3, 3, device="cuda") / (9 * c) ** 0.5).contiguous(memory_format=torch.channels_last))
    wd = torch.nn.Parameter(torch.randn(k, c, 1, 1, device="cuda") / c ** 0.5)
    y, yd = ops.conv_s2_block(x, w, wd)
    x64 = x.detach().double().requires_grad_()
    w64, wd64 = w.detach().bfloat16().double().requires_grad_(), wd.detach().bfloat16().double().requires_grad_()
    ry, ryd = F.conv2d(x64, w64, None, 2, 1), F.conv2d(x64, wd64, None, 2, 0)
    assert y.shape == ry.shape and yd.shape == ryd.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    for a, b in ((y, ry), (yd, ryd)):
        assert (a.double() - b).abs().max() <= 2.0 ** -8 * max(1.0, b.abs().max().item())
    gy, gyd = torch.randn_like(y), torch.randn_like(yd)
    dx, dw, dwd = torch.autograd.grad((y, yd), (x, w, wd), (gy, gyd))
    rdx, rdw, rdwd = torch.autograd.grad((ry, ryd), (x64, w64, wd64), (gy.double(), gyd.double()))
    assert dx.dtype == torch.bfloat16 and dx.shape == x.shape
    assert (dx.double() - rdx).abs().max() <= 2.0 ** -8 * max(1.0, rdx.abs().max().item())
    rows = n * y.shape[2] * y.shape[3]
    for a, b, p in ((dw, rdw, w), (dwd, rdwd, wd)):
        assert a.dtype == torch.float32 and a.shape == p.shape and a.stride() == p.stride()       # the parameter's own memory format
        assert (a.double() - b).abs().max() <= 2e-5 * max(1.0, b.abs().max().item()) * max(1.0, rows ** 0.5 / 64)
    y2, yd2 = ops.conv_s2_block(x, w, wd)
    g2 = torch.autograd.grad((y2, yd2), (x, w, wd), (gy, gyd))
    assert torch.equal(y, y2) and torch.equal(yd, yd2) and all(torch.equal(a, b) for a, b in zip((dx, dw, dwd), g2))
    # a contiguous (KCRS) parameter: the same numbers in its layout
    wc = torch.nn.Parameter(w.detach().contiguous())
    y3, yd3 = ops.conv_s2_block(x, wc, wd)
    g3 = torch.autograd.grad((y3, yd3), (x, wc, wd), (gy, gyd))
    assert torch.equal(y, y3) and torch.equal(g3[0], dx) and g3[1].stride() == wc.stride() and torch.equal(g3[1], dw)
    # C ABI: the class offsets inside the data-gradient pack follow how the pack was BUILT (with_down), not whether dy_down is given:
    # dy_down = NULL on a pack with the branch is a zero branch gradient (== passing zeros, bit for bit); dy_down without a branch pack is refused
    lib = ops.load()
    _, _, dg = ops.conv_s2_packs(w.detach(), wd.detach())
    gyb = gy.bfloat16().contiguous(memory_format=torch.channels_last)
    zeros = torch.zeros_like(gyb)
    dx_null, dx_zero = torch.empty_like(x), torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.gdkvm_conv_s2_dgrad(gyb.data_ptr(), None, dg.data_ptr(), dx_null.data_ptr(), n, c, hh, ww, k, 1, ops.BF16, st) == 0
    assert lib.gdkvm_conv_s2_dgrad(gyb.data_ptr(), zeros.data_ptr(), dg.data_ptr(), dx_zero.data_ptr(), n, c, hh, ww, k, 1, ops.BF16, st) == 0
    assert torch.equal(dx_null, dx_zero)
    rdx_main = torch.autograd.grad(F.conv2d(x64, w64, None, 2, 1), x64, gyb.double())[0]
    assert (dx_null.double() - rdx_main).abs().max() <= 2.0 ** -8 * max(1.0, rdx_main.abs().max().item())
    assert lib.gdkvm_conv_s2_dgrad(gyb.data_ptr(), zeros.data_ptr(), dg.data_ptr(), dx_zero.data_ptr(), n, c, hh, ww, k, 0, ops.BF16, st) == -6   # GDKVM_ERR_ARG


@pytest.mark.parametrize("case", [(2, 64, 64, 12, 10, 3, 1, 1), (3, 128, 40, 9, 9, 1, 2, 0), (1, 64, 72, 16, 16, 5, 2, 2), (2, 64, 64, 8, 8, 3, 3, 0)])
def test_strided_weight_gradient_any_window(hip, case):
    """ops.conv_wgrad_strided (gdkvm_conv_wgrad_strided) for windows / strides / paddings beyond the block's: against fp64 autograd."""
    from gdkvm_amd import ops
    n, c, k, hh, ww, r, stride, pad = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, hh, ww, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w64 = torch.randn(k, c, r, r, device="cuda", dtype=torch.float64, requires_grad=True)
    ry = F.conv2d(x.double(), w64, None, stride, pad)
    gy = torch.randn(ry.shape, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    (rdw,) = torch.autograd.grad(ry, w64, gy.double())
    for like in (torch.empty(k, c, r, r, device="cuda"), torch.empty(k, c, r, r, device="cuda").contiguous(memory_format=torch.channels_last)):
        dw = ops.conv_wgrad_strided(x, gy, like, stride, pad)
        assert dw.stride() == like.stride()
        assert (dw.double() - rdw).abs().max() <= 2e-5 * max(1.0, rdw.abs().max().item())


def test_training_step_is_deterministic(hip):
    """With the strided layers on the hand-written kernels no gradient of a training step accumulates atomically any more: two runs of the
    same steps from the same start give the same losses and the same weights, bit for bit (what makes a DDP-wrapped step at world size 1
    equal to the bare one: tests/test_zz_nccl_gpu.py)."""
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from gdkvm_amd.train import train_step
    frames, target = _ellipse_batches(3, 2, 4, 112, 33)

    def run():
        torch.manual_seed(34)
        model = GDKVM(GDKVMConfig()).cuda().train().to(memory_format=torch.channels_last)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        losses = [train_step(model, opt, frames[i], target[i], torch.bfloat16).item() for i in range(3)]
        grads = {n_: p.grad.detach().clone() for n_, p in model.named_parameters()}
        return losses, grads, {n_: p.detach().clone() for n_, p in model.named_parameters()}

    l1, g1, w1 = run()
    l2, g2, w2 = run()
    assert l1 == l2, (l1, l2)
    diff = [n_ for n_ in g1 if not torch.equal(g1[n_], g2[n_])]
    assert not diff, diff
    assert all(torch.equal(w1[n_], w2[n_]) for n_ in w1)
