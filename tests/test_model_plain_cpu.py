"""The CPU reference module (oracle/model_ref.GDKVMRef: the product's nn.Module with the memory path on the oracle) against an
INDEPENDENT restatement of the architecture (oracle/model_plain.plain_forward: one function of the state_dict, float64,
torch.nn.functional + the numpy oracle, no code shared with gdkvm_amd).  A wiring error in the product's module -- the wrong skip
tensor into a decoder stage, a residual taken before instead of after the downsample, projections mixed up -- would be shared by
GDKVM and GDKVMRef and invisible to every comparison between them; it is not shared by this one."""
import pytest
import torch


def _ref(cfg, seed):
    from oracle.model_ref import GDKVMRef
    torch.manual_seed(seed)
    ref = GDKVMRef(cfg).eval()
    ref.math = "f64"
    for m in ref.modules():                      # non-trivial BatchNorm statistics and affine parameters
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.8, 1.25)
            m.weight.data.uniform_(0.8, 1.2); m.bias.data.normal_(0, 0.1)
    return ref


@pytest.mark.parametrize("case", [dict(widths=(16, 32, 64), pixel_dim=64, value_dim=32, key_dim=64, heads=1, num_classes=2, rule="delta_sequential"),
                                  dict(widths=(16, 16, 32), pixel_dim=32, value_dim=16, key_dim=64, heads=2, num_classes=4, rule="gated_linear"),
                                  dict(widths=(8, 16, 32), pixel_dim=32, value_dim=16, key_dim=64, heads=1, num_classes=3, rule="delta_parallel")])
def test_reference_module_equals_the_plain_restatement(case):
    from gdkvm_amd.model import GDKVMConfig
    from oracle.model_plain import plain_forward
    cfg = GDKVMConfig(**case)
    ref = _ref(cfg, seed=len(case["rule"]))
    g = torch.Generator().manual_seed(5)
    frames = torch.rand(2, 3, 3, 64, 80, generator=g)
    mask0 = (torch.rand(2, 1, 64, 80, generator=g) > 0.5).float()
    s0 = 0.2 * torch.randn(2, cfg.heads, cfg.key_dim, cfg.value_dim, generator=g)
    kw = dict(heads=cfg.heads, key_dim=cfg.key_dim, value_dim=cfg.value_dim, rule=cfg.rule)
    for m0, st in ((None, None), (mask0, s0)):
        with torch.no_grad():
            lr, sr = ref(frames, mask0=m0, state=st, return_state=True)
        lp, sp = plain_forward(ref.state_dict(), frames, mask0=m0, state=st, **kw)
        assert lp.shape == lr.shape and sp.shape == sr.shape
        scale = max(1.0, lp.abs().max().item())
        assert (lp - lr.double()).abs().max().item() <= 2e-4 * scale          # (the reference's convolutions are float32)
        assert (sp - sr.double()).abs().max().item() <= 2e-4 * max(1.0, sp.abs().max().item())
    with torch.no_grad():
        low = ref(frames, _lowres=True)
    lowp, _ = plain_forward(ref.state_dict(), frames, lowres=True, **kw)
    assert lowp.shape == low.shape and (lowp - low.double()).abs().max().item() <= 2e-4 * max(1.0, lowp.abs().max().item())


def test_the_plain_restatement_notices_a_rewired_skip():
    """The check has teeth: the stride-4 feature reaches the output only through the last decoder stage's skip input, and moving it
    (what a mis-wired skip would amount to) changes the restatement's result."""
    from gdkvm_amd.model import GDKVMConfig
    from oracle.model_plain import plain_forward
    cfg = GDKVMConfig(widths=(16, 16, 32), pixel_dim=32, value_dim=16)
    ref = _ref(cfg, seed=1)
    frames = torch.rand(1, 2, 3, 64, 64)
    sd = ref.state_dict()
    base, _ = plain_forward(sd, frames, value_dim=16)
    bad = dict(sd)
    bad["encoder.layer1.1.bn2.bias"] = sd["encoder.layer1.1.bn2.bias"] + 0.5         # f4 reaches the output only through decoder.up4's skip
    other, _ = plain_forward(bad, frames, value_dim=16)
    assert (other - base).abs().max().item() > 1e-3


@pytest.mark.parametrize("case", [dict(widths=(16, 32, 64), pixel_dim=64, value_dim=32, rule="delta_sequential", num_classes=2),
                                  dict(widths=(16, 16, 32), pixel_dim=32, value_dim=16, heads=2, rule="gated_linear", num_classes=3)])
def test_reference_gradients_equal_the_plain_restatement(case):
    """The gradient oracle is independent too: one training step's loss and gradients from the reference module (the product's wiring,
    torch autograd, train-mode BatchNorm, gdkvm_amd.train.segmentation_loss) against oracle.model_plain.plain_loss_and_grads (the
    architecture, the objective and its ignore-label rule written a second time in float64, sharing nothing with gdkvm_amd).  The GPU
    tests compare the product's HIP backward with the reference module; this ties that module to an independent derivation -- with
    unlabelled pixels (label 255) in the batch, which both sides must leave out of every sum."""
    from gdkvm_amd.model import GDKVMConfig
    from gdkvm_amd.train import segmentation_loss
    from oracle.model_plain import plain_loss_and_grads
    cfg = GDKVMConfig(**case)
    ref = _ref(cfg, seed=7).train()
    g = torch.Generator().manual_seed(11)
    frames = torch.rand(2, 2, 3, 64, 64, generator=g)
    target = torch.randint(0, cfg.num_classes, (2, 2, 64, 64), generator=g)
    target[:, 1, :20] = 255                                                    # unlabelled rows of the second frame
    loss = segmentation_loss(ref(frames), target)
    loss.backward()
    lp, gp = plain_loss_and_grads(ref.state_dict(), frames, target, heads=cfg.heads, key_dim=cfg.key_dim, value_dim=cfg.value_dim, rule=cfg.rule)
    assert abs(loss.item() - lp.item()) <= 2e-5 * max(1.0, abs(lp.item()))
    seen = 0
    for n, p in ref.named_parameters():
        if p.grad is None or n not in gp:
            assert (p.grad is None or p.grad.abs().max() == 0) and (n not in gp or gp[n].abs().max() == 0), n
            continue
        scale = max(gp[n].abs().max().item(), 1e-7)
        assert (p.grad.double() - gp[n]).abs().max().item() <= 2e-3 * scale, (n, (p.grad.double() - gp[n]).abs().max().item(), scale)
        seen += 1
    assert seen >= 60                                                          # every layer of the model took part
