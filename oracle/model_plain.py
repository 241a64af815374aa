"""plain_forward -- the whole module restated as ONE function of a state_dict, sharing no code with gdkvm_amd.

TEST INFRASTRUCTURE ONLY (tests/ and smoke()).  oracle/model_ref.GDKVMRef subclasses the product's nn.Module and swaps the memory
path for the oracle: handy (same parameters, same autograd), but the layer wiring -- which block feeds which, which skip tensor a
decoder stage concatenates, where the residual enters -- is then the product's own code on both sides of every comparison, and a
wiring error is invisible.  This file writes the architecture of SURVEY.md Appendix A / DESIGN.md §1 down a second time, from the
parameter names alone, as torch.nn.functional calls in float64 on the CPU (eval-mode BatchNorm from the running statistics), with
the memory path on the numpy oracle (oracle/gdkvm_oracle.py, float64).  tests/test_model_plain_cpu.py pins GDKVMRef to it; the GPU
parity tests compare the product's fused inference build with it directly.

    frames [B,T,C,H,W] -> encoder (ResNet-18-style trunk to stride 16: f4, f8, f16)
      f16 -> key / query / value / gate 1x1 projections per token, decay gate from the token mean
      LKVA read + GDR write over the T frames (oracle.scan), KPFF (oracle.kpff) on (key feature, read-out, f16)
      decoder: [upsample x2 ; f8] -> two 3x3 convs -> [upsample x2 ; f4] -> two 3x3 convs -> 1x1 head -> bilinear to HxW
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from oracle import gdkvm_oracle as O

_RULE_IDS = {"gated_linear": 0, "delta_parallel": 1, "delta_sequential": 2}


def _t(sd, name):
    return sd[name].detach().to("cpu", torch.float64)


def _bn(sd, prefix, x, eps=1e-5):
    return F.batch_norm(x, _t(sd, prefix + ".running_mean"), _t(sd, prefix + ".running_var"), _t(sd, prefix + ".weight"),
                        _t(sd, prefix + ".bias"), False, 0.0, eps)


def _block(sd, p, x, stride):
    """conv3x3(stride) - bn - relu - conv3x3 - bn, plus the input (through a strided 1x1 conv + bn when the shape changes), relu."""
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, _t(sd, p + ".conv1.weight"), None, stride, 1)))
    y = _bn(sd, p + ".bn2", F.conv2d(y, _t(sd, p + ".conv2.weight"), None, 1, 1))
    if (p + ".down.0.weight") in sd:
        x = _bn(sd, p + ".down.1", F.conv2d(x, _t(sd, p + ".down.0.weight"), None, stride, 0))
    return F.relu(y + x)


def _up(sd, p, x, skip):
    x = torch.cat([F.interpolate(x, size=skip.shape[-2:], mode="bilinear", align_corners=False), skip], 1)
    x = F.relu(_bn(sd, p + ".conv.1", F.conv2d(x, _t(sd, p + ".conv.0.weight"), None, 1, 1)))
    return F.relu(_bn(sd, p + ".conv.4", F.conv2d(x, _t(sd, p + ".conv.3.weight"), None, 1, 1)))


def _tokens(x):
    return x.permute(0, 2, 3, 1).reshape(x.shape[0], x.shape[2] * x.shape[3], x.shape[1])


@torch.no_grad()
def plain_forward(sd, frames, heads=1, key_dim=64, value_dim=256, rule="delta_sequential", mask0=None, state=None, lowres=False,
                  taps=None, override=None, normalizer=False, normalizer_eps=1e-6, mask_feedback=False):
    """sd: state_dict of the un-fused module (gdkvm_amd.model.GDKVM(...).state_dict()); frames [B,T,C,H,W].
    Returns (logits [B,T,ncls,H,W] float64 -- stride-4 logits if lowres --, final state [B,Hh,Dk,Dv] float64).
    taps (a dict): receives the intermediate stages -- f4, f8, f16 [BT,C,h,w], k, q, v, beta, alpha (token-major), r (read-out
    [BT,N,Hh*Dv]), fused (KPFF output [BT,N,Cp]), dec (stride-4 decoder feature) -- as float64 tensors.  override (a dict with any
    of f4 / f8 / f16 / r / fused): that stage is taken from the caller (e.g. from the product's bf16 build) instead of being
    computed, everything after it runs here in float64: tools/stage_error.py attributes the build's mask flips this way.
    normalizer (SURVEY A.1 flag): z carried beside S, read-out / (|q . z| + eps); `state` and the returned state are then [B,Hh,Dk,Dv+1]
    (S with z as one more column).  mask_feedback: the per-frame step mode -- frame t's value carries the embedding of
    the mask predicted for frame t (mask0 for frame 0 when given): read, KPFF, decoder, argmax, write, frame by frame; taps["mask"]
    then holds the predicted masks [B,T,H,W] (uint8) and taps["margin"] the gap between the two largest full-resolution logits."""
    taps = {} if taps is None else taps
    override = override or {}
    ov = lambda name, val: override[name].detach().to("cpu", torch.float64) if name in override else val
    B, T, C, H, W = frames.shape
    Hh, Dk, Dv = heads, key_dim, value_dim
    x = frames.detach().to("cpu", torch.float64).reshape(B * T, C, H, W)
    # encoder
    x = F.relu(_bn(sd, "encoder.stem.1", F.conv2d(x, _t(sd, "encoder.stem.0.weight"), None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    f4 = _block(sd, "encoder.layer1.1", _block(sd, "encoder.layer1.0", x, 1), 1)
    f8 = _block(sd, "encoder.layer2.1", _block(sd, "encoder.layer2.0", f4, 2), 1)
    f16 = _block(sd, "encoder.layer3.1", _block(sd, "encoder.layer3.0", f8, 2), 1)
    f4, f8, f16 = ov("f4", f4), ov("f8", f8), ov("f16", f16)
    taps.update(f4=f4, f8=f8, f16=f16)
    h, w = f16.shape[-2:]
    N = h * w
    # per-token projections of the stride-16 feature
    p_tok = _tokens(f16)                                                   # [BT, N, Cp]
    lin = lambda name: p_tok @ _t(sd, name + ".weight").reshape(sd[name + ".weight"].shape[0], -1).T + _t(sd, name + ".bias")
    k_tok = lin("key_proj")                                                # [BT, N, Hh*Dk]: also KPFF's local key feature
    q = lin("query_proj")
    v = lin("value_proj").reshape(B, T, N, Hh * Dv)
    if mask0 is not None:
        m = F.adaptive_avg_pool2d(mask0.detach().to("cpu", torch.float64), (h, w))
        me = _tokens(F.conv2d(m, _t(sd, "mask_embed.weight")))             # [B, N, Hh*Dv]: joins the first frame's values
        v = torch.cat([v[:, :1] + me.unsqueeze(1), v[:, 1:]], 1)
    beta = lin("gate_proj").reshape(B, T, N, Hh)                           # logits, per token and head
    alpha = (p_tok.mean(1) @ _t(sd, "decay_proj.weight").T + _t(sd, "decay_proj.bias")).reshape(B, T, Hh)   # per frame and head
    # memory: LKVA read + GDR write (flags 3: L2-normalised q / k, gates given as logits), then KPFF
    s0 = None if state is None else state.detach().cpu().double().numpy()
    if mask_feedback:
        return _plain_feedback(sd, (B, T, H, W, h, w, Hh, Dk, Dv), f4, f8, p_tok, k_tok, q, v, alpha, beta, s0, _RULE_IDS[rule], mask0 is not None,
                               lowres, taps)
    if normalizer:
        r, s, z = O.scan_normalizer(q.reshape(B, T, N, Hh, Dk).numpy(), k_tok.reshape(B, T, N, Hh, Dk).numpy(), v.reshape(B, T, N, Hh, Dv).numpy(),
                                    alpha.numpy(), beta.numpy(), None if s0 is None else s0[..., :Dv], None if s0 is None else s0[..., Dv],
                                    _RULE_IDS[rule], 3, normalizer_eps)
        s = np.concatenate([s, z[..., None]], -1)
    else:
        r, s = O.scan(q.reshape(B, T, N, Hh, Dk).numpy(), k_tok.reshape(B, T, N, Hh, Dk).numpy(), v.reshape(B, T, N, Hh, Dv).numpy(),
                      alpha.numpy(), beta.numpy(), s0, _RULE_IDS[rule], 3)
    r = ov("r", torch.from_numpy(np.asarray(r, np.float64)).reshape(B * T, N, Hh * Dv)).reshape(B * T, N, Hh * Dv)
    fused = O.kpff(k_tok.numpy(), r.numpy(), p_tok.numpy(), *(_t(sd, "kpff." + n).numpy() for n in ("wa", "ba", "wl", "wg")), h, w)
    fused = ov("fused", torch.from_numpy(np.asarray(fused, np.float64)).reshape(B * T, N, -1)).reshape(B * T, N, -1)
    taps.update(k=k_tok, q=q, v=v.reshape(B * T, N, Hh * Dv), beta=beta, alpha=alpha, r=r, fused=fused)
    fmap = fused.reshape(B * T, h, w, -1).permute(0, 3, 1, 2)
    # decoder
    y = _up(sd, "decoder.up4", _up(sd, "decoder.up8", fmap, f8), f4)
    taps["dec"] = y
    logits = F.conv2d(y, _t(sd, "decoder.head.weight"), _t(sd, "decoder.head.bias"))
    if not lowres:
        logits = F.interpolate(logits, size=(H, W), mode="bilinear", align_corners=False)
    return logits.reshape(B, T, -1, *logits.shape[-2:]), torch.from_numpy(np.asarray(s, np.float64))


def _plain_feedback(sd, dims, f4, f8, p_tok, k_tok, q, v, alpha, beta, s0, rule_id, have_mask0, lowres, taps):
    """The per-frame step mode restated (gdkvm_amd.model.GDKVM._forward_feedback is the product's): for t = 0 .. T-1
         R_t = Qn_t S_{t-1};  F_t = KPFF(K_t, R_t, P_t);  logits_t = decoder(F_t, f8_t, f4_t);  mask_t = argmax(bilinear(logits_t))
         v_t += mask_embed(adaptive_avg_pool(mask_t != 0))          (frame 0 keeps the mask0 embedding plain_forward already added)
         S_t = GDR(S_{t-1}, K_t, v_t)
    everything in float64, frame-major tensors as plain_forward made them ([B*T, ...] with clip-major order)."""
    B, T, H, W, h, w, Hh, Dk, Dv = dims
    N = h * w
    qn = O.l2_normalize(q.reshape(B, T, N, Hh, Dk).numpy())
    k5 = k_tok.reshape(B, T, N, Hh, Dk).numpy()
    v5 = v.reshape(B, T, N, Hh * Dv).numpy().copy()
    a_np, b_np = alpha.numpy(), beta.numpy()
    S = np.zeros((B, Hh, Dk, Dv)) if s0 is None else s0.copy()
    w_embed = _t(sd, "mask_embed.weight").reshape(-1).numpy()
    kp = [_t(sd, "kpff." + n).numpy() for n in ("wa", "ba", "wl", "wg")]
    k3, p3 = k_tok.reshape(B, T, N, -1), p_tok.reshape(B, T, N, -1)
    f8_5, f4_5 = f8.reshape(B, T, *f8.shape[1:]), f4.reshape(B, T, *f4.shape[1:])
    lows, masks, margins = [], [], []
    for t in range(T):
        r_t = np.einsum("bnhd,bhdc->bnhc", qn[:, t], S).reshape(B, N, Hh * Dv)
        fused = O.kpff(k3[:, t].numpy(), r_t, p3[:, t].numpy(), *kp, h, w)
        fmap = torch.from_numpy(np.asarray(fused, np.float64)).reshape(B, h, w, -1).permute(0, 3, 1, 2)
        y = _up(sd, "decoder.up4", _up(sd, "decoder.up8", fmap, f8_5[:, t]), f4_5[:, t])
        low = F.conv2d(y, _t(sd, "decoder.head.weight"), _t(sd, "decoder.head.bias"))
        full = F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)
        m_t = full.argmax(1)                                                   # ties -> lowest class
        top2 = full.topk(min(2, full.shape[1]), 1).values
        margins.append(top2[:, 0] - top2[:, -1])
        lows.append(low)
        masks.append(m_t.to(torch.uint8))
        if not (t == 0 and have_mask0):
            v5[:, t] += O.mask_cell_mean((m_t != 0).numpy(), h, w)[:, :, None] * w_embed[None, None, :]
        _, S = O.scan(qn[:, t:t + 1], k5[:, t:t + 1], v5[:, t:t + 1].reshape(B, 1, N, Hh, Dv), a_np[:, t:t + 1], b_np[:, t:t + 1], S, rule_id, 3)
    low = torch.stack(lows, 1)
    if taps is not None:
        taps.update(mask=torch.stack(masks, 1), margin=torch.stack(margins, 1))
    logits = low if lowres else F.interpolate(low.reshape(B * T, *low.shape[2:]), size=(H, W), mode="bilinear", align_corners=False).reshape(B, T, -1, H, W)
    return logits, torch.from_numpy(np.asarray(S, np.float64))


# ------------------------------------------------------------------------------------------------------------------------------------
# The TRAINING objective restated the same way (round 5): train-mode BatchNorm (batch statistics), the memory path on the differentiable
# float64 restatement (oracle/torch_ref.py), cross-entropy + soft Dice with unlabelled pixels left out of every sum -- and autograd
# through all of it.  GDKVMRef's gradients come from the product's own module wiring (it subclasses gdkvm_amd.model.GDKVM); these do not.

def _bn_train(p, prefix, x, eps=1e-5):
    return F.batch_norm(x, None, None, p[prefix + ".weight"], p[prefix + ".bias"], True, 0.0, eps)


def _block_train(p, q, x, stride):
    y = F.relu(_bn_train(p, q + ".bn1", F.conv2d(x, p[q + ".conv1.weight"], None, stride, 1)))
    y = _bn_train(p, q + ".bn2", F.conv2d(y, p[q + ".conv2.weight"], None, 1, 1))
    if (q + ".down.0.weight") in p:
        x = _bn_train(p, q + ".down.1", F.conv2d(x, p[q + ".down.0.weight"], None, stride, 0))
    return F.relu(y + x)


def _up_train(p, q, x, skip):
    x = torch.cat([F.interpolate(x, size=skip.shape[-2:], mode="bilinear", align_corners=False), skip], 1)
    x = F.relu(_bn_train(p, q + ".conv.1", F.conv2d(x, p[q + ".conv.0.weight"], None, 1, 1)))
    return F.relu(_bn_train(p, q + ".conv.4", F.conv2d(x, p[q + ".conv.3.weight"], None, 1, 1)))


def plain_objective(logits, target, dice_weight=1.0, eps=1.0):
    """mean CE over the labelled pixels + dice_weight * (1 - mean_c (2 I_c + eps) / (P_c + O_c + eps)), sums over the whole batch;
    a label outside [0, C) is an unlabelled pixel and enters NO sum (include/gdkvm.h, gdkvm_seg_loss_fwd).  logits [B,T,C,H,W] float64."""
    B, T, C, H, W = logits.shape
    lg = logits.reshape(B * T, C, H, W)
    tg = target.reshape(B * T, H, W).long()
    lab = (tg >= 0) & (tg < C)
    lse = torch.logsumexp(lg, 1)
    picked = torch.gather(lg, 1, torch.where(lab, tg, torch.zeros_like(tg)).unsqueeze(1)).squeeze(1)
    ce = ((lse - picked) * lab).sum() / lab.sum().clamp_min(1)
    p = torch.softmax(lg, 1) * lab.unsqueeze(1)
    oh = torch.stack([(tg == c) & lab for c in range(C)], 1).to(lg.dtype)
    dice = 1.0 - ((2 * (p * oh).sum((0, 2, 3)) + eps) / (p.sum((0, 2, 3)) + oh.sum((0, 2, 3)) + eps)).mean()
    return ce + dice_weight * dice


def plain_loss_and_grads(sd, frames, target, heads=1, key_dim=64, value_dim=256, rule="delta_sequential", dice_weight=1.0, eps=1.0):
    """(loss, {parameter name: d loss / d parameter}) of ONE training step's objective in float64 on the CPU, train-mode BatchNorm, from
    the state_dict alone.  Parameters that do not reach the loss (mask_embed without a first-frame mask) are absent from the dict."""
    from oracle import torch_ref
    p = {k: v.detach().to("cpu", torch.float64).clone().requires_grad_(v.is_floating_point() and "running_" not in k)
         for k, v in sd.items() if v.is_floating_point()}
    B, T, C, H, W = frames.shape
    Hh, Dk, Dv = heads, key_dim, value_dim
    x = frames.detach().to("cpu", torch.float64).reshape(B * T, C, H, W)
    x = F.relu(_bn_train(p, "encoder.stem.1", F.conv2d(x, p["encoder.stem.0.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    f4 = _block_train(p, "encoder.layer1.1", _block_train(p, "encoder.layer1.0", x, 1), 1)
    f8 = _block_train(p, "encoder.layer2.1", _block_train(p, "encoder.layer2.0", f4, 2), 1)
    f16 = _block_train(p, "encoder.layer3.1", _block_train(p, "encoder.layer3.0", f8, 2), 1)
    h, w = f16.shape[-2:]
    N = h * w
    p_tok = _tokens(f16)
    lin = lambda name: p_tok @ p[name + ".weight"].reshape(p[name + ".weight"].shape[0], -1).T + p[name + ".bias"]
    k_tok, q = lin("key_proj"), lin("query_proj")
    v = lin("value_proj").reshape(B, T, N, Hh, Dv)
    beta = lin("gate_proj").reshape(B, T, N, Hh)
    alpha = (p_tok.mean(1) @ p["decay_proj.weight"].T + p["decay_proj.bias"]).reshape(B, T, Hh)
    r, _ = torch_ref.scan(q.reshape(B, T, N, Hh, Dk), k_tok.reshape(B, T, N, Hh, Dk), v, alpha, beta, None, _RULE_IDS[rule], 3)
    fused = torch_ref.kpff(k_tok, r.reshape(B * T, N, Hh * Dv), p_tok, p["kpff.wa"], p["kpff.ba"], p["kpff.wl"], p["kpff.wg"], h, w)
    fmap = fused.reshape(B * T, h, w, -1).permute(0, 3, 1, 2)
    y = _up_train(p, "decoder.up4", _up_train(p, "decoder.up8", fmap, f8), f4)
    logits = F.interpolate(F.conv2d(y, p["decoder.head.weight"], p["decoder.head.bias"]), size=(H, W), mode="bilinear", align_corners=False)
    loss = plain_objective(logits.reshape(B, T, -1, H, W), target.detach().cpu(), dice_weight, eps)
    names = [k for k, t in p.items() if t.requires_grad]
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    return loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}
