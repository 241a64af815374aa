#!/usr/bin/env python3
"""Which stage owns the mask flips of the bf16 inference build?  (VERDICT r02 item 2 asked for `tools/stage_error.py`; it lives
under tests/ because it runs the oracle, which only tests/, smoke() and bench.py's cpu_baseline leg may do.)

    python tests/stage_error.py [--clips 2] [--frames 8] [--size 112] [--classes 2] [--out gpurun_out/stage_error.json]

Runs the fused bf16 build (the product, on the GPU) and oracle/model_plain.plain_forward (float64, CPU) on the same clips and
weights, taps every stage of both -- stride-4/8/16 encoder features, k / q / v projections, LKVA read-out R, KPFF output F, the
stride-4 decoder logits -- and prints
  1. per stage: max and mean absolute error and the error relative to the stage's RMS;
  2. an attribution of the argmax flips: plain_forward re-run with the product's stages substituted one after the other
     (encoder features -> + read-out -> + KPFF output -> the product's own logits), everything downstream in float64, and the
     share of pixels whose class differs from the all-float64 run at each step."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def product_taps(model, frames):
    """One forward of the (fused bf16) product with its stages recorded."""
    taps = {}
    hook = model.encoder.register_forward_hook(lambda m, i, o: taps.update(f4=o[0], f8=o[1], f16=o[2]))
    scan0, fuse0 = model._memory_scan, model._fuse

    def scan(q, k, v, al, be, st, **kw):
        r, s = scan0(q, k, v, al, be, st, **kw)
        B, T, N = q.shape[:3]
        taps.update(q=q.reshape(B * T, N, -1), k=k.reshape(B * T, N, -1), v=v.reshape(B * T, N, -1), r=r.reshape(B * T, N, -1),
                    beta=be, alpha=al)
        return r, s

    def fuse(local, glob, pixel, h, w):
        out = fuse0(local, glob, pixel, h, w)
        taps["fused"] = out
        return out

    model._memory_scan, model._fuse = scan, fuse
    try:
        with torch.no_grad():
            taps["logits"] = model(frames, _lowres=True)
    finally:
        hook.remove()
        model._memory_scan, model._fuse = scan0, fuse0
    return {k: v.detach().float().cpu().double() for k, v in taps.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=112)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stage_error.json"))
    args = ap.parse_args()
    from gdkvm_amd import ops
    from gdkvm_amd.model import GDKVM, GDKVMConfig
    from oracle.model_plain import plain_forward
    ops.require_native()
    B, T, S = args.clips, args.frames, args.size
    torch.manual_seed(args.seed)
    cfg = GDKVMConfig(num_classes=args.classes)
    model = GDKVM(cfg).eval()
    for m in model.modules():                               # non-trivial BatchNorm statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.8, 1.25)
    g = torch.Generator().manual_seed(1000 + args.seed)
    u = torch.rand(B, T, 3, S, S, generator=g)
    frames = (u * torch.sqrt(-2.0 * torch.log(torch.rand(B, T, 1, S, S, generator=g).clamp_min(1e-7))) * 0.25).clamp_(0, 1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    lp, _ = plain_forward(sd, frames, lowres=True)
    with torch.no_grad():                                   # balance the random-init head: every class present in the reference masks
        med = lp.flatten(3).median(-1).values.mean((0, 1))
        model.decoder.head.bias -= med.float()
        model.decoder.head.bias.copy_(model.decoder.head.bias.bfloat16().float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = {}
    lp, _ = plain_forward(sd, frames, lowres=True, taps=ref)
    ref["logits"] = lp.reshape(B * T, -1, *lp.shape[-2:])

    dev = torch.device("cuda")
    fused = model.to(dev).to(memory_format=torch.channels_last).fuse_for_inference().to(torch.bfloat16)
    got = product_taps(fused, frames.to(dev).to(torch.bfloat16))
    got["logits"] = got["logits"].reshape(B * T, -1, *got["logits"].shape[-2:])

    def up_mask(lg):                                        # [BT, C, h, w] stride-4 logits -> full-resolution class map
        return F.interpolate(lg, size=(S, S), mode="bilinear", align_corners=False).argmax(1)

    stages = []
    for name in ("f4", "f8", "f16", "k", "q", "v", "beta", "alpha", "r", "fused", "logits"):
        a, b = got[name].reshape(ref[name].shape), ref[name]
        err = (a - b).abs()
        rms = b.pow(2).mean().sqrt().item()
        stages.append({"stage": name, "max_abs": err.max().item(), "mean_abs": err.mean().item(), "ref_rms": rms,
                       "mean_rel_to_rms": err.mean().item() / max(rms, 1e-30)})
    m_ref = up_mask(ref["logits"])
    fg = [(m_ref == c).float().mean().item() for c in range(args.classes)]

    def flips(override):
        lg, _ = plain_forward(sd, frames, lowres=True, override=override)
        return (up_mask(lg.reshape(B * T, -1, *lg.shape[-2:])) != m_ref).float().mean().item()

    enc = {k: got[k] for k in ("f4", "f8", "f16")}
    attribution = [
        {"product_stages": "encoder features f4 / f8 / f16 (bf16 convolutions)", "flip_fraction": flips(enc)},
        {"product_stages": "+ read-out R (projections and scan on the bf16 feature, R rounded to bf16)", "flip_fraction": flips({**enc, "r": got["r"]})},
        {"product_stages": "+ KPFF output F (bf16 weights and pooled feature)", "flip_fraction": flips({**enc, "fused": got["fused"]})},
        {"product_stages": "the product's own logits (+ decoder convolutions, head)", "flip_fraction": (up_mask(got["logits"]) != m_ref).float().mean().item()},
    ]
    # the same decoder fed float64 encoder skips but the product's fused map: isolates the memory path + KPFF from the encoder skips
    attribution.append({"product_stages": "only F from the product (float64 f4 / f8 skips into the decoder)",
                        "flip_fraction": flips({"fused": got["fused"]})})
    margin = ref["logits"].sort(1, descending=True).values
    margin = F.interpolate((margin[:, 0] - margin[:, 1]).unsqueeze(1), size=(S, S), mode="bilinear", align_corners=False)
    out = {"shape": [B, T, S, S], "classes": args.classes, "class_fractions_reference": fg, "stages": stages, "attribution": attribution,
           "reference_margin_quantiles": {q: margin.flatten().quantile(q).item() for q in (0.01, 0.02, 0.05, 0.5)}}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(f"clips {B} x {T} frames {S}x{S}, {args.classes} classes; reference class shares {[round(x, 3) for x in fg]}")
    print(f"{'stage':8s} {'max |err|':>11s} {'mean |err|':>11s} {'ref rms':>9s} {'mean/rms':>9s}")
    for s_ in stages:
        print(f"{s_['stage']:8s} {s_['max_abs']:11.3e} {s_['mean_abs']:11.3e} {s_['ref_rms']:9.3f} {s_['mean_rel_to_rms']:9.2e}")
    print("flips against the all-float64 masks, product stages substituted cumulatively:")
    for a_ in attribution:
        print(f"  {a_['flip_fraction'] * 100:6.3f} %   {a_['product_stages']}")
    print("reference |top1 - top2| logit margin quantiles:", {k: round(v, 4) for k, v in out["reference_margin_quantiles"].items()})


if __name__ == "__main__":
    main()
