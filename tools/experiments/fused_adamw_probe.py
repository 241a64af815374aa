#!/usr/bin/env python3
"""Does torch.optim.AdamW(fused=True) pair parameter and gradient elements correctly when they are channels_last?  (round 5: the fused
optimiser trained the model ~10x slower than the default one with the same step magnitudes.)"""
import json
import torch

dev = torch.device("cuda", 0)
out = []
for name, shape, cl_p, cl_g in (("contig/contig", (64, 32, 3, 3), False, False), ("cl/cl", (64, 32, 3, 3), True, True),
                                ("cl/contig-grad", (64, 32, 3, 3), True, False), ("contig/cl-grad", (64, 32, 3, 3), False, True),
                                ("2d", (64, 32), False, False)):
    torch.manual_seed(0)
    w0 = torch.randn(shape, device=dev)
    g0 = torch.randn(shape, device=dev)
    res = {}
    for kind, kw in (("default", {}), ("fused", {"fused": True}), ("fused_capturable", {"fused": True, "capturable": True})):
        p = torch.nn.Parameter(w0.clone().contiguous(memory_format=torch.channels_last) if cl_p and w0.dim() == 4 else w0.clone())
        p.grad = g0.clone().contiguous(memory_format=torch.channels_last) if cl_g and w0.dim() == 4 else g0.clone()
        opt = torch.optim.AdamW([p], lr=1e-2, weight_decay=0.0, **kw)
        try:
            opt.step()
        except RuntimeError as e:
            res[kind] = {'error': str(e)[:80]}
            continue
        d = (p.detach() - w0)
        res[kind] = {"sign_agrees_with_minus_grad": float(((d < 0) == (g0 > 0)).float().mean()), "param_strides": list(p.stride()), "grad_strides": list(p.grad.stride()),
                     "state_strides": list(opt.state[p]["exp_avg"].stride())}
    out.append({"case": name, **res})
for o in out:
    print(json.dumps(o))
