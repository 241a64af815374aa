#!/usr/bin/env python3
"""SURVEY §8f row n4, measured: everything the memory path derives from the pixel feature, in its two forms, at the cfg2 shapes
(512 frames x 49 tokens x 256 channels, bf16), HIP events on the launch stream:
    three launches  gdkvm_proj_rows + gdkvm_gate_logits + gdkvm_scan_fwd (norms inside the frame-parallel kernel)
    fused           gdkvm_proj_gates + gdkvm_scan_fwd_normed
    python3 tools/n4_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402
from tools.config_sweep import ev_time  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
B, T, N, Hh, Dk, Dv, Cp = 16, 32, 49, 1, 64, 256, 256
g = torch.Generator(device=dev).manual_seed(1)
p = torch.randn(B * T, N, Cp, device=dev, generator=g).bfloat16()
w = torch.randn(2 * Dk + Dv, Cp, device=dev, generator=g) / Cp ** 0.5
b = torch.randn(2 * Dk + Dv, device=dev, generator=g)
wg, bg = torch.randn(Hh, Cp, device=dev, generator=g) / Cp ** 0.5, torch.zeros(Hh, device=dev)
wd, bd = torch.randn(Hh, Cp, device=dev, generator=g) / Cp ** 0.5, 2 + torch.zeros(Hh, device=dev)
pack = ops.pack_rows_weight(w)
ws = torch.empty(ops.scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=dev)
r = torch.empty(B, T, N, Hh, Dv, device=dev, dtype=torch.bfloat16)
s = torch.empty(B, Hh, Dk, Dv, device=dev)
sh = lambda t, c: t.reshape(B, T, N, Hh, c)


def three():
    k, q, v = ops.proj_rows(p.reshape(B * T * N, Cp), pack, b, (Dk, Dk, Dv))
    be, al = ops.gate_logits(p, wg, bg, wd, bd)
    return ops.scan_fwd(sh(q, Dk), sh(k, Dk), sh(v, Dv), al.reshape(B, T, Hh), be.reshape(B, T, N, Hh), flags=3, workspace=ws, out=r, state_out=s)


def fused():
    (k, q, v), (be, al), nm = ops.proj_gates(p, pack, b, wg, bg, wd, bd, Hh, Dk, Dv)
    return ops.scan_fwd(sh(q, Dk), sh(k, Dk), sh(v, Dv), al.reshape(B, T, Hh), be.reshape(B, T, N, Hh), flags=3, workspace=ws, out=r, state_out=s,
                        norms=nm)


r3, s3 = three(); r3 = r3.clone(); s3 = s3.clone()
rf, sf = fused()
print(f"max |dS| fused vs three launches {(sf - s3).abs().max().item():.2e}   max |dR| {(rf.float() - r3.float()).abs().max().item():.2e}")
(k, q, v), (be, al), nm = ops.proj_gates(p, pack, b, wg, bg, wd, bd, Hh, Dk, Dv)
args = (sh(q, Dk), sh(k, Dk), sh(v, Dv), al.reshape(B, T, Hh), be.reshape(B, T, N, Hh))
rows = [("gdkvm_proj_rows", lambda: ops.proj_rows(p.reshape(B * T * N, Cp), pack, b, (Dk, Dk, Dv))),
        ("gdkvm_gate_logits", lambda: ops.gate_logits(p, wg, bg, wd, bd)),
        ("gdkvm_proj_gates", lambda: ops.proj_gates(p, pack, b, wg, bg, wd, bd, Hh, Dk, Dv)),
        ("gdkvm_scan_prep", lambda: ops.scan_prep(*args[:3], args[4], ws, flags=3)),
        ("gdkvm_scan_fwd", lambda: ops.scan_fwd(*args, flags=3, workspace=ws, out=r, state_out=s)),
        ("gdkvm_scan_fwd_normed", lambda: ops.scan_fwd(*args, flags=3, workspace=ws, out=r, state_out=s, norms=nm)),
        ("three launches: proj_rows + gate_logits + scan_fwd", three),
        ("fused: proj_gates + scan_fwd_normed", fused)]
for name, fn in rows:
    print(f"{name:55s} {ev_time(fn, iters=iters):8.1f} us")

# The full fold (K / Q / V computed inside the frame-parallel kernel) must keep the frame's value tile (64 x 256 bf16 = 32 KiB) next to
# the kernel's own 72 KiB: 104 KiB, ONE workgroup per CU instead of two.  What that occupancy alone costs, before any projection work:
#     GDKVM_PREP_LDS_PAD_KB=32 python3 tools/n4_bench.py   (the gdkvm_scan_prep row)
if os.environ.get("GDKVM_PREP_LDS_PAD_KB"):
    print(f"(gdkvm_scan_prep above ran with {os.environ['GDKVM_PREP_LDS_PAD_KB']} KiB of extra LDS per workgroup)")
