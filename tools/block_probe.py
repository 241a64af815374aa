#!/usr/bin/env python3
"""gdkvm_scan_fwd as overlapping time blocks (csrc/gdr_pipeline.hip) against the plain prep -> scan -> read-out sequence: time per call
(HIP events on the launch stream) and bit-identity of read-outs and final state, per shape and block plan.
usage: block_probe.py [cfg2|cfg3|cfg5|cfg5c ...]   (GDKVM_SCAN_BLOCKS / GDKVM_SCAN_BLOCK_LIST are set by the tool itself)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdkvm_amd import ops  # noqa: E402
from tools.config_sweep import ev_time  # noqa: E402

SHAPES = {"cfg2": (16, 32, 49), "cfg3": (8, 20, 256), "cfg5": (2, 512, 256), "cfg5c": (2, 32, 256), "cfg5h": (2, 128, 256), "cfg5l": (1, 300, 200)}
PLANS = {
    "cfg2": ["1", "2", "4", "L8", "L4,12", "L4,8,16"],
    "cfg3": ["1", "P", "P140", "S1", "S2", "S4", "S3", "S5", "S6"],
    "cfg5": ["1", "P", "P140", "S1", "S2", "S4", "S3", "S5", "S6"],
    "cfg5c": ["1", "P"],
    "cfg5h": ["1", "P", "P140"],
    "cfg5l": ["1", "P"],
}


GRAPH = os.environ.get("BLOCK_PROBE_GRAPH", "1") == "1"


def set_plan(p):
    os.environ.pop("GDKVM_SCAN_BLOCKS", None)
    os.environ.pop("GDKVM_SCAN_BLOCK_LIST", None)
    os.environ.pop("GDKVM_PIPE_SCAN_LDS_KB", None)
    os.environ.pop("GDKVM_PIPE_STAGES", None)
    os.environ["GDKVM_SCAN_PIPE"] = "0"
    if p.startswith("S"):                                  # timing only: a subset of the concurrent form's stages (1 fold, 2 recurrence, 4 read-out)
        os.environ["GDKVM_SCAN_PIPE"] = "1"
        os.environ["GDKVM_PIPE_STAGES"] = p[1:]
    elif p.startswith("P"):                                # the concurrent form (flags), optionally with the serial workgroups' LDS padded
        os.environ["GDKVM_SCAN_PIPE"] = "1"
        if len(p) > 1:
            os.environ["GDKVM_PIPE_SCAN_LDS_KB"] = p[1:]
    elif p.startswith("L"):
        os.environ["GDKVM_SCAN_BLOCK_LIST"] = p[1:]
    else:
        os.environ["GDKVM_SCAN_BLOCKS"] = p


def main():
    dev = torch.device("cuda")
    names = sys.argv[1:] or ["cfg5", "cfg3", "cfg2"]
    Dv = 256
    for name in names:
        B, T, N = SHAPES[name]
        for dt in (torch.bfloat16, torch.float32) if os.environ.get("BLOCK_PROBE_F32") else (torch.bfloat16,):
            g = torch.Generator(device=dev).manual_seed(1)
            q, k = (torch.randn(B, T, N, 1, 64, device=dev, generator=g).to(dt) for _ in range(2))
            v = torch.randn(B, T, N, 1, Dv, device=dev, generator=g).to(dt)
            al = 2 + torch.randn(B, T, 1, device=dev, generator=g)
            be = torch.randn(B, T, N, 1, device=dev, generator=g)
            s0 = torch.randn(B, 1, 64, Dv, device=dev, generator=g)
            ws = ops.new_workspace(B, T, 1, N, 64, Dv, dev)
            r = torch.empty(B, T, N, 1, Dv, device=dev, dtype=dt)
            s = torch.empty(B, 1, 64, Dv, device=dev)
            es = q.element_size()
            alg = B * T * (es * N * (2 * 64 + 2 * Dv) + 4 * (1 + N)) + B * 2 * 4 * 64 * Dv

            def run():
                ops.scan_fwd(q, k, v, al, be, s0, flags=3, workspace=ws, out=r, state_out=s)

            ref = None
            for plan in (os.environ["BLOCK_PROBE_PLANS"].split(";") if os.environ.get("BLOCK_PROBE_PLANS") else PLANS[name]):
                set_plan(plan)
                r.zero_(); s.zero_()
                run(); torch.cuda.synchronize()
                if ref is None:
                    ref = (r.clone(), s.clone())
                same = torch.equal(r, ref[0]) and torch.equal(s, ref[1])
                if not same and not plan.startswith("S"):                               # where: which frames' read-outs differ, and the final state
                    d = (r.float() - ref[0].float()).abs().amax(dim=(2, 3, 4))                     # [B, T]
                    bad_t = (d != 0).any(0).nonzero().flatten().tolist()
                    print(f"   read-out: {int((r != ref[0]).sum())} of {r.numel()} differ, max |d| {float(d.max()):.3g}, nan {int(torch.isnan(r.float()).sum())}, "
                          f"frames {bad_t[:12]}{'...' if len(bad_t) > 12 else ''} ({len(bad_t)} of {T});  state: {int((s != ref[1]).sum())} of {s.numel()} differ, "
                          f"max |d| {float((s - ref[1]).abs().max()):.3g}", flush=True)
                t = ev_time(run, iters=20 if T < 256 else 10)
                tg = float("nan")
                if GRAPH:                                  # the same call captured once and replayed: what the host's enqueue time hides
                    try:
                        gr = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gr):
                            run()
                        r.zero_(); s.zero_()
                        gr.replay(); torch.cuda.synchronize()
                        same = same and torch.equal(r, ref[0]) and torch.equal(s, ref[1])
                        tg = ev_time(gr.replay, iters=20 if T < 256 else 10)
                    except Exception as e:                 # noqa: BLE001
                        print("graph capture failed:", str(e).splitlines()[0], flush=True)
                print(f"{name} {B}x{T}x{N} {'bf16' if dt == torch.bfloat16 else 'f32 '} blocks {plan:24s} {t:8.1f} us  frac {alg / (t * 1e-6) / 8e12:.4f}  "
                      f"graph replay {tg:8.1f} us  frac {alg / (tg * 1e-6) / 8e12:.4f}  bit-identical {same}", flush=True)
            set_plan("1")


if __name__ == "__main__":
    main()
