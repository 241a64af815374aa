"""GDKVMRef -- the CPU reference of the whole module: the SAME encoder/decoder definitions and state_dict as
gdkvm_amd.model.GDKVM (PyTorch CPU convolutions), with the memory path replaced by the CPU oracle
(oracle/gdkvm_oracle.c through ctypes).  TEST INFRASTRUCTURE ONLY: used by tests/, smoke() and bench.py's
cpu_baseline leg; the product never imports this (the dependency points from the oracle to the product's
layer definitions, never the other way)."""
from __future__ import annotations

import numpy as np
import torch

from gdkvm_amd.model import GDKVM, _RULES
from oracle import c_oracle


class GDKVMRef(GDKVM):
    math = "f32"          # arithmetic of the C oracle: "f32" (CPU-baseline speed) or "f64" (parity checks)

    def _memory_scan(self, q, k, v, alpha_logit, beta_logit, state):
        if torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v, alpha_logit, beta_logit)):
            from oracle import torch_ref                      # differentiable restatement: the gradient oracle
            r, s = torch_ref.scan(q.double(), k.double(), v.double(), alpha_logit.double(), beta_logit.double(),
                                  None if state is None else state.double(), _RULES[self.cfg.rule], 3)
            return r.to(q.dtype), s.float()
        args = [t.detach().float().cpu().numpy() for t in (q, k, v, alpha_logit, beta_logit)]
        s0 = None if state is None else state.detach().float().cpu().numpy()
        r, s = c_oracle.scan(*args, s0, _RULES[self.cfg.rule], 3, math=self.math)
        return torch.from_numpy(r).to(q.dtype), torch.from_numpy(s)

    def _fuse(self, local, glob, pixel, h, w):
        p = self.kpff
        if torch.is_grad_enabled() and (pixel.requires_grad or p.wa.requires_grad):
            from oracle import torch_ref
            return torch_ref.kpff(*(t.double() for t in (local, glob, pixel, p.wa, p.ba, p.wl, p.wg)), h, w).to(pixel.dtype)
        f = c_oracle.kpff(*(t.detach().float().cpu().numpy() for t in (local, glob, pixel, p.wa, p.ba, p.wl, p.wg)),
                          h, w, math=self.math)
        return torch.from_numpy(f).to(pixel.dtype)

    @torch.no_grad()
    def segment(self, frames, target=None, **kw):
        B, T, _, H, W = frames.shape
        lowres = self.forward(frames, _lowres=True, **kw)
        ncls, hl, wl = lowres.shape[2:]
        tgt = None if target is None else target.reshape(B * T, H, W).numpy()
        mask, counts = c_oracle.upsample_argmax_dice(lowres.reshape(B * T, ncls, hl, wl).float().numpy(), H, W, tgt)
        return (torch.from_numpy(mask).reshape(B, T, H, W),
                None if counts is None else torch.from_numpy(counts).reshape(B, T, ncls, 3))
