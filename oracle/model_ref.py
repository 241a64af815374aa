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

    def _memory_scan(self, q, k, v, alpha_logit, beta_logit, state, norms=None, readout=True):
        if self.cfg.normalizer:                               # SURVEY A.1 flag: module state = [S | z]
            args = [t.detach().float().cpu().numpy() for t in (q, k, v, alpha_logit, beta_logit)]
            Dv = v.shape[-1]
            s0 = None if state is None else state[..., :Dv].detach().float().cpu().numpy()
            z0 = None if state is None else state[..., Dv].detach().float().cpu().numpy()
            r, s, z = c_oracle.scan_normalizer(*args, s0, z0, _RULES[self.cfg.rule], 3, self.cfg.normalizer_eps, math=self.math)
            return torch.from_numpy(r).to(q.dtype), torch.cat([torch.from_numpy(s), torch.from_numpy(z).unsqueeze(-1)], -1)
        if torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v, alpha_logit, beta_logit)):
            from oracle import torch_ref                      # differentiable restatement: the gradient oracle
            r, s = torch_ref.scan(q.double(), k.double(), v.double(), alpha_logit.double(), beta_logit.double(),
                                  None if state is None else state.double(), _RULES[self.cfg.rule], 3)
            return r.to(q.dtype), s.float()
        args = [t.detach().float().cpu().numpy() for t in (q, k, v, alpha_logit, beta_logit)]
        s0 = None if state is None else state.detach().float().cpu().numpy()
        r, s = c_oracle.scan(*args, s0, _RULES[self.cfg.rule], 3, math=self.math)
        return torch.from_numpy(r).to(q.dtype), torch.from_numpy(s)

    # ---- the per-frame step mode (cfg.mask_feedback): read, predicted-mask embedding, mask from stride-4 logits -- all on the CPU oracle
    def _memory_read(self, q, state, norms=None):
        from oracle import gdkvm_oracle as O
        qn = O.l2_normalize(q.detach().double().cpu().numpy())                        # [B,N,Hh,Dk]
        r = np.einsum("bnhd,bhdc->bnhc", qn, state.detach().double().cpu().numpy())
        return torch.from_numpy(r).to(q.dtype)

    def _embed_mask_(self, v, mask, h, w):
        from oracle import gdkvm_oracle as O
        m = mask.cpu().numpy()
        pooled = O.mask_cell_mean((m != 0) & (m != 255), h, w)                        # [B, N]
        wv = self.mask_embed.weight.detach().double().reshape(-1).numpy()
        return (v.double() + torch.from_numpy(pooled[:, :, None] * wv[None, None, :])).to(v.dtype)

    def _mask_from_lowres(self, lowres, H, W, target=None, mask_out=None, counts_out=None):
        m, c = c_oracle.upsample_argmax_dice(lowres.detach().float().cpu().numpy(), H, W, None if target is None else target.cpu().numpy())
        return torch.from_numpy(m), (None if c is None else torch.from_numpy(c))

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
        if self.cfg.mask_feedback:
            _, mask, counts, _ = self._forward_feedback(frames, kw.get("mask0"), kw.get("state"), target=target, masks_only=True)
            return mask, counts
        lowres = self.forward(frames, _lowres=True, **kw)
        ncls, hl, wl = lowres.shape[2:]
        tgt = None if target is None else target.reshape(B * T, H, W).numpy()
        mask, counts = c_oracle.upsample_argmax_dice(lowres.reshape(B * T, ncls, hl, wl).float().numpy(), H, W, tgt)
        return (torch.from_numpy(mask).reshape(B, T, H, W),
                None if counts is None else torch.from_numpy(counts).reshape(B, T, ncls, 3))
