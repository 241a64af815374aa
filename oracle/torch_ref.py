"""Differentiable fp64 restatement of the memory path in plain torch (CPU autograd) -- the GRADIENT oracle for
SURVEY.md §8 row a7.  TEST INFRASTRUCTURE ONLY (same rules as oracle/gdkvm_oracle.py: never imported by gdkvm_amd/).
Semantics are those of oracle/gdkvm_oracle.py (SPEC-v0; parity unpinned, see that header);
tests/test_oracle_kat.py::test_torch_ref_matches_numpy_oracle ties the two together, so autograd through this file
yields the derivatives of exactly the function the HIP forward computes."""
from __future__ import annotations

import torch

EPS_NORM = 1e-12
KPFF_SCALES = (1, 2, 4)


def scan(q, k, v, alpha, beta, s0=None, rule=2, flags=0):
    """q,k [B,T,N,Hh,Dk] v [B,T,N,Hh,Dv] alpha [B,T,Hh] beta [B,T,N,Hh] -> (R [B,T,N,Hh,Dv], S_T [B,Hh,Dk,Dv]).
    Token-sequential definition, vectorised over clips and heads."""
    if flags & 1:
        q = q * torch.rsqrt((q * q).sum(-1, keepdim=True) + EPS_NORM)
        k = k * torch.rsqrt((k * k).sum(-1, keepdim=True) + EPS_NORM)
    if flags & 2:
        alpha, beta = torch.sigmoid(alpha), torch.sigmoid(beta)
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    S = torch.zeros(B, Hh, Dk, Dv, dtype=q.dtype) if s0 is None else s0
    R = []
    for t in range(T):
        qt, kt, vt = (x[:, t].permute(0, 2, 1, 3) for x in (q, k, v))          # [B,Hh,N,D]
        bt = beta[:, t].permute(0, 2, 1)                                        # [B,Hh,N]
        R.append((qt @ S).permute(0, 2, 1, 3))
        S = alpha[:, t, :, None, None] * S
        if rule == 0:
            S = S + kt.transpose(-1, -2) @ (bt[..., None] * vt)
        elif rule == 1:
            S = S + kt.transpose(-1, -2) @ (bt[..., None] * (vt - kt @ S))
        else:
            for i in range(N):
                ki, vi = kt[:, :, i], vt[:, :, i]                               # [B,Hh,Dk], [B,Hh,Dv]
                e = vi - (ki[..., None, :] @ S)[..., 0, :]
                S = S + bt[:, :, i, None, None] * ki[..., :, None] * e[..., None, :]
    return torch.stack(R, 1), S


def multiscale_pool(G, h, w):
    BT, N, C = G.shape
    g = G.reshape(BT, h, w, C)
    acc = g.clone()
    for s in KPFF_SCALES[1:]:
        out = torch.empty_like(g)
        for y0 in range(0, h, s):
            for x0 in range(0, w, s):
                out[:, y0:y0 + s, x0:x0 + s] = g[:, y0:y0 + s, x0:x0 + s].mean((1, 2), keepdim=True)
        acc = acc + out
    return (acc / len(KPFF_SCALES)).reshape(BT, N, C)


def kpff(L, G, P, Wa, ba, Wl, Wg, h, w):
    Cp = P.shape[-1]
    Gms = multiscale_pool(G, h, w)
    g = torch.sigmoid(torch.cat([P, L, Gms], -1) @ Wa.T + ba)
    return P + g[..., :Cp] * (L @ Wl.T) + g[..., Cp:] * (Gms @ Wg.T)
