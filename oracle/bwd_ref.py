"""numpy fp64 restatement of the BACKWARD algorithm the HIP kernels implement -- the
reverse-time recurrence on dS plus the per-frame gradient assembly through the WY factors.  TEST INFRASTRUCTURE ONLY.
It is checked against autograd of oracle/torch_ref.py (tests/test_oracle_kat.py), which proves the derivation in
DESIGN.md §2.5 before any kernel is compared with it.

Forward per frame (Kn, Qn normalised; a, b activated gates; T = (I + tril(diag(b) Kn Kn^T, -1))^-1):
    Wt = T diag(b) Kn   Ut = T diag(b) V   R = Qn S   X = Wt S   U = Ut - a X   S' = a S + Kn^T U
Backward, given dR_t and dS' (gradient w.r.t. the state AFTER the frame):
    dS   = a (dS' - Wt^T (Kn dS')) + Qn^T dR                 (reverse recurrence: same shape as the forward one)
    dU   = Kn dS'      dKn = U dS'^T      dWt = -a dU S^T      dQn = dR S^T      da = <S,dS'> - <X,dU>
    Z    = T^T [dWt | dU]         db = rowsum(Z * [Kn|V]) + rowsum(tril(dA,-1) * (Kn Kn^T))
    dA   = -tril(Z [Wt|Ut]^T, -1) M = diag(b) dA       dKn += b Z_K + M Kn + M^T Kn       dV = b Z_V
then through the L2 normalisation and the sigmoids."""
from __future__ import annotations

import numpy as np

EPS = 1e-12


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def scan_backward(q, k, v, alpha, beta, s0, dR, dS_T, rule=2, flags=0):
    q, k, v, alpha, beta, dR = (np.asarray(x, np.float64) for x in (q, k, v, alpha, beta, dR))
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    dalpha, dbeta = np.zeros_like(alpha), np.zeros_like(beta)
    ds0 = np.zeros((B, Hh, Dk, Dv))
    for b in range(B):
        for h in range(Hh):
            # ---- forward replay, saving the state before every frame (the HIP forward writes s_hist)
            S = np.zeros((Dk, Dv)) if s0 is None else np.array(s0[b, h], np.float64)
            hist, fac = [], []
            for t in range(T):
                qt, kt, vt = q[b, t, :, h], k[b, t, :, h], v[b, t, :, h]
                qinv = 1 / np.sqrt((qt * qt).sum(-1) + EPS) if flags & 1 else np.ones(N)
                kinv = 1 / np.sqrt((kt * kt).sum(-1) + EPS) if flags & 1 else np.ones(N)
                Qn, Kn = qt * qinv[:, None], kt * kinv[:, None]
                a = _sig(alpha[b, t, h]) if flags & 2 else alpha[b, t, h]
                bt = _sig(beta[b, t, :, h]) if flags & 2 else beta[b, t, :, h]
                G = Kn @ Kn.T
                A = np.tril(bt[:, None] * G, -1) if rule == 2 else np.zeros((N, N))
                Tm = np.linalg.inv(np.eye(N) + A)
                Wt = Tm @ (bt[:, None] * Kn) if rule != 0 else np.zeros((N, Dk))
                Ut = Tm @ (bt[:, None] * vt)
                hist.append(S.copy()); fac.append((Qn, Kn, qinv, kinv, a, bt, G, Tm, Wt, Ut))
                S = a * S + Kn.T @ (Ut - a * (Wt @ S))
            # ---- reverse recurrence + per-frame assembly
            dSn = np.zeros((Dk, Dv)) if dS_T is None else np.array(dS_T[b, h], np.float64)
            for t in range(T - 1, -1, -1):
                Qn, Kn, qinv, kinv, a, bt, G, Tm, Wt, Ut = fac[t]
                S = hist[t]
                vt, dRt = v[b, t, :, h], dR[b, t, :, h]
                X = Wt @ S
                U = Ut - a * X
                dU = Kn @ dSn
                dKn = U @ dSn.T
                dWt = -a * dU @ S.T
                dQn = dRt @ S.T
                da = (S * dSn).sum() - (X * dU).sum()
                dS_prev = a * (dSn - Wt.T @ dU) + Qn.T @ dRt
                # through T
                dY = np.concatenate([dWt, dU], 1)
                Z = Tm.T @ dY
                X0 = np.concatenate([Kn, vt], 1)
                Y = np.concatenate([Wt, Ut], 1)
                if rule == 0:
                    Z[:, :Dk] = 0.0                                   # Wt == 0 does not depend on the inputs
                db = (Z * X0).sum(1)
                dKn = dKn + bt[:, None] * Z[:, :Dk]
                dv[b, t, :, h] = bt[:, None] * Z[:, Dk:]
                if rule == 2:
                    dA = -np.tril(Z @ Y.T, -1)
                    db = db + (dA * G).sum(1)
                    M = bt[:, None] * dA
                    dKn = dKn + M @ Kn + M.T @ Kn
                # normalisation and gates
                if flags & 1:
                    dk[b, t, :, h] = kinv[:, None] * (dKn - Kn * (Kn * dKn).sum(1, keepdims=True))
                    dq[b, t, :, h] = qinv[:, None] * (dQn - Qn * (Qn * dQn).sum(1, keepdims=True))
                else:
                    dk[b, t, :, h], dq[b, t, :, h] = dKn, dQn
                dalpha[b, t, h] = da * a * (1 - a) if flags & 2 else da
                dbeta[b, t, :, h] = db * bt * (1 - bt) if flags & 2 else db
                dSn = dS_prev
            ds0[b, h] = dSn
    return dq, dk, dv, dalpha, dbeta, ds0
