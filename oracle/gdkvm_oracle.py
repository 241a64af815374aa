"""CPU oracle (numpy) for the GDKVM memory path: LKVA read, GDR write, scan, KPFF, argmax+Dice.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gdkvm_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and there only as the
checker.  The product path is the HIP library behind ``include/gdkvm.h``.

PARITY UNPINNED.  The reference snapshot (/root/reference) is the project *website*: it contains no
implementation, test, fixture or golden vector of this path (the code lives in the
un-vendored, un-pinned repo named at /root/reference/README.md:1 and excluded at
/root/reference/.gitignore:73-76).  What this file restates is therefore the builder's SPEC-v0
(SURVEY.md Appendix A), derived from the only reference text that describes the method:
  * /root/reference/README.md:20  (LKVA "model inter-frame correlations", GDR "store intermediate
    memory states", KPFF "integrate local and global features at multiple scales"),
  * /root/reference/website/src/content/homepage/en.json:20  (LKVA = "state transition matrix", GDR
    "dynamically managing memory", KPFF "fuses the local key feature, the global key feature with the
    pixel feature"),
  * /root/repo/BASELINE.json north_star ("S_t = a*S_{t-1} + b*k_t v_t^T style outer-product update",
    "Q*K^T softmax-free linear attention").
The oracle is pinned by the analytic known-answer tests of SURVEY.md A.6 (tests/test_oracle_kat.py),
by fp64-vs-fp32 agreement, and by the independent scalar C restatement in oracle/gdkvm_oracle.c.

Layouts (token-major, channel innermost -- what an NHWC 1x1 conv emits):
  q, k   [B, T, N, Hh, Dk]      v, r  [B, T, N, Hh, Dv]
  alpha  [B, T, Hh]             beta  [B, T, N, Hh]
  state  [B, Hh, Dk, Dv]
"""
from __future__ import annotations

import numpy as np

RULE_GATED_LINEAR = 0          # S <- a S + sum_i b_i k_i v_i^T               (literal BASELINE.json formula)
RULE_DELTA_PARALLEL = 1        # all tokens of a frame erase against the decayed state (A = 0)
RULE_DELTA_SEQUENTIAL = 2      # token-sequential delta rule inside the frame (default)

FLAG_NORMALIZE_QK = 1          # q,k <- x * rsqrt(sum x^2 + EPS_NORM)           
FLAG_GATE_LOGITS = 2           # alpha,beta given as logits; gate = sigmoid(x)   
EPS_NORM = 1e-12


def sigmoid(x):
    x = np.asarray(x)
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(x.dtype)


def l2_normalize(x):
    """x * rsqrt(sum(x^2) + EPS_NORM) over the last axis."""
    x = np.asarray(x)
    ss = np.sum(x.astype(np.float64) ** 2, axis=-1, keepdims=True)
    return (x / np.sqrt(ss + EPS_NORM)).astype(x.dtype)


def prologue(q, k, alpha, beta, flags):
    """Row a5: key/query normalisation and gate activation, selected by ``flags``."""
    if flags & FLAG_NORMALIZE_QK:
        q, k = l2_normalize(q), l2_normalize(k)
    if flags & FLAG_GATE_LOGITS:
        alpha, beta = sigmoid(alpha), sigmoid(beta)
    return q, k, alpha, beta


def lkva_read(q_t, S):
    """Row a1: R_t = Q_t . S_{t-1}.   q_t [N, Dk], S [Dk, Dv] -> [N, Dv]."""
    return q_t @ S


def gdr_write_sequential(S, k_t, v_t, alpha_t, beta_t, rule=RULE_DELTA_SEQUENTIAL):
    """Row a2, definitional token loop.  S [Dk,Dv], k_t [N,Dk], v_t [N,Dv], beta_t [N]."""
    S = alpha_t * S
    N = k_t.shape[0]
    if rule == RULE_GATED_LINEAR:
        for i in range(N):
            S = S + beta_t[i] * np.outer(k_t[i], v_t[i])
        return S
    if rule == RULE_DELTA_PARALLEL:
        E = v_t - k_t @ S
        return S + k_t.T @ (beta_t[:, None] * E)
    for i in range(N):
        e = v_t[i] - S.T @ k_t[i]
        S = S + beta_t[i] * np.outer(k_t[i], e)
    return S


def wy_factors(k_t, v_t, beta_t, rule=RULE_DELTA_SEQUENTIAL):
    """State-independent per-frame factors  Wt = T diag(b) K,  Ut = T diag(b) V  with
    T = (I + tril(diag(b) K K^T, -1))^{-1}  (SURVEY A.3).  U = Ut - Wt S'  then  S = S' + K^T U."""
    N = k_t.shape[0]
    bK = beta_t[:, None] * k_t
    bV = beta_t[:, None] * v_t
    if rule == RULE_GATED_LINEAR:
        return np.zeros_like(bK), bV
    if rule == RULE_DELTA_PARALLEL:
        return bK, bV
    A = np.tril(bK @ k_t.T, -1)
    Wt = np.empty_like(bK)
    Ut = np.empty_like(bV)
    for i in range(N):                      # forward substitution, row by row
        Wt[i] = bK[i] - A[i, :i] @ Wt[:i]
        Ut[i] = bV[i] - A[i, :i] @ Ut[:i]
    return Wt, Ut


def gdr_write_chunk(S, k_t, v_t, alpha_t, beta_t, rule=RULE_DELTA_SEQUENTIAL):
    """Row a2, frame (WY) form of SURVEY A.3 -- the form the HIP kernels use."""
    Wt, Ut = wy_factors(k_t, v_t, beta_t, rule)
    Sd = alpha_t * S
    U = Ut - Wt @ Sd
    return Sd + k_t.T @ U


def scan(q, k, v, alpha, beta, s0=None, rule=RULE_DELTA_SEQUENTIAL, flags=0, form="sequential",
         dtype=np.float64):
    """Row a3: for each frame read then write.  Returns (R [B,T,N,Hh,Dv], S_T).

    ``form`` = "sequential" (definition) or "chunk" (WY).  ``dtype`` is the arithmetic type."""
    q = np.asarray(q, dtype=dtype); k = np.asarray(k, dtype=dtype); v = np.asarray(v, dtype=dtype)
    alpha = np.asarray(alpha, dtype=dtype); beta = np.asarray(beta, dtype=dtype)
    q, k, alpha, beta = prologue(q, k, alpha, beta, flags)
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    S = np.zeros((B, Hh, Dk, Dv), dtype=dtype) if s0 is None else np.array(s0, dtype=dtype)
    R = np.empty((B, T, N, Hh, Dv), dtype=dtype)
    write = gdr_write_sequential if form == "sequential" else gdr_write_chunk
    for b in range(B):
        for h in range(Hh):
            Sbh = S[b, h]
            for t in range(T):
                R[b, t, :, h, :] = lkva_read(q[b, t, :, h, :], Sbh)
                Sbh = write(Sbh, k[b, t, :, h, :], v[b, t, :, h, :], alpha[b, t, h], beta[b, t, :, h], rule)
            S[b, h] = Sbh
    return R, S


NORMALIZER_EPS = 1e-6          # SPEC-v0 default of the `normalizer` flag (SURVEY A.1 names the flag, not the value)


def scan_normalizer(q, k, v, alpha, beta, s0=None, z0=None, rule=RULE_DELTA_SEQUENTIAL, flags=0, eps=NORMALIZER_EPS, form="sequential",
                    dtype=np.float64):
    """SURVEY A.1 with the `normalizer` flag: z [B,Hh,Dk] is carried with "the same recurrence on v == 1" and the read-out is
    R_t / (|Q_t z_{t-1}| + eps).  Restated literally: one more value channel that is 1 for every token rides through ``scan``; its
    state column is z, its read-out is Q_t z_{t-1}.  (The scalar C oracle carries z explicitly: the two check each other.)
    Returns (R [B,T,N,Hh,Dv], S_T, z_T)."""
    v = np.asarray(v, dtype=dtype)
    B, T, N, Hh, Dv = v.shape
    Dk = np.asarray(q).shape[-1]
    v1 = np.concatenate([v, np.ones((B, T, N, Hh, 1), dtype=dtype)], axis=-1)
    s_aug = None
    if s0 is not None or z0 is not None:
        s_aug = np.zeros((B, Hh, Dk, Dv + 1), dtype=dtype)
        if s0 is not None:
            s_aug[..., :Dv] = s0
        if z0 is not None:
            s_aug[..., Dv] = z0
    R1, S1 = scan(q, k, v1, alpha, beta, s_aug, rule, flags, form, dtype)
    return R1[..., :Dv] / (np.abs(R1[..., Dv:]) + eps), S1[..., :Dv].copy(), S1[..., Dv].copy()


def mask_cell_mean(mask_fg, h, w):
    """adaptive_avg_pool2d of a foreground indicator [F, H, W] (bool / 0-1) to h x w cells: rows [floor(i H / h), ceil((i + 1) H / h)),
    columns likewise -> [F, h*w] float64 (the pooled first-frame / fed-back mask of the module's mask embedding)."""
    m = np.asarray(mask_fg, dtype=np.float64)
    F_, H, W = m.shape
    out = np.empty((F_, h, w), dtype=np.float64)
    for i in range(h):
        y0, y1 = (i * H) // h, -((-(i + 1) * H) // h)
        for j in range(w):
            x0, x1 = (j * W) // w, -((-(j + 1) * W) // w)
            out[:, i, j] = m[:, y0:y1, x0:x1].mean(axis=(1, 2))
    return out.reshape(F_, h * w)


# ----------------------------------------------------------------------------------------------- KPFF
KPFF_SCALES = (1, 2, 4)


def multiscale_pool(G, h, w):
    """G [BT, N=h*w, C] -> mean over s in KPFF_SCALES of (s x s cell average broadcast back).
    Cells are anchored at (0,0); edge cells average only the tokens that exist."""
    BT, N, C = G.shape
    g = G.reshape(BT, h, w, C).astype(np.float64)
    acc = np.zeros_like(g)
    for s in KPFF_SCALES:
        if s == 1:
            acc += g
            continue
        out = np.empty_like(g)
        for y0 in range(0, h, s):
            for x0 in range(0, w, s):
                cell = g[:, y0:y0 + s, x0:x0 + s, :]
                out[:, y0:y0 + s, x0:x0 + s, :] = cell.mean(axis=(1, 2), keepdims=True)
        acc += out
    return (acc / len(KPFF_SCALES)).reshape(BT, N, C).astype(G.dtype)


def kpff(L, G, P, Wa, ba, Wl, Wg, h, w, dtype=np.float64):
    """Row a4.  L [BT,N,Ck] local key feature, G [BT,N,Cv] global (read-out) feature,
    P [BT,N,Cp] pixel feature.  Wa [2Cp, Cp+Ck+Cv], ba [2Cp], Wl [Cp,Ck], Wg [Cp,Cv].
      Gms = multiscale_pool(G);  g = sigmoid([P;L;Gms] Wa^T + ba) -> (g_l, g_g)
      F = P + g_l * (L Wl^T) + g_g * (Gms Wg^T)"""
    L = np.asarray(L, dtype=dtype); G = np.asarray(G, dtype=dtype); P = np.asarray(P, dtype=dtype)
    Wa = np.asarray(Wa, dtype=dtype); ba = np.asarray(ba, dtype=dtype)
    Wl = np.asarray(Wl, dtype=dtype); Wg = np.asarray(Wg, dtype=dtype)
    Cp = P.shape[-1]
    Gms = multiscale_pool(G, h, w)
    X = np.concatenate([P, L, Gms], axis=-1)
    g = sigmoid(X @ Wa.T + ba)
    return P + g[..., :Cp] * (L @ Wl.T) + g[..., Cp:] * (Gms @ Wg.T)


# --------------------------------------------------------------------------------------- argmax + Dice
def argmax_mask(logits):
    """Row a6.  logits [BT, ncls, H, W] -> uint8 mask; ties -> lowest class index (np.argmax rule)."""
    return np.argmax(np.asarray(logits), axis=1).astype(np.uint8)


def dice_counts(mask, target, ncls):
    """Integer counts per (image, class): |A n B|, |A|, |B| -- exact, so the GPU must match bit for bit."""
    BT = mask.shape[0]
    inter = np.zeros((BT, ncls), dtype=np.int64)
    psum = np.zeros((BT, ncls), dtype=np.int64)
    tsum = np.zeros((BT, ncls), dtype=np.int64)
    m = mask.reshape(BT, -1); t = np.asarray(target).reshape(BT, -1)
    for c in range(ncls):
        a = (m == c); b = (t == c)
        inter[:, c] = (a & b).sum(1); psum[:, c] = a.sum(1); tsum[:, c] = b.sum(1)
    return inter, psum, tsum


def dice_from_counts(inter, psum, tsum, eps=1e-6):
    """Dice_c = (2|AnB| + eps) / (|A| + |B| + eps): both empty -> 1, disjoint -> ~0, identical -> 1."""
    inter = np.asarray(inter, dtype=np.float64)
    return (2.0 * inter + eps) / (np.asarray(psum, np.float64) + np.asarray(tsum, np.float64) + eps)


# ------------------------------------------------------------------------------------------ bf16 helper
def to_bf16_f32(x):
    """Round an fp32 array to bfloat16 (round-to-nearest-even) and return it widened back to fp32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32)
