// gdr_scan_bwd.hip -- SURVEY.md §8 row a7: backward of the LKVA read + GDR write (gdkvm_scan_fwd).
//
// With the forward per frame   R = Qn S,  X = Wt S,  U = Ut - a X,  S' = a S + Kn^T U   (Wt = T b Kn, Ut = T b V):
//
//  gdr_bwd_scan_kernel   reverse-time recurrence on dS, one workgroup per (clip, head, 16-column slice) like the
//                        forward scan -- it has the same shape:
//                            dS = a (dS' - Wt^T (Kn dS')) + Qn^T dR
//                        (two dependent products per frame + a state-independent term), exact fp32 MFMA, with the
//                        accumulator-as-operand trick of the forward kernel.  Writes dS' of every frame (ds_hist).
//  gdr_bwd_frame_kernel  everything else is frame-local given S (s_hist, saved by the forward) and dS' (ds_hist):
//                            dU = Kn dS'   dKn = U dS'^T   dWt = -a dU S^T   dQn = dR S^T   da = <S,dS'> - <X,dU>
//                            Z = T^T [dWt | dU]  (back substitution)   db, dV, dKn through diag(b) and A = tril(b Kn Kn^T)
//                        then the L2-normalisation and sigmoid derivatives.  One workgroup per (clip, frame, head),
//                        fully parallel over frames; fp32 VALU over LDS tiles (first version: written for clarity and
//                        exactness, not yet for MFMA throughput -- the backward is ~17 MFLOP per frame).
//
// Derivation and its numpy restatement: oracle/bwd_ref.py (checked against autograd in tests/test_oracle_kat.py).
#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

struct BwdScanArgs {
    const void* q; const float* qinv; const float* knT; const float* wt; const float* alpha;
    const void* d_r; const float* ds_out; float* ds_hist; float* ds_in;
    int T, Hh, N, Dv, flags, nb;
};

template <int IO>
__global__ __launch_bounds__(256) void gdr_bwd_scan_kernel(BwdScanArgs a)
{
    __shared__ __attribute__((aligned(16))) f32x4 s_D[4 * 64];                       // dS image (B operand)
    __shared__ __attribute__((aligned(16))) f32x4 s_V[(GDKVM_MAX_N / 16) * 64];      // V' = -a Kn dS tiles

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsl = a.Dv / 16, N = a.N, Hh = a.Hh, Dv = a.Dv, T = a.T, nb = a.nb, NP = 16 * a.nb;
    const int bh = blockIdx.x / nsl, sl = blockIdx.x % nsl;
    const int b = bh / Hh, h = bh % Hh;

    f32x4 dacc = {0.f, 0.f, 0.f, 0.f};                    // rows 16w+4g+r of dS, column 16*sl + li
    if (a.ds_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dacc[r] = a.ds_out[((size_t)bh * GDKVM_DK + 16 * w + 4 * g + r) * Dv + 16 * sl + li];
    }
    s_D[w * 64 + lane] = dacc;
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const size_t bt = (size_t)b * T + t, fh = bt * Hh + h;
        float alpha = a.alpha[fh];
        if (a.flags & GDKVM_FLAG_GATE_LOGITS) alpha = 1.0f / (1.0f + expf(-alpha));
        const float* knT = a.knT + fh * GDKVM_DK * NP;
        const float* wt = a.wt + fh * NP * GDKVM_DK;
        const float* qinv = a.qinv + fh * NP;
        {   // dS' of this frame (gradient w.r.t. the state after frame t)
            float* hp = a.ds_hist + (fh * GDKVM_DK + 16 * w + 4 * g) * Dv + 16 * sl + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) hp[(size_t)r * Dv] = dacc[r];
        }
        f32x4 dreg[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) dreg[m] = s_D[m * 64 + lane];
        for (int tt = w; tt < nb; tt += 4) {              // Y = Kn dS' for token tile tt;  V' = -a Y
            f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    y = mfma4(knT[(size_t)(16 * m + 4 * g + r) * NP + 16 * tt + li], dreg[m][r], y);
            s_V[tt * 64 + lane] = y * (-alpha);
        }
        __syncthreads();
        f32x4 acc = dacc * alpha;
        for (int tt = 0; tt < nb; ++tt) {
            const f32x4 vb = s_V[tt * 64 + lane];
            f32x4 rb;                                      // dR[token 16tt+4g+r][col]  (B operand of Qn^T dR)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * tt + 4 * g + r;
                rb[r] = n < N ? load1<IO>(a.d_r, ((bt * N + n) * Hh + h) * Dv + 16 * sl + li) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * tt + 4 * g + r;         // A operands: column 16w+li of Wt / Qn, token n
                acc = mfma4(wt[(size_t)n * GDKVM_DK + 16 * w + li], vb[r], acc);
                const float qn = n < N ? load1<IO>(a.q, ((bt * N + n) * Hh + h) * GDKVM_DK + 16 * w + li) * qinv[n] : 0.f;
                acc = mfma4(qn, rb[r], acc);
            }
        }
        dacc = acc;
        s_D[w * 64 + lane] = dacc;
        __syncthreads();
    }
    if (a.ds_in) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.ds_in[((size_t)bh * GDKVM_DK + 16 * w + 4 * g + r) * Dv + 16 * sl + li] = dacc[r];
    }
}

// -------------------------------------------------------------------------------------------------------------
struct BwdFrameArgs {
    const void* q; const void* k; const void* v; const float* alpha; const float* beta;
    const float* qinv; const float* knT; const float* wt; const float* ut;
    const float* s_hist; const float* ds_hist; const void* d_r;
    void* d_q; void* d_k; void* d_v; float* d_alpha; float* d_beta;
    int T, Hh, N, Dv, rule, flags;
};

constexpr int BF_LD = 65;                                  // padded leading dimension of the 64x64 LDS tiles
constexpr int BF_TILE = 64 * BF_LD;

__device__ __forceinline__ float quad_sum(float x)         // sum over the 4 adjacent lanes that share a row
{
    x += __shfl_xor(x, 1);
    x += __shfl_xor(x, 2);
    return x;
}

template <int IO>
__global__ __launch_bounds__(256) void gdr_bwd_frame_kernel(BwdFrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Kn = sm;                  float* Wt = sm + BF_TILE;        float* Qn = sm + 2 * BF_TILE;
    float* bA = sm + 3 * BF_TILE;    float* bB = sm + 4 * BF_TILE;    float* bC = sm + 5 * BF_TILE;
    float* bD = sm + 6 * BF_TILE;    float* bE = sm + 7 * BF_TILE;    float* bF = sm + 8 * BF_TILE;
    float* s_beta = sm + 9 * BF_TILE; float* s_kinv = s_beta + 64;    float* s_qinv = s_kinv + 64;
    float* s_red = s_qinv + 64;                                       // 8 floats

    const int tid = threadIdx.x;
    const int row = tid >> 2, c0 = (tid & 3) * 16;         // this thread's 1 x 16 strip of every 64 x 64 result
    const int fh = blockIdx.x, h = fh % a.Hh;
    const size_t bt = fh / a.Hh;
    const int N = a.N, Hh = a.Hh, Dv = a.Dv, NP = 64, NB = 4;
    const bool seq = a.rule == GDKVM_RULE_DELTA_SEQUENTIAL, lin = a.rule == GDKVM_RULE_GATED_LINEAR;
    const bool normalize = a.flags & GDKVM_FLAG_NORMALIZE_QK, logits = a.flags & GDKVM_FLAG_GATE_LOGITS;
    float alpha = a.alpha[fh];
    if (logits) alpha = 1.0f / (1.0f + expf(-alpha));

    // ---- stage the frame's factors:  Kn (from Kn^T), Wt, Qn = q * qinv, gates, key norms ---------------------
    if (tid < 64) {
        float bta = 0.f, kinv = 0.f;
        if (tid < N) {
            bta = a.beta[(bt * N + tid) * Hh + h];
            if (logits) bta = 1.0f / (1.0f + expf(-bta));
            kinv = 1.f;
            if (normalize) {
                float ss = 0.f;
                for (int c = 0; c < GDKVM_DK; ++c) {
                    const float x = load1<IO>(a.k, ((bt * N + tid) * Hh + h) * GDKVM_DK + c);
                    ss += x * x;
                }
                kinv = 1.0f / sqrtf(ss + GDKVM_EPS_NORM);
            }
        }
        s_beta[tid] = bta; s_kinv[tid] = kinv; s_qinv[tid] = a.qinv[(size_t)fh * NP + tid];
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63;              // token i, channel d
        Wt[i * BF_LD + d] = a.wt[((size_t)fh * NP + i) * GDKVM_DK + d];
        Kn[d * BF_LD + i] = a.knT[((size_t)fh * GDKVM_DK + i) * NP + d];       // idx = (channel i, token d) here
        Qn[i * BF_LD + d] = i < N ? load1<IO>(a.q, ((bt * N + i) * Hh + h) * GDKVM_DK + d) * s_qinv[i] : 0.f;
    }
    __syncthreads();

    float dKn[16], dWt[16], dQn[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) dKn[j] = dWt[j] = dQn[j] = 0.f;
    float da = 0.f;
    const float* s_prev = a.s_hist + (size_t)fh * GDKVM_DK * Dv;
    const float* ds_now = a.ds_hist + (size_t)fh * GDKVM_DK * Dv;
    const f32x4* ut_img = reinterpret_cast<const f32x4*>(a.ut + (size_t)fh * NP * Dv);
    const int nchunk = (Dv + 63) / 64;

    // load a 64 x CW chunk of [S ; dS' ; dR ; Ut] into bA / bB / bC / bD  (rows: channel d or token i; cols: chunk)
    auto load_chunk = [&](int cb, int CW, bool want_dr) {
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int i = idx >> 6, c = idx & 63;
            const bool in = c < CW;
            bA[i * BF_LD + c] = in ? s_prev[(size_t)i * Dv + cb + c] : 0.f;
            bB[i * BF_LD + c] = in ? ds_now[(size_t)i * Dv + cb + c] : 0.f;
            if (want_dr) bC[i * BF_LD + c] = (in && i < N) ? load1<IO>(a.d_r, ((bt * N + i) * Hh + h) * Dv + cb + c) : 0.f;
            float u = 0.f;
            if (in) {                                       // de-image: token i = 16I + 4g + r, column = 16ct + li
                const int ct = (cb + c) >> 4, lane = ((i >> 2) & 3) * 16 + ((cb + c) & 15);
                u = ut_img[((size_t)ct * NB + (i >> 4)) * 64 + lane][i & 3];
            }
            bD[i * BF_LD + c] = u;
        }
    };
    // X = Wt S, U = Ut - a X (-> bD in place), dU = Kn dS' (-> bE); also the da partial of this thread's strip
    auto xu_du = [&](bool want_da) {
        float x[16], du[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = du[j] = 0.f;
        for (int d = 0; d < 64; ++d) {
            const float wv = Wt[row * BF_LD + d], kv = Kn[row * BF_LD + d];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                x[j] += wv * bA[d * BF_LD + c0 + j];
                du[j] += kv * bB[d * BF_LD + c0 + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (want_da) da += bA[row * BF_LD + c0 + j] * bB[row * BF_LD + c0 + j] - x[j] * du[j];
            bD[row * BF_LD + c0 + j] -= alpha * x[j];
            bE[row * BF_LD + c0 + j] = du[j];
        }
    };

    // ---- phase 1: contractions over Dv, 64 columns at a time --------------------------------------------------
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cb = 64 * ch, CW = min(64, Dv - cb);
        load_chunk(cb, CW, true);
        __syncthreads();
        xu_du(true);
        __syncthreads();
        for (int c = 0; c < CW; ++c) {
            const float u = bD[row * BF_LD + c], du = bE[row * BF_LD + c], dr = bC[row * BF_LD + c];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float sp = bA[(c0 + j) * BF_LD + c];
                dKn[j] += u * bB[(c0 + j) * BF_LD + c];
                dWt[j] -= alpha * du * sp;
                dQn[j] += dr * sp;
            }
        }
        __syncthreads();
    }
    {   // d alpha
        da = quad_sum(da);
        for (int o = 4; o < 64; o <<= 1) da += __shfl_xor(da, o);
        if ((tid & 63) == 0) s_red[tid >> 6] = da;
        __syncthreads();
        if (tid == 0) {
            const float d = s_red[0] + s_red[1] + s_red[2] + s_red[3];
            a.d_alpha[fh] = logits ? d * alpha * (1.f - alpha) : d;
        }
    }

    // ---- phase 2: through T = (I + tril(b Kn Kn^T, -1))^-1 ---------------------------------------------------
    float dA[16], dbeta = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) dA[j] = 0.f;
    if (seq) {                                              // Gram matrix -> bF
        float gm[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) gm[j] = 0.f;
        for (int d = 0; d < 64; ++d) {
            const float kv = Kn[row * BF_LD + d];
#pragma unroll
            for (int j = 0; j < 16; ++j) gm[j] += kv * Kn[(c0 + j) * BF_LD + d];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) bF[row * BF_LD + c0 + j] = gm[j];
    }
    __syncthreads();
    // column chunks of dY = [dWt | dU]:  cc = 0 is the key block (Y = Wt, X0 = Kn), cc >= 1 the value chunks
    for (int cc = lin ? 1 : 0; cc <= nchunk; ++cc) {
        const int cb = 64 * (cc - 1), CW = cc == 0 ? 64 : min(64, Dv - cb);
        if (cc == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) bE[row * BF_LD + c0 + j] = dWt[j];          // dY chunk -> bE
        } else {
            load_chunk(cb, CW, false);
            __syncthreads();
            xu_du(false);                                   // bE = dU chunk; bD = U chunk (need Ut: rebuilt below)
            __syncthreads();
            for (int idx = tid; idx < 64 * 64; idx += 256) {                          // Y chunk = Ut -> bD, X0 chunk = V -> bC
                const int i = idx >> 6, c = idx & 63;
                float u = 0.f, vv = 0.f;
                if (c < CW) {
                    const int ct = (cb + c) >> 4, lane = ((i >> 2) & 3) * 16 + ((cb + c) & 15);
                    u = ut_img[((size_t)ct * NB + (i >> 4)) * 64 + lane][i & 3];
                    if (i < N) vv = load1<IO>(a.v, ((bt * N + i) * Hh + h) * Dv + cb + c);
                }
                bD[i * BF_LD + c] = u;
                bC[i * BF_LD + c] = vv;
            }
        }
        __syncthreads();
        if (seq) {
            // Z = T^T dY:  z_i = dy_i - sum_{j>i} A_ji z_j, i descending; column = tid>>2, the j range split over 4 lanes
            const int c = tid >> 2, part = tid & 3;
            for (int i = 62; i >= 0; --i) {
                float sacc = 0.f;
                for (int j = i + 1 + part; j < 64; j += 4) sacc += s_beta[j] * bF[j * BF_LD + i] * bE[j * BF_LD + c];
                sacc = quad_sum(sacc);
                if (part == 0) bE[i * BF_LD + c] -= sacc;
                // only the 4 adjacent lanes of this column read the row just written: order it inside the wave
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            __syncthreads();                                // other waves own the other columns of Z
        }
        // now bE = Z chunk.  d beta, d(X0) and dA
        {
            const float bi = s_beta[row];
            const float* X0 = cc == 0 ? Kn : bC;
            const float* Y = cc == 0 ? Wt : bD;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float z = bE[row * BF_LD + c0 + j];
                dbeta += z * X0[row * BF_LD + c0 + j];
                if (cc == 0) dKn[j] += bi * z;
                else if (c0 + j < CW && row < N) store1<IO>(a.d_v, ((bt * N + row) * Hh + h) * Dv + cb + c0 + j, bi * z);
            }
            if (seq)
                for (int c = 0; c < CW; ++c) {
                    const float z = bE[row * BF_LD + c];
#pragma unroll
                    for (int j = 0; j < 16; ++j) dA[j] -= z * Y[(c0 + j) * BF_LD + c];
                }
        }
        __syncthreads();
    }
    if (lin) {                                              // Wt == 0: the key block contributes nothing through T
        // (dKn keeps only U dS'^T)
    }
    if (seq) {
        const float bi = s_beta[row];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float m = (c0 + j < row) ? dA[j] : 0.f;   // strictly lower: column < row
            dbeta += m * bF[row * BF_LD + c0 + j];
            bE[row * BF_LD + c0 + j] = bi * m;              // M = diag(b) dA
        }
        __syncthreads();
        for (int j2 = 0; j2 < 64; ++j2) {
            const float m1 = bE[row * BF_LD + j2], m2 = bE[j2 * BF_LD + row];
#pragma unroll
            for (int j = 0; j < 16; ++j) dKn[j] += (m1 + m2) * Kn[j2 * BF_LD + c0 + j];
        }
    }
    // ---- gates, L2 normalisation, stores ---------------------------------------------------------------------
    dbeta = quad_sum(dbeta);
    if ((tid & 3) == 0 && row < N) {
        const float bi = s_beta[row];
        a.d_beta[(bt * N + row) * Hh + h] = logits ? dbeta * bi * (1.f - bi) : dbeta;
    }
    float dotk = 0.f, dotq = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        dotk += Kn[row * BF_LD + c0 + j] * dKn[j];
        dotq += Qn[row * BF_LD + c0 + j] * dQn[j];
    }
    dotk = quad_sum(dotk); dotq = quad_sum(dotq);
    if (row < N) {
        const float ki = s_kinv[row], qi = s_qinv[row];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float gk = dKn[j], gq = dQn[j];
            if (normalize) {
                gk = ki * (gk - Kn[row * BF_LD + c0 + j] * dotk);
                gq = qi * (gq - Qn[row * BF_LD + c0 + j] * dotq);
            }
            store1<IO>(a.d_k, ((bt * N + row) * Hh + h) * GDKVM_DK + c0 + j, gk);
            store1<IO>(a.d_q, ((bt * N + row) * Hh + h) * GDKVM_DK + c0 + j, gq);
        }
    }
}

}  // namespace

extern "C" size_t gdkvm_scan_bwd_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv)
{
    if (B <= 0 || T <= 0 || Hh <= 0 || N <= 0 || Dk <= 0 || Dv <= 0) return 16;
    return (size_t)B * T * Hh * Dk * Dv * sizeof(float) + 16;
}

extern "C" int gdkvm_scan_bwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                              const float* s_hist, const void* fwd_workspace, size_t fwd_workspace_bytes,
                              const void* d_r, const float* d_s_out,
                              void* d_q, void* d_k, void* d_v, float* d_alpha, float* d_beta, float* d_s_in,
                              void* bwd_workspace, size_t bwd_workspace_bytes,
                              int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    if (int rc = check_common("scan_bwd", B, T, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (rule < 0 || rule > 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_bwd: rule=%d", rule);
    if (B == 0) return GDKVM_OK;
    if (T == 0 || N == 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_bwd: T and N must be positive");
    if (N > 64) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_bwd: N=%d > 64 tokens per frame is not supported yet", N);
    if (int rc = check_ptrs("scan_bwd", {q, k, v, alpha, beta, s_hist, fwd_workspace, d_r, d_q, d_k, d_v, d_alpha, d_beta, bwd_workspace},
                            {d_s_out, d_s_in})) return rc;
    WsView ws;
    if (int rc = carve("scan_bwd", const_cast<void*>(fwd_workspace), fwd_workspace_bytes, B, T, Hh, N, Dk, Dv, &ws)) return rc;
    if (bwd_workspace_bytes < gdkvm_scan_bwd_workspace_bytes(B, T, Hh, N, Dk, Dv) - 16)
        return gdkvm_fail(GDKVM_ERR_WORKSPACE, "scan_bwd: backward workspace %zu < %zu bytes", bwd_workspace_bytes,
                          gdkvm_scan_bwd_workspace_bytes(B, T, Hh, N, Dk, Dv));
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* ds_hist = static_cast<float*>(bwd_workspace);

    BwdScanArgs sa{q, ws.qinv, ws.knT, ws.wt, alpha, d_r, d_s_out, ds_hist, d_s_in, T, Hh, N, Dv, flags, ws.nb};
    const dim3 grid((unsigned)(B * Hh * (Dv / 16)));
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gdr_bwd_scan_kernel<GDKVM_F32>), grid, dim3(256), 0, st, sa);
    else hipLaunchKernelGGL((gdr_bwd_scan_kernel<GDKVM_BF16>), grid, dim3(256), 0, st, sa);
    GDKVM_LAUNCH_CHECK("gdr_bwd_scan_kernel");

    BwdFrameArgs fa{q, k, v, alpha, beta, ws.qinv, ws.knT, ws.wt, ws.ut, s_hist, ds_hist, d_r,
                    d_q, d_k, d_v, d_alpha, d_beta, T, Hh, N, Dv, rule, flags};
    const size_t lds = (size_t)(9 * BF_TILE + 3 * 64 + 8) * sizeof(float);
    const void* fn = io_dtype == GDKVM_F32 ? reinterpret_cast<const void*>(gdr_bwd_frame_kernel<GDKVM_F32>)
                                           : reinterpret_cast<const void*>(gdr_bwd_frame_kernel<GDKVM_BF16>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "scan_bwd: LDS attribute: %s", hipGetErrorString(e));
    const dim3 fgrid((unsigned)(B * T * Hh));
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gdr_bwd_frame_kernel<GDKVM_F32>), fgrid, dim3(256), lds, st, fa);
    else hipLaunchKernelGGL((gdr_bwd_frame_kernel<GDKVM_BF16>), fgrid, dim3(256), lds, st, fa);
    GDKVM_LAUNCH_CHECK("gdr_bwd_frame_kernel");
    return GDKVM_OK;
}
