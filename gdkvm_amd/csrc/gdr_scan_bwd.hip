// gdr_scan_bwd.hip -- SURVEY.md §8 row a7: backward of the LKVA read + GDR write (gdkvm_scan_fwd).
//
// With the forward per frame   R = Qn S,  X = Wt S,  U = Ut - a X,  S' = a S + Kn^T U   (Wt = T b Kn, Ut = T b V):
//
//  reverse scan          the reverse-time recurrence on dS has the forward's affine shape,
//                            dS = a (dS' - Wt^T (Kn dS')) + Qn^T dR  =  a P^T dS' + Gb ,
//                        and runs on the forward's serial kernel (gdr_scan.hip: gdr_affine_scan_kernel in reverse mode, P^T
//                        images from the training-mode fold, Gb = Qn^T dR from gdr_bwd_g_kernel).  Writes dS' of every
//                        frame (ds_hist).
//  gdr_bwd_frame_kernel  everything else is frame-local given S (s_hist, saved by the forward) and dS' (ds_hist):
//                            dU = Kn dS'   dKn = U dS'^T   dWt = -a dU S^T   dQn = dR S^T   da = <S,dS'> - <X,dU>
//                            Z = T^T [dWt | dU]  (back substitution)   db, dV, dKn through diag(b) and A = tril(b Kn Kn^T)
//                        then the L2-normalisation and sigmoid derivatives.  One workgroup per (clip, frame, head),
//                        fully parallel over frames; the eight 64x64x64 contractions run on exact-fp32 MFMA with both
//                        operands read straight from padded LDS tiles, the back substitution on VALU.
//
// Derivation and its numpy restatement: oracle/bwd_ref.py (checked against autograd in tests/test_oracle_kat.py).
#include <type_traits>

#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

int gdr_launch_reverse_scan(const WsView& ws, const float* alpha, const void* d_r, const float* ds_out, float* ds_hist,
                            float* ds_in, float* gb, int B, int T, int Hh, int N, int Dv, int io_dtype, int flags, hipStream_t st);

namespace {

struct BwdFrameArgs {
    const void* q; const void* k; const void* v; const float* alpha; const float* beta;
    const float* qinv; const float* kn; const float* wt; const float* ut; const float* tii;
    const float* s_hist; const float* ds_hist; const void* d_r;
    void* d_q; void* d_k; void* d_v; float* d_alpha; float* d_beta;
    int T, Hh, N, Dv, rule, flags;
};

constexpr int BF_LD = 68;                                  // leading dimension of the 64 x 64 LDS tiles: rows stay 16-byte
constexpr int BF_TILE = 64 * BF_LD;                        // aligned (b128 operand reads) and shift by 4 banks per row

// acc[nt] += A[16mt .. +15, 0..63] * B over one 64-deep contraction, A row-major with k contiguous.
//   BT = false:  B given as [k][col]   (one ds_read_b32 per MFMA)
//   BT = true :  B given as [col][k]   (ds_read_b128, like A)
// k is visited in the order 16kb + 4g + r on BOTH operands (the accumulator-as-operand permutation of the forward).
template <bool BT>
__device__ __forceinline__ void tile_gemm(f32x4 (&acc)[4], const float* A, const float* B, int mt, int li, int g)
{
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + (16 * mt + li) * BF_LD + 16 * kb + 4 * g);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if constexpr (BT) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(B + (16 * nt + li) * BF_LD + 16 * kb + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nt] = mfma4(a4[r], b4[r], acc[nt]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nt] = mfma4(a4[r], B[(16 * kb + 4 * g + r) * BF_LD + 16 * nt + li], acc[nt]);
            }
        }
    }
}

template <int I, class F>
__device__ __forceinline__ void static_for_desc(F&& f)     // f(I), f(I-1), ..., f(0) with compile-time indices
{
    f(std::integral_constant<int, I>{});
    if constexpr (I > 0) static_for_desc<I - 1>(f);
}

__device__ __forceinline__ float row16_sum(float x)        // sum over the 16 lanes (li) that hold one accumulator row
{
    x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
    return x;
}

// One workgroup (4 waves) per (clip, frame, head).  Wave w owns the 16 x 64 row band w of every 64 x 64 result as four
// accumulator tiles: element (reg r of tile nt, lane (li, g)) = [row 16w + 4g + r][col 16nt + li].
template <int IO>
__global__ __launch_bounds__(256) void gdr_bwd_frame_kernel(BwdFrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Kn = sm;                  float* Wt = sm + BF_TILE;
    float* b0 = sm + 2 * BF_TILE;    float* b1 = sm + 3 * BF_TILE;    float* b2 = sm + 4 * BF_TILE;
    float* b3 = sm + 5 * BF_TILE;    float* b4 = sm + 6 * BF_TILE;    float* b5 = sm + 7 * BF_TILE;
    float* s_beta = sm + 8 * BF_TILE; float* s_kinv = s_beta + 64;    float* s_qinv = s_kinv + 64;
    float* s_red = s_qinv + 64;                                       // 8 floats
    float* s_tii = s_red + 8;                                         // [4][16][16] diagonal-block inverses T_II

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = blockIdx.x, h = fh % a.Hh;
    const size_t bt = fh / a.Hh;
    const int N = a.N, Hh = a.Hh, Dv = a.Dv, NP = 64, NB = 4;
    const bool seq = a.rule == GDKVM_RULE_DELTA_SEQUENTIAL, lin = a.rule == GDKVM_RULE_GATED_LINEAR;
    const bool normalize = a.flags & GDKVM_FLAG_NORMALIZE_QK, logits = a.flags & GDKVM_FLAG_GATE_LOGITS;
    float alpha = a.alpha[fh];
    if (logits) alpha = 1.0f / (1.0f + expf(-alpha));
    auto at = [&](int nt, int r) { return (16 * w + 4 * g + r) * BF_LD + 16 * nt + li; };   // LDS offset of an accumulator element

    // ---- stage the frame's factors:  Kn (from Kn^T), Wt, gates, key norms ---------------------------------
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
    for (int idx = tid; idx < NB * 256; idx += 256) s_tii[idx] = a.tii[(size_t)fh * NB * 256 + idx];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                           // Wt and Kn (the training prep's natural-layout copy) rows
        const int e = tid + 256 * u, i = e >> 4, d = (e & 15) * 4;
        *reinterpret_cast<f32x4*>(Wt + i * BF_LD + d) = *reinterpret_cast<const f32x4*>(a.wt + ((size_t)fh * NP + i) * GDKVM_DK + d);
        *reinterpret_cast<f32x4*>(Kn + i * BF_LD + d) = *reinterpret_cast<const f32x4*>(a.kn + ((size_t)fh * NP + i) * GDKVM_DK + d);
    }
    __syncthreads();

    f32x4 dKn[4], dWt[4], dQn[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dKn[nt] = dWt[nt] = dQn[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float da = 0.f;
    const float* s_prev = a.s_hist + (size_t)fh * GDKVM_DK * Dv;
    const float* ds_now = a.ds_hist + (size_t)fh * GDKVM_DK * Dv;
    const f32x4* ut_img = reinterpret_cast<const f32x4*>(a.ut + (size_t)fh * NP * Dv);
    const int nchunk = (Dv + 63) / 64;

    // Staging helpers: every thread issues all of its 16-byte loads back to back (a `for idx += 256` loop of scalar loads
    // serialises on memory latency: 16 round trips per call).
    auto load4io = [&](const void* base, size_t off) { return load4<IO>(base, off); };
    // Ut chunk (column tiles cb/16 .. ) -> dst, read in image order: one float4 = 4 tokens (r) of one column
    auto load_ut = [&](float* dst, int cb, int CW) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;                      // (ct_local, I, lane)
            const int ln = e & 63, I = (e >> 6) & 3, ctl = e >> 8;
            const bool in = 16 * ctl < CW;
            const f32x4 v4 = ut_img[((size_t)((cb >> 4) + (in ? ctl : 0)) * NB + I) * 64 + ln];
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(16 * I + 4 * (ln >> 4) + r) * BF_LD + 16 * ctl + (ln & 15)] = in ? v4[r] : 0.f;
        }
    };
    // rows x 64-column chunk of a row-major fp32 matrix [64][Dv] -> dst
    auto load_rows_f32 = [&](float* dst, const float* src, int cb, int CW) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, i = e >> 4, c = (e & 15) * 4;
            const bool in = c < CW;
            f32x4 v4 = *reinterpret_cast<const f32x4*>(src + (size_t)i * Dv + cb + (in ? c : 0));
            if (!in) v4 = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(dst + i * BF_LD + c) = v4;
        }
    };
    // token rows x 64-column chunk of an I/O tensor [N][Hh][C] (C = Dv or Dk) -> dst, optionally scaled per row
    auto load_rows_io = [&](float* dst, const void* src, int C, int cb, int CW, const float* rowscale) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, i = e >> 4, c = (e & 15) * 4;
            const bool in = c < CW && i < N;
            f32x4 v4 = load4io(src, ((bt * N + min(i, N - 1)) * Hh + h) * C + cb + (c < CW ? c : 0));
            const float sc = in ? (rowscale ? rowscale[i] : 1.f) : 0.f;
            *reinterpret_cast<f32x4*>(dst + i * BF_LD + c) = v4 * sc;
        }
    };
    // 64 x CW chunk of S -> b0, dS' -> b1, (dR -> b2), Ut -> b3   (rows: channel d or token i)
    auto load_chunk = [&](int cb, int CW, bool want_dr) {
        load_rows_f32(b0, s_prev, cb, CW);
        load_rows_f32(b1, ds_now, cb, CW);
        if (want_dr) load_rows_io(b2, a.d_r, Dv, cb, CW, nullptr);
        load_ut(b3, cb, CW);
    };
    // X = Wt S (registers), dU = Kn dS' -> b4, U = Ut - a X -> b3 in place; optionally the da partial
    auto xu_du = [&](bool want_da) {
        f32x4 x[4], du[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) x[nt] = du[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_gemm<false>(x, Wt, b0, w, li, g);
        tile_gemm<false>(du, Kn, b1, w, li, g);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = at(nt, r);
                if (want_da) da += b0[o] * b1[o] - x[nt][r] * du[nt][r];
                b3[o] -= alpha * x[nt][r];
                b4[o] = du[nt][r];
            }
    };

    // ---- phase 1: contractions over Dv, 64 columns at a time --------------------------------------------------
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cb = 64 * ch, CW = min(64, Dv - cb);
        load_chunk(cb, CW, a.d_r != nullptr);
        __syncthreads();
        xu_du(true);
        __syncthreads();
        tile_gemm<true>(dKn, b3, b1, w, li, g);             // dKn += U dS'^T
        tile_gemm<true>(dWt, b4, b0, w, li, g);             // dWt += dU S^T      (scaled by -a below)
        if (a.d_r) tile_gemm<true>(dQn, b2, b0, w, li, g);  // dQn += dR S^T   (state-only mode: no read-out, no dQ)
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dWt[nt] *= -alpha;
    {   // d alpha
        for (int o = 1; o < 64; o <<= 1) da += __shfl_xor(da, o);
        if (lane == 0) s_red[w] = da;
        __syncthreads();
        if (tid == 0) {
            const float d = s_red[0] + s_red[1] + s_red[2] + s_red[3];
            a.d_alpha[fh] = logits ? d * alpha * (1.f - alpha) : d;
        }
    }

    // ---- phase 2: through T = (I + tril(b Kn Kn^T, -1))^-1 ---------------------------------------------------
    f32x4 dA[4];
    float dbeta[4] = {0.f, 0.f, 0.f, 0.f};                  // per accumulator row r (token 16w + 4g + r), partial over li
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dA[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (seq) {                                              // Gram matrix -> b5
        f32x4 gm[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) gm[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_gemm<true>(gm, Kn, Kn, w, li, g);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) b5[at(nt, r)] = gm[nt][r];
    }
    __syncthreads();
    // column chunks of dY = [dWt | dU]:  cc = 0 is the key block (Y = Wt, X0 = Kn), cc >= 1 the value chunks
    for (int cc = lin ? 1 : 0; cc <= nchunk; ++cc) {
        const int cb = 64 * (cc - 1), CW = cc == 0 ? 64 : min(64, Dv - cb);
        if (cc == 0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[at(nt, r)] = dWt[nt][r];                  // dY chunk -> b4
        } else {
            load_chunk(cb, CW, false);
            __syncthreads();
            xu_du(false);                                   // b4 = dU chunk (b3 = U is rebuilt as Ut below)
            __syncthreads();
            load_ut(b3, cb, CW);                             // Y chunk = Ut -> b3
            load_rows_io(b2, a.v, Dv, cb, CW, nullptr);      // X0 chunk = V -> b2
        }
        __syncthreads();
        if (seq) {
            // Z = T^T dY by blocked BACK substitution, the mirror image of the forward prep: wave w owns columns
            // 16w..16w+15 (columns never interact); for I = 3..0:  Z_I = T_II^T (dY_I - sum_{J>I} A_JI^T Z_J), every
            // solved Z_J staying in accumulator registers as the B operand of the next product (k = 4g + r).
            f32x4 Z[4];
            static_for_desc<3>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                f32x4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = b4[(16 * I + 4 * g + r) * BF_LD + 16 * w + li];
#pragma unroll
                for (int J = I + 1; J < 4; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * J + 4 * g + r;                               // A = -A_JI^T[i = li][j]
                        acc = mfma4(-s_beta[j] * b5[j * BF_LD + 16 * I + li], Z[J][r], acc);
                    }
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) z = mfma4(s_tii[I * 256 + (4 * g + r) * 16 + li], acc[r], z);   // T_II^T[li][4g+r]
                Z[I] = z;
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[(16 * I + 4 * g + r) * BF_LD + 16 * w + li] = z[r];
            });
            __syncthreads();                                // other waves own the other columns of Z
        }
        // now b4 = Z chunk.  d beta, d(X0) and dA
        {
            const float* X0 = cc == 0 ? Kn : b2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * w + 4 * g + r;
                const float bi = s_beta[row];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int o = at(nt, r), col = 16 * nt + li;
                    const float z = b4[o];
                    dbeta[r] += z * X0[o];
                    if (cc == 0) dKn[nt][r] += bi * z;
                    else if (col < CW && row < N) store1<IO>(a.d_v, ((bt * N + row) * Hh + h) * Dv + cb + col, bi * z);
                }
            }
            if (seq) tile_gemm<true>(dA, b4, cc == 0 ? Wt : b3, w, li, g);              // accumulates +Z Y^T; negated below
        }
        __syncthreads();
    }
    if (seq) {
        // M = diag(b) * tril(-Z Y^T, -1);  d beta += rowsum(tril(dA) * G);  dKn += (M + M^T) Kn
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + 4 * g + r;
            const float bi = s_beta[row];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float m = (16 * nt + li < row) ? -dA[nt][r] : 0.f;                  // strictly lower: column < row
                dbeta[r] += m * b5[at(nt, r)];
                b0[at(nt, r)] = bi * m;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) b1[at(nt, r)] = b0[at(nt, r)] + b0[(16 * nt + li) * BF_LD + 16 * w + 4 * g + r];
        __syncthreads();
        tile_gemm<false>(dKn, b1, Kn, w, li, g);
    }
    // ---- gates, L2 normalisation, stores ---------------------------------------------------------------------
    if (a.d_q) load_rows_io(b2, a.q, GDKVM_DK, 0, 64, s_qinv);                          // Qn = q * qinv -> b2
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * g + r;
        const float db = row16_sum(dbeta[r]);
        float dotk = 0.f, dotq = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { dotk += Kn[at(nt, r)] * dKn[nt][r]; if (a.d_q) dotq += b2[at(nt, r)] * dQn[nt][r]; }
        dotk = row16_sum(dotk); dotq = row16_sum(dotq);
        if (row < N) {
            const float bi = s_beta[row], ki = s_kinv[row], qi = s_qinv[row];
            if (li == 0) a.d_beta[(bt * N + row) * Hh + h] = logits ? db * bi * (1.f - bi) : db;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float gk = dKn[nt][r];
                if (normalize) gk = ki * (gk - Kn[at(nt, r)] * dotk);
                store1<IO>(a.d_k, ((bt * N + row) * Hh + h) * GDKVM_DK + 16 * nt + li, gk);
                if (a.d_q) {
                    float gq = dQn[nt][r];
                    if (normalize) gq = qi * (gq - b2[at(nt, r)] * dotq);
                    store1<IO>(a.d_q, ((bt * N + row) * Hh + h) * GDKVM_DK + 16 * nt + li, gq);
                }
            }
        }
    }
}

// [FH][Dk][Dv] row-major (a gradient with respect to the state before every frame) -> the accumulator images the serial kernel
// takes as its additive term: out[((fh*nsl + sl)*4 + w)*64 + lane][r] = in[fh][16w + 4g + r][16sl + li]
__global__ void gdr_rows_to_img_kernel(const float* in, float* out, size_t n_img, int Dv)
{
    const int nsl = Dv / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_img; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), li = lane & 15, g = lane >> 4, w = (int)((i >> 6) & 3);
        const size_t fs = i >> 8, fh = fs / nsl;
        const int sl = (int)(fs - fh * nsl);
        const float* src = in + (fh * GDKVM_DK + 16 * w + 4 * g) * Dv + 16 * sl + li;
        reinterpret_cast<f32x4*>(out)[i] = f32x4{src[0], src[Dv], src[2 * (size_t)Dv], src[3 * (size_t)Dv]};
    }
}

}  // namespace

extern "C" size_t gdkvm_scan_bwd_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv)
{
    if (B <= 0 || T <= 0 || Hh <= 0 || N <= 0 || Dk <= 0 || Dv <= 0) return 16;
    return 2 * (size_t)B * T * Hh * Dk * Dv * sizeof(float) + 16;       // dS' of every frame + Gb = Qn^T dR of every frame
}

// Shared body of gdkvm_scan_bwd (read-out gradient d_r, query gradient d_q) and gdkvm_scan_state_bwd (no read-out: q, d_r,
// d_q NULL; instead d_hist, the gradient with respect to the state before every frame, enters the reverse recurrence as its
// additive term).
static int scan_bwd_impl(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                         const float* s_hist, const void* fwd_workspace, size_t fwd_workspace_bytes,
                         const void* d_r, const float* d_hist, const float* d_s_out,
                         void* d_q, void* d_k, void* d_v, float* d_alpha, float* d_beta, float* d_s_in,
                         void* bwd_workspace, size_t bwd_workspace_bytes,
                         int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    if (int rc = check_common("scan_bwd", B, T, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (rule < 0 || rule > 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_bwd: rule=%d", rule);
    if (B == 0) return GDKVM_OK;
    if (T == 0 || N == 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_bwd: T and N must be positive");
    if (N > 64) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_bwd: N=%d > 64 tokens per frame: use gdkvm_scan_train_fwd / gdkvm_scan_train_bwd", N);
    if (int rc = check_ptrs("scan_bwd", {k, v, alpha, beta, s_hist, fwd_workspace, d_k, d_v, d_alpha, d_beta, bwd_workspace},
                            {q, d_r, d_q, d_hist, d_s_out, d_s_in})) return rc;
    if ((d_r != nullptr) != (d_q != nullptr) || (d_q && !q))
        return gdkvm_fail(GDKVM_ERR_ARG, "scan_bwd: q, d_r and d_q come together");
    WsView ws;
    if (int rc = carve("scan_bwd", const_cast<void*>(fwd_workspace), fwd_workspace_bytes, B, T, Hh, N, Dk, Dv, &ws)) return rc;
    if (bwd_workspace_bytes < gdkvm_scan_bwd_workspace_bytes(B, T, Hh, N, Dk, Dv) - 16)
        return gdkvm_fail(GDKVM_ERR_WORKSPACE, "scan_bwd: backward workspace %zu < %zu bytes", bwd_workspace_bytes,
                          gdkvm_scan_bwd_workspace_bytes(B, T, Hh, N, Dk, Dv));
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* ds_hist = static_cast<float*>(bwd_workspace);
    float* gb = ds_hist + (size_t)B * T * Hh * Dk * Dv;

    // the additive term of the reverse recurrence when it does not come from a read-out: d_hist as images, or zero
    if (!d_r) {
        const size_t n_img = (size_t)B * T * Hh * (Dv / 16) * 4 * 64;
        if (d_hist) {
            hipLaunchKernelGGL(gdr_rows_to_img_kernel, dim3((unsigned)((n_img + 255) / 256 > 8192 ? 8192 : (n_img + 255) / 256)), dim3(256), 0, st,
                               d_hist, gb, n_img, Dv);
            GDKVM_LAUNCH_CHECK("gdr_rows_to_img_kernel");
        } else {
            if (int rc = gdkvm_zero_async(gb, n_img * 16, st)) return rc;
        }
    }
    // reverse recurrence on the forward's serial kernel (operands from the training-mode workspace)
    if (int rc = gdr_launch_reverse_scan(ws, alpha, d_r, d_s_out, ds_hist, d_s_in, gb, B, T, Hh, N, Dv, io_dtype, flags, st)) return rc;

    BwdFrameArgs fa{q, k, v, alpha, beta, ws.qinv, ws.kn, ws.wt, ws.ut, ws.tii, s_hist, ds_hist, d_r,
                    d_q, d_k, d_v, d_alpha, d_beta, T, Hh, N, Dv, rule, flags};
    const size_t lds = (size_t)(8 * BF_TILE + 3 * 64 + 8 + 4 * 256) * sizeof(float);
    const void* fn = io_dtype == GDKVM_F32 ? reinterpret_cast<const void*>(gdr_bwd_frame_kernel<GDKVM_F32>)
                                           : reinterpret_cast<const void*>(gdr_bwd_frame_kernel<GDKVM_BF16>);
    {
        static std::atomic<unsigned long long> done_mask[2];
        if (int rc = gdr_lds_optin(fn, done_mask[io_dtype == GDKVM_F32 ? 0 : 1], lds, "scan_bwd")) return rc;
    }
    const dim3 fgrid((unsigned)(B * T * Hh));
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gdr_bwd_frame_kernel<GDKVM_F32>), fgrid, dim3(256), lds, st, fa);
    else hipLaunchKernelGGL((gdr_bwd_frame_kernel<GDKVM_BF16>), fgrid, dim3(256), lds, st, fa);
    GDKVM_LAUNCH_CHECK("gdr_bwd_frame_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_scan_bwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                              const float* s_hist, const void* fwd_workspace, size_t fwd_workspace_bytes,
                              const void* d_r, const float* d_s_out,
                              void* d_q, void* d_k, void* d_v, float* d_alpha, float* d_beta, float* d_s_in,
                              void* bwd_workspace, size_t bwd_workspace_bytes,
                              int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    if (!q || !d_r || !d_q) return gdkvm_fail(GDKVM_ERR_ARG, "scan_bwd: null pointer");
    return scan_bwd_impl(q, k, v, alpha, beta, s_hist, fwd_workspace, fwd_workspace_bytes, d_r, nullptr, d_s_out, d_q, d_k, d_v,
                         d_alpha, d_beta, d_s_in, bwd_workspace, bwd_workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream);
}

extern "C" int gdkvm_scan_state_bwd(const void* k, const void* v, const float* alpha, const float* beta,
                                    const float* s_hist, const void* fwd_workspace, size_t fwd_workspace_bytes,
                                    const float* d_hist, const float* d_s_out,
                                    void* d_k, void* d_v, float* d_alpha, float* d_beta, float* d_s_in,
                                    void* bwd_workspace, size_t bwd_workspace_bytes,
                                    int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    return scan_bwd_impl(nullptr, k, v, alpha, beta, s_hist, fwd_workspace, fwd_workspace_bytes, nullptr, d_hist, d_s_out, nullptr, d_k, d_v,
                         d_alpha, d_beta, d_s_in, bwd_workspace, bwd_workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream);
}
