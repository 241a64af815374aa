// gdr_prep.hip -- the frame-parallel side of the GDR memory path for gfx950: every frame folded
// into the affine map of the state,  S' = a P S + G,  P = I - Kn^T Wt,  G = Kn^T Ut  (SPEC-v0, SURVEY.md A.3, with
// U = Ut - a Wt S substituted).  Kernels, in file order:
//
//  gdr_prep_kernel     training only (N <= 64): the WY factors Wt = T b Kn, Ut = T b V by a blocked forward substitution that
//                      lives entirely in MFMA accumulators, plus the extra operand layouts and T_II the backward consumes.
//  gdr_fold_kernel     training: P and G (and P^T for the backward) from the stored WY factors.
//  gdr_prepm_kernel    inference: P and G directly, M = Kn^T T b by a back substitution on four Kn tiles, G = M V.
//  gdr_compose_kernel  frames of more than 64 tokens: composition of the 64-token chunks' affine maps.
// The serial recurrence that consumes P and G is in gdr_scan.hip; the workspace layout in gdr_ws.hpp.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <initializer_list>
#include <type_traits>

#include <cstdlib>
#include "gdkvm_common.hpp"
#include "gdr_device.hpp"
#include "gdr_ws.hpp"

namespace {


struct PrepArgs {
    const void* q; const void* k; const void* v; const float* beta;
    float* wt; float* knT; float* ut; float* qinv;
    float* kn; float* wtT; float* qnT; float* tii;      // training mode only (GDKVM_FLAG_TRAIN); so is wt
    float* wti;                                         // Wt as accumulator images, for the fold kernel
    int T, Hh, N, Dv, rule, flags;
};

__device__ __forceinline__ int pair_slot(int I, int J) { return I * (I - 1) / 2 + J; }   // J < I

// LDS carve for NB token tiles (floats): kinv[NP] beta[NP] qinv[NP] pad[NP] | negA[NB(NB-1)/2][64][4] | Ld[NB][64][4] | Tm[NB][64][4]
__host__ __device__ constexpr size_t prep_lds_bytes(int NB)
{
    return (size_t)(4 * 16 * NB + (NB * (NB - 1) / 2 + 2 * NB) * 256) * sizeof(float);
}

// TPR = column tiles a wave solves at once.  Their operands are fetched at kernel entry (latency hidden behind
// the norm / Gram / T_II phases) and their substitution chains are interleaved (TPR independent MFMA chains).
template <int NB, int IO, int TPR>
__global__ __launch_bounds__(256, (NB == 4 ? 2 : 1)) void gdr_prep_kernel(PrepArgs a)
{
    constexpr int NP = 16 * NB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_kinv = smem;
    float* s_beta = smem + NP;
    float* s_qinv = smem + 2 * NP;                        // (+ NP floats of padding keeps the images 16-byte aligned)
    f32x4* s_negA = reinterpret_cast<f32x4*>(smem + 4 * NP);
    f32x4* s_Ld = s_negA + (NB * (NB - 1) / 2) * 64;
    f32x4* s_Tm = s_Ld + NB * 64;

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = blockIdx.x;                       // (b*T + t)*Hh + h
    const int h = fh % a.Hh;
    const size_t bt = fh / a.Hh;
    const int N = a.N, Hh = a.Hh, Dv = a.Dv;
    const bool seq = a.rule == GDKVM_RULE_DELTA_SEQUENTIAL;
    const int ntile = GDKVM_DK / 16 + Dv / 16;       // column tiles: 4 of K (-> Wt, Kn^T) then Dv/16 of V (-> Ut)
    const int nround = (ntile + 4 * TPR - 1) / (4 * TPR);

    // raw operands of this wave's column tiles for one round: x[i][I][r] = X[token 16I+4g+r][col 16c+li]
    float xr[TPR][NB][4];
    auto fetch_round = [&](int rd) {
#pragma unroll
        for (int i = 0; i < TPR; ++i) {
            const int c = min((rd * TPR + i) * 4 + w, ntile - 1);          // clamp: surplus slots refetch a real tile
            const bool isK = c < GDKVM_DK / 16;
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = min(16 * I + 4 * g + r, N - 1);          // padding tokens: real row, zeroed by beta = 0
                    xr[i][I][r] = isK ? load1<IO>(a.k, ((bt * N + n) * Hh + h) * GDKVM_DK + 16 * c + li)
                                      : load1<IO>(a.v, ((bt * N + n) * Hh + h) * Dv + 16 * (c - 4) + li);
                }
        }
    };
    fetch_round(0);

    // ---- phase 0 (a5 prologue): inverse key / query norms and gates ----------------------------------
    for (int n = tid; n < NP; n += 256) {
        float kinv = 0.f, qinv = 0.f, bta = 0.f;
        if (n < N) {
            kinv = qinv = 1.f;
            if (a.flags & GDKVM_FLAG_NORMALIZE_QK) {
                float sk = 0.f, sq = 0.f;
#pragma unroll
                for (int c = 0; c < GDKVM_DK; c += 4) {
                    const f32x4 x = load4<IO>(a.k, ((bt * N + n) * Hh + h) * GDKVM_DK + c);
                    const f32x4 y = load4<IO>(a.q, ((bt * N + n) * Hh + h) * GDKVM_DK + c);
                    sk += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
                    sq += y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
                }
                kinv = 1.0f / sqrtf(sk + GDKVM_EPS_NORM);
                qinv = 1.0f / sqrtf(sq + GDKVM_EPS_NORM);
            }
            bta = a.beta[(bt * N + n) * Hh + h];
            if (a.flags & GDKVM_FLAG_GATE_LOGITS) bta = 1.0f / (1.0f + expf(-bta));
        }
        s_kinv[n] = kinv;
        s_beta[n] = bta;
        s_qinv[n] = qinv;
        a.qinv[(size_t)fh * NP + n] = qinv;
    }
    __syncthreads();

    if (seq) {
        // ---- phase 1: Gram blocks, transposed:  C = K_J K_I^T, lane (i,g) reg r = k_{J,4g+r} . k_{I,i}
        //      = the A-operand image of A_IJ[i][4g+r];  stored negated and scaled by b_i (row gate).
        for (int p = w; p < NB * (NB + 1) / 2; p += 4) {
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= p) ++I;
            const int J = p - I * (I + 1) / 2;
            const int nI = 16 * I + li, nJ = 16 * J + li;
            f32x4 kI[4], kJ[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                kI[m] = load4<IO>(a.k, ((bt * N + min(nI, N - 1)) * Hh + h) * GDKVM_DK + 16 * m + 4 * g);
                kJ[m] = load4<IO>(a.k, ((bt * N + min(nJ, N - 1)) * Hh + h) * GDKVM_DK + 16 * m + 4 * g);
            }
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (m & 1) acc1 = mfma4(kJ[m][r], kI[m][r], acc1);
                    else acc0 = mfma4(kJ[m][r], kI[m][r], acc0);
                }
            f32x4 acc = acc0 + acc1;
            const float rowscale = s_kinv[nI] * s_beta[nI];                 // 0 for padding rows (kinv = beta = 0)
            const f32x4 kinvJ = *reinterpret_cast<const f32x4*>(s_kinv + 16 * J + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] *= rowscale * kinvJ[r];
            if (J < I) {
                s_negA[pair_slot(I, J) * 64 + lane] = -acc;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * g + r >= li) acc[r] = 0.f;   // strictly lower: col < row
                s_Ld[I * 64 + lane] = acc;
            }
        }
        __syncthreads();
        // ---- phase 2: T_II = (I + L_II)^-1 by forward substitution; 16 threads per block, one column each
        {
            const int I = tid >> 4, j = tid & 15;
            if (I < NB) {
                float t[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = (i == j) ? 1.f : 0.f;
#pragma unroll
                for (int i = 1; i < 16; ++i) {
                    float sm = 0.f;
#pragma unroll
                    for (int gg = 0; gg * 4 < i; ++gg) {
                        const f32x4 Lr = s_Ld[I * 64 + gg * 16 + i];        // L[i][4gg .. 4gg+3]
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (4 * gg + r < i) sm += Lr[r] * t[4 * gg + r];
                    }
                    t[i] = (i > j) ? -sm : t[i];
                }
                float* Tm = reinterpret_cast<float*>(s_Tm + I * 64);
#pragma unroll
                for (int i = 0; i < 16; ++i) Tm[((j >> 2) * 16 + i) * 4 + (j & 3)] = t[i];   // image of T[i][j]
                if (a.flags & GDKVM_FLAG_TRAIN) {                                        // T_II row-major for the backward
                    float* tg = a.tii + ((size_t)fh * NB + I) * 256;
#pragma unroll
                    for (int i = 0; i < 16; ++i) tg[i * 16 + j] = t[i];
                }
            }
        }
        __syncthreads();
    }

    // ---- phase 3: blocked forward substitution entirely in accumulators, TPR column tiles interleaved ----
    float* wt = a.wt + (size_t)fh * NP * GDKVM_DK;
    f32x4* wti = reinterpret_cast<f32x4*>(a.wti) + (size_t)fh * NP * GDKVM_DK / 4;
    float* knT = a.knT + (size_t)fh * GDKVM_DK * NP;
    f32x4* ut = reinterpret_cast<f32x4*>(a.ut + (size_t)fh * NP * Dv);
    const bool train = a.flags & GDKVM_FLAG_TRAIN;
    float* kn_nat = a.kn + (size_t)fh * NP * GDKVM_DK;
    float* wtT = a.wtT + (size_t)fh * GDKVM_DK * NP;
    if (train) {                                          // Qn^T for the backward's Qn^T dR: 4 tokens per 16-byte store
        float* qnT = a.qnT + (size_t)fh * GDKVM_DK * NP;
        for (int idx = tid; idx < GDKVM_DK * (NP / 4); idx += 256) {
            const int d = idx / (NP / 4), n0 = (idx - d * (NP / 4)) * 4;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[r] = n0 + r < N ? load1<IO>(a.q, ((bt * N + n0 + r) * Hh + h) * GDKVM_DK + d) * s_qinv[n0 + r] : 0.f;
            *reinterpret_cast<f32x4*>(qnT + (size_t)d * NP + n0) = o;
        }
    }
    for (int rd = 0; rd < nround; ++rd) {
        if (rd > 0) fetch_round(rd);
        f32x4 Y[TPR][NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            const int n0 = 16 * I + 4 * g;
            const f32x4 bt4 = *reinterpret_cast<const f32x4*>(s_beta + n0);
            const f32x4 ki4 = *reinterpret_cast<const f32x4*>(s_kinv + n0);
            f32x4 acc[TPR];
#pragma unroll
            for (int i = 0; i < TPR; ++i) {
                const int c = (rd * TPR + i) * 4 + w;
                const bool isK = c < GDKVM_DK / 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = isK ? xr[i][I][r] * ki4[r] : xr[i][I][r];
                if (isK) {
                    *reinterpret_cast<f32x4*>(knT + (size_t)(16 * c + li) * NP + n0) = acc[i];   // Kn^T, 4 tokens
                    if (train) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) kn_nat[(size_t)(n0 + r) * GDKVM_DK + 16 * c + li] = acc[i][r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] *= bt4[r];
            }
            if (seq) {
#pragma unroll
                for (int J = 0; J < I; ++J) {
                    const f32x4 na = s_negA[pair_slot(I, J) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int i = 0; i < TPR; ++i) acc[i] = mfma4(na[r], Y[i][J][r], acc[i]);
                }
                const f32x4 t4 = s_Tm[I * 64 + lane];
                f32x4 y[TPR];
#pragma unroll
                for (int i = 0; i < TPR; ++i) y[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < TPR; ++i) y[i] = mfma4(t4[r], acc[i][r], y[i]);
#pragma unroll
                for (int i = 0; i < TPR; ++i) acc[i] = y[i];
            }
#pragma unroll
            for (int i = 0; i < TPR; ++i) {
                const int c = (rd * TPR + i) * 4 + w;
                const bool isK = c < GDKVM_DK / 16;
                if (isK && a.rule == GDKVM_RULE_GATED_LINEAR) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                Y[i][I] = acc[i];
                if (c < ntile) {
                    if (isK) {
                        wti[((size_t)c * NB + I) * 64 + lane] = acc[i];
                        if (train) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) wt[(size_t)(n0 + r) * GDKVM_DK + 16 * c + li] = acc[i][r];
                            *reinterpret_cast<f32x4*>(wtT + (size_t)(16 * c + li) * NP + n0) = acc[i];
                        }
                    } else {
                        ut[((size_t)(c - 4) * NB + I) * 64 + lane] = acc[i];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The frame as ONE affine map of the state.  Substituting U = Ut - a Wt S into S' = a S + Kn^T U gives
//        S' = a (I - Kn^T Wt) S + Kn^T Ut  =  a P S + G ,
// and neither P [Dk,Dk] nor G [Dk,Dv] depends on S: they belong to the state-independent (frame-parallel) side.  The
// serial chain per frame drops from two dependent GEMMs with an LDS exchange between them (and a cost that grows with the
// token count) to one [Dk,Dk]x[Dk,16] product per slice -- 16 fp32 MFMA per state wave, one barrier, independent of N.
//
// gdr_fold_kernel: P = I - Kn^T Wt and G = Kn^T Ut for one frame-head; wave w owns row tile w (Dk rows 16w..16w+15), the
// 4 + Dv/16 column tiles are split over gridDim.y workgroups.  A operands are rows of knT (k = 16I + 4g + r, the
// permutation under which the Ut images -- prep's accumulator layout -- are B operands as stored).
struct FoldArgs { const float* wti; const float* knT; const float* ut; float* pp; float* gg; float* ppt; int Dv; float* gmax; };

// grid (FH, 1 + ceil(Dv/64)): block y = 0 folds the four Wt tiles into P, block y > 0 four Ut tiles into G.  Every operand
// of the block's four tiles is requested up front (20 16-byte loads per lane), then 4 x 4NB MFMA run back to back.
template <int NB, int FMT>
__global__ __launch_bounds__(256, 2) void gdr_fold_kernel(FoldArgs a)
{
    constexpr int NP = 16 * NB, NT = fmt_terms(FMT);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t fh = blockIdx.x;
    const int nsl = a.Dv / 16;
    const bool isP = blockIdx.y == 0;
    const int c0 = isP ? 0 : 4 * ((int)blockIdx.y - 1);               // first of this block's column tiles
    const int nlim = isP ? GDKVM_DK / 16 : nsl;
    const f32x4* img = isP ? reinterpret_cast<const f32x4*>(a.wti) + fh * (NP * (size_t)GDKVM_DK / 4)
                           : reinterpret_cast<const f32x4*>(a.ut) + fh * (NP * (size_t)a.Dv / 4);
    f32x4 ka[NB];
    {
        const float* kp = a.knT + (fh * GDKVM_DK + 16 * w + li) * NP + 4 * g;
#pragma unroll
        for (int I = 0; I < NB; ++I) ka[I] = *reinterpret_cast<const f32x4*>(kp + 16 * I);
    }
    constexpr int TB = NB <= 8 ? 4 : 2;                                // tiles whose operands are in registers together
#pragma unroll
    for (int j0 = 0; j0 < 4; j0 += TB) {
        f32x4 y[TB][NB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int c = min(c0 + j0 + j, nlim - 1);
#pragma unroll
            for (int I = 0; I < NB; ++I) y[j][I] = img[((size_t)c * NB + I) * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int c = c0 + j0 + j;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int I = 0; I < NB; ++I) {
                    if (I & 1) acc1 = mfma4(ka[I][r], y[j][I][r], acc1);
                    else acc0 = mfma4(ka[I][r], y[j][I][r], acc0);
                }
            const f32x4 o = acc0 + acc1;
            if (c < nlim) {
                if (isP) {
                    {   // P^T for the backward's reverse recurrence dS = a P^T dS' + Qn^T dR: this lane's four values are
                        // P^T[16c + li][k = 16w + 4g + r], four consecutive k of row li of row tile c
                        __bf16 t3[3][4];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            split3(((16 * w + 4 * g + r == 16 * c + li) ? 1.f : 0.f) - o[r], t3[0][r], t3[1][r], t3[2][r]);
                        uint2* pt = reinterpret_cast<uint2*>(a.ppt) + (fh * 4 + c) * (size_t)(3 * SPLIT_IMG);
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) pt[sp * SPLIT_IMG + split_slot(w, g, li)] = pack_bf16x4(t3[sp]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                  // P[16w + 4g + r][k = 16c + li] -> term images of row tile w
                        const int row = 16 * w + 4 * g + r, col = 16 * c + li;
                        unsigned short tt[3];
                        OpFmt<FMT>::split1((row == col ? 1.f : 0.f) - o[r], tt);
                        unsigned short* img = reinterpret_cast<unsigned short*>(a.pp) + fh * (size_t)(4 * 3 * SPLIT_IMG * 4) + w * (NT * SPLIT_IMG * 4);
                        const int e = split_slot(c, li >> 2, 4 * g + r) * 4 + (li & 3);      // k0 = 16c + 4(li>>2), element li&3
#pragma unroll
                        for (int sp = 0; sp < NT; ++sp) img[sp * SPLIT_IMG * 4 + e] = tt[sp];
                    }
                } else {
                    reinterpret_cast<f32x4*>(a.gg)[((fh * nsl + c) * 4 + w) * 64 + lane] = o * OpFmt<FMT>::STATE;   // the scan carries S * STATE
                    const float mx = wave_max_nonneg(absmax4(o));      // range bookkeeping of the pair16 recurrence (gdr_ws.hpp: gmax)
                    if (lane == 0) a.gmax[(fh * nsl + c) * 4 + w] = mx;
                }
            }
        }
    }
}

template <int NB>
void launch_fold(const FoldArgs& fa, int FH, bool wide, hipStream_t st)
{
    const dim3 grid((unsigned)FH, (unsigned)(1 + (fa.Dv / 16 + 3) / 4));
    if (wide) hipLaunchKernelGGL((gdr_fold_kernel<NB, FMT_SPLIT3>), grid, dim3(256), 0, st, fa);
    else hipLaunchKernelGGL((gdr_fold_kernel<NB, FMT_PAIR16>), grid, dim3(256), 0, st, fa);
}

// ------------------------------------------------------------------------------------------------------------------
// gdr_prepm_kernel -- the frame-parallel side for INFERENCE (N <= 128), producing the affine map of the frame directly:
//        M = Kn^T T diag(b)  [Dk, N]          P = I - M Kn          G = M V
// (P = I - Kn^T Wt and G = Kn^T Ut with Wt = T b Kn, Ut = T b V substituted).  Only the FOUR column tiles of Kn go through
// the triangular solve -- M^T = diag(b) T^T Kn is a BACK substitution with the transposed Gram blocks -- and V is used
// raw: its tiles, fetched in the accumulator layout, are B operands as loaded.  Wt and Ut never exist, nothing but P, G
// and qinv is written.  (Training keeps gdr_prep_kernel + gdr_fold_kernel: the backward consumes Wt, Ut, T_II.)
//   phase 0   norms and gates (as gdr_prep_kernel)
//   phase 1   Gram blocks: L_II (row gate, strictly lower) and, for J > I, -L_JI as the A image of (L_JI)^T
//   phase 2   T_II = (I + L_II)^-1 by forward substitution, stored as the A image of T_II^T
//   phase 3   wave w: Kn column tile w;  Z_I = T_II^T (Kn_I - sum_{J>I} L_JI^T Z_J), I = NB-1 .. 0;  M^T_I = b_I Z_I;
//             Kn and M^T tiles -> LDS as accumulator images
//   phase 4   wave w: P^T tiles (w, m) = delta - Kn_w^T-image x M^T_m  (the C layout of P^T IS the state waves' A layout
//             of P), then G tiles (m, cV) = M_m x V_cV for cV = w, w+4, ...
//             bf16 I/O, N <= 64: V is exactly bf16, so G runs on v_mfma_f32_16x16x32_bf16 with M split into three bf16
//             terms (split3): 6 MFMA of 16 cycles per tile instead of 16 of 32, same accuracy.
struct PrepMArgs {
    const void* q; const void* k; const void* v; const float* beta;
    float* qinv; float* pp; float* gg;
    float* gmax;                                        // max |G| of the frame's final map per 16-column slice (gdr_ws.hpp)
    const float* norms;                                 // [rows][Hh][2] inverse key / query norms from gdkvm_proj_gates, or NULL: computed here
    float* x0; float* ppc; float* ggc;                  // frames of more than 64 tokens: chunk 0 -> x0, chunk c >= 1 -> ppc/ggc[c-1]
    int T, Hh, N, Dv, rule, flags, np_total;            // N = tokens of the frame, np_total = its padded count (qinv row length)
    int nchunk;                                         // FUSE: 64-token chunks per frame, walked by ONE workgroup (else gridDim.y)
    int grid3;                                          // launched as (8, T, B / 8): frame = (x + 8 z) T + y, no division (one chunk, one head)
#ifdef GDKVM_DIAG
    unsigned long long* diag;
#endif
};

__host__ __device__ constexpr bool prepm_split(int NB, int IO) { return NB == 4 && IO == GDKVM_BF16; }
// bf16 I/O, chunk-parallel (not the fused walk): the COMPACT layout -- two regions re-used along the kernel's phases, 40 KiB instead
// of 72.7, so that THREE workgroups share a CU (cfg3: 640 chunk workgroups in one round instead of 1.25):
//   R1 16 KiB   phases 0-3: kinv beta qinv pad | negB pairs | Ld | TmT (15 KiB)   end of phase 3: mt [4][NB]   phase 4b: V staging
//   R2 24 KiB   phase 0: the raw K rows (17 KiB)                                   end of phase 3: m3 [3][4][NB/2]
// (mt and m3 are written behind a barrier that follows every wave's last read of R1 / R2, and are read into registers once, at the
// start of phase 4, behind which another barrier releases R1 for the V tiles.)
__host__ __device__ constexpr bool prepm_compact(int NB, int IO, bool fuse) { return prepm_split(NB, IO) && !fuse; }
__host__ __device__ constexpr size_t prepm_lds_bytes(int NB, int IO, bool fuse = false)
{   // kinv beta qinv pad | negB pairs | Ld | TmT | kni [4][NB] | mt [4][NB]   (images of 64 x f32x4) | m3 [3][4][NB/2] (bf16 arm)
    // | fused chunk walk: wave 3's column tile (two term images)
    if (prepm_compact(NB, IO, fuse)) return (size_t)(4 * NB * 256 + 3 * 4 * (NB / 2) * 256) * sizeof(float);
    return (size_t)(4 * 16 * NB + (NB * (NB - 1) / 2 + 2 * NB + 8 * NB + (prepm_split(NB, IO) ? 3 * 4 * (NB / 2) : 0)) * 256) * sizeof(float)
           + (fuse ? 2 * SPLIT_IMG * 8 : 0);
}



// W3: three workgroups per CU (the compact LDS layout allows it; 168 registers) -- chosen by the launch when there are more workgroups
// than two per CU hold (cfg3: 640 chunk workgroups); with at most two per CU the kernel takes the registers instead (M's term images
// fetched once for the P and the G tiles).
template <int NB, int IO, int FMT, bool FUSE = false, bool W3 = false>
__global__ __launch_bounds__(256, (NB == 4 ? (W3 ? 3 : 2) : 1))
__attribute__((amdgpu_waves_per_eu(1, (W3 ? 3 : 2)))) void gdr_prepm_kernel(PrepMArgs a)
{
    static_assert(!W3 || prepm_compact(NB, IO, FUSE), "three workgroups per CU need the compact LDS layout");
    constexpr int NP = 16 * NB, NT = fmt_terms(FMT);
    static_assert(!FUSE || (prepm_split(NB, IO) && FMT == FMT_PAIR16), "the fused chunk walk is built for bf16 I/O on pair16 operands");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool COMPACT = prepm_compact(NB, IO, FUSE);  // (layout: prepm_lds_bytes)
    float* s_kinv = smem;
    float* s_beta = smem + NP;
    // bf16 I/O: beta * gs and kinv / gs, gs the power of two G is handed to the scan with (a single-chunk frame: the scan's state
    // scale) -- M is formed as (M gs) and Kn^T as (Kn^T / gs), so P = I - (M gs)(Kn / gs) is unchanged to the bit and the G tiles
    // leave the MFMA scaled, one multiply per element less in the kernel's busiest phase
    float* s_beta_s = smem + 2 * NP;
    float* s_kinv_s = smem + 3 * NP;
    f32x4* s_negB = reinterpret_cast<f32x4*>(smem + 4 * NP);
    f32x4* s_Ld = s_negB + (NB * (NB - 1) / 2) * 64;
    f32x4* s_TmT = s_Ld + NB * 64;
    f32x4* s_r2 = reinterpret_cast<f32x4*>(smem) + 4 * NB * 64;          // COMPACT: region R2, behind the 16 KiB of R1
    f32x4* s_kni = COMPACT ? reinterpret_cast<f32x4*>(smem) : s_TmT + NB * 64;   // (V staging tiles of phase 4b)
    f32x4* s_mt = COMPACT ? reinterpret_cast<f32x4*>(smem) : s_kni + 4 * NB * 64;
    constexpr bool SPLIT = prepm_split(NB, IO);
    constexpr int KS = NB / 2;                            // 32-token k-steps of the bf16 MFMA
    uint2* s_m3 = reinterpret_cast<uint2*>(COMPACT ? s_r2 : s_mt + 4 * NB * 64);   // [3 terms][4 m][KS][64 lanes][2 halves]: A images of M
    constexpr int KLD = GDKVM_DK + 4;
    // the K rows [NP][KLD] as fp32 until the end of phase 3 overwrites the region (COMPACT: at the END of R2)
    float* s_K = COMPACT ? smem + (prepm_lds_bytes(NB, IO, FUSE) / sizeof(float) - NP * KLD) : reinterpret_cast<float*>(s_kni);
    // bf16 I/O: the same rows as they came, [NP][KRP bytes] (pitch 16 x odd: a b128 read of 16 rows touches every bank once) -- the Gram
    // blocks' operand images are read from here by every wave, not fetched by every wave from memory (4 x 8 KB of the 72 KB a workgroup
    // pulled through its CU's vector L1 at entry, where all eight waves of the CU queue).  Lives until the barrier that ends phase 1:
    // over T^T and the gap in front of the fp32 rows (COMPACT), or inside the M^T tiles (behind the 2 KB tail of the fp32 rows)
    constexpr int KRP = 2 * GDKVM_DK + 16;
    char* s_Kraw = COMPACT ? reinterpret_cast<char*>(s_TmT) : reinterpret_cast<char*>(s_mt) + 2048;
    static_assert(!COMPACT || (4 * NP + (NB * (NB - 1) / 2 + NB) * 256) * 4 + NP * KRP <= (int)prepm_lds_bytes(NB, IO, false) - NP * KLD * 4,
                  "COMPACT: the raw key rows end in front of the fp32 ones");
    static_assert(COMPACT || (NP * KLD * 4 - 4 * NB * 1024 <= 2048 && 2048 + NP * KRP <= 4 * NB * 1024), "the raw key rows fit the M^T tiles");
    static_assert(NP * KLD <= (COMPACT ? 3 * 4 * (NB / 2) : 8 * NB) * 256, "the K staging tile aliases kni + mt (COMPACT: m3)");
    static_assert(4 * NP + (NB * (NB - 1) / 2 + 2 * NB) * 256 <= 4 * NB * 256, "COMPACT: the phase 0-3 scratch fits under mt");

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // every argument the prologue needs is fetched HERE, in one batch of scalar loads: left to the compiler they are fetched where first
    // used, one basic block after the other -- four dependent round trips to the scalar cache before the first key row is requested
    {
        const unsigned gdx = gridDim.x, gdy = gridDim.y;
        asm volatile("" :: "s"(a.q), "s"(a.k), "s"(a.v), "s"(a.beta), "s"(a.qinv), "s"(a.pp), "s"(a.gg), "s"(a.gmax), "s"(a.norms),
                     "s"(a.T), "s"(a.Hh), "s"(a.N), "s"(a.Dv), "s"(a.rule), "s"(a.flags), "s"(a.np_total), "s"(a.nchunk), "s"(a.grid3), "s"(gdx), "s"(gdy));
    }
    // XCD-aware: the serial kernel runs clip-head bh on XCD bh % 8 (when their count is a multiple of 8); fold the frames of
    // that clip-head on the same XCD so its P and G are read from the L2 they were written through (speed only)
    int fh = blockIdx.x;
    if (a.grid3) {
        // (the common case -- one head, frames of at most 64 tokens, clips a multiple of 8 -- comes as a 3-D grid whose x is the XCD:
        // the integer divisions of the 1-D decoding stood between the kernel's entry and its first load)
        fh = ((int)blockIdx.x + 8 * (int)blockIdx.z) * (int)gridDim.y + (int)blockIdx.y;
    } else {
        const int BH = (int)(gridDim.x / a.T), per_clip = a.T * a.Hh;       // gridDim.x = B * T * Hh; BH = B * Hh clip-heads
        if (BH % 8 == 0 && a.Hh == 1) {
            const int x = blockIdx.x, xcd = x & 7, idx = x >> 3;
            fh = (xcd + 8 * (idx / per_clip)) * per_clip + idx % per_clip;
        }
    }
    const int h = a.Hh == 1 ? 0 : fh % a.Hh;
    const int Ntot = a.N, nchunk = FUSE ? a.nchunk : (a.grid3 ? 1 : (int)gridDim.y);
    // FUSE (frames of more than 64 tokens, many frames): this workgroup walks the frame's chunks itself and carries the frame's
    // running map [P | G] <- P_c [P | G] + [0 | G_c] in accumulators -- wave w owns COLUMN tile w of P and column tiles w, w+4, ..
    // of G (all four row tiles of each: X[m][j]), which is the layout the chunk's own P tiles and G tiles are born in, and a
    // column's four row tiles are exactly the rows a B image of that column needs: the re-split never leaves the wave.  The chunk
    // maps P_c, G_c (80 KB per chunk) are never written: gdr_compose_kernel re-read them from HBM (cfg5: 2 x 328 MB).
    constexpr int XJ = FUSE ? 5 : 1;
    f32x4 X[FUSE ? 4 : 1][XJ];
    float xsplit_max = 0.f;                               // FUSE: largest |running map entry| this wave re-split into fp16 pairs
    const int tid_k = tid;
#if defined(GDKVM_DIAG) && defined(GDKVM_DIAG_TWICE)
    // diagnostic: the whole body twice over the same frame (trip count opaque: one copy of the code), second pass stamped into row T - 1 --
    // what the first pass pays for instructions and data met for the first time
    int npass__ = 2;
    asm volatile("" : "+s"(npass__));
    for (int pass__ = 0; pass__ < npass__; ++pass__)
#endif
    for (int chunk = (FUSE || a.grid3) ? 0 : (int)blockIdx.y, chunk_end = FUSE ? nchunk : chunk + 1; chunk < chunk_end; ++chunk) {
    // FUSE: the lane ids are re-derived per chunk from an opaque copy -- otherwise every lane-dependent address of the body is
    // hoisted out of the chunk loop as an invariant and held (then spilled) across it: ~100 registers the running map needs
    int tid_o = tid_k;
#if defined(GDKVM_DIAG) && defined(GDKVM_DIAG_TWICE)
    asm volatile("" : "+v"(tid_o));
    __syncthreads();
#endif
    if constexpr (FUSE) asm volatile("" : "+v"(tid_o));
    const int tid = tid_o, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int tok0 = chunk * NP;
    const size_t bt = (size_t)(a.Hh == 1 ? fh : fh / a.Hh) * Ntot + tok0;   // row of this chunk's first token; rows are addressed bt*1 + n below
    const int N = min(NP, Ntot - tok0), Hh = a.Hh, Dv = a.Dv, nsl = Dv / 16;
    const bool seq = a.rule == GDKVM_RULE_DELTA_SEQUENTIAL;
    const bool p_identity = a.rule == GDKVM_RULE_GATED_LINEAR;
    // the frame's final G goes to the scan in the scan's scale (it carries S * STATE); chunk maps headed for the composition stay raw
    const float gscale = nchunk == 1 ? OpFmt<FMT>::STATE : 1.0f, gscale_inv = nchunk == 1 ? OpFmt<FMT>::STATE_INV : 1.0f;
#if defined(GDKVM_DIAG) && defined(GDKVM_DIAG_TWICE)
    const int t = a.T - pass__;
#else
    const int t = a.T;                                    // diagnostic builds: stamps go to row T of the buffer
#endif
    (void)t;
    DIAG_STAMP(0);
#if defined(GDKVM_DIAG) && defined(GDKVM_DIAG_SPAN)
    // diagnostic: every workgroup's entry and exit time (a chip-wide counter) behind the stamp rows -- when the workgroups of one launch
    // start, how long each lives, what the launch adds around them
    if (a.diag && tid == 0) {
        unsigned long long t__;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");
        a.diag[(size_t)(a.T + 1) * 8 + 2 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))] = t__;
    }
#endif

    // this wave's first V tile, raw, in the accumulator layout: x[I][r] = V[token 16I+4g+r][16cV+li]
    float xk[NB][4], xv[SPLIT ? 1 : 2][SPLIT ? 1 : NB][4];
    auto load_v = [&](int cV, float (&d)[SPLIT ? 1 : NB][4]) __attribute__((always_inline)) {
        cV = min(cV, nsl - 1);
#pragma unroll
        for (int I = 0; I < (SPLIT ? 1 : NB); ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                d[I][r] = load1<IO>(a.v, ((bt + min(16 * I + 4 * g + r, N - 1)) * Hh + h) * Dv + 16 * cV + li);
    };
    // SPLIT: a V tile (64 tokens x 16 columns bf16 = 2 KiB) is fetched as 128 row-major 16-byte pieces, two per lane, staged in
    // a wave-private LDS tile and read back TRANSPOSED (ds_read_b64_tr_b16: four rows x 16 columns per 16-lane group, column-
    // major) as the B operand -- 2 vector loads + 2 LDS writes + 4 LDS reads per tile instead of 16 two-byte gathers.
    uint4 vA0, vA1, vB0, vB1;                              // (scalars, not an array: an array of these ends up in scratch)
    // the register-rich build fetches the wave's first FOUR V tiles (Dv <= 256: all of them) at kernel entry, behind the key rows: with
    // one tile in flight ahead of the MFMAs, each of the wave's tiles exposed most of a memory round trip (stamps: 6.4 k cycles of G
    // phase for 1.5 k of MFMA issue)
    constexpr bool VPRE = SPLIT && !FUSE && !W3;
    uint4 vC0, vC1, vD0, vD1;
    // (a uniform base and two 32-bit lane offsets: as 64-bit lane addresses they cost four registers across the G loop, which the
    // three-workgroups-per-CU build of this kernel does not have)
    const bf16_t* vrow = static_cast<const bf16_t*>(a.v) + (bt * Hh + h) * Dv;
#ifdef GDKVM_ABL_PREP_VSAME                                 // (tools/abl_scan.py prep: every lane pair reads token 0's piece -- one row segment per load)
    const unsigned voff0 = 8u * (lane & 1), voff1 = 8u * (lane & 1);
#else
    const unsigned voff0 = (unsigned)min(lane >> 1, N - 1) * (unsigned)(Hh * Dv) + 8u * (lane & 1);
    const unsigned voff1 = (unsigned)min(32 + (lane >> 1), N - 1) * (unsigned)(Hh * Dv) + 8u * (lane & 1);
#endif
    auto load_vraw = [&](int cV, uint4& d0, uint4& d1) __attribute__((always_inline)) {
        cV = min(cV, nsl - 1);
        const bf16_t* vp = vrow + 16 * cV;
        d0 = *reinterpret_cast<const uint4*>(vp + voff0);
        d1 = *reinterpret_cast<const uint4*>(vp + voff1);
    };

    // bf16 I/O: the Gram blocks of phase 1 run on the bf16 MFMA straight from the key rows -- token tile X as an operand image is
    // 16 rows x 16 bytes per k-step, the A image of K_X and the B image of K_X^T at once, and products of bf16 are exact in fp32 --
    // 2 MFMAs of 16 cycles per block instead of 16 exact-fp32 ones of 32 fed from the LDS staging tile.  Rows past N are clamped
    // duplicates, not zeros: every Gram entry that involves one is multiplied by its kinv = 0 / beta = 0 below.
    bf16x8 kimg[IO == GDKVM_BF16 ? NB : 1][2];            // (read from the raw rows in LDS behind the phase 0 barrier)

    // ---- phase 0 (a5 prologue): ONE pass over the k and q rows by all 256 threads (4 threads per token, 16 channels each):
    //      K staged in LDS as fp32 for the Gram blocks and the Kn tiles, inverse norms by a 4-lane reduction, gates
#pragma unroll
    for (int rep = 0; rep < NP / 64; ++rep) {
        const int n = rep * 64 + (tid >> 2), qd = tid & 3;
        float sk = 0.f, sq = 0.f;
        const bool given = a.norms != nullptr;            // (uniform) the norms came with the projections: q is not read here at all
        // every load of the phase is requested before the first is consumed (no branch in between: a conditional load starts its round
        // trip only when the ones in front of it are back): gate, given norms, key (and query) rows, rows past N clamped and zeroed
        // (a uniform row index plus a small lane part: addresses as scalar base + 32-bit lane offset -- as one 64-bit lane expression
        // each cost a 64-bit multiply-add per load, in front of the loads everything else waits for)
        const size_t row_u = bt * Hh + h;                  // (uniform)
        const unsigned row_l = (unsigned)min(n, N - 1) * (unsigned)Hh;
        const size_t rown = row_u + row_l;
        float bta_raw = (a.beta + row_u)[row_l];
        float2 nn = {1.f, 1.f};
        if (given) nn = reinterpret_cast<const float2*>(a.norms + row_u * 2)[row_l];
        f32x4 xs[4], ys[4];
        uint4 kraw0, kraw1;
        if constexpr (IO == GDKVM_BF16) {
            const char* kb = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.k) + row_u * GDKVM_DK);
            const unsigned ko = row_l * (GDKVM_DK * 2) + 32u * qd;
            kraw0 = *reinterpret_cast<const uint4*>(kb + ko); kraw1 = *reinterpret_cast<const uint4*>(kb + ko + 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (IO != GDKVM_BF16) xs[j] = load4<IO>(a.k, rown * GDKVM_DK + 16 * qd + 4 * j);
            ys[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!given) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ys[j] = load4<IO>(a.q, rown * GDKVM_DK + 16 * qd + 4 * j);
        }
        if constexpr (VPRE) {
            if (rep == 0) {
                load_vraw(w, vA0, vA1);
                load_vraw(w + 4, vB0, vB1);
                load_vraw(w + 8, vC0, vC1);
                load_vraw(w + 12, vD0, vD1);
            }
        }
        DIAG_STAMP2(0);                                    // every load of the phase requested
        if constexpr (IO == GDKVM_BF16) {
            // the raw rows are all the later phases read (rows past N: the clamped duplicates, as fetched -- finite, and gated out by
            // kinv = beta = 0); widened to fp32 only for the norms, and only when they did not come with the projections
            *reinterpret_cast<uint4*>(s_Kraw + n * KRP + 32 * qd) = kraw0;
            *reinterpret_cast<uint4*>(s_Kraw + n * KRP + 32 * qd + 16) = kraw1;
            if (!given) {
                const unsigned rw[8] = {kraw0.x, kraw0.y, kraw0.z, kraw0.w, kraw1.x, kraw1.y, kraw1.z, kraw1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float lo = __uint_as_float(rw[j] << 16), hi = __uint_as_float(rw[j] & 0xffff0000u);
                    sk += lo * lo + hi * hi;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) sq += ys[j][0] * ys[j][0] + ys[j][1] * ys[j][1] + ys[j][2] * ys[j][2] + ys[j][3] * ys[j][3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 x = xs[j], y = ys[j];
                if (n >= N) { x = f32x4{0.f, 0.f, 0.f, 0.f}; y = x; }
                *reinterpret_cast<f32x4*>(s_K + n * KLD + 16 * qd + 4 * j) = x;
                sk += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
                sq += y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
            }
        }
        DIAG_STAMP2(1);                                    // key rows arrived, converted, staged
        if (!given) {
            sk += __shfl_xor(sk, 1); sq += __shfl_xor(sq, 1);
            sk += __shfl_xor(sk, 2); sq += __shfl_xor(sq, 2);
        }
        if (qd == 0) {
            float kinv = 0.f, qinv = 0.f, bta = 0.f;
            if (n < N) {
                kinv = qinv = 1.f;
                if (given) {
                    kinv = nn.x; qinv = nn.y;
                } else if (a.flags & GDKVM_FLAG_NORMALIZE_QK) {
                    kinv = 1.0f / sqrtf(sk + GDKVM_EPS_NORM);
                    qinv = 1.0f / sqrtf(sq + GDKVM_EPS_NORM);
                }
                bta = bta_raw;
                if (a.flags & GDKVM_FLAG_GATE_LOGITS) bta = 1.0f / (1.0f + expf(-bta));
            }
            s_kinv[n] = kinv;
            s_beta[n] = bta;
            if constexpr (SPLIT) {
                s_beta_s[n] = bta * gscale;
                s_kinv_s[n] = kinv * gscale_inv;
            }
            a.qinv[(size_t)fh * a.np_total + tok0 + n] = qinv;
        }
    }
    DIAG_STAMP2(2);                                        // norms and gates done
    __syncthreads();
    DIAG_STAMP(1);
    if constexpr (IO == GDKVM_BF16) {
#pragma unroll
        for (int X = 0; X < NB; ++X)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                kimg[X][ks] = *reinterpret_cast<const bf16x8*>(s_Kraw + (16 * X + li) * KRP + 64 * ks + 16 * g);
    }
    if constexpr (VPRE) {}                                 // (fetched at entry)
    else if constexpr (SPLIT) load_vraw(w, vA0, vA1);     // first V tile: in flight behind phases 1-3
    else load_v(w, xv[0]);
    // bf16 I/O: this wave's 16 key channels of every token, TRANSPOSED out of the raw rows (ds_read_b64_tr_b16: a 16-lane group
    // addresses four rows x 16 columns and lane i receives column i of the four rows): four tokens of channel 16w + li per read --
    // the Kn tiles' grouping (tokens 16I + 4g + r) in four reads, the Kn^T images' (tokens 32ks + 8g + j) in four more, instead of
    // thirty-two 4-byte reads of an fp32 copy of the rows (which bf16 I/O no longer writes)
    uint2 ktr[IO == GDKVM_BF16 ? 8 : 1];
    if constexpr (IO == GDKVM_BF16) {
        const unsigned a0 = (unsigned)(uintptr_t)(s_Kraw + (4 * g + (li >> 2)) * KRP + (16 * w + 4 * (li & 3)) * 2);      // tokens 4g + q
        const unsigned a1 = (unsigned)(uintptr_t)(s_Kraw + (8 * g + (li >> 2)) * KRP + (16 * w + 4 * (li & 3)) * 2);      // tokens 8g + q
        static_assert(NB != 4 || (48 * KRP < 65536 && 36 * KRP < 65536), "offsets fit the instruction's 16 bits");
        asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                     "ds_read_b64_tr_b16 %1, %8 offset:%c10\n\t"
                     "ds_read_b64_tr_b16 %2, %8 offset:%c11\n\t"
                     "ds_read_b64_tr_b16 %3, %8 offset:%c12\n\t"
                     "ds_read_b64_tr_b16 %4, %9\n\t"
                     "ds_read_b64_tr_b16 %5, %9 offset:%c13\n\t"
                     "ds_read_b64_tr_b16 %6, %9 offset:%c14\n\t"
                     "ds_read_b64_tr_b16 %7, %9 offset:%c15\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(ktr[0]), "=&v"(ktr[1]), "=&v"(ktr[2]), "=&v"(ktr[3]), "=&v"(ktr[4]), "=&v"(ktr[5]), "=&v"(ktr[6]), "=&v"(ktr[7])
                     : "v"(a0), "v"(a1), "n"(16 * KRP), "n"(32 * KRP), "n"(48 * KRP), "n"(4 * KRP), "n"(32 * KRP), "n"(36 * KRP) : "memory");
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            xk[I][0] = __uint_as_float(ktr[I].x << 16); xk[I][1] = __uint_as_float(ktr[I].x & 0xffff0000u);
            xk[I][2] = __uint_as_float(ktr[I].y << 16); xk[I][3] = __uint_as_float(ktr[I].y & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) xk[I][r] = s_K[(16 * I + 4 * g + r) * KLD + 16 * w + li];
    }
    // bf16 I/O (round 3): the P tiles of phase 4 run on the bf16 MFMA too -- Kn^T (this wave's 16 key channels x 64 tokens) and M as
    // three-term operands, six term products per 32-token k-step: 12 MFMAs of 16 cycles per tile instead of 16 exact-fp32 ones of
    // 32, and the fp32 M^T tiles (16 KB of LDS, 64 registers) are not needed at all.  Here: the A images of Kn^T -- lane (g, li) holds
    // tokens 32 ks + 8 g .. + 7 of channel 16 w + li, the other token grouping than xk's -- which are also the B images of Kn.
    uint4 knT[SPLIT ? 3 : 1][SPLIT ? KS : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 ki0 = *reinterpret_cast<const f32x4*>(s_kinv_s + 32 * ks + 8 * g), ki1 = *reinterpret_cast<const f32x4*>(s_kinv_s + 32 * ks + 8 * g + 4);
            // (ktr[4 + 2 ks], ktr[5 + 2 ks]: tokens 32 ks + 8 g + 0..3 and + 4..7 of channel 16 w + li)
            const uint2 t0 = ktr[IO == GDKVM_BF16 ? 4 + 2 * ks : 0], t1 = ktr[IO == GDKVM_BF16 ? 5 + 2 * ks : 0];
            f32x4 v0 = {__uint_as_float(t0.x << 16), __uint_as_float(t0.x & 0xffff0000u), __uint_as_float(t0.y << 16), __uint_as_float(t0.y & 0xffff0000u)};
            f32x4 v1 = {__uint_as_float(t1.x << 16), __uint_as_float(t1.x & 0xffff0000u), __uint_as_float(t1.y << 16), __uint_as_float(t1.y & 0xffff0000u)};
            v0 *= ki0; v1 *= ki1;
            uint2 h0, m0, l0, h1, m1, l1;
            split3x4(v0, h0, m0, l0);
            split3x4(v1, h1, m1, l1);
            knT[0][ks] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            knT[1][ks] = make_uint4(m0.x, m0.y, m1.x, m1.y);
            knT[2][ks] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    }

    DIAG_STAMP2(3);                                        // Kn^T images built
    constexpr bool WAVE_BLOCKS = IO == GDKVM_BF16 && NB == 4;
    f32x4* s_T = WAVE_BLOCKS ? s_Ld : s_TmT;              // where phase 3 finds the T^T images
    if (seq && WAVE_BLOCKS) {
        // ---- phases 1 + 2, bf16 I/O: wave w takes diagonal block w -- its Gram block L_ww, then T_ww = (I + L_ww)^-1 right away, in the
        //      same wave (no barrier between the two) -- and one or two of the six blocks below the diagonal, which fill the gaps of the
        //      substitution's dependent chain.  Every block index is a compile-time constant of its wave's branch (the run-time loop
        //      over blocks picked the operand images through sixteen predicated branches per block and took 2.7 k cycles for ten blocks
        //      of two MFMAs; the substitution ran on sixteen lanes per block of wave 0 alone, the other waves waiting: 2.0 k).
        //      Substitution on all 64 lanes: lane (jg, q, c) = (lane >> 4, (lane >> 2) & 3, lane & 3) works on column j = 4 jg + c of
        //      T and holds its rows 4q .. 4q+3; row i is  delta_ij - sum_k L[i][k] T[k][j], each lane summing its own four k and the
        //      four q-lanes of a column combined by two row rotations (DPP).
        if constexpr (WAVE_BLOCKS) {
        auto gram = [&](auto Ac, auto Bc) __attribute__((always_inline)) {
            constexpr int A = decltype(Ac)::value, B = decltype(Bc)::value;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kimg[A][ks], kimg[B][ks], acc, 0, 0, 0);
            return acc;
        };
        auto below = [&](auto Ic, auto Jc) __attribute__((always_inline)) {   // lane (g,li) reg r = -L_IJ[4g+r][li], I > J
            constexpr int I = decltype(Ic)::value, J = decltype(Jc)::value;
            f32x4 acc = gram(Ic, Jc);
            const f32x4 kiI = *reinterpret_cast<const f32x4*>(s_kinv + 16 * I + 4 * g);
            const f32x4 btI = *reinterpret_cast<const f32x4*>(s_beta + 16 * I + 4 * g);
            const float colscale = s_kinv[16 * J + li];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] *= -(kiI[r] * btI[r] * colscale);
            s_negB[pair_slot(I, J) * 64 + lane] = acc;
        };
        static_for<0, 4>([&](auto wc) {
            constexpr int I = decltype(wc)::value;
            if (w != I) return;
            // L_II, row-major in this wave's 1 KB of Ld: lane (g,li) holds row li, columns 4g .. 4g+3 (zero on and above the diagonal)
            f32x4 acc = gram(wc, wc);
            {
                const float rowscale = s_kinv[16 * I + li] * s_beta[16 * I + li];
                const f32x4 kinvJ = *reinterpret_cast<const f32x4*>(s_kinv + 16 * I + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = (4 * g + r >= li) ? 0.f : acc[r] * rowscale * kinvJ[r];
            }
            s_Ld[I * 64 + li * 4 + g] = acc;
            if constexpr (I == 0) { below(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); below(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}); }
            if constexpr (I == 1) { below(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}); below(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}); }
            if constexpr (I == 2) below(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
            if constexpr (I == 3) below(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
            const int q = (lane >> 2) & 3, j = 4 * (lane >> 4) + (lane & 3);
            // (rows of L fetched LG at a time: all fifteen at once in the register-rich build, four where registers are short)
            constexpr int LG = (W3 || FUSE) ? 4 : 15;
            f32x4 tq = {0.f, 0.f, 0.f, 0.f};                                          // T[4q + r][j]
            if (q == 0) tq[0] = (j == 0) ? 1.f : 0.f;
#pragma unroll
            for (int i0 = 1; i0 < 16; i0 += LG) {
                f32x4 Lr[LG];
#pragma unroll
                for (int i = i0; i < i0 + LG && i < 16; ++i) Lr[i - i0] = s_Ld[I * 64 + i * 4 + q];   // L[i][4q .. 4q+3]  (written above by this wave's own lanes)
#pragma unroll
                for (int i = i0; i < i0 + LG && i < 16; ++i) {
                    float sm = Lr[i - i0][0] * tq[0];
                    sm = fmaf(Lr[i - i0][1], tq[1], sm);
                    sm = fmaf(Lr[i - i0][2], tq[2], sm);
                    sm = fmaf(Lr[i - i0][3], tq[3], sm);
                    sm += __int_as_float(dpp_i32<0x124>(__float_as_int(sm)));
                    sm += __int_as_float(dpp_i32<0x128>(__float_as_int(sm)));
                    const float val = (i > j) ? -sm : ((i == j) ? 1.f : 0.f);
                    if (q == (i >> 2)) tq[i & 3] = val;
                }
            }
            // lane (q, j) reg r = T[4q + r][j]: the A image of T^T -- over this wave's own L block (every lane has its rows in registers),
            // NOT in the T^T region: the raw key rows live there until the barrier below, and another wave may still be fetching its images
            s_T[I * 64 + q * 16 + j] = tq;
        });
        }
        DIAG_STAMP(2);
        __syncthreads();
    } else if (seq) {
        // ---- phase 1
        for (int p = w; p < NB * (NB + 1) / 2; p += 4) {
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= p) ++I;
            const int J = p - I * (I + 1) / 2;                                  // J <= I
            const int nI = 16 * I + li, nJ = 16 * J + li;
            f32x4 kI[4], kJ[4];
            if constexpr (IO != GDKVM_BF16) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    kI[m] = *reinterpret_cast<const f32x4*>(s_K + nI * KLD + 16 * m + 4 * g);
                    kJ[m] = *reinterpret_cast<const f32x4*>(s_K + nJ * KLD + 16 * m + 4 * g);
                }
            }
            // (bf16 I/O: block images picked by a compile-time index -- a run-time one would put the register array in scratch)
            auto gram16 = [&](int A, int B) __attribute__((always_inline)) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                static_for<0, NB>([&](auto ac) {
                    static_for<0, NB>([&](auto bc) {
                        if (decltype(ac)::value == A && decltype(bc)::value == B) {
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
                                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kimg[IO == GDKVM_BF16 ? decltype(ac)::value : 0][ks],
                                                                              kimg[IO == GDKVM_BF16 ? decltype(bc)::value : 0][ks], acc, 0, 0, 0);
                        }
                    });
                });
                return acc;
            };
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (J == I) {                      // lane (g,li) reg r = k_{I,4g+r} . k_{I,li}: the image of L_II[li][4g+r]
                if constexpr (IO == GDKVM_BF16) acc0 = gram16(J, I);
                else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (m & 1) acc1 = mfma4(kJ[m][r], kI[m][r], acc1);
                        else acc0 = mfma4(kJ[m][r], kI[m][r], acc0);
                    }
                }
                f32x4 acc = acc0 + acc1;
                const float rowscale = s_kinv[nI] * s_beta[nI];
                const f32x4 kinvJ = *reinterpret_cast<const f32x4*>(s_kinv + 16 * J + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = (4 * g + r >= li) ? 0.f : acc[r] * rowscale * kinvJ[r];
                s_Ld[I * 64 + lane] = acc;
            } else {                           // lane (g,li) reg r = L_IJ[4g+r][li] = b_{I,4g+r} kn_{I,4g+r} . kn_{J,li}  (I > J)
                if constexpr (IO == GDKVM_BF16) acc0 = gram16(I, J);
                else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (m & 1) acc1 = mfma4(kI[m][r], kJ[m][r], acc1);
                        else acc0 = mfma4(kI[m][r], kJ[m][r], acc0);
                    }
                }
                f32x4 acc = acc0 + acc1;
                const f32x4 kiI = *reinterpret_cast<const f32x4*>(s_kinv + 16 * I + 4 * g);
                const f32x4 btI = *reinterpret_cast<const f32x4*>(s_beta + 16 * I + 4 * g);
                const float colscale = s_kinv[nJ];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] *= -(kiI[r] * btI[r] * colscale);
                s_negB[pair_slot(I, J) * 64 + lane] = acc;
            }
        }
        __syncthreads();
        DIAG_STAMP(2);
        // ---- phase 2: T_II by forward substitution (16 threads per block, one column each), stored transposed
        {
            const int I = tid >> 4, j = tid & 15;
            if (I < NB) {
                float t[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = (i == j) ? 1.f : 0.f;
#pragma unroll
                for (int i = 1; i < 16; ++i) {
                    float sm = 0.f;
#pragma unroll
                    for (int gg = 0; gg * 4 < i; ++gg) {
                        const f32x4 Lr = s_Ld[I * 64 + gg * 16 + i];
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (4 * gg + r < i) sm += Lr[r] * t[4 * gg + r];
                    }
                    t[i] = (i > j) ? -sm : t[i];
                }
                float* Tm = reinterpret_cast<float*>(s_TmT + I * 64);
#pragma unroll
                for (int i = 0; i < 16; ++i) Tm[((i >> 2) * 16 + j) * 4 + (i & 3)] = t[i];   // lane (i>>2, j) reg i&3 = T[i][j]
            }
        }
        __syncthreads();
    }
    DIAG_STAMP(3);

    // ---- phase 3: back substitution on Kn column tile w
    if (!seq) __syncthreads();                            // every wave has taken its Kn tile out of the staging tile (kni aliases it)
    f32x4 KN[NB], Z[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        const f32x4 ki4 = *reinterpret_cast<const f32x4*>(s_kinv + 16 * I + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) KN[I][r] = xk[I][r] * ki4[r];
    }
    if (seq) {
        // k-step r of a product with token block J covers its tokens 4g + r: in the LAST block only the steps r < N - 16 (NB - 1) meet a
        // real token (rows of padding tokens are exactly zero in Kn, hence in Z) -- 49 tokens: one step of four, 12 of the 40 MFMAs less
        const int rl = min(max(N - 16 * (NB - 1), 0), 4);
        static_for<0, NB>([&](auto ic) {
            constexpr int I = NB - 1 - decltype(ic)::value;
            f32x4 acc = KN[I];
            static_for<I + 1, NB>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                const f32x4 nb = s_negB[pair_slot(J, I) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (J < NB - 1 || r < rl) acc = mfma4(nb[r], Z[J][r], acc);
            });
            const f32x4 t4 = s_T[I * 64 + lane];
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (I < NB - 1 || r < rl) z = mfma4(t4[r], acc[r], z);
            Z[I] = z;
        });
    } else {
#pragma unroll
        for (int I = 0; I < NB; ++I) Z[I] = KN[I];
    }
    f32x4 btI[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) btI[I] = *reinterpret_cast<const f32x4*>((SPLIT ? s_beta_s : s_beta) + 16 * I + 4 * g);
    // (COMPACT: m3 below overwrites the K rows in R2 -- every wave took its copies right behind the phase 0 barrier and has passed
    // the barriers of phases 1 and 2, or the one in front of phase 3, since)
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        const f32x4 bt4 = btI[I];
        const f32x4 mtI = Z[I] * bt4;
        if constexpr (!SPLIT) s_mt[(w * NB + I) * 64 + lane] = mtI;      // (fp32 I/O: the exact fp32 products of phase 4 read M^T back)
        if constexpr (SPLIT) {             // this lane's 4 tokens 16I+4g+r of row 16w+li are half (g&1) of A lane (2(I&1)+(g>>1), li), ks = I>>1
            __bf16 hh[4], mm[4], ll[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) split3(mtI[r], hh[r], mm[r], ll[r]);
            const int slot = w * KS * 128 + split_slot(I, g, li);
            s_m3[slot] = pack_bf16x4(hh);
            s_m3[4 * KS * 128 + slot] = pack_bf16x4(mm);
            s_m3[2 * 4 * KS * 128 + slot] = pack_bf16x4(ll);
        }
    }
    __syncthreads();
    DIAG_STAMP(4);

    // ---- phase 4: P images and G tiles
    const int nlast = N - 16 * (NB - 1);                  // real tokens in the last block (<= 0: none)
    f32x4 mt[SPLIT ? 1 : 4][SPLIT ? 1 : NB];
    if constexpr (!SPLIT) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int I = 0; I < NB; ++I) mt[m][I] = s_mt[(m * NB + I) * 64 + lane];
    }
    // SPLIT: row tile m of M as three term images, [term][k step] (A images of M = B images of M^T)
    auto m_terms = [&](int m, uint4 (&d)[3][SPLIT ? KS : 1]) __attribute__((always_inline)) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
#pragma unroll
            for (int ks = 0; ks < (SPLIT ? KS : 1); ++ks)
                d[sp][ks] = *reinterpret_cast<const uint4*>(&s_m3[sp * 4 * KS * 128 + ((m * KS + ks) * 64 + lane) * 2]);
    };
    // chunk 0 of a chunked frame starts the composition as [P | G] accumulator tiles; under delta_parallel (every token sees
    // the frame's old state) the chunks combine additively, P = I - sum_c (I - P_c), so all of them are written as tiles
    const bool first_of_many = nchunk > 1 && chunk == 0;
    const bool p_tiles = nchunk > 1 && (chunk == 0 || (!FUSE && a.rule == GDKVM_RULE_DELTA_PARALLEL));
    // FUSE: P_c of a later chunk goes to LDS as A images (over the M^T tiles, which every wave holds in registers by then), and
    // each wave re-splits its columns of the running map through a private 4 KB tile (waves 0-2: over the Gram / T blocks, which
    // phase 3 is done with; wave 3: 4 KB appended to the allocation)
    uint2* s_pc = reinterpret_cast<uint2*>(s_mt);
    uint2* s_wx = w < 3 ? reinterpret_cast<uint2*>(s_negB) + w * (NT * SPLIT_IMG) : s_m3 + 3 * 4 * KS * 128;
    static_assert(!FUSE || 3 * NT * SPLIT_IMG * 8 <= (NB * (NB - 1) / 2 + 2 * NB) * 1024, "three wave tiles fit the Gram / T blocks");
    if constexpr (FUSE) __syncthreads();                  // every wave has its M^T tiles in registers: their LDS is free
    uint2* pp = (chunk == 0 ? reinterpret_cast<uint2*>(a.pp) + (size_t)fh * (4 * 3 * SPLIT_IMG)
                            : reinterpret_cast<uint2*>(a.ppc) + ((size_t)fh * (nchunk - 1) + chunk - 1) * (4 * 3 * SPLIT_IMG));
    f32x4* x0 = reinterpret_cast<f32x4*>(a.x0) + (size_t)fh * (4 + nsl) * 4 * 64;
    f32x4* ptile = chunk == 0 ? x0 : reinterpret_cast<f32x4*>(pp);      // P tiles of a later chunk take the place of its images
    // chunk-parallel bf16 path: all four row tiles of M's term images are fetched ONCE, here, for the P tiles and for the G tiles of
    // phase 4b (which kept them in 96 registers anyway): the four P products are then independent MFMA chains with no LDS round trip
    // between them (stamps: the P phase took 4.3 k cycles for 0.8 k of MFMA issue with the images re-read per row tile)
    constexpr bool MALL = SPLIT && !FUSE && !W3;
    uint4 mall[MALL ? 4 : 1][3][SPLIT ? KS : 1];
    if constexpr (MALL) {
#pragma unroll
        for (int m = 0; m < 4; ++m) m_terms(m, mall[m]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        uint4 mterm[3][SPLIT ? KS : 1];
        if constexpr (SPLIT) {
            if (!p_identity) {
                if constexpr (MALL) {
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) mterm[sp][ks] = mall[m][sp][ks];
                } else m_terms(m, mterm);
                if (!p_tiles) acc0 = OpFmt<FMT_SPLIT3>::product(knT, mterm);      // (Kn^T)_w (M^T)_m: rows = key channels, columns = rows of M
            }
        } else if (!p_identity) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int I = 0; I < NB - 1; ++I) {
                    if (I & 1) acc1 = mfma4(KN[I][r], mt[m][I][r], acc1);
                    else acc0 = mfma4(KN[I][r], mt[m][I][r], acc0);
                }
            // k-step (I, r) covers tokens 16I + 4g + r: in the last block only the first nlast steps hold real tokens
            if (nlast > 0) { acc0 = mfma4(KN[NB - 1][0], mt[m][NB - 1][0], acc0);
                if (nlast > 1) { acc1 = mfma4(KN[NB - 1][1], mt[m][NB - 1][1], acc1);
                    if (nlast > 2) { acc0 = mfma4(KN[NB - 1][2], mt[m][NB - 1][2], acc0);
                        if (nlast > 3) acc1 = mfma4(KN[NB - 1][3], mt[m][NB - 1][3], acc1); } } }
        }
        if (p_tiles) {                         // the composition wants chunk 0 as [P | G] accumulator tiles: P tile (m, w),
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};          // the same product with the operands swapped
            if (!p_identity) {
                if constexpr (SPLIT) b0 = OpFmt<FMT_SPLIT3>::product(mterm, knT);
                else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int I = 0; I < NB; ++I) {
                        if (I & 1) b1 = mfma4(mt[m][I][r], KN[I][r], b1);
                        else b0 = mfma4(mt[m][I][r], KN[I][r], b0);
                    }
                }
            }
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = ((16 * m + 4 * g + r == 16 * w + li) ? 1.f : 0.f) - (b0[r] + b1[r]);
            if constexpr (FUSE) X[m][0] = o;
            else ptile[(w * 4 + m) * 64 + lane] = o;
            continue;
        }
        // lane (g,li) reg r = P[16m + li][k = 16w + 4g + r]: four consecutive k of row li of row tile m -> its term images
        f32x4 pv = -(acc0 + acc1);
        if (m == w) {                                      // (uniform: the identity meets only the diagonal tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = ((li == 4 * g + r) ? 1.f : 0.f) - (acc0[r] + acc1[r]);
        }
        uint2 tt[3];
        OpFmt<FMT>::split4(pv, tt);
        const int e = split_slot(w, g, li);
#pragma unroll
        for (int sp = 0; sp < NT; ++sp) {
            if constexpr (FUSE) s_pc[(m * NT + sp) * SPLIT_IMG + e] = tt[sp];
            else pp[(m * NT + sp) * SPLIT_IMG + e] = tt[sp];
        }
    }
    DIAG_STAMP(5);
    // FUSE, chunk c >= 1: column tile j of the running map as B images (this wave's private tile), then row tile m of
    // P_c X[:, j] (+ the chunk's own G tile) -- gdr_compose_kernel's step, without leaving the workgroup
    auto col_images = [&](auto jc, uint4 (&xb)[NT][2]) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value < XJ ? decltype(jc)::value : 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            uint2 tt[3];
            if constexpr (FUSE) xsplit_max = fmaxf(xsplit_max, absmax4(X[m][j]));      // (what goes into fp16 pairs here: checked at the end)
            OpFmt<FMT>::split4(X[FUSE ? m : 0][j] * OpFmt<FMT>::STATE, tt);
            const int e = split_slot(m, g, li);
#pragma unroll
            for (int sp = 0; sp < NT; ++sp) s_wx[sp * SPLIT_IMG + e] = tt[sp];
        }
#pragma unroll
        for (int sp = 0; sp < NT; ++sp)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xb[sp][ks] = *reinterpret_cast<const uint4*>(&s_wx[sp * SPLIT_IMG + (ks * 64 + lane) * 2]);
    };
    auto pc_times = [&](int m, const uint4 (&xb)[NT][2]) __attribute__((always_inline)) {
        uint4 pa[NT][2];
#pragma unroll
        for (int sp = 0; sp < NT; ++sp)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) pa[sp][ks] = *reinterpret_cast<const uint4*>(&s_pc[(m * NT + sp) * SPLIT_IMG + (ks * 64 + lane) * 2]);
        return OpFmt<FMT>::STATE_INV * OpFmt<FMT>::product(pa, xb);
    };
    if constexpr (FUSE) {
        if (chunk > 0) {
            __syncthreads();                               // P_c complete in LDS
            uint4 xb[NT][2];
            col_images(std::integral_constant<int, 0>{}, xb);
#pragma unroll
            for (int m = 0; m < 4; ++m) X[m][0] = pc_times(m, xb);
        }
    }
    f32x4* gg = first_of_many ? x0 + 4 * 4 * 64
              : (chunk == 0 ? reinterpret_cast<f32x4*>(a.gg) + (size_t)fh * nsl * 4 * 64
                            : reinterpret_cast<f32x4*>(a.ggc) + ((size_t)fh * (nchunk - 1) + chunk - 1) * nsl * 4 * 64);
    // the frame's FINAL map (one chunk): max |G| per slice goes to the scan, which sizes the state's fp16-pair exponent by it
    const bool final_g = nchunk == 1;
    auto g_tiles = [&](int cV, const float (&x)[SPLIT ? 1 : NB][4]) __attribute__((always_inline)) {
        float gm = 0.f;
        if constexpr (!SPLIT)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int I = 0; I < NB - 1; ++I) {
                    if (I & 1) acc1 = mfma4(mt[m][I][r], x[I][r], acc1);
                    else acc0 = mfma4(mt[m][I][r], x[I][r], acc0);
                }
            if (nlast > 0) { acc0 = mfma4(mt[m][NB - 1][0], x[NB - 1][0], acc0);
                if (nlast > 1) { acc1 = mfma4(mt[m][NB - 1][1], x[NB - 1][1], acc1);
                    if (nlast > 2) { acc0 = mfma4(mt[m][NB - 1][2], x[NB - 1][2], acc0);
                        if (nlast > 3) acc1 = mfma4(mt[m][NB - 1][3], x[NB - 1][3], acc1); } } }
            if (cV < nsl) gg[((size_t)cV * 4 + m) * 64 + lane] = (acc0 + acc1) * gscale;
            gm = fmaxf(gm, absmax4(acc0 + acc1));
        }
        if (final_g && cV < nsl) {
            gm = wave_max_nonneg(gm);
            const int gi = __builtin_amdgcn_readfirstlane((fh * nsl + cV) * 4);
            if (lane == 0) *reinterpret_cast<f32x4*>(a.gmax + gi) = f32x4{gm, 0.f, 0.f, 0.f};
        }
    };
    if constexpr (SPLIT) {
        // the A images of M: resident for all four row tiles (96 registers), or -- FUSE, where the running map takes 80 -- re-read
        // per row tile (six 16-byte LDS reads against twelve MFMAs)
        bf16x8 am[FUSE ? 1 : 4][KS][3];
        auto am_load = [&](int m, bf16x8 (&d)[KS][3]) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    d[ks][sp] = *reinterpret_cast<const bf16x8*>(&s_m3[sp * 4 * KS * 128 + ((m * KS + ks) * 64 + lane) * 2]);
        };
        // (COMPACT: R1 stages the V tiles from here on; every wave left its phase 0-3 scratch behind at the barrier that ends phase 3)
        if constexpr (MALL) {                              // (the copies fetched in front of the P tiles)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) am[m][ks][sp] = __builtin_bit_cast(bf16x8, mall[m][sp][ks]);
        } else if constexpr (!FUSE) {
#pragma unroll
            for (int m = 0; m < 4; ++m) am_load(m, am[m]);
        }
        // emit(m, tile): what becomes of G tile (m, cV) -- stored (one chunk, or a chunk map), or folded into the running map
        auto g_tiles3 = [&](int cV, const bf16x8 (&x)[KS], auto&& emit) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if constexpr (FUSE) am_load(m, am[0]);
                const bf16x8 (&amm)[KS][3] = am[FUSE ? 0 : m];
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {          // smallest terms first
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amm[ks][2], x[ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amm[ks][1], x[ks], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amm[ks][0], x[ks], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amm[ks][0], x[ks], acc0, 0, 0, 0);
                }
                emit(m, acc0 + acc1);
            }
        };
        char* vst = reinterpret_cast<char*>(s_kni) + w * 4096;          // two wave-private 2 KiB tiles (the K staging area is free)
        auto stage = [&](int buf, const uint4& d0, const uint4& d1) __attribute__((always_inline)) {
            *reinterpret_cast<uint4*>(vst + buf * 2048 + lane * 16) = d0;              // piece p = lane: token p>>1, half p&1
            *reinterpret_cast<uint4*>(vst + buf * 2048 + 1024 + lane * 16) = d1;       // piece p = lane + 64
        };
        auto read_b = [&](int buf, bf16x8 (&x)[KS]) __attribute__((always_inline)) {
            // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the group's 4-row block; the group of lanes
            // 16g.. takes rows 32ks + 8g + 4*half + (0..3); lane i receives column i of those rows
            const unsigned addr = (unsigned)(uintptr_t)(vst + buf * 2048 + ((8 * g + (li >> 2)) * 16 + 4 * (li & 3)) * 2);
            uint2 r00, r01, r10, r11;
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                         "ds_read_b64_tr_b16 %1, %4 offset:128\n\t"
                         "ds_read_b64_tr_b16 %2, %4 offset:1024\n\t"
                         "ds_read_b64_tr_b16 %3, %4 offset:1152\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r00), "=&v"(r01), "=&v"(r10), "=&v"(r11) : "v"(addr) : "memory");
            x[0] = __builtin_bit_cast(bf16x8, make_uint4(r00.x, r00.y, r01.x, r01.y));
            x[1] = __builtin_bit_cast(bf16x8, make_uint4(r10.x, r10.y, r11.x, r11.y));
        };
        bf16x8 xb[KS];
        if constexpr (!FUSE) {
            auto g_pair = [&](int cV, uint4& a0, uint4& a1, uint4& b0, uint4& b1, int nextA, int nextB) __attribute__((always_inline)) {
                if (nextB >= 0) load_vraw(nextB, b0, b1);     // (one-ahead scheme: tile B is requested here, tile A of the next trip below)
                stage(0, a0, a1);
                read_b(0, xb);
                float gmA = 0.f, gmB = 0.f;
                g_tiles3(cV, xb, [&](int m, const f32x4& t) { if (cV < nsl) gg[((size_t)cV * 4 + m) * 64 + lane] = t; absmax4_into(gmA, t); });
                if (nextA >= 0) load_vraw(nextA, a0, a1);
                stage(1, b0, b1);
                read_b(1, xb);
                g_tiles3(cV + 4, xb, [&](int m, const f32x4& t) { if (cV + 4 < nsl) gg[((size_t)(cV + 4) * 4 + m) * 64 + lane] = t; absmax4_into(gmB, t); });
                if (final_g) {
                    gmA = wave_max_nonneg(gmA) * gscale_inv; gmB = wave_max_nonneg(gmB) * gscale_inv;   // (the tiles came out of the MFMA scaled)
                    // (the index as a scalar: as a vector expression its 64-bit address lived in registers across the loop and spilled)
                    const int gi = __builtin_amdgcn_readfirstlane((fh * nsl + cV) * 4);
                    if (lane == 0) {
                        if (cV < nsl) *reinterpret_cast<f32x4*>(a.gmax + gi) = f32x4{gmA, 0.f, 0.f, 0.f};
                        if (cV + 4 < nsl) *reinterpret_cast<f32x4*>(a.gmax + gi + 16) = f32x4{gmB, 0.f, 0.f, 0.f};
                    }
                }
            };
            if constexpr (VPRE) {
                // tiles w, w+4 (A, B) and w+8, w+12 (C, D) are here or on their way; a wider V goes on in the same four registers pairs,
                // each refilled as soon as its tile is staged
                for (int cV = w; cV < nsl; cV += 16) {
                    const bool more = cV + 16 < nsl;       // (uniform)
                    g_pair(cV, vA0, vA1, vB0, vB1, -1, -1);
                    if (more) { load_vraw(cV + 16, vA0, vA1); load_vraw(cV + 20, vB0, vB1); }
                    if (cV + 8 < nsl) g_pair(cV + 8, vC0, vC1, vD0, vD1, -1, -1);
                    if (more) { load_vraw(cV + 24, vC0, vC1); load_vraw(cV + 28, vD0, vD1); }
                }
            } else {
                for (int cV = w; cV < nsl; cV += 8) g_pair(cV, vA0, vA1, vB0, vB1, cV + 8, cV + 4);
            }
        } else {
            // column tile cV = w + 4 (j - 1) of G is X[.][j], j = 1 .. 4 (Dv <= 256): compile-time indices, run-time bounds.
            // Two column tiles per trip share every row tile's operand fetch -- M's term images (6 reads) and P_c's (4): with one
            // column per fetch the walk read 4x the chunk-parallel kernel's LDS bytes and two co-resident workgroups got in each
            // other's way (measured: 69 us for 256 frames, 107 us for 512).
            auto gtile = [&](const bf16x8 (&amm)[KS][3], const bf16x8 (&x)[KS]) __attribute__((always_inline)) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {          // smallest terms first
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amm[ks][2], x[ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amm[ks][1], x[ks], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amm[ks][0], x[ks], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amm[ks][0], x[ks], acc0, 0, 0, 0);
                }
                return acc0 + acc1;
            };
            static_for<0, 2>([&](auto tc) {
                constexpr int t2 = decltype(tc)::value, jA = 1 + 2 * t2, jB = 2 + 2 * t2;
                const int cV = w + 8 * t2;
                if (cV < nsl) {
                    const bool twoB = cV + 4 < nsl;
                    bf16x8 xbB[KS];
                    stage(0, vA0, vA1);
                    if (t2 == 0) {                          // (the first trip's second tile is not in flight yet)
                        load_vraw(cV + 4, vB0, vB1);
                        load_vraw(cV + 8, vA0, vA1);
                        stage(1, vB0, vB1);
                        load_vraw(cV + 12, vB0, vB1);
                    } else {
                        stage(1, vB0, vB1);
                    }
                    read_b(0, xb);
                    read_b(1, xbB);
                    uint4 xiA[NT][2], xiB[NT][2];
                    if (chunk > 0) {
                        col_images(std::integral_constant<int, jA>{}, xiA);
                        col_images(std::integral_constant<int, jB>{}, xiB);
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        am_load(m, am[0]);
                        const f32x4 tA = gtile(am[0], xb), tB = gtile(am[0], xbB);
                        if (chunk == 0) {
                            X[m][jA] = tA;
                            X[m][jB] = tB;
                        } else {
                            uint4 pa[NT][2];
#pragma unroll
                            for (int sp = 0; sp < NT; ++sp)
#pragma unroll
                                for (int ks = 0; ks < 2; ++ks) pa[sp][ks] = *reinterpret_cast<const uint4*>(&s_pc[(m * NT + sp) * SPLIT_IMG + (ks * 64 + lane) * 2]);
                            X[m][jA] = OpFmt<FMT>::STATE_INV * OpFmt<FMT>::product(pa, xiA) + tA;
                            if (twoB) X[m][jB] = OpFmt<FMT>::STATE_INV * OpFmt<FMT>::product(pa, xiB) + tB;
                        }
                    }
                }
            });
        }
    } else {
        for (int cV = w; cV < nsl; cV += 8) {     // two V tiles per trip: the next tile's loads are in flight behind the MFMAs
            load_v(cV + 4, xv[1]);
            g_tiles(cV, xv[0]);
            load_v(cV + 8, xv[0]);
            g_tiles(cV + 4, xv[1]);
        }
    }
    DIAG_STAMP(6);
#if defined(GDKVM_DIAG) && defined(GDKVM_DIAG_SPAN)
    __syncthreads();
    if (a.diag && tid == 0) {
        unsigned long long t__;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");
        a.diag[(size_t)(a.T + 1) * 8 + 2 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 1] = t__;
    }
#endif
    if constexpr (FUSE) __syncthreads();                  // the next chunk's staging tile overwrites what this one's phase 4 read
    }   // chunks
    if constexpr (FUSE) {
        // the frame's map in the formats the scan consumes (as gdr_compose_kernel's last step): P tile (m, w) -> the term images of
        // row tile m, G tiles in the scan's scale
        const int nsl = a.Dv / 16;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            unsigned short* img = reinterpret_cast<unsigned short*>(a.pp) + (size_t)fh * (4 * 3 * SPLIT_IMG * 4) + m * (NT * SPLIT_IMG * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned short tt[3];
                OpFmt<FMT>::split1(X[m][0][r], tt);
                const int e = split_slot(w, li >> 2, 4 * g + r) * 4 + (li & 3);
#pragma unroll
                for (int sp = 0; sp < NT; ++sp) img[sp * SPLIT_IMG * 4 + e] = tt[sp];
            }
        }
        f32x4* gout = reinterpret_cast<f32x4*>(a.gg) + (size_t)fh * nsl * 4 * 64;
        // an intermediate of the composition beyond the fp16 pair's range (|x| 2^-4 >= 65504: values ~1e6 times the usual) is
        // reported as +inf and the scan answers with NaNs -- never a silently saturated map
        const bool sat = wave_max_nonneg(xsplit_max) * OpFmt<FMT>::STATE >= PAIR_SAT;
        static_for<1, 5>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int cV = w + 4 * (j - 1);
            if (cV < nsl) {
                float gm = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    gout[((size_t)cV * 4 + m) * 64 + lane] = X[m][j] * OpFmt<FMT>::STATE;
                    gm = fmaxf(gm, absmax4(X[m][j]));
                }
                gm = sat ? __builtin_inff() : wave_max_nonneg(gm);
                if (lane == 0) *reinterpret_cast<f32x4*>(a.gmax + ((size_t)fh * nsl + cV) * 4) = f32x4{gm, 0.f, 0.f, 0.f};
            }
        });
    }
}

template <int NB, int IO, int FMT, bool FUSE = false, bool W3 = false>
int launch_prepm_fmt(const PrepMArgs& pa, int FH, int nchunk, hipStream_t st)
{
    size_t lds = prepm_lds_bytes(NB, IO, FUSE);
    // diagnostics (tools/n4_bench.py fold): extra KiB of LDS per workgroup, to measure what a design that keeps more per frame
    // resident -- the projections folded into this kernel need the frame's feature and value tiles: 104 KiB, one workgroup per CU --
    // would pay in occupancy before any of its own work
    static const int lds_pad_kb = [] { const char* e = getenv("GDKVM_PREP_LDS_PAD_KB"); return e ? atoi(e) : 0; }();
    if (lds_pad_kb > 0 && lds + (size_t)lds_pad_kb * 1024 <= 160 * 1024) lds += (size_t)lds_pad_kb * 1024;
    if (lds > 64 * 1024) {
        // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device; lock-free cache as in gdr_scan.hip
        static std::atomic<unsigned long long> done_mask{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "gdr_prepm: hipGetDevice");
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gdr_prepm_kernel<NB, IO, FMT, FUSE, W3>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "gdr_prepm: LDS attribute: %s", hipGetErrorString(e));
            done_mask.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    PrepMArgs la = pa;
    dim3 grid(FH, FUSE ? 1 : nchunk);
    la.grid3 = 0;
    if (!FUSE && nchunk == 1 && pa.Hh == 1 && pa.T <= 65535 && FH % (8 * pa.T) == 0 && FH / (8 * pa.T) <= 65535) {
        la.grid3 = 1;
        grid = dim3(8, pa.T, FH / (8 * pa.T));
    }
    hipLaunchKernelGGL((gdr_prepm_kernel<NB, IO, FMT, FUSE, W3>), grid, dim3(256), lds, st, la);
    GDKVM_LAUNCH_CHECK("gdr_prepm_kernel");
    return GDKVM_OK;
}
template <int NB, int IO>
int launch_prepm(const PrepMArgs& pa, int FH, int nchunk, bool wide, bool fuse, hipStream_t st)
{
    if constexpr (prepm_split(NB, IO)) {
        if (fuse && !wide) return launch_prepm_fmt<NB, IO, FMT_PAIR16, true>(pa, FH, nchunk, st);
        // more workgroups than two per CU hold: the three-per-CU build (fewer registers, M's images re-read per row tile)
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        }
        bool w3 = (long)FH * nchunk > 2L * cus;
        if (const char* e = getenv("GDKVM_PREP_W3")) w3 = e[0] == '1';           // "0" / "1": overrides the choice (A/B runs; same results)
        if (w3)
            return wide ? launch_prepm_fmt<NB, IO, FMT_SPLIT3, false, true>(pa, FH, nchunk, st) : launch_prepm_fmt<NB, IO, FMT_PAIR16, false, true>(pa, FH, nchunk, st);
    }
    return wide ? launch_prepm_fmt<NB, IO, FMT_SPLIT3>(pa, FH, nchunk, st) : launch_prepm_fmt<NB, IO, FMT_PAIR16>(pa, FH, nchunk, st);
}

// ------------------------------------------------------------------------------------------------------------------
// gdr_compose_kernel -- frames of more than 64 tokens.  The tokens of a frame act on the state in order, so the frame's
// affine map is the composition of its 64-token chunks' maps:  [P | G] <- P_c [P | G] + [0 | G_c],  c = 1 .. nchunk-1,
// starting from chunk 0.  Columns never mix: one workgroup carries four column tiles of [P | G] (64 x 64 fp32 in
// accumulators, wave w = row tile w) through all steps, exactly like the serial scan carries S -- P_c as three-term A
// images (as prepm wrote them), the running columns re-split into three-term B images through LDS each step.
// Output: the final P as term images (pp) and G as accumulator images (gg), the formats the scan consumes.
struct ComposeArgs { const float* x0; const float* ppc; const float* ggc; float* pp; float* gg; int Dv, nchunk, additive; float* gmax; };

template <int FMT>
__global__ __launch_bounds__(256, 2) void gdr_compose_kernel(ComposeArgs a)       // (256 registers: MFMA results in VGPRs; with 512 they land in AGPRs and are copied out)
{
    constexpr int NT = fmt_terms(FMT);
    __shared__ __attribute__((aligned(16))) uint2 s_X3[4 * NT * SPLIT_IMG];       // [col tile j][term] B images (pair16: at 2^-4)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t fh = blockIdx.x;
    const int nsl = a.Dv / 16, ncol = 4 + nsl, c0 = 4 * blockIdx.y;               // this block's column tiles c0 .. c0+3 of [P | G]
    f32x4 X[4];
    float xsplit_max = 0.f;                                // pair16: largest |entry| this wave re-split into fp16 pairs (checked at the end)
    const f32x4* x0 = reinterpret_cast<const f32x4*>(a.x0) + fh * ncol * 4 * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) X[j] = x0[((size_t)min(c0 + j, ncol - 1) * 4 + w) * 64 + lane];
    for (int c = 1; c < a.nchunk && a.additive; ++c) {     // delta_parallel: [P | G] += [P_c - I | G_c]  (ppc holds P_c tiles)
        const size_t ci = fh * (a.nchunk - 1) + (c - 1);
        const f32x4* pc = reinterpret_cast<const f32x4*>(a.ppc) + ci * (size_t)(GDKVM_DK * GDKVM_DK * 3 / 8);
        const f32x4* gc = reinterpret_cast<const f32x4*>(a.ggc) + ci * nsl * 4 * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = min(c0 + j, ncol - 1);
            f32x4 d = col < 4 ? pc[(col * 4 + w) * 64 + lane] : gc[((size_t)(col - 4) * 4 + w) * 64 + lane];
            if (col < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] -= (16 * w + 4 * g + r == 16 * col + li) ? 1.f : 0.f;
            }
            X[j] += d;
        }
    }
    for (int c = 1; c < a.nchunk && !a.additive; ++c) {
        const size_t ci = fh * (a.nchunk - 1) + (c - 1);
        const uint4* pimg = reinterpret_cast<const uint4*>(a.ppc) + ci * (4 * 3 * 2 * 64) + w * (NT * 2 * 64) + lane;
        uint4 pa[NT][2];
#pragma unroll
        for (int sp = 0; sp < NT; ++sp)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) pa[sp][ks] = pimg[(sp * 2 + ks) * 64];
        const f32x4* gc = reinterpret_cast<const f32x4*>(a.ggc) + ci * nsl * 4 * 64;
        f32x4 gadd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c0 + j;
            gadd[j] = gc[((size_t)min(max(col - 4, 0), nsl - 1) * 4 + w) * 64 + lane];
            if (col < 4) gadd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                      // rows 16w + 4g + r of column tile j -> its term images
            uint2 tt[3];
            if constexpr (FMT == FMT_PAIR16) xsplit_max = fmaxf(xsplit_max, absmax4(X[j]));
            OpFmt<FMT>::split4(X[j] * OpFmt<FMT>::STATE, tt);
            const int e = split_slot(w, g, li);
#pragma unroll
            for (int sp = 0; sp < NT; ++sp) s_X3[(j * NT + sp) * SPLIT_IMG + e] = tt[sp];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 xb[NT][2];
#pragma unroll
            for (int sp = 0; sp < NT; ++sp)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) xb[sp][ks] = *reinterpret_cast<const uint4*>(&s_X3[(j * NT + sp) * SPLIT_IMG + (ks * 64 + lane) * 2]);
            X[j] = OpFmt<FMT>::STATE_INV * OpFmt<FMT>::product(pa, xb) + gadd[j];
        }
        __syncthreads();                                   // the images are rewritten in the next step
    }
    const bool sat = wave_max_nonneg(xsplit_max) * OpFmt<FMT>::STATE >= PAIR_SAT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = c0 + j;
        if (col >= ncol) continue;
        if (col < 4) {                                     // P[16w + 4g + r][k = 16 col + li] -> term images of row tile w
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned short tt[3];
                OpFmt<FMT>::split1(X[j][r], tt);
                unsigned short* img = reinterpret_cast<unsigned short*>(a.pp) + fh * (size_t)(4 * 3 * SPLIT_IMG * 4) + w * (NT * SPLIT_IMG * 4);
                const int e = split_slot(col, li >> 2, 4 * g + r) * 4 + (li & 3);
#pragma unroll
                for (int sp = 0; sp < NT; ++sp) img[sp * SPLIT_IMG * 4 + e] = tt[sp];
            }
        } else {
            reinterpret_cast<f32x4*>(a.gg)[((fh * nsl + (col - 4)) * 4 + w) * 64 + lane] = X[j] * OpFmt<FMT>::STATE;   // the scan carries S * STATE
            // range bookkeeping (gdr_ws.hpp: gmax): this wave's row tile; +inf when a composition step left the fp16 pair's range
            const float mx = sat ? __builtin_inff() : wave_max_nonneg(absmax4(X[j]));
            if (lane == 0) a.gmax[(fh * nsl + (col - 4)) * 4 + w] = mx;
        }
    }
}

template <int NB, int IO, int TPR>
int launch_prep(const PrepArgs& pa, int FH, hipStream_t st)
{
    const size_t lds = prep_lds_bytes(NB);
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done_mask{0};
        if (int rc = gdr_lds_optin(reinterpret_cast<const void*>(gdr_prep_kernel<NB, IO, TPR>), done_mask, lds, "gdr_prep")) return rc;
    }
    hipLaunchKernelGGL((gdr_prep_kernel<NB, IO, TPR>), dim3(FH), dim3(256), lds, st, pa);
    GDKVM_LAUNCH_CHECK("gdr_prep_kernel");
    return GDKVM_OK;
}

}  // namespace

extern "C" size_t gdkvm_scan_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv)
{
    if (gdr_narrow_keys(Dk) && B > 0 && T > 0 && Hh > 0 && N > 0 && Dv > 0 && N <= GDKVM_MAX_N)      // (gdkvm_scan_fwd's zero-extended copies)
        return gdr_up256(gdr_workspace_bytes(B, T, Hh, N, GDKVM_DK, Dv)) + gdr_narrow_extra_bytes(B, T, Hh, N, Dv);
    if (gdr_wide_keys(Dk)) return GDKVM_WS_TAIL;          // (gdr_general.hip keeps its state in LDS)
    return gdr_workspace_bytes(B, T, Hh, N, Dk, Dv);
}

static int scan_prep_impl(const void* q, const void* k, const void* v, const float* beta, const float* norms, void* workspace, size_t workspace_bytes,
                          int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    if (int rc = check_common("scan_prep", B, T, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (norms && (!(flags & GDKVM_FLAG_NORMALIZE_QK) || (flags & GDKVM_FLAG_TRAIN) || !gdkvm_aligned16(norms)))
        return gdkvm_fail(GDKVM_ERR_ARG, "scan_prep_normed: norms go with GDKVM_FLAG_NORMALIZE_QK, inference (no GDKVM_FLAG_TRAIN), 16-byte aligned");
    if (rule < 0 || rule > 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_prep: rule=%d", rule);
    if (B == 0 || T == 0 || N == 0) return GDKVM_OK;
    if (int rc = check_ptrs("scan_prep", {q, k, v, beta, workspace}, {})) return rc;
    WsView ws;
    if (int rc = carve("scan_prep", workspace, workspace_bytes, B, T, Hh, N, Dk, Dv, &ws)) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!(flags & GDKVM_FLAG_TRAIN) || ws.nchunk > 1) {  // P and G directly (frames of > 64 tokens: per 64-token chunk, then composed)
        PrepMArgs pm{q, k, v, beta, ws.qinv, ws.pp, ws.gg, ws.gmax, norms, ws.x0, ws.ppc, ws.ggc, T, Hh, N, Dv, rule, flags, 16 * ws.nb, ws.nchunk};
#ifdef GDKVM_DIAG
        pm.diag = g_gdkvm_diag_buf;
#endif
        const bool wide = flags & GDKVM_FLAG_WIDE_RANGE;
        // Frames of more than 64 tokens: with enough frames to fill the device on their own, ONE workgroup walks a frame's chunks and
        // composes its map in registers (no chunk maps through HBM, no compose kernel); with few frames the chunks run as separate
        // workgroups (4x the parallelism) and gdr_compose_kernel stitches them.  bf16 I/O on pair16 operands, Dv <= 256 in
        // multiples of 64 (five column tiles of accumulators per wave), not delta_parallel (its chunks add up instead).
        bool fuse = false;
        if (ws.nchunk > 1 && io_dtype == GDKVM_BF16 && !wide && rule != GDKVM_RULE_DELTA_PARALLEL && Dv % 64 == 0 && Dv <= 256) {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) {
                int n = 0;
                if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
            }
            fuse = (long)B * T * Hh >= cus;               // measured crossover at N = 256, Dv = 256: 256 frames (54 us either way)
            if (const char* e = getenv("GDKVM_PREP_FUSE")) fuse = e[0] == '1';     // "0" / "1": overrides the choice by frame count (tests, A/B)
        }
        if (int rc = io_dtype == GDKVM_F32 ? launch_prepm<4, GDKVM_F32>(pm, B * T * Hh, ws.nchunk, wide, false, st)
                                           : launch_prepm<4, GDKVM_BF16>(pm, B * T * Hh, ws.nchunk, wide, fuse, st)) return rc;
        if (ws.nchunk > 1 && !fuse) {
            ComposeArgs ca{ws.x0, ws.ppc, ws.ggc, ws.pp, ws.gg, Dv, ws.nchunk, rule == GDKVM_RULE_DELTA_PARALLEL, ws.gmax};
            const dim3 cgrid((unsigned)(B * T * Hh), (unsigned)((4 + Dv / 16 + 3) / 4));
            if (wide) hipLaunchKernelGGL(gdr_compose_kernel<FMT_SPLIT3>, cgrid, dim3(256), 0, st, ca);
            else hipLaunchKernelGGL(gdr_compose_kernel<FMT_PAIR16>, cgrid, dim3(256), 0, st, ca);
            GDKVM_LAUNCH_CHECK("gdr_compose_kernel");
        }
        return GDKVM_OK;
    }
    // training, <= 64 tokens: the WY factors the backward consumes, then folded
    PrepArgs pa{q, k, v, beta, ws.wt, ws.knT, ws.ut, ws.qinv, ws.kn, ws.wtT, ws.qnT, ws.tii, ws.wti, T, Hh, N, Dv, rule, flags};
    if (int rc = io_dtype == GDKVM_F32 ? launch_prep<4, GDKVM_F32, 5>(pa, B * T * Hh, st) : launch_prep<4, GDKVM_BF16, 5>(pa, B * T * Hh, st)) return rc;
    FoldArgs fa{ws.wti, ws.knT, ws.ut, ws.pp, ws.gg, ws.ppt, Dv, ws.gmax};
    launch_fold<4>(fa, B * T * Hh, flags & GDKVM_FLAG_WIDE_RANGE, st);
    GDKVM_LAUNCH_CHECK("gdr_fold_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_scan_prep(const void* q, const void* k, const void* v, const float* beta, void* workspace, size_t workspace_bytes,
                               int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    return scan_prep_impl(q, k, v, beta, nullptr, workspace, workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream);
}

extern "C" int gdkvm_scan_prep_normed(const void* q, const void* k, const void* v, const float* beta, const float* norms,
                                      void* workspace, size_t workspace_bytes,
                                      int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    if (!norms) return gdkvm_fail(GDKVM_ERR_ARG, "scan_prep_normed: null norms");
    return scan_prep_impl(q, k, v, beta, norms, workspace, workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream);
}

extern "C" int gdkvm_scan_fwd_normed(const void* q, const void* k, const void* v, const float* alpha, const float* beta, const float* norms,
                                     const float* s_in, void* r_out, float* s_out, void* workspace, size_t workspace_bytes,
                                     int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    if (Dk != GDKVM_DK) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd_normed: Dk=%d (the norms path is built for Dk=%d)", Dk, GDKVM_DK);
    if (rule == GDKVM_RULE_DELTA_PARALLEL) flags |= GDKVM_FLAG_WIDE_RANGE;      // as gdkvm_scan_fwd
    if (int rc = gdkvm_scan_prep_normed(q, k, v, beta, norms, workspace, workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream)) return rc;
    return gdkvm_scan_apply(q, alpha, s_in, r_out, s_out, nullptr, workspace, workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, flags, stream);
}
