// gdr_scan.hip -- LKVA read + gated-delta-rule write for gfx950: the serial recurrence
//     R_t = Qn_t S_{t-1},      S_t = a_t P_t S_{t-1} + G_t
// on the per-frame affine maps (P, G) that gdr_prep.hip folds.  Kernels, in file order:
//
//  gdr_affine_scan_kernel  one workgroup per (clip, head, 16-column slice of Dv), 8 waves in two roles (state / read-out),
//                          one barrier per frame.  Also the backward's reverse recurrence (reverse mode) and
//                          the state-transition matrix (transition mode).
//  gdr_readout_kernel      frames of more than 64 tokens: the LKVA read-out, frame-parallel, from state images the serial
//                          kernel dumps.
//  gdr_decay_kernel        no tokens: the state only decays.
//  gdr_bwd_g_kernel        backward: Gb = Qn^T dR per frame, the additive term of the reverse recurrence.
// Products that dominate a kernel run on v_mfma_f32_16x16x32_bf16 with fp32 operands carried as three bf16 terms (split3,
// gdr_device.hpp); everything else is exact fp32 on v_mfma_f32_16x16x4_f32.  Workspace layout: gdr_ws.hpp.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <initializer_list>
#include <type_traits>

#include "gdkvm_common.hpp"
#include "gdr_device.hpp"
#include "gdr_ws.hpp"

#ifdef GDKVM_DIAG
unsigned long long* g_gdkvm_diag_buf = nullptr;
extern "C" void gdkvm_diag_set_buffer(unsigned long long* p) { g_gdkvm_diag_buf = p; }
#endif

namespace {

// gdr_affine_scan_kernel: the serial recurrence on the folded operands.  8 waves per workgroup in two fixed roles:
//   state waves  0-3   S_t tile (rows 16w.., this slice's 16 columns) = a_t * (P_t[rows 16w..] S_{t-1}) + G_t tile; the B
//                      operand is the four waves' accumulator tiles as published in LDS (double-buffered by frame parity:
//                      one barrier per frame).  Each wave fetches its own operands -- its row tile of the P images, its G
//                      tile and a_t -- from global memory straight into registers, AFF_PD frames ahead, the loads issued
//                      behind the frame's S reads.  (Round 1 and the first half of round 2 streamed them through an LDS ring
//                      filled by four LDS-DMA loader waves: measured, the ring's 40 KB of LDS traffic per frame cost the
//                      chain as much as the loads' issue slots do here, and it held 120 KB of LDS and four more waves.)
//   read waves   4-7   R_t = (Qn_t S_{t-1}) * qinv from the same images, with their own register prefetch of q.
// Barriers are raw s_barrier + lgkmcnt(0): __syncthreads() would add vmcnt(0) and drain the prefetch queue every frame.
struct AffArgs {
    const void* q; const float* alpha; const float* s_in;
    const float* pp; const float* gg; const float* qinv;
    void* r_out; float* s_out; float* s_hist; char* trash;
    int init_identity, zero_g;                       // transition mode: S_0 = I, and gg is ONE zero tile (all strides 0)
    int T, Hh, N, Dv, flags, BH;
    int reverse;                                     // visit the frames last to first (the backward's reverse recurrence)
    float* simg;                                     // DEFER: per-frame operand images of the state for gdr_readout_kernel
    const float* gmax;                               // pair16: max |G| per frame and slice from the frame-parallel side (NULL: default exponent)
    float* esc;                                      // pair16: 2^e of this (clip-head, slice) -- NaN when the call's bound is not finite -- for
                                                     // gdr_readout_kernel (DEFER) and for gdkvm_scan_status
};
// LDS (16-byte units): S term images [2 parities][NT terms][2 ksteps][64] | (fp32 I/O on split3) S fp32 images [2][4][64]
template <int IO> struct RItem;                                        // read-out operands of one 16-token tile
template <> struct RItem<GDKVM_F32> { f32x4 q[4]; float qinv; };
template <> struct RItem<GDKVM_BF16> { bf16x8 q[2]; float qinv; };
// (NT = terms of the operand format: 3 = split3 bf16, 2 = pair16 fp16)
__host__ __device__ constexpr int aff_s_f4(int NT) { return 2 * NT * 2 * 64; }     // S term images, two parities
__host__ __device__ constexpr size_t aff_lds_bytes(int IO, int NT)
{
    return (size_t)(aff_s_f4(NT) + (IO == GDKVM_F32 && NT == 3 ? 2 * 4 * 64 : 0)) * 16;
}
#ifndef AFF_PD_FRAMES
#define AFF_PD_FRAMES 2
#endif
#ifndef AFF_NBUF
#define AFF_NBUF 4
#endif
constexpr int AFF_PD = AFF_PD_FRAMES;                // operand prefetch distance of the state waves, in frames (read waves: AFF_NBUF - 1)
constexpr int AFF_THREADS = 512;

__device__ __forceinline__ void aff_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DEFER (frames of more than 64 tokens): every slice workgroup would re-read the whole frame's q -- 32 KB at 256 tokens, and the
// CU's vector-memory path saturates (1.35 us per frame) -- so the read waves only dump the state's operand images (4 KB per
// frame and slice) and gdr_readout_kernel does the read-out frame-parallel, reading q once per frame.
template <int IO, int FMT, bool DEFER, bool SAVE>
__global__ __launch_bounds__(AFF_THREADS) void gdr_affine_scan_kernel(AffArgs a)
{
    constexpr int NB = 4, NP = 16 * NB, JT = 1, NBUF = AFF_NBUF, DEPTH = NBUF - 1, UFR = NBUF / JT;
    constexpr int NT = fmt_terms(FMT);
    constexpr bool PAIR = FMT == FMT_PAIR16;
    // the kernel carries S' = S * st_scale; G arrives scaled by OpFmt<FMT>::STATE (2^-4 under pair16, the default exponent)
    extern __shared__ __attribute__((aligned(16))) f32x4 aff_smem[];
    uint2* s_S3 = reinterpret_cast<uint2*>(aff_smem);      // [parity][term] images of SPLIT_IMG uint2: B operand of the 16x16x32 MFMA
    constexpr bool EXACT = IO == GDKVM_F32 && !PAIR;       // fp32 I/O on full-range operands: exact fp32 read-out from fp32 images of S
    f32x4* s_Sf = aff_smem + aff_s_f4(NT);                 // EXACT only: accumulator images of S

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & 3, role = wave >> 2;             // 0 state, 1 read-out
    const int nsl = a.Dv / 16, N = a.N, Hh = a.Hh, Dv = a.Dv, T = a.T;
    int bh, sl;
    {
        const int x = blockIdx.x;                         // XCD-aware: the slices of one (clip, head) share an L2
        if (a.BH % 8 == 0) { bh = (x & 7) + 8 * ((x >> 3) / nsl); sl = (x >> 3) % nsl; }
        else { bh = x / nsl; sl = x % nsl; }
    }
    const int b = bh / Hh, h = bh % Hh;
    const size_t fh0 = (size_t)b * T * Hh + h;
    constexpr int ESZ = IO == GDKVM_F32 ? 4 : 2;

    // ---- the state's exponent (pair16).  fp16 pairs hold |S * 2^-e| < 65504, so e must fit this call's state: columns of S never
    // mix and every frame's map is a contraction in the 2-norm (delta_sequential: products of I - b k k^T with |k| = 1, b in [0,1];
    // gated_linear: P = I), hence for this 16-column slice  |S_t| <= 8 (max|S_0| + sum_t max|G_t|)  elementwise for every t (8 =
    // sqrt(Dk)).  The frame-parallel side left max|G_t| per slice in the workspace; every wave forms the same bound from the same
    // numbers (no exchange) and takes e = max(4, ceil(log2 bound) - 15): 4, the format's default, for anything up to ~5e5 -- every
    // ordinary input, so chunked calls of a clip stay bit-identical to one call -- and beyond that whatever the state needs (exact:
    // powers of two).  A bound that is not finite (a composition step of a > 64-token frame overflowed) poisons the call with NaNs.
    float st_scale = OpFmt<FMT>::STATE, st_inv = OpFmt<FMT>::STATE_INV;
    bool rescaled = false;
    if constexpr (PAIR) {
        if (a.gmax) {
            float gs = 0.f, sm = 0.f;
            const f32x4* gm = reinterpret_cast<const f32x4*>(a.gmax) + fh0 * nsl + sl;
            for (int t = lane; t < T; t += 64) {
                const f32x4 x = gm[(size_t)t * Hh * nsl];
                gs += fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
            }
            if (a.s_in) {
                const f32x4* sp = reinterpret_cast<const f32x4*>(a.s_in + ((size_t)bh * GDKVM_DK + lane) * Dv + 16 * sl);
#pragma unroll
                for (int c = 0; c < 4; ++c) sm = fmaxf(sm, absmax4(sp[c]));
            }
            const float bound = 8.0f * (wave_max_nonneg(sm) + wave_sum(gs));
            const int eb = (int)((__float_as_uint(bound) >> 23) & 0xffu) - 126;       // bound < 2^eb
            const int e = __builtin_amdgcn_readfirstlane(bound <= 3.0e38f ? max(4, eb - 15) : -1);
            if (e != 4) {
                rescaled = true;
                st_scale = e < 0 ? __builtin_nanf("") : __uint_as_float((unsigned)(127 - e) << 23);
                st_inv = e < 0 ? __builtin_nanf("") : __uint_as_float((unsigned)(127 + e) << 23);
            }
        }
    }
    const float STATE = st_scale, STATE_INV = st_inv;
    if constexpr (PAIR) {                                  // every workgroup leaves its slice's 2^e (gdr_ws.hpp: esc)
        if (a.esc && wave == 0 && lane == 0) a.esc[(size_t)bh * ((nsl + 3) & ~3) + sl] = STATE_INV;
    }
    const float gfix = st_scale * OpFmt<FMT>::STATE_INV;   // what G (prepared at the default exponent) is multiplied by: 1 unless rescaled

    // Publishing S: the three bf16 terms of this wave's rows 16w + 4g + r (k of the next product) as B images; the fp32
    // arm also keeps the accumulator image for its exact fp32 read-out.
    auto publish_state = [&](int par, const f32x4& sv) __attribute__((always_inline)) {
        uint2 t3[3];
        OpFmt<FMT>::split4(sv, t3);
        const int e = split_slot(w, g, li);
#pragma unroll
        for (int sp = 0; sp < NT; ++sp) s_S3[(par * NT + sp) * SPLIT_IMG + e] = t3[sp];
        if constexpr (EXACT) s_Sf[par * 256 + w * 64 + lane] = sv;
    };
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
    if (role == 0) {
        if (a.s_in) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[r] = STATE * a.s_in[((size_t)bh * GDKVM_DK + 16 * w + 4 * g + r) * Dv + 16 * sl + li];
        } else if (a.init_identity) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[r] = (16 * w + 4 * g + r == 16 * sl + li) ? STATE : 0.f;
        }
        publish_state(0, sacc);
    }

    if (role == 1) {
        // ------------------------------------------------------------------------------ read-out waves
        if constexpr (DEFER) {                             // dump the images of S_{t-1}: 1 KiB per read wave and frame
            uint4* dst = reinterpret_cast<uint4*>(a.simg) + ((fh0 * nsl + sl) * 4 + w) * 64 + lane;
            const size_t d_fstride = (size_t)Hh * nsl * 4 * 64;
            aff_barrier();
            for (int t = 0; t < T; ++t) {
                const int par = t & 1;
                uint4 img;
                if constexpr (EXACT) img = *reinterpret_cast<const uint4*>(&s_Sf[par * 256 + w * 64 + lane]);
                else img = *reinterpret_cast<const uint4*>(&s_S3[(par * NT + (w >> 1)) * SPLIT_IMG + ((w & 1) * 64 + lane) * 2]);
                dst[(size_t)(a.reverse ? T - 1 - t : t) * d_fstride] = img;
                aff_barrier();
            }
            return;
        }
        if (!a.r_out) {                                    // states only (transition matrices, segment end states): just keep the
            aff_barrier();                                 // barrier count
            for (int t = 0; t < T; ++t) aff_barrier();
            return;
        }
        const int last_item = T * JT - 1;
        const char* qbase = static_cast<const char*>(a.q) + (((size_t)b * T * N * Hh + h) * GDKVM_DK) * ESZ;
        const size_t q_fstride = (size_t)N * Hh * GDKVM_DK * ESZ;
        const float* qinv_lane = a.qinv + fh0 * NP + li;
        // The read-out is computed TRANSPOSED, R^T = S^T Qn^T: the S images are also the A operand of S^T and the q rows as
        // loaded are also the B operand of Qn^T, so only the two MFMA arguments swap -- and lane (g, li) ends up with columns
        // 4g..4g+3 of token li: one 8- or 16-byte store per lane into the token's row instead of four scattered 2-byte ones
        // (the store issue made the read waves the slowest role of a frame).
        char* rbase = static_cast<char*>(a.r_out) + ((size_t)b * T * N * Hh * Dv + h * Dv + 16 * sl + 4 * g) * ESZ;
        const size_t r_fstride = (size_t)N * Hh * Dv * ESZ;
        auto load_q = [&](int item, RItem<IO>& d) __attribute__((always_inline)) {
            item = min(item, last_item);
            const int t = item / JT, tt = w + 4 * (item - t * JT);
            const int nq = min(16 * tt + li, N - 1);
            const char* p = qbase + t * q_fstride + (size_t)nq * (Hh * GDKVM_DK * ESZ);
            if constexpr (EXACT) {
#pragma unroll
                for (int m = 0; m < 4; ++m) d.q[m] = *reinterpret_cast<const f32x4*>(p + 64 * m + 16 * g);
            } else if constexpr (IO == GDKVM_F32) {        // pair16: channels 32ks + 8g .. +7 as q[2ks], q[2ks + 1]
#pragma unroll
                for (int m = 0; m < 4; ++m) d.q[m] = *reinterpret_cast<const f32x4*>(p + 128 * (m >> 1) + 32 * g + 16 * (m & 1));
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) d.q[ks] = *reinterpret_cast<const bf16x8*>(p + 64 * ks + 16 * g);
            }
            d.qinv = qinv_lane[(size_t)t * Hh * NP + 16 * tt];
        };
        RItem<IO> qb[NBUF];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) load_q(i, qb[i]);
        aff_barrier();
        // B operand of the frame's read-out, fetched once per frame: fp32 arm the four accumulator images (k = 16m + 4g + r,
        // exact fp32 MFMA); bf16 arm the three term images (q is exactly bf16: two bf16 MFMA per k step).
        struct SB { f32x4 f[4]; bf16x8 t[2][2]; };
        auto load_sb = [&](int par, SB& sb) __attribute__((always_inline)) {
            if constexpr (EXACT) {
#pragma unroll
                for (int m = 0; m < 4; ++m) sb.f[m] = s_Sf[par * 256 + m * 64 + lane];
            } else {
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)         // split3: h + m, 16 bits of S, far inside the bf16 output's 2^-9; pair16: both terms
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        sb.t[sp][ks] = *reinterpret_cast<const bf16x8*>(&s_S3[(par * NT + sp) * SPLIT_IMG + (ks * 64 + lane) * 2]);
            }
        };
        // pair16: the frame's q tile as halves (exact power-of-two scaling of the token's row) and the read-out's final factor
        // are prepared BEFORE the barrier that releases the frame -- q was fetched three frames ago and does not depend on S --
        // so that behind the barrier only the S images, four MFMAs and the store stand on the read wave's path.
        struct QOp { f16x8 qh[2], ql[2]; float rscale; };   // (ql: fp32 I/O only -- q as a pair16 as well)
        auto prepare = [&](const RItem<IO>& it, QOp& o) __attribute__((always_inline)) {
            o.rscale = it.qinv * STATE_INV;
            if constexpr (PAIR) {
                const float sc = pow2_floor(it.qinv);
                o.rscale = it.qinv * pow2_inv(sc) * STATE_INV;
                if constexpr (IO == GDKVM_BF16) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) o.qh[ks] = bf16x8_to_f16(it.q[ks], sc);
                } else {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        uint2 h0, l0, h1, l1;
                        pair16x4(it.q[2 * ks] * sc, h0, l0);
                        pair16x4(it.q[2 * ks + 1] * sc, h1, l1);
                        o.qh[ks] = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                        o.ql[ks] = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
                    }
                }
            }
        };
        auto read_item = [&](int t, int j, const SB& sb, const RItem<IO>& cur, const QOp& op, RItem<IO>& nxt) __attribute__((always_inline)) {
            const int tt = w + 4 * j;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EXACT) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (m & 1) acc1 = mfma4(sb.f[m][r], cur.q[m][r], acc1);
                        else acc0 = mfma4(sb.f[m][r], cur.q[m][r], acc0);
                    }
            } else if constexpr (PAIR) {                   // acc0: hh; acc1: the cross terms carried at 2^11 (fp32 I/O: q is a pair too)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sb.t[1][ks]), op.qh[ks], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sb.t[0][ks]), op.qh[ks], acc0, 0, 0, 0);
                    if constexpr (IO == GDKVM_F32)
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sb.t[0][ks]), op.ql[ks], acc1, 0, 0, 0);
                }
                acc1 *= PAIR_LO_INV;
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sb.t[1][ks], cur.q[ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sb.t[0][ks], cur.q[ks], acc1, 0, 0, 0);
                }
            }
            const f32x4 accR = (acc0 + acc1) * op.rscale;
            const int nr = 16 * tt + li;                   // this lane's token; its columns 16sl + 4g .. +3
            // rows of padding tokens go to a write-only slot, one address per lane (a branch would make the vmcnt counting
            // conservative; one shared address serialises the lanes of a mostly-padded tile in the memory system)
            char* p = (nr < N && a.r_out) ? rbase + t * r_fstride + (size_t)nr * (Hh * Dv * ESZ) : a.trash + lane * (4 * ESZ);
            if constexpr (IO == GDKVM_F32) *reinterpret_cast<f32x4*>(p) = accR;
            else *reinterpret_cast<uint2*>(p) = make_uint2(cvt_pk_bf16(accR[0], accR[1]), cvt_pk_bf16(accR[2], accR[3]));
            load_q(t * JT + j + DEPTH, nxt);
        };
        QOp qop[2];
        prepare(qb[0], qop[0]);
        auto frame = [&](int t, auto fc) __attribute__((always_inline)) {
            constexpr int F = decltype(fc)::value;
            SB sb;
            load_sb(t & 1, sb);
            read_item(t, 0, sb, qb[F % NBUF], qop[F & 1], qb[(F + DEPTH) % NBUF]);
            prepare(qb[(F + 1) % NBUF], qop[(F + 1) & 1]);   // next frame's operand, off the released path
            aff_barrier();                                 // S_{t-1} consumed / S_t published
        };
        int t0 = 0;
        for (; t0 + UFR <= T; t0 += UFR)
            static_for<0, UFR>([&](auto fc) { frame(t0 + decltype(fc)::value, fc); });
        static_for<0, UFR - 1>([&](auto fc) {
            if (t0 + decltype(fc)::value < T) frame(t0 + decltype(fc)::value, fc);
        });
        return;
    }

    // ---------------------------------------------------------------------------------- state waves
    __builtin_amdgcn_s_setprio(2);                        // the S -> S chain goes first at the SIMD's issue port
    const bool gate_logits = a.flags & GDKVM_FLAG_GATE_LOGITS;
    aff_barrier();
    // P row-tile images, G tile and a_t of frame t + AFF_PD are fetched into registers during frame t, so after the barrier only
    // the S term images stand between the wave and its MFMAs
    struct POp { uint4 pa[NT][2]; f32x4 gt; float al; };
    {
        // (a frame-head's P slot is sized for three terms whatever the format; row tile w is 2 NT KiB contiguous, lane-linear)
        // The three operand streams are walked by RUNNING pointers -- a uniform base per stream, advanced by one frame's stride after each
        // fetch until the last frame is reached (further prefetches repeat it), plus a constant 32-bit lane offset: six scalar adds per
        // frame where the index arithmetic (clamp, reversal, three 64-bit products) took 25 instructions between the S reads and the MFMAs
        const long dir = a.reverse ? -1 : 1;
        const size_t f_first = a.reverse ? (size_t)(T - 1) : 0;
        long pp_step = dir * (long)Hh * (GDKVM_DK * GDKVM_DK * 3 / 2) * 4;                             // bytes
        long gg_step = a.zero_g ? 0 : dir * (long)Hh * nsl * 4 * 64 * 4 * 4;
        long al_step = dir * (long)Hh * 4;
        const char* pp_cur = reinterpret_cast<const char*>(a.pp + (fh0 + f_first * Hh) * (GDKVM_DK * GDKVM_DK * 3 / 2) + (size_t)w * (NT * 2 * 64) * 4);
        const char* gg_cur = reinterpret_cast<const char*>(a.gg + (a.zero_g ? 0 : ((fh0 + f_first * Hh) * nsl + sl) * 4 + w) * 64 * 4);
        const char* al_cur = reinterpret_cast<const char*>(a.alpha + fh0 + f_first * Hh);
        const unsigned lane16 = lane * 16;
        int f_cur = 0;                                     // frames fetched so far - 1, clamped: the pointers address frame min(f_cur, T - 1)
        auto fetch_op = [&](int, POp& d) __attribute__((always_inline)) {
#pragma unroll
            for (int sp = 0; sp < NT; ++sp)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) d.pa[sp][ks] = *reinterpret_cast<const uint4*>(pp_cur + lane16 + 1024 * (sp * 2 + ks));
            d.gt = *reinterpret_cast<const f32x4*>(gg_cur + lane16);
            d.al = *reinterpret_cast<const float*>(al_cur);
            if (f_cur >= T - 1) pp_step = gg_step = al_step = 0;      // (uniform; taken at the clip's end only)
            ++f_cur;
            pp_cur += pp_step;
            gg_cur += gg_step;
            al_cur += al_step;
        };
        POp od[AFF_PD + 1];
#pragma unroll
        for (int i = 0; i < AFF_PD; ++i) fetch_op(i, od[i]);
        auto frame = [&](int t, const POp& op, POp& far, auto scaled_c) __attribute__((always_inline)) {
            const int par = t & 1;
            if constexpr (SAVE) {
                float* hp = a.s_hist + ((fh0 + (size_t)(a.reverse ? T - 1 - t : t) * Hh) * GDKVM_DK + 16 * w + 4 * g) * Dv + 16 * sl + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) hp[(size_t)r * Dv] = STATE_INV * sacc[r];
            }
            uint4 sb[NT][2];
#pragma unroll
            for (int sp = 0; sp < NT; ++sp)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    sb[sp][ks] = *reinterpret_cast<const uint4*>(&s_S3[(par * NT + sp) * SPLIT_IMG + (ks * 64 + lane) * 2]);
            __builtin_amdgcn_sched_barrier(0);
            fetch_op(t + AFF_PD, far);                     // issued behind the S reads: the issue overlaps their latency
            __builtin_amdgcn_sched_barrier(0);
            const float alpha = gate_logits ? fast_sigmoid(op.al) : op.al;
            if constexpr (PAIR) {
                // S = alpha (acc0 + 2^-11 acc1) + G on PACKED fp32 FMAs (two values an instruction; the same two roundings as the scalar
                // form): four instructions on the chain instead of eight
                f32x4 acc0, acc1;
                OpFmt<FMT>::product_pair(op.pa, sb, acc0, acc1);
                const f32x2_t lo2 = {PAIR_LO_INV, PAIR_LO_INV}, al2 = {alpha, alpha};
                f32x2_t g01 = {op.gt[0], op.gt[1]}, g23 = {op.gt[2], op.gt[3]};
                if constexpr (decltype(scaled_c)::value) { g01 *= (f32x2_t){gfix, gfix}; g23 *= (f32x2_t){gfix, gfix}; }   // (exact: a power of two)
                const f32x2_t c01 = __builtin_elementwise_fma((f32x2_t){acc1[0], acc1[1]}, lo2, (f32x2_t){acc0[0], acc0[1]});
                const f32x2_t c23 = __builtin_elementwise_fma((f32x2_t){acc1[2], acc1[3]}, lo2, (f32x2_t){acc0[2], acc0[3]});
                const f32x2_t s01 = __builtin_elementwise_fma(al2, c01, g01), s23 = __builtin_elementwise_fma(al2, c23, g23);
                sacc = f32x4{s01[0], s01[1], s23[0], s23[1]};
            } else {
                const f32x4 ps = OpFmt<FMT>::product(op.pa, sb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (decltype(scaled_c)::value) sacc[r] = alpha * ps[r] + op.gt[r] * gfix;   // (exact: a power of two)
                    else sacc[r] = alpha * ps[r] + op.gt[r];
                }
            }
            publish_state(par ^ 1, sacc);
            aff_barrier();
        };
        constexpr int NR = AFF_PD + 1, UFD = NR % 2 == 0 ? NR : 2 * NR;
        // two copies of the frame loop: the ordinary one (default exponent: G as prepared, the chain as it always was) and the one
        // that rescales G for a state beyond the default range
        auto run = [&](auto scaled_c) __attribute__((always_inline)) {
            int t0 = 0;
            for (; t0 + UFD <= T; t0 += UFD)
                static_for<0, UFD>([&](auto fc) {
                    constexpr int F = decltype(fc)::value;
                    frame(t0 + F, od[F % NR], od[(F + AFF_PD) % NR], scaled_c);
                });
            static_for<0, UFD - 1>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
                if (t0 + F < T) frame(t0 + F, od[F % NR], od[(F + AFF_PD) % NR], scaled_c);
            });
        };
        if constexpr (PAIR) {
            if (rescaled) run(std::true_type{});
            else run(std::false_type{});
        } else {
            run(std::false_type{});
        }
        if (a.s_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a.s_out[((size_t)bh * GDKVM_DK + 16 * w + 4 * g + r) * Dv + 16 * sl + li] = STATE_INV * sacc[r];
        }
    }
}

template <int IO, int FMT, bool DEFER, bool SV>
int launch_affine(const AffArgs& sa, dim3 grid, hipStream_t st)
{
    constexpr int NT = fmt_terms(FMT);                     // (8 or 20 KiB of LDS: no opt-in needed)
    hipLaunchKernelGGL((gdr_affine_scan_kernel<IO, FMT, DEFER, SV>), grid, dim3(AFF_THREADS), aff_lds_bytes(IO, NT), st, sa);
    GDKVM_LAUNCH_CHECK("gdr_affine_scan_kernel");
    return GDKVM_OK;
}

template <int IO, int FMT>
int launch_affine_fmt(bool defer, bool save, const AffArgs& sa, dim3 grid, hipStream_t st)
{
    if (defer) return save ? launch_affine<IO, FMT, true, true>(sa, grid, st) : launch_affine<IO, FMT, true, false>(sa, grid, st);
    return save ? launch_affine<IO, FMT, false, true>(sa, grid, st) : launch_affine<IO, FMT, false, false>(sa, grid, st);
}
template <int IO>
int launch_affine_any(bool wide, bool defer, bool save, const AffArgs& sa, dim3 grid, hipStream_t st)
{
    return wide ? launch_affine_fmt<IO, FMT_SPLIT3>(defer, save, sa, grid, st) : launch_affine_fmt<IO, FMT_PAIR16>(defer, save, sa, grid, st);
}

// gdr_readout_kernel -- LKVA read-out for frames of more than 64 tokens, frame-parallel: R_t = (Qn_t S_{t-1}) from the operand
// images the serial kernel dumped.  One workgroup per (frame-head, 8 column tiles); a wave keeps the images of its two column
// tiles in registers and walks the frame's token tiles, so q is read once per workgroup and nothing goes through LDS.  Same
// arithmetic and operation order as the in-scan read-out (R^T = S^T Qn^T: pair16 terms on the f16 MFMA, or exact fp32).
struct ReadoutArgs { const void* q; const float* qinv; const float* simg; void* r_out; int Hh, N, Dv, NP; const float* esc; int T; };

template <int IO, int FMT>
__global__ __launch_bounds__(256, 2) void gdr_readout_kernel(ReadoutArgs a)       // (256 registers: MFMA results in VGPRs; with 512 they land in AGPRs and are copied out)
{
    constexpr bool PAIR = FMT == FMT_PAIR16;
    constexpr int ESZ = IO == GDKVM_F32 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t fh = blockIdx.x;
    const int h = (int)(fh % a.Hh), N = a.N, Dv = a.Dv, nsl = Dv / 16;
    const size_t bt = fh / a.Hh;
    const int c0 = (blockIdx.y * 4 + w) * 2;
    if (c0 >= nsl) return;
    const bool two = c0 + 1 < nsl;
    constexpr bool EXACT = IO == GDKVM_F32 && !PAIR;       // (as in the scan: fp32 images + exact fp32 MFMA only on full-range operands)
    const uint4* img = reinterpret_cast<const uint4*>(a.simg) + (fh * nsl + c0) * 4 * 64 + lane;
    uint4 sb[2][4];                                        // [col tile][bf16: term*2 + ks | fp32: k tile m]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) sb[c][i] = img[((c && two) ? 4 * 64 : 0) + i * 64];
    const char* qbase = static_cast<const char*>(a.q) + ((bt * N * a.Hh + h) * GDKVM_DK) * ESZ;
    const float* qinv = a.qinv + fh * a.NP;
    char* rbase = static_cast<char*>(a.r_out) + ((bt * N * a.Hh + h) * (size_t)Dv + 16 * c0 + 4 * g) * ESZ;
    const size_t rowq = (size_t)a.Hh * GDKVM_DK * ESZ, rowr = (size_t)a.Hh * Dv * ESZ;
    // token tiles [tt0, tt1) of the frame: with few frames the tokens are split over gridDim.z workgroups (cfg3: 160 frames x 2
    // column groups left half the CUs idle on a 16-tile latency chain; the state images are re-read from L2 per split)
    const int ntt_all = (N + 15) / 16, per_z = (ntt_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int tt0 = (int)blockIdx.z * per_z, ntt = min(ntt_all, tt0 + per_z);
    if (tt0 >= ntt) return;
    struct QT { uint4 q[IO == GDKVM_F32 ? 4 : 2]; float qi; };
    auto load_q = [&](int tt, QT& d) __attribute__((always_inline)) {
        const int nq = min(16 * min(tt, ntt_all - 1) + li, N - 1);
        const char* p = qbase + (size_t)nq * rowq;
        if constexpr (IO == GDKVM_F32 && PAIR) {           // channels 32ks + 8g .. +7 as q[2ks], q[2ks + 1]
#pragma unroll
            for (int i = 0; i < 4; ++i) d.q[i] = *reinterpret_cast<const uint4*>(p + 128 * (i >> 1) + 32 * g + 16 * (i & 1));
        } else {
#pragma unroll
            for (int i = 0; i < (IO == GDKVM_F32 ? 4 : 2); ++i) d.q[i] = *reinterpret_cast<const uint4*>(p + 64 * i + 16 * g);
        }
        d.qi = qinv[min(16 * min(tt, ntt_all - 1) + li, a.NP - 1)];
    };
    // (the dumped images are those of S * 2^-e: pair16 -- the exponent the serial kernel chose for this clip-head and column tile)
    float sinv[2] = {OpFmt<FMT>::STATE_INV, OpFmt<FMT>::STATE_INV};
    if constexpr (PAIR) {
        const size_t bh = (fh / ((size_t)a.T * a.Hh)) * a.Hh + h;
        const float* ep = a.esc + bh * ((nsl + 3) & ~3);
        sinv[0] = ep[c0];
        sinv[1] = ep[two ? c0 + 1 : c0];
    }
    auto tile = [&](int tt, const QT& d) __attribute__((always_inline)) {
        float rscale = d.qi;
        f16x8 qh[2], ql[2];
        if constexpr (PAIR) {
            const float sc = pow2_floor(d.qi);
            rscale = d.qi * pow2_inv(sc);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (IO == GDKVM_BF16) qh[ks] = bf16x8_to_f16(__builtin_bit_cast(bf16x8, d.q[ks]), sc);
                else {
                    uint2 h0, l0, h1, l1;
                    pair16x4(__builtin_bit_cast(f32x4, d.q[2 * ks]) * sc, h0, l0);
                    pair16x4(__builtin_bit_cast(f32x4, d.q[2 * ks + 1]) * sc, h1, l1);
                    qh[ks] = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                    ql[ks] = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EXACT) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 sm = __builtin_bit_cast(f32x4, sb[c][m]), qm = __builtin_bit_cast(f32x4, d.q[m]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (m & 1) acc1 = mfma4(sm[r], qm[r], acc1);
                        else acc0 = mfma4(sm[r], qm[r], acc0);
                    }
                }
            } else if constexpr (PAIR) {                   // pair16 images of S (h, l at 2^11), q -> fp16 under a power-of-two scaling
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sb[c][2 + ks]), qh[ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sb[c][ks]), qh[ks], acc1, 0, 0, 0);
                    if constexpr (IO == GDKVM_F32)
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sb[c][ks]), ql[ks], acc0, 0, 0, 0);
                }
                acc0 *= PAIR_LO_INV;
            } else {                                       // split3: the h and m term images, q exactly bf16
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sb[c][2 + ks]), __builtin_bit_cast(bf16x8, d.q[ks]), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sb[c][ks]), __builtin_bit_cast(bf16x8, d.q[ks]), acc1, 0, 0, 0);
                }
            }
            const f32x4 accR = (acc0 + acc1) * (rscale * sinv[c]);
            const int nr = 16 * tt + li;
#ifdef GDKVM_ABL_RO_NOSTORE                                 // (tools/abl_scan.py: stores only where a value is NaN -- never, but not provably)
            if (nr < N && (c == 0 || two) && accR[0] != accR[0]) {
#else
            if (nr < N && (c == 0 || two)) {
#endif
                char* p = rbase + (size_t)nr * rowr + c * 16 * ESZ;
                if constexpr (IO == GDKVM_F32) *reinterpret_cast<f32x4*>(p) = accR;
                else *reinterpret_cast<uint2*>(p) = make_uint2(cvt_pk_bf16(accR[0], accR[1]), cvt_pk_bf16(accR[2], accR[3]));
            }
        }
    };
    QT qa, qb;
    load_q(tt0, qa);
#ifdef GDKVM_ABL_RO_NOQ                                     // (tools/abl_scan.py: the first two token tiles' q rows for every tile)
    load_q(tt0 + 1, qb);
    for (int tt = tt0; tt < ntt; tt += 2) {
        tile(tt, qa);
        if (tt + 1 < ntt) tile(tt + 1, qb);
    }
#else
    for (int tt = tt0; tt < ntt; tt += 2) {
        load_q(tt + 1, qb);
        tile(tt, qa);
        load_q(tt + 2, qa);
        if (tt + 1 < ntt) tile(tt + 1, qb);
    }
#endif
}

// gdr_readout_rows_kernel -- the same read-out for bf16 I/O with every global access a run of whole rows (round 6).  Ablation of the
// kernel above at cfg5 (tools/abl_scan.py ro, profiles/r06_ad_readout_ablation.txt): 82 us of which the R stores cost ~80 and the q loads
// ~25-45 -- not their bytes (134 + 33 MB) but their SHAPE: a wave's store is 16 rows x 32 bytes, its q load 16 rows x 64 bytes, every
// wave of the frame's workgroups loads the same q rows, and the memory pipeline is paid per row segment.  Here the workgroup moves both
// through LDS: q rows of TWO token tiles (32 tokens x 128 bytes) arrive as one 16-byte load per thread (a contiguous 4 KB when Hh = 1),
// requested one pair ahead; the R tiles of the pair (32 tokens x the workgroup's 128 columns) are written to LDS in the accumulator
// layout and leave as 16 bytes per thread, four full 256-byte row pieces per wave and instruction.  One barrier per pair of token tiles
// (q and R stage double-buffered: a stage written in iteration p was last read before the barrier of iteration p - 1).  Arithmetic, operand
// images and operation order are those of gdr_readout_kernel: bit-identical results.
constexpr int RO_QP = 2 * GDKVM_DK + 16;                  // q stage row pitch, bytes (16 x odd: a b128 read of 16 rows touches every bank once)
constexpr int RO_RP = 256 + 32;                           // R stage row pitch, bytes (72 dwords = 8 mod 64: b64 writes of 16 rows x 4 groups in two passes)
constexpr size_t RO_LDS = 2 * 32 * RO_QP + 2 * 32 * RO_RP;

template <int FMT>
__global__ __launch_bounds__(256, 2) void gdr_readout_rows_kernel(ReadoutArgs a)
{
    constexpr bool PAIR = FMT == FMT_PAIR16;
    extern __shared__ __attribute__((aligned(16))) char ro_smem[];
    char* s_q = ro_smem;                                   // [2][32][RO_QP]
    char* s_r = ro_smem + 2 * 32 * RO_QP;                  // [2][32][RO_RP]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t fh = blockIdx.x;
    const int h = (int)(fh % a.Hh), N = a.N, Dv = a.Dv, nsl = Dv / 16;
    const size_t bt = fh / a.Hh;
    const int c0 = (blockIdx.y * 4 + w) * 2;
    const bool one = c0 < nsl, two = c0 + 1 < nsl;          // (a wave without columns still stages q rows and stores R rows: no early exit)
    uint4 sb[2][4];                                        // [col tile][term*2 + ks]
    {
        const uint4* img = reinterpret_cast<const uint4*>(a.simg) + (fh * nsl + (one ? c0 : 0)) * 4 * 64 + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) sb[c][i] = img[((c && two) ? 4 * 64 : 0) + i * 64];
    }
    const char* qbase = static_cast<const char*>(a.q) + ((bt * N * a.Hh + h) * GDKVM_DK) * 2;
    const float* qinv = a.qinv + fh * a.NP;
    const size_t rowq = (size_t)a.Hh * GDKVM_DK * 2, rowr = (size_t)a.Hh * Dv * 2;
    const int col0 = 128 * (int)blockIdx.y;                 // first column of the workgroup
    char* rbase = static_cast<char*>(a.r_out) + ((bt * N * a.Hh + h) * (size_t)Dv + col0) * 2;
    const int nchunks = min(16, (Dv - col0) / 8);           // 16-byte pieces of a row that exist (Dv is a multiple of 16)
    const int ntt_all = (N + 15) / 16, per_z = (ntt_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int tt0 = (int)blockIdx.z * per_z, ntt = min(ntt_all, tt0 + per_z);
    if (tt0 >= ntt) return;                                 // (whole workgroup)
    float sinv[2] = {OpFmt<FMT>::STATE_INV, OpFmt<FMT>::STATE_INV};
    if constexpr (PAIR) {
        const size_t bh = (fh / ((size_t)a.T * a.Hh)) * a.Hh + h;
        const float* ep = a.esc + bh * ((nsl + 3) & ~3);
        sinv[0] = ep[one ? c0 : 0];
        sinv[1] = ep[two ? c0 + 1 : (one ? c0 : 0)];
    }
    // staging roles: thread -> (token of the pair, 16-byte piece of its q row)
    const int q_tok = tid >> 3, q_pc = tid & 7;
    auto fetch_q = [&](int tt) __attribute__((always_inline)) {
        const int nq = min(16 * tt + q_tok, N - 1);
        return *reinterpret_cast<const uint4*>(qbase + (size_t)nq * rowq + 16 * q_pc);
    };
    auto fetch_qi = [&](int tt, int j) __attribute__((always_inline)) { return qinv[min(16 * min(tt + j, ntt_all - 1) + li, a.NP - 1)]; };
    uint4 qn = fetch_q(tt0);
    float qi0 = fetch_qi(tt0, 0), qi1 = fetch_qi(tt0, 1);
    *reinterpret_cast<uint4*>(s_q + q_tok * RO_QP + 16 * q_pc) = qn;
    __syncthreads();
    int par = 0;
    for (int tt = tt0; tt < ntt; tt += 2, par ^= 1) {
        const bool more = tt + 2 < ntt;
        float qi_n0 = 0.f, qi_n1 = 0.f;
        if (more) {
            qn = fetch_q(tt + 2);
            qi_n0 = fetch_qi(tt + 2, 0);
            qi_n1 = fetch_qi(tt + 2, 1);
        }
        const char* sq = s_q + par * 32 * RO_QP;
        char* sr = s_r + par * 32 * RO_RP;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (tt + j >= ntt || !one) continue;
            const float qi = j ? qi1 : qi0;
            uint4 qv[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qv[ks] = *reinterpret_cast<const uint4*>(sq + (16 * j + li) * RO_QP + 64 * ks + 16 * g);
            float rscale = qi;
            f16x8 qh[2];
            if constexpr (PAIR) {
                const float sc = pow2_floor(qi);
                rscale = qi * pow2_inv(sc);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) qh[ks] = bf16x8_to_f16(__builtin_bit_cast(bf16x8, qv[ks]), sc);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (PAIR) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sb[c][2 + ks]), qh[ks], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sb[c][ks]), qh[ks], acc1, 0, 0, 0);
                    }
                    acc0 *= PAIR_LO_INV;
                } else {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sb[c][2 + ks]), __builtin_bit_cast(bf16x8, qv[ks]), acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sb[c][ks]), __builtin_bit_cast(bf16x8, qv[ks]), acc1, 0, 0, 0);
                    }
                }
                const f32x4 accR = (acc0 + acc1) * (rscale * sinv[c]);
                if (c == 0 || two)
                    *reinterpret_cast<uint2*>(sr + (16 * j + li) * RO_RP + (2 * w + c) * 32 + 8 * g) =
                        make_uint2(cvt_pk_bf16(accR[0], accR[1]), cvt_pk_bf16(accR[2], accR[3]));
            }
        }
        if (more) *reinterpret_cast<uint4*>(s_q + (par ^ 1) * 32 * RO_QP + q_tok * RO_QP + 16 * q_pc) = qn;
        __syncthreads();
        // the pair's R rows: 32 tokens x 16 pieces of 16 bytes; thread -> pieces tid and tid + 256 (a wave: four whole row pieces of 256 bytes)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int id = tid + 256 * r, row = id >> 4, pc = id & 15;
            const int nr = 16 * tt + row;
            if (16 * tt + row < 16 * ntt && nr < N && pc < nchunks)
                *reinterpret_cast<uint4*>(rbase + (size_t)nr * rowr + 16 * pc) = *reinterpret_cast<const uint4*>(sr + row * RO_RP + 16 * pc);
        }
        qi0 = qi_n0;
        qi1 = qi_n1;
    }
}

// N == 0: no tokens -> the state only decays, S_T = S_0 * prod_t alpha_t (no read-out rows exist)
__global__ void gdr_decay_kernel(const float* alpha, const float* s_in, float* s_out, int T, int Hh, int per_bh, int flags)
{
    const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
    float f = 1.f;
    for (int t = 0; t < T; ++t) {
        float al = alpha[((size_t)b * T + t) * Hh + h];
        if (flags & GDKVM_FLAG_GATE_LOGITS) al = 1.0f / (1.0f + expf(-al));
        f *= al;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_bh; i += gridDim.x * blockDim.x)
        s_out[(size_t)bh * per_bh + i] = s_in ? s_in[(size_t)bh * per_bh + i] * f : 0.f;
}

}  // namespace

extern "C" int gdkvm_scan_apply(const void* q, const float* alpha, const float* s_in, void* r_out, float* s_out,
                                float* s_hist, const void* workspace, size_t workspace_bytes,
                                int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int flags, void* stream)
{
    if (int rc = check_common("scan_apply", B, T, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (B == 0) return GDKVM_OK;
    const bool have_tokens = T > 0 && N > 0;
    WsView ws{};
    if (have_tokens) {
        if (int rc = check_ptrs("scan_apply", {q, alpha, workspace}, {r_out, s_in, s_out, s_hist})) return rc;
        if (int rc = carve("scan_apply", const_cast<void*>(workspace), workspace_bytes, B, T, Hh, N, Dk, Dv, &ws)) return rc;
    } else {
        if (T > 0 && !alpha) return gdkvm_fail(GDKVM_ERR_ARG, "scan_apply: null alpha");
        if (int rc = check_ptrs("scan_apply", {}, {alpha, s_in, s_out})) return rc;
    }
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!have_tokens) {
        if (s_hist && T > 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_apply: s_hist needs N > 0");
        if (s_out) {
            hipLaunchKernelGGL(gdr_decay_kernel, dim3(4, (unsigned)(B * Hh)), dim3(256), 0, st, alpha, s_in, s_out, T, Hh,
                               GDKVM_DK * Dv, flags);
            GDKVM_LAUNCH_CHECK("gdr_decay_kernel");
        }
        return GDKVM_OK;
    }
    const bool defer = ws.nb > 4 && r_out != nullptr;     // > 64 tokens per frame: read-out by its own frame-parallel kernel
    AffArgs sa{q, alpha, s_in, ws.pp, ws.gg, ws.qinv, defer ? nullptr : r_out, s_out, s_hist, ws.trash, 0, 0, T, Hh, N, Dv, flags, B * Hh, 0, ws.simg,
               ws.gmax, ws.esc};
    const dim3 grid((unsigned)(B * Hh * (Dv / 16)));
    const bool wide = flags & GDKVM_FLAG_WIDE_RANGE;
    if (int rc = io_dtype == GDKVM_F32 ? launch_affine_any<GDKVM_F32>(wide, defer, s_hist != nullptr, sa, grid, st)
                                       : launch_affine_any<GDKVM_BF16>(wide, defer, s_hist != nullptr, sa, grid, st)) return rc;
    if (defer) {
        ReadoutArgs ra{q, ws.qinv, ws.simg, r_out, Hh, N, Dv, 16 * ws.nb, ws.esc, T};
        // enough workgroups for two per CU: split the frame's token tiles when frames x column groups alone do not give them
        unsigned ny = (unsigned)((Dv / 16 + 7) / 8), nz = 1;
        {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) {
                int n = 0;
                if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
            }
            const long wg = (long)B * T * Hh * ny, ntt = (N + 15) / 16;
            while (wg * nz < 2L * cus && 2 * nz <= (unsigned)(ntt / 2)) nz *= 2;       // (at least two token tiles per workgroup)
        }
        const dim3 rgrid((unsigned)(B * T * Hh), ny, nz);
        static const bool rows_form = !getenv("GDKVM_READOUT_PLAIN");           // (diagnostic: the round-5 kernel for bf16 I/O too)
        if (io_dtype == GDKVM_F32 && wide) hipLaunchKernelGGL((gdr_readout_kernel<GDKVM_F32, FMT_SPLIT3>), rgrid, dim3(256), 0, st, ra);
        else if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gdr_readout_kernel<GDKVM_F32, FMT_PAIR16>), rgrid, dim3(256), 0, st, ra);
        else if (!rows_form && wide) hipLaunchKernelGGL((gdr_readout_kernel<GDKVM_BF16, FMT_SPLIT3>), rgrid, dim3(256), 0, st, ra);
        else if (!rows_form) hipLaunchKernelGGL((gdr_readout_kernel<GDKVM_BF16, FMT_PAIR16>), rgrid, dim3(256), 0, st, ra);
        else if (wide) hipLaunchKernelGGL((gdr_readout_rows_kernel<FMT_SPLIT3>), rgrid, dim3(256), RO_LDS, st, ra);
        else hipLaunchKernelGGL((gdr_readout_rows_kernel<FMT_PAIR16>), rgrid, dim3(256), RO_LDS, st, ra);
        GDKVM_LAUNCH_CHECK("gdr_readout_kernel");
    }
    return GDKVM_OK;
}

// The one SYNCHRONISING entry point of the scan: did the last gdkvm_scan_fwd / gdkvm_scan_apply on this workspace stay inside the range
// of its fp16-pair operands?  Every call is asynchronous, so a range failure cannot come back as its return code; it is loud in the data
// (NaNs) and visible here: the serial kernel leaves 2^e per (clip-head, 16-column slice) in the workspace, NaN where the bound on the state
// -- or a chunk composition on the way to it -- was not finite.  Copies those words back on `stream`, waits for it, and returns
// GDKVM_ERR_RANGE (with the remedy in gdkvm_last_error) or GDKVM_OK.
extern "C" int gdkvm_scan_status(const void* workspace, size_t workspace_bytes, int B, int T, int Hh, int N, int Dk, int Dv, int flags, void* stream)
{
    if (gdr_wide_keys(Dk)) {                               // fp32 recurrence (gdr_general.hip): there is no operand range to leave
        const hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
        return e == hipSuccess ? GDKVM_OK : gdkvm_fail(GDKVM_ERR_LAUNCH, "scan_status: %s", hipGetErrorString(e));
    }
    if (gdr_narrow_keys(Dk)) Dk = GDKVM_DK;                // (gdkvm_scan_fwd ran the Dk = 64 kernels on the same base layout)
    if (int rc = check_common("scan_status", B, T, Hh, N, Dk, Dv, GDKVM_F32, flags)) return rc;
    if (B == 0 || T == 0 || N == 0 || (flags & GDKVM_FLAG_WIDE_RANGE)) return GDKVM_OK;        // nothing ran / full-range operands: no bound to break
    if (int rc = check_ptrs("scan_status", {workspace}, {})) return rc;
    WsView ws;
    if (int rc = carve("scan_status", const_cast<void*>(workspace), workspace_bytes, B, T, Hh, N, Dk, Dv, &ws)) return rc;
    const int nsl = Dv / 16, nsl4 = (nsl + 3) & ~3;
    const size_t words = (size_t)B * Hh * nsl4;
    float* host = static_cast<float*>(malloc(words * sizeof(float)));
    if (!host) return gdkvm_fail(GDKVM_ERR_LAUNCH, "scan_status: out of host memory");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(host, ws.esc, words * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    int bad = -1;
    if (e == hipSuccess)
        for (size_t i = 0; i < words && bad < 0; ++i)
            if ((int)(i % nsl4) < nsl && host[i] != host[i]) bad = (int)(i / nsl4);
    free(host);
    if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "scan_status: %s", hipGetErrorString(e));
    if (bad >= 0)
        return gdkvm_fail(GDKVM_ERR_RANGE, "scan: the state of clip-head %d left the range of the fp16-pair operands (a bound that is not finite, or a "
                                           "chunk composition of a frame of more than 64 tokens beyond it); the results are NaNs -- pass GDKVM_FLAG_WIDE_RANGE", bad);
    return GDKVM_OK;
}

extern "C" int gdkvm_scan_transition(const void* q, const float* alpha, float* phi_out, const void* workspace, size_t workspace_bytes,
                                     int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int flags, void* stream)
{
    if (int rc = check_common("scan_transition", B, T, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (B == 0) return GDKVM_OK;
    if (T == 0 || N == 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_transition: T and N must be positive");
    if (int rc = check_ptrs("scan_transition", {q, alpha, phi_out, workspace}, {})) return rc;
    WsView ws;
    if (int rc = carve("scan_transition", const_cast<void*>(workspace), workspace_bytes, B, T, Hh, N, Dk, Dv, &ws)) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the recurrence on Dk columns from S = I with G = 0 and no read-out:  Phi = prod_t a_t P_t
    if (int rc = gdkvm_zero_async(ws.zero, 64 * 4 * sizeof(float), st)) return rc;       // the one G tile every frame and slice reads
    AffArgs sa{q, alpha, nullptr, ws.pp, ws.zero, ws.qinv, nullptr, phi_out, nullptr, ws.trash, 1, 1, T, Hh, N, GDKVM_DK, flags, B * Hh, 0, nullptr};
    const dim3 grid((unsigned)(B * Hh * (GDKVM_DK / 16)));
    const bool wide = flags & GDKVM_FLAG_WIDE_RANGE;
    return io_dtype == GDKVM_F32 ? launch_affine_any<GDKVM_F32>(wide, false, false, sa, grid, st)
                                 : launch_affine_any<GDKVM_BF16>(wide, false, false, sa, grid, st);
}

// gdr_stitch_kernel -- the sequential part of the time-segmented scan (row n3): with Phi_c and S_loc_c of every segment known,
//     start_0 = S_in,   start_{c+1} = Phi_c start_c + S_loc_c,
// one workgroup per (clip, head, 16-column slice), wave w owning row tile w; exact fp32 MFMA.  The four accumulator tiles are
// exchanged through LDS as lane-linear images, which ARE the B operand under the k permutation "register r of lane (li, g) of
// image m is k = 16m + 4g + r" (the same trick as the scan's exact read-out); double-buffered by segment parity: one barrier each.
// 13.9 us for 16 segments at cfg5 (0.87 us per step).  Not the operands' round trip: requested one segment ahead, or four ahead in a ring of
// register sets behind a raw LDS-only barrier, the kernel took 13.9 / 14.5 us (round 6) -- a step is 16 dependent fp32 MFMAs (32 cycles
// each), the image exchange through LDS and four strided stores, on 32 workgroups; left in its simple form.
__global__ __launch_bounds__(256) void gdr_stitch_kernel(const float* phi, const float* s_loc, const float* s_in, float* starts, float* s_end,
                                                          int S, int Hh, int Dv)
{
    __shared__ f32x4 s_img[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsl = Dv / 16, sl = blockIdx.x % nsl, bh = blockIdx.x / nsl, b = bh / Hh, h = bh % Hh;
    auto tile_ptr = [&](const float* base, size_t mat) { return base + mat * GDKVM_DK * Dv + (size_t)(16 * w + 4 * g) * Dv + 16 * sl + li; };
    f32x4 cur = {0.f, 0.f, 0.f, 0.f};
    if (s_in) {
        const float* p = tile_ptr(s_in, (size_t)bh);
#pragma unroll
        for (int r = 0; r < 4; ++r) cur[r] = p[(size_t)r * Dv];
    }
    for (int c = 0; c < S; ++c) {
        const size_t mat = ((size_t)b * S + c) * Hh + h;
        float* st = const_cast<float*>(tile_ptr(starts, mat));
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(size_t)r * Dv] = cur[r];
        s_img[c & 1][w][lane] = cur;
        const float* lp = tile_ptr(s_loc, mat);
        f32x4 nxt;
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[r] = lp[(size_t)r * Dv];
        f32x4 pa[4];                                       // Phi_c[row 16w + li][16m + 4g .. +3]
#pragma unroll
        for (int m = 0; m < 4; ++m) pa[m] = *reinterpret_cast<const f32x4*>(phi + mat * GDKVM_DK * GDKVM_DK + (size_t)(16 * w + li) * GDKVM_DK + 16 * m + 4 * g);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 bv = s_img[c & 1][m][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) nxt = mfma4(pa[m][r], bv[r], nxt);
        }
        cur = nxt;
    }
    if (s_end) {
        float* p = const_cast<float*>(tile_ptr(s_end, (size_t)bh));
#pragma unroll
        for (int r = 0; r < 4; ++r) p[(size_t)r * Dv] = cur[r];
    }
}

extern "C" int gdkvm_scan_stitch(const float* phi, const float* s_loc, const float* s_in, float* starts, float* s_end,
                                 int B, int S, int Hh, int Dk, int Dv, void* stream)
{
    if (B < 0 || S <= 0 || Hh <= 0 || Dv <= 0 || Dv % 16) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_stitch: B=%d S=%d Hh=%d Dv=%d", B, S, Hh, Dv);
    if (Dk != GDKVM_DK) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_stitch: Dk=%d unsupported (kernels are built for Dk=%d)", Dk, GDKVM_DK);
    if (B == 0) return GDKVM_OK;
    if (int rc = check_ptrs("scan_stitch", {phi, s_loc, starts}, {s_in, s_end})) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    hipLaunchKernelGGL(gdr_stitch_kernel, dim3((unsigned)(B * Hh * (Dv / 16))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       phi, s_loc, s_in, starts, s_end, S, Hh, Dv);
    GDKVM_LAUNCH_CHECK("gdr_stitch_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_scan_fwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                              const float* s_in, void* r_out, float* s_out, float* s_hist, void* workspace, size_t workspace_bytes,
                              int B, int T, int Hh, int N, int Dk, int Dv,
                              int io_dtype, int rule, int flags, void* stream)
{
    if (gdr_wide_keys(Dk)) {
        // key widths 72 .. 256: the definitional recurrence, one workgroup per state slice (gdr_general.hip); inference only
        if (s_hist) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd: s_hist (training) needs Dk <= %d", GDKVM_DK);
        if (int rc = check_ptrs("scan_fwd", {q, k, v, alpha, beta}, {s_in, r_out, s_out})) return rc;
        return gdr_general_scan_fwd(q, k, v, alpha, beta, s_in, r_out, s_out, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, static_cast<hipStream_t>(stream));
    }
    if (gdr_narrow_keys(Dk) && B > 0 && T > 0 && N > 0) {
        // key widths 8 .. 56: the Dk = 64 kernels on zero-extended copies of q, k and the state (gdr_ws.hpp)
        if (s_hist) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd: s_hist needs Dk=%d (training pads the keys itself)", GDKVM_DK);
        if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "scan_fwd: io_dtype=%d", io_dtype);
        if (Hh <= 0 || Dv <= 0 || Dv % 16 || N > GDKVM_MAX_N) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd: Hh=%d N=%d Dv=%d", Hh, N, Dv);
        if (int rc = check_ptrs("scan_fwd", {q, k, v, alpha, beta, workspace}, {s_in, r_out, s_out})) return rc;
        const size_t base = gdr_up256(gdr_workspace_bytes(B, T, Hh, N, GDKVM_DK, Dv));
        if (workspace_bytes < base + gdr_narrow_extra_bytes(B, T, Hh, N, Dv))
            return gdkvm_fail(GDKVM_ERR_WORKSPACE, "scan_fwd: workspace %zu < %zu bytes", workspace_bytes, base + gdr_narrow_extra_bytes(B, T, Hh, N, Dv));
        if (int rc = gdkvm_check_device()) return rc;
        hipStream_t st = static_cast<hipStream_t>(stream);
        const size_t es = io_dtype == GDKVM_F32 ? 4 : 2, rows = (size_t)B * T * N * Hh, BH = (size_t)B * Hh;
        char* p = static_cast<char*>(workspace) + base;
        char* q_p = p;  p += gdr_up256(rows * GDKVM_DK * 4);
        char* k_p = p;  p += gdr_up256(rows * GDKVM_DK * 4);
        float* si_p = reinterpret_cast<float*>(p);  p += gdr_up256(BH * GDKVM_DK * Dv * 4);
        float* so_p = reinterpret_cast<float*>(p);
        const size_t rq = Dk * es / 16, rq64 = GDKVM_DK * es / 16, sq = (size_t)Dk * Dv * 4 / 16, sq64 = (size_t)GDKVM_DK * Dv * 4 / 16;
        if (int rc = gdr_block_copy(q, q_p, rows, rq, rq64, rq, rq64, st)) return rc;
        if (int rc = gdr_block_copy(k, k_p, rows, rq, rq64, rq, rq64, st)) return rc;
        if (s_in) if (int rc = gdr_block_copy(s_in, si_p, BH, sq, sq64, sq, sq64, st)) return rc;
        if (int rc = gdkvm_scan_fwd(q_p, k_p, v, alpha, beta, s_in ? si_p : nullptr, r_out, s_out ? so_p : nullptr, nullptr, workspace, base,
                                    B, T, Hh, N, GDKVM_DK, Dv, io_dtype, rule, flags, stream)) return rc;
        return s_out ? gdr_block_copy(so_p, s_out, BH, sq64, sq, sq, sq, st) : GDKVM_OK;
    }
    if (s_hist) flags |= GDKVM_FLAG_TRAIN;              // the backward reads extra operand layouts from the workspace
    if (rule == GDKVM_RULE_DELTA_PARALLEL) flags |= GDKVM_FLAG_WIDE_RANGE;      // not contractive: full-range operands
    if (int rc = gdkvm_scan_prep(q, k, v, beta, workspace, workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream)) return rc;
    return gdkvm_scan_apply(q, alpha, s_in, r_out, s_out, s_hist, workspace, workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, flags, stream);
}

// Backward pass (called by gdkvm_scan_bwd).  The reverse-time recurrence has the forward's affine shape,
//     dS = a (dS' - Wt^T (Kn dS')) + Qn^T dR  =  a P^T dS' + Gb,      Gb = Qn^T dR  [Dk, Dv] per frame,
// so it runs on gdr_affine_scan_kernel itself: P^T images from the training-mode fold, Gb tiles from the kernel below,
// frames visited last to first, no read-out, and the state BEFORE every step (= dS' of that frame) written like s_hist.
struct BwdGArgs { const float* qnT; const void* d_r; float* gb; int Hh, N, Dv; };

template <int IO>
__global__ __launch_bounds__(256) void gdr_bwd_g_kernel(BwdGArgs a)
{
    constexpr int NB = 4, NP = 64;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t fh = blockIdx.x;
    const int h = (int)(fh % a.Hh), N = a.N, Dv = a.Dv, nsl = Dv / 16;
    const size_t bt = fh / a.Hh;
    f32x4 qa[NB];                                          // A operand: Qn^T rows 16w + li, k = token 16I + 4g + r (padding columns are 0)
#pragma unroll
    for (int I = 0; I < NB; ++I) qa[I] = *reinterpret_cast<const f32x4*>(a.qnT + (fh * GDKVM_DK + 16 * w + li) * NP + 16 * I + 4 * g);
    for (int cV = blockIdx.y; cV < nsl; cV += gridDim.y) {
        float x[NB][4];
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[I][r] = load1<IO>(a.d_r, ((bt * N + min(16 * I + 4 * g + r, N - 1)) * a.Hh + h) * Dv + 16 * cV + li);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                if (I & 1) acc1 = mfma4(qa[I][r], x[I][r], acc1);
                else acc0 = mfma4(qa[I][r], x[I][r], acc0);
            }
        reinterpret_cast<f32x4*>(a.gb)[((fh * nsl + cV) * 4 + w) * 64 + lane] = acc0 + acc1;
    }
}

int gdr_launch_reverse_scan(const WsView& ws, const float* alpha, const void* d_r, const float* ds_out, float* ds_hist,
                            float* ds_in, float* gb, int B, int T, int Hh, int N, int Dv, int io_dtype, int flags, hipStream_t st)
{
    if (d_r) {                                             // (d_r == NULL: the caller has filled gb itself)
        BwdGArgs ga{ws.qnT, d_r, gb, Hh, N, Dv};
        const dim3 ggrid((unsigned)(B * T * Hh), (unsigned)((Dv / 16 + 3) / 4));
        if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gdr_bwd_g_kernel<GDKVM_F32>), ggrid, dim3(256), 0, st, ga);
        else hipLaunchKernelGGL((gdr_bwd_g_kernel<GDKVM_BF16>), ggrid, dim3(256), 0, st, ga);
        GDKVM_LAUNCH_CHECK("gdr_bwd_g_kernel");
    }
    AffArgs sa{nullptr, alpha, ds_out, ws.ppt, gb, nullptr, nullptr, ds_in, ds_hist, ws.trash, 0, 0, T, Hh, N, Dv, flags, B * Hh, 1, nullptr};
    const dim3 grid((unsigned)(B * Hh * (Dv / 16)));
    // gradients have no natural magnitude: the reverse recurrence keeps the full-range three-term bf16 operands (P^T images)
    return io_dtype == GDKVM_F32 ? launch_affine<GDKVM_F32, FMT_SPLIT3, false, true>(sa, grid, st)
                                 : launch_affine<GDKVM_BF16, FMT_SPLIT3, false, true>(sa, grid, st);
}
