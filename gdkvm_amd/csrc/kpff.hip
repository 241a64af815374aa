// kpff.hip -- SURVEY.md §8 row a4 / Appendix A.5: Key-Pixel Feature Fusion, one fused kernel.
//
//   Gms = mean_{s in 1,2,4} cellmean_s(G)            (multi-scale "global key feature")
//   g   = sigmoid([P ; L ; Gms] Wa^T + ba) = (g_l | g_g)
//   F   = P + g_l * (L Wl^T) + g_g * (Gms Wg^T)
//
// One workgroup owns a tile of <= 64 tokens made of whole 4-row bands of the h x w grid (or the whole
// frame when it has <= 64 tokens), so every 2x2 and 4x4 pooling cell is tile-local; the four channel mixes run as ONE
// MFMA pass over the concatenated input [P ; L ; Gms] with bias, sigmoid, gating and the residual in the epilogue.
// Three arms, chosen by gdkvm_kpff_fwd from the I/O type, the channel counts and the workspace:
//   kpff_bf16_kernel<NT>  bf16 I/O, channels % 32 == 0: v_mfma_f32_16x16x32_bf16, NT 64-token tiles per 8-wave workgroup as
//                         bf16 B images in LDS, weights re-packed in MFMA-fragment order and streamed through a static
//                         register ring (the inference path; also gdkvm_proj_rows' machinery)
//   kpff_split_kernel     fp32 I/O with a workspace: every operand as two bf16 terms, three bf16 MFMAs per product
//   kpff_kernel<IO>       exact fp32 MFMA (v_mfma_f32_16x16x4_f32): rows staged in LDS as fp32 (padded rows -> conflict-free
//                         16-byte operand reads), G pooled in place, weights read row-major from L2 with 16-byte loads; odd
//                         channel counts, or a caller that hands over no workspace
// plus the training forward (saves gates / mixes / pooled feature) and the elementwise halves of the backward.
#include <atomic>
#include <type_traits>
#include "gdkvm_common.hpp"

namespace {

struct KpffSave { void* gates; void* lp; void* gp; void* gms; };   // training: [M,2Cp] [M,Cp] [M,Cp] [M,Cv] (io dtype) or NULL

struct KpffArgs {
    const void* L; const void* G; const void* P;
    const float* wa; const float* ba; const float* wl; const float* wg;
    void* out;
    int Ck, Cv, Cp, h, w, rows_per_tile, tiles_per_frame;
    KpffSave sv;
    int cols_per_tile, col_tiles;     // grids wider than 16 columns: a tile is a row band x a block of 16 columns
};

template <int I, int E, class F>
__device__ __forceinline__ void kpff_static_for(F&& f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        kpff_static_for<I + 1, E>(f);
    }
}

constexpr int KPFF_TM = 64;       // tokens per workgroup tile
constexpr int KPFF_PAD = 4;       // row padding in floats: stride % 64 == 4 -> 16 rows cover all 64 banks

template <int IO>
__global__ __launch_bounds__(256) void kpff_kernel(KpffArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_x[];   // [KPFF_TM][Cin + PAD]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ck = a.Ck, Cv = a.Cv, Cp = a.Cp, Cin = Cp + Ck + Cv, ld = Cin + KPFF_PAD;
    const int f = blockIdx.x / a.tiles_per_frame, tf = blockIdx.x % a.tiles_per_frame;
    const int rt = tf / a.col_tiles, ct = tf - rt * a.col_tiles;
    const int N = a.h * a.w;
    const int row0 = rt * a.rows_per_tile, col0 = ct * a.cols_per_tile;
    const int nrows = min(a.rows_per_tile, a.h - row0), W = min(a.cols_per_tile, a.w - col0);      // W: the TILE's width
    const int ntok = nrows * W;
    // token tok of the tile (row-major inside the tile) -> token of the frame
    auto gtok = [&](int tok) { const int ty = tok / W; return (row0 + ty) * a.w + col0 + (tok - ty * W); };

    // ---- stage [P ; L ; G] rows (4 channels per thread, 16-byte LDS stores) -------------------------
    {
        const int q4 = Cin / 4;
        for (int idx = tid; idx < KPFF_TM * q4; idx += 256) {
            const int tok = idx / q4, c = (idx - tok * q4) * 4;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (tok < ntok) {
                const size_t row = (size_t)f * N + gtok(tok);
                if (c < Cp) x = load4<IO>(a.P, row * Cp + c);
                else if (c < Cp + Ck) x = load4<IO>(a.L, row * Ck + (c - Cp));
                else x = load4<IO>(a.G, row * Cv + (c - Cp - Ck));
            }
            *reinterpret_cast<f32x4*>(s_x + (size_t)tok * ld + c) = x;
        }
    }
    __syncthreads();
    // ---- multi-scale pooling of G in place: one thread per (4x4 cell, channel) -----------------------
    {
        const int cw = (W + 3) / 4, chh = (nrows + 3) / 4;
        float* gx = s_x + Cp + Ck;
        for (int idx = tid; idx < cw * chh * Cv; idx += 256) {
            const int c = idx % Cv, cell = idx / Cv;
            const int y0 = (cell / cw) * 4, x0 = (cell % cw) * 4;
            const int y1 = min(y0 + 4, nrows), x1 = min(x0 + 4, W);
            float s4 = 0.f, s2[4] = {0.f, 0.f, 0.f, 0.f};
            int n2[4] = {0, 0, 0, 0};
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const float v = gx[(size_t)(y * W + x) * ld + c];
                    const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                    s2[q] += v; n2[q] += 1; s4 += v;
                }
            const float m4 = s4 / (float)((y1 - y0) * (x1 - x0));
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                    float* p = gx + (size_t)(y * W + x) * ld + c;
                    *p = (*p + s2[q] / (float)n2[q] + m4) * (1.0f / 3.0f);
                }
        }
    }
    __syncthreads();
    if (a.sv.gms) {                                            // training: keep the pooled feature for the backward
        for (int idx = tid; idx < ntok * Cv; idx += 256) {
            const int tok = idx / Cv, c = idx - tok * Cv;
            store1<IO>(a.sv.gms, ((size_t)f * N + gtok(tok)) * Cv + c, s_x[(size_t)tok * ld + Cp + Ck + c]);
        }
    }

    // ---- fused channel mixes: wave owns output channels o = 64*chunk + 16*wave + li -----------------
    const int kbP = Cp / 16, kbL = Ck / 16, kbG = Cv / 16;
    for (int chunk = 0; chunk * 64 < Cp; ++chunk) {
        const int ob = chunk * 64 + 16 * w_id;            // wave-uniform
        if (ob >= Cp) break;
        const int o = ob + li;
        f32x4 gl[4], gg[4], lp[4], gp[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gl[mt] = gg[mt] = lp[mt] = gp[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wal = a.wa + (size_t)o * Cin + 4 * g;
        const float* wag = a.wa + (size_t)(Cp + o) * Cin + 4 * g;
        const float* xa = s_x + (size_t)li * ld + 4 * g;

        // The weight fragments (16 bytes per lane and stream, row-major rows of Wa / Wl / Wg in L2) are fetched THREE k-blocks
        // ahead into a ring of register sets with static indices: round 1 loaded them right in front of the MFMAs that use them,
        // one exposed L2 round trip per k-block -- 36 of them per 64 output channels, most of the arm's 390 us.
        auto run = [&](int kb0, int n, const float* wmix, f32x4* mix, auto hasmix) __attribute__((always_inline)) {
            constexpr bool MIX = decltype(hasmix)::value;
            constexpr int WD = 3;
            if (n <= 0) return;
            f32x4 rb[WD][3];
            auto wl3 = [&](int j, f32x4 (&d)[3]) __attribute__((always_inline)) {
                j = min(j, n - 1);
                d[0] = *reinterpret_cast<const f32x4*>(wal + 16 * (kb0 + j));
                d[1] = *reinterpret_cast<const f32x4*>(wag + 16 * (kb0 + j));
                if constexpr (MIX) d[2] = *reinterpret_cast<const f32x4*>(wmix + 16 * j);
            };
#pragma unroll
            for (int d = 0; d < WD; ++d) wl3(d, rb[d]);
            // token fragments: one set, tile mt's fragment for the next k-block requested right behind the MFMAs that used this one's
            f32x4 av[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(xa + (size_t)mt * 16 * ld + 16 * kb0);
            auto body = [&](int i, auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                const int kbn = kb0 + min(i + 1, n - 1);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gl[mt] = mfma4(av[mt][r], rb[j][0][r], gl[mt]);
                        gg[mt] = mfma4(av[mt][r], rb[j][1][r], gg[mt]);
                        if constexpr (MIX) mix[mt] = mfma4(av[mt][r], rb[j][2][r], mix[mt]);
                    }
                    av[mt] = *reinterpret_cast<const f32x4*>(xa + (size_t)mt * 16 * ld + 16 * kbn);
                }
                wl3(i + WD, rb[j]);
            };
            int i = 0;
            for (; i + WD <= n; i += WD) kpff_static_for<0, WD>([&](auto jc) { body(i + decltype(jc)::value, jc); });
            const int rem = n - i;
            kpff_static_for<0, WD - 1>([&](auto jc) { if (decltype(jc)::value < rem) body(i + decltype(jc)::value, jc); });
        };
        run(0, kbP, nullptr, nullptr, std::false_type{});
        run(kbP, kbL, a.wl + (size_t)o * Ck + 4 * g, lp, std::true_type{});
        run(kbP + kbL, kbG, a.wg + (size_t)o * Cv + 4 * g, gp, std::true_type{});

        const float bl = a.ba[o], bg = a.ba[Cp + o];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = 16 * mt + 4 * g + r;
                if (tok < ntok) {
                    const float sl = 1.0f / (1.0f + expf(-(gl[mt][r] + bl)));
                    const float sg = 1.0f / (1.0f + expf(-(gg[mt][r] + bg)));
                    const float y = s_x[(size_t)tok * ld + o] + sl * lp[mt][r] + sg * gp[mt][r];
                    const size_t grow = (size_t)f * N + gtok(tok);
                    store1<IO>(a.out, grow * Cp + o, y);
                    if (a.sv.gates) {
                        store1<IO>(a.sv.gates, grow * 2 * Cp + o, sl);
                        store1<IO>(a.sv.gates, grow * 2 * Cp + Cp + o, sg);
                        store1<IO>(a.sv.lp, grow * Cp + o, lp[mt][r]);
                        store1<IO>(a.sv.gp, grow * Cp + o, gp[mt][r]);
                    }
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------------
// bf16 arm: same tiling, operands in bf16 on v_mfma_f32_16x16x32_bf16 (fp32 accumulate), 16x the fp32 MFMA rate.
// The tile is staged as bf16 (72 KB -> two workgroups per CU), pooled in place (fp32 math, bf16 store), and the
// weights are read as bf16 from a workspace copy made by kpff_pack_weights_kernel in MFMA-fragment order (a B
// fragment = 8 consecutive k of one output channel = one 16-byte load, a wave's 64 fragments contiguous).  Lane l = 16g + i:
//   A = X[row i][k 8g..8g+7],  B = W[col i][k 8g..8g+7],  C/D reg r = D[row 4g + r][col i].

#ifdef KPFF_STAMPS
// Diagnostic build only (tools/stamp_kpff.py): lane 0 of every wave of the first 64 workgroups stamps s_memtime at the phase
// boundaries of kpff_bf16_kernel into a buffer of its own ([block][wave][16]).  Never compiled into the product library.
__device__ unsigned long long* g_kpff_stamps = nullptr;
extern "C" void gdkvm_kpff_diag_set_buffer(unsigned long long* p) { hipMemcpyToSymbol(HIP_SYMBOL(g_kpff_stamps), &p, sizeof(p)); }
#define KPFF_STAMP(slot)                                                                                           \
    do {                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        unsigned long long t__;                                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                                \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (g_kpff_stamps && blockIdx.x < 64 && (threadIdx.x & 63) == 0) g_kpff_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = t__; \
    } while (0)
#else
#define KPFF_STAMP(slot) do {} while (0)
#endif

struct KpffBf16Args {
    const bf16_t* L; const bf16_t* G; const bf16_t* P;
    const bf16_t* wa; const float* ba; const bf16_t* wl; const bf16_t* wg;
    bf16_t* out;
    int Ck, Cv, Cp, h, w, rows_per_tile, tiles_per_frame;
    KpffSave sv;
    int cols_per_tile, col_tiles;
};

// Row padding of the bf16 token tiles, in elements.  The MFMA B fragments are ds_read_b128 with lane (g, li) at row li, byte 16 g:
// the hardware serves such a read in four groups of 16 lanes, each holding ALL sixteen rows with g differing by at most one
// (MI355X_MICROARCH.md §LDS: {0-3, 12-15, 20-27}, ...), so the sixteen 16-byte pieces are conflict-free exactly when the row pitch
// is 32 bytes mod 64 -- channel counts are multiples of 32 (64 bytes), hence 16 elements of padding.  (Rounds 1-2 padded by 8: a
// pitch of 16 bytes mod 64 makes every fragment read a two-way conflict -- SQ_LDS_BANK_CONFLICT was 48 % of SQ_LDS_IDX_ACTIVE in
// kpff_bf16_kernel, profiles/r03_d_hotpath_cfg2_pmc_lds.csv.)
constexpr int KPFF_PAD16 = 16;

// Weights are re-packed to bf16 in MFMA-fragment order: for output tile ot (16 channels) and k-step ks (32 inputs) the
// 64 lanes' B fragments (lane 16g+i = W[16ot+i][32ks+8g .. +7]) are contiguous, so a wave's B load is one 1 KiB access.
// (Reading the row-major matrix directly makes each wave-instruction touch 16 rows x 64 B: measured 9 B/clk/CU.)
// dst_lo (fp32 arm on splits, below): a second pack of the same layout with the bf16 remainders w - bf16(w).
__global__ void kpff_pack_weights_kernel(const float* wa, const float* wl, const float* wg, bf16_t* dst, bf16_t* dst_lo,
                                         int Cp, int Ck, int Cv)
{
    const int Cin = Cp + Ck + Cv;
    const size_t na = (size_t)2 * Cp * Cin, nl = (size_t)Cp * Ck, ng = (size_t)Cp * Cv, n = na + nl + ng;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float* src; int K; size_t e;
        if (i < na) { src = wa; K = Cin; e = i; }
        else if (i < na + nl) { src = wl; K = Ck; e = i - na; }
        else { src = wg; K = Cv; e = i - na - nl; }
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        const size_t blk = e >> 9;                          // = ot * (K/32) + ks
        const int KS = K / 32, ks = (int)(blk % KS), ot = (int)(blk / KS);
        const float v = src[(size_t)(16 * ot + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4) + j];
        const bf16_t hi = f32_to_bf16(v);
        dst[i] = hi;
        if (dst_lo) dst_lo[i] = f32_to_bf16(v - bf16_to_f32(hi));
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// One software-pipelined pass of `n` k-steps (32 inputs each, starting at k-step ks0 of the LDS tile) against NS weight
// streams in fragment order, for MT token tiles.  The WEIGHTS are the A operand and the tokens the B operand, so an
// accumulator register holds D[channel 4g+r][token i]: a lane ends up with 4 consecutive channels of one token (8-byte
// stores, one 8-byte LDS read for the residual) instead of one channel of 4 tokens (four 2-byte stores).
// Weight fragments are prefetched KPFF_WD k-steps ahead (L2 latency ~600-800 cycles vs ~130-400 cycles of MFMA per step) into
// a ring of KPFF_WD register sets with STATIC indices: k-step i uses set i % KPFF_WD and refills it for k-step i + KPFF_WD.
// (Until round 2 the sets were rotated with register copies -- b[0] = b[1]; ...; b[3] = load -- and the copy of the freshly
// loaded set made the compiler wait for the newest fetch every k-step: s_waitcnt vmcnt(0) / vmcnt(1) inside the loop.)
// OT output tiles at once: a token fragment read from LDS feeds NS*OT MFMAs (the kernel is LDS-read-bound at OT = 1: every
// 32-deep k-step re-reads MT KiB of tokens for MT*NS MFMAs); ot_stride = elements between consecutive output tiles' packs.
#ifndef KPFF_WD_STEPS
#define KPFF_WD_STEPS 4
#endif
constexpr int KPFF_WD = KPFF_WD_STEPS;               // ring depth = k-steps per unrolled trip

// The ring of weight register sets of one pass.  A caller may own it and PRIME it (kpff_prime: the pass's first KPFF_WD k-steps
// requested) long before the pass runs -- at kernel entry, or right behind the previous output tile's pass of the same kind -- so that
// the pass opens on fragments that have landed instead of on an L2 round trip (round 3 stamps: the two short passes of an output
// tile took 4.9 k cycles against an MFMA floor of 2.6 k, the long one ~1 k over its floor: one exposed prologue each).
template <int NS, int OT> struct KRing { bf16x8 b0[KPFF_WD][OT]; bf16x8 b1[KPFF_WD][OT]; };

#ifdef KPFF_ABL_WSAME                                       // ablation: every weight fragment from one L1-resident KiB
#define KPFF_WOFF(x) ((size_t)0 * (x))
#else
#define KPFF_WOFF(x) (x)
#endif
template <int NS, int OT>
__device__ __forceinline__ void kpff_wload(const bf16_t* w0, const bf16_t* w1, size_t ot_stride, int slot_ks, int n, bf16x8 (&d0)[OT], bf16x8 (&d1)[OT])
{
    const size_t off = KPFF_WOFF((size_t)min(slot_ks, n - 1) * 512);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        d0[o] = *reinterpret_cast<const bf16x8*>(w0 + o * ot_stride + off);
        if constexpr (NS == 2) d1[o] = *reinterpret_cast<const bf16x8*>(w1 + o * ot_stride + off);
    }
}
template <int NS, int OT>
__device__ __forceinline__ void kpff_prime(KRing<NS, OT>& r, const bf16_t* w0, const bf16_t* w1, size_t ot_stride, int n)
{
#pragma unroll
    for (int d = 0; d < KPFF_WD; ++d) kpff_wload<NS, OT>(w0, w1, ot_stride, d, n, r.b0[d], r.b1[d]);
}

template <int NS, int MT, int OT, bool PRIMED = false>
__device__ __forceinline__ void kpff_stream(const bf16_t* xb, int ld, int ks0, int n, const bf16_t* w0, const bf16_t* w1, size_t ot_stride,
                                            f32x4 (&acc0)[OT][MT], f32x4 (&acc1)[OT][MT], KRing<NS, OT>& ring)
{
    if (n <= 0) return;
    auto& b0 = ring.b0;
    auto& b1 = ring.b1;
    auto wload = [&](int slot_ks, bf16x8 (&d0)[OT], bf16x8 (&d1)[OT]) __attribute__((always_inline)) {
        kpff_wload<NS, OT>(w0, w1, ot_stride, slot_ks, n, d0, d1);
    };
    if constexpr (!PRIMED) {
#pragma unroll
        for (int d = 0; d < KPFF_WD; ++d) wload(d, b0[d], b1[d]);
    }
    bf16x8 xa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xa[mt] = *reinterpret_cast<const bf16x8*>(xb + (size_t)mt * 16 * ld + 32 * ks0);
    // One register set of token fragments, refilled fragment by fragment: tile mt's fragment for the NEXT k-step is requested
    // right behind the MFMAs that consumed this k-step's -- MT - 1 tiles of MFMAs (>= the LDS latency) before it is needed, and
    // LDS returns in order, so the wait in front of each tile's MFMAs is a counted lgkmcnt(MT - 1), never a drain.  (A second
    // full set -- a k-step ahead -- cost 32 more registers and spilled; a plain loop over one set waited lgkmcnt(0) per read.)
    auto body = [&](int i, auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value % KPFF_WD;   // ring set of k-step i
        const int ksn = ks0 + min(i + 1, n - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int o = 0; o < OT; ++o) {
                acc0[o][mt] = mfma_bf16(b0[j][o], xa[mt], acc0[o][mt]);
                if constexpr (NS == 2) acc1[o][mt] = mfma_bf16(b1[j][o], xa[mt], acc1[o][mt]);
            }
            xa[mt] = *reinterpret_cast<const bf16x8*>(xb + (size_t)mt * 16 * ld + 32 * ksn);
        }
        wload(i + KPFF_WD, b0[j], b1[j]);                  // refill the weight set just used
#ifndef KPFF_NO_SCHED_GROUPS
        // Pin the order written above.  Left alone, the machine scheduler clusters: all of the k-step's MFMAs first, then the MT
        // LDS reads in one batch -- and the next k-step opens with s_waitcnt lgkmcnt(MT - 1) on a read issued a few cycles earlier:
        // a full LDS round trip exposed per k-step (round 3 disassembly; the MFMA pipe was 40 % busy inside this loop).  One group
        // per token tile: its NS * OT MFMAs, then its fragment read; the weight loads last.
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            __builtin_amdgcn_sched_group_barrier(0x008, NS * OT, 0);       // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x020, NS * OT, 0);           // VMEM read
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    int i = 0;
    for (; i + KPFF_WD <= n; i += KPFF_WD)                 // (i stays a multiple of KPFF_WD: the set ids are static)
        kpff_static_for<0, KPFF_WD>([&](auto jc) { body(i + decltype(jc)::value, jc); });
    const int rem = n - i;
    kpff_static_for<0, KPFF_WD - 1>([&](auto jc) {
        if (decltype(jc)::value < rem) body(i + decltype(jc)::value, jc);
    });
}

template <int NS, int MT, int OT>
__device__ __forceinline__ void kpff_stream(const bf16_t* xb, int ld, int ks0, int n, const bf16_t* w0, const bf16_t* w1, size_t ot_stride,
                                            f32x4 (&acc0)[OT][MT], f32x4 (&acc1)[OT][MT])
{
    KRing<NS, OT> ring;
    kpff_stream<NS, MT, OT, false>(xb, ld, ks0, n, w0, w1, ot_stride, acc0, acc1, ring);
}

// NT = 64-token tiles per workgroup (4*NT waves).  NT = 2 halves the weight traffic per token: at 64 tokens per
// workgroup the kernel sits at the L2 balance point (0.75 MB of weights per 64 tokens ~ 64 flop per L2 byte).
// OT = output tiles a wave accumulates at once; the workgroup has 4*NT/OT waves (NT = 2, OT = 2: four waves with up to 512
// registers each, 64 accumulator tiles per wave).
#ifndef KPFF_OT
#define KPFF_OT 1
#endif
// SB = rows a sub-tile occupies in the workgroup's tile: 64, or 56 when both sub-tiles hold at most 56 tokens (two 7x7 frames:
// 98 tokens in SEVEN 16-row token tiles instead of eight -- an eighth off the MFMA passes, the fragment reads and the epilogues).
template <int NT, int OT, int SB = KPFF_TM>
__global__ __launch_bounds__(256 * NT / OT, (NT == 1 ? 2 : 1)) void kpff_bf16_kernel(KpffBf16Args a, int total_tiles)
{
    constexpr int NTHR = 256 * NT / OT, TMW = SB * NT, MT = (TMW + 15) / 16;
    static_assert(SB == KPFF_TM || NT == 2, "packed sub-tiles: two per workgroup");
    extern __shared__ __attribute__((aligned(16))) bf16_t s_xb[];   // [TMW][Cin + PAD16]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ck = a.Ck, Cv = a.Cv, Cp = a.Cp, Cin = Cp + Ck + Cv, ld = Cin + KPFF_PAD16;
    const int N = a.h * a.w;
    // sub-tile s of this workgroup = global tile NT*blockIdx.x + s  ->  (frame, row band, block of 16 columns)
    int t_f[NT], t_n0[NT], t_ntok[NT], t_nrows[NT], t_w[NT];
#pragma unroll
    for (int sb = 0; sb < NT; ++sb) {
        const int tile = NT * blockIdx.x + sb;
        const bool valid = tile < total_tiles;
        const int f = valid ? tile / a.tiles_per_frame : 0, tf = valid ? tile % a.tiles_per_frame : 0;
        const int rt = tf / a.col_tiles, ct = tf - rt * a.col_tiles;
        const int row0 = rt * a.rows_per_tile, col0 = ct * a.cols_per_tile;
        t_nrows[sb] = valid ? min(a.rows_per_tile, a.h - row0) : 0;
        t_w[sb] = min(a.cols_per_tile, a.w - col0);                       // the tile's width
        t_f[sb] = f; t_n0[sb] = row0 * a.w + col0; t_ntok[sb] = t_nrows[sb] * t_w[sb];
    }
    // token tok of sub-tile sb (row-major inside the tile) -> token of the frame
    // (Integer divisions by run-time divisors cost ~30 vector instructions each and sat in every staged piece and every epilogue
    // row: a tile as wide as the grid -- the usual case -- needs none, and the others use an exact float reciprocal, valid for the
    // small non-negative indices here: floor(n / d) == (int)((n + 0.5f) * (1.0f / d)) for n < 2^20.)
    const bool full_w = a.cols_per_tile >= a.w;               // uniform: the tile's tokens are consecutive tokens of the frame
    float inv_tw[NT];
#pragma unroll
    for (int sb = 0; sb < NT; ++sb) inv_tw[sb] = 1.0f / (float)max(t_w[sb], 1);
    auto gtok = [&](int sb, int tok) {
#ifdef KPFF_ABL_INTDIV                                          // ablation: the integer divisions of rounds 1-2
        { const int ty = tok / t_w[sb]; return t_n0[sb] + ty * a.w + (tok - ty * t_w[sb]); }
#endif
        if (full_w) return t_n0[sb] + tok;
        const int ty = (int)(((float)tok + 0.5f) * inv_tw[sb]);
        return t_n0[sb] + ty * a.w + (tok - ty * t_w[sb]);
    };

    // The weight ring of the long pass (the gate mixes), primed for this wave's first output tile before anything else: the fragments
    // travel while the tile is staged and pooled.
    const int ksP = Cp / 32, ksL = Ck / 32, ksG = Cv / 32, KSa = Cin / 32;
    constexpr int OB_STEP = 16 * OT * (NTHR / 64);
    auto wa_g = [&](int ob) { return a.wa + ((size_t)(ob / 16) * KSa * 64 + lane) * 8; };
    auto wl_g = [&](int ob) { return a.wl + ((size_t)(ob / 16) * ksL * 64 + lane) * 8; };
    auto wg_g = [&](int ob) { return a.wg + ((size_t)(ob / 16) * ksG * 64 + lane) * 8; };
    KRing<2, OT> ring_g;
#ifndef KPFF_SKIP_GEMM
    if (16 * OT * w_id < Cp) kpff_prime<2, OT>(ring_g, wa_g(16 * OT * w_id), wa_g(Cp + 16 * OT * w_id), (size_t)KSa * 512, KSa);
#endif
    KPFF_STAMP(0);
    // ---- stage the [P ; L] rows as they are (8 channels = 16 bytes per thread, 4 loads in flight) -----------
    {
        const int q8 = (Cp + Ck) / 8, total = TMW * q8;
        const float inv_q8 = 1.0f / (float)q8;
        for (int base = tid; base < total; base += 4 * NTHR) {
            uint4 x[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTHR;
#ifdef KPFF_ABL_INTDIV
                const int trow = idx / q8, c = (idx - trow * q8) * 8;
#else
                const int trow = (int)(((float)idx + 0.5f) * inv_q8), c = (idx - trow * q8) * 8;
#endif
                const int sb = trow >= SB ? 1 : 0, tok = trow - SB * sb;
                dst[u] = idx < total ? trow * ld + c : -1;
                x[u] = make_uint4(0u, 0u, 0u, 0u);
                if (idx < total && tok < t_ntok[sb < NT ? sb : 0]) {
                    const size_t row = (size_t)t_f[sb] * N + gtok(sb, tok);
                    const bf16_t* src = c < Cp ? a.P + row * Cp + c : a.L + row * Ck + (c - Cp);
                    x[u] = *reinterpret_cast<const uint4*>(src);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) *reinterpret_cast<uint4*>(s_xb + dst[u]) = x[u];
        }
    }
    KPFF_STAMP(1);
#ifndef KPFF_SKIP_POOL
    // ---- G: pooled (multi-scale) straight from global memory: one thread per (4x4 cell, group of four channels), the cell's 16 tokens
    //      as 16 independent 8-byte loads, fp32 math (per channel the sums run over the cell's tokens in raster order), one 8-byte store
    //      per token into the tile.  Both sub-tiles share one index space, so at 7x7 tokens and 256 channels every thread has exactly
    //      one item.  (Until the end of round 2 G was staged like P and L and pooled in place in a second pass over LDS; until round 3
    //      a thread took a channel PAIR -- twice the vector-memory instructions for the same bytes: 11.4 k of the kernel's 70 k cycles.)
    const int cv4 = Cv / 4;
    int items[NT], cwv[NT];
    int itot = 0;
#pragma unroll
    for (int sb = 0; sb < NT; ++sb) {
        cwv[sb] = (t_w[sb] + 3) / 4;
        items[sb] = cwv[sb] * ((t_nrows[sb] + 3) / 4) * cv4;
        itot += items[sb];
    }
    struct PoolItem { int sb, c, y0, x0; };
    auto pool_locate = [&](int it) __attribute__((always_inline)) {
        int sb = 0, idx = it;
#pragma unroll
        for (int k = 0; k + 1 < NT; ++k)
            if (idx >= items[k] && sb == k) { idx -= items[k]; sb = k + 1; }
        const int cell = idx / cv4, cy = cell / cwv[sb];
        return PoolItem{sb, (idx - cell * cv4) * 4, cy * 4, (cell - cy * cwv[sb]) * 4};
    };
    auto pool_issue = [&](const PoolItem& pi, uint2 (&u)[16], bool (&ok)[16]) __attribute__((always_inline)) {
        const int nrows = t_nrows[pi.sb], W = t_w[pi.sb];
        const bf16_t* gsrc = a.G + ((size_t)t_f[pi.sb] * N + t_n0[pi.sb]) * Cv;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int y = pi.y0 + (p >> 2), x = pi.x0 + (p & 3);
            ok[p] = y < nrows && x < W;
            u[p] = ok[p] ? *reinterpret_cast<const uint2*>(gsrc + (size_t)(y * a.w + x) * Cv + pi.c) : make_uint2(0u, 0u);
        }
    };
    auto pool_finish = [&](const PoolItem& pi, const uint2 (&u)[16], const bool (&ok)[16]) __attribute__((always_inline)) {
        const int W = t_w[pi.sb];
        bf16_t* gx = s_xb + (size_t)pi.sb * SB * ld + Cp + Ck;
        float s2[4][4], s4[4] = {0.f, 0.f, 0.f, 0.f};
        float n2[4] = {0.f, 0.f, 0.f, 0.f}, n4 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) s2[q][j] = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int q = ((p >> 3) << 1) | ((p >> 1) & 1);
            const float v[4] = {__uint_as_float(u[p].x << 16), __uint_as_float(u[p].x & 0xffff0000u),
                                __uint_as_float(u[p].y << 16), __uint_as_float(u[p].y & 0xffff0000u)};
            const float m = ok[p] ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s2[q][j] += v[j]; s4[j] += v[j]; }
            n2[q] += m; n4 += m;
        }
        const float i4 = 1.0f / n4;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int q = ((p >> 3) << 1) | ((p >> 1) & 1);
            const float i2 = 1.0f / fmaxf(n2[q], 1.f);
            const float v[4] = {__uint_as_float(u[p].x << 16), __uint_as_float(u[p].x & 0xffff0000u),
                                __uint_as_float(u[p].y << 16), __uint_as_float(u[p].y & 0xffff0000u)};
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = (v[j] + s2[q][j] * i2 + s4[j] * i4) * (1.0f / 3.0f);
            const int y = pi.y0 + (p >> 2), x = pi.x0 + (p & 3);
            if (ok[p]) *reinterpret_cast<uint2*>(gx + (size_t)(y * W + x) * ld + pi.c) =
                make_uint2((unsigned)f32_to_bf16(r[0]) | ((unsigned)f32_to_bf16(r[1]) << 16), (unsigned)f32_to_bf16(r[2]) | ((unsigned)f32_to_bf16(r[3]) << 16));
        }
    };
    // (Measured and withdrawn in round 3: requesting the cell's loads here and doing the arithmetic behind the [P ; L] part of the first
    // output tile's gate mixes.  vmcnt counts in order, so the first weight fragment of the mixes waits for the sixteen older pooling
    // loads anyway: 37.6 -> 42.2 us.)
    for (int it = tid; it < itot; it += NTHR) {
        uint2 u[16];
        bool ok[16];
        const PoolItem pi = pool_locate(it);
        pool_issue(pi, u, ok);
        pool_finish(pi, u, ok);
    }
#endif
    KPFF_STAMP(2);
    __syncthreads();
    KPFF_STAMP(3);
    if (a.sv.gms) {                                            // training: keep the pooled feature for the backward
        bf16_t* gms = static_cast<bf16_t*>(a.sv.gms);
#pragma unroll
        for (int sb = 0; sb < NT; ++sb)
            for (int idx = tid; idx < t_ntok[sb] * Cv; idx += NTHR) {
                const int tok = idx / Cv, c = idx - tok * Cv;
                gms[((size_t)t_f[sb] * N + gtok(sb, tok)) * Cv + c] = s_xb[(size_t)(sb * SB + tok) * ld + Cp + Ck + c];
            }
    }

    // ---- fused channel mixes: wave owns output channels 16*(4*NT*chunk + wave) .. +15 for all TMW tokens ------
    for (int ob0 = 16 * OT * w_id; ob0 < Cp; ob0 += OB_STEP) {
        f32x4 gl[OT][MT], gg[OT][MT], lp[OT][MT], gp[OT][MT];
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) gl[o][mt] = gg[o][mt] = lp[o][mt] = gp[o][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16_t* xb = s_xb + (size_t)li * ld + 8 * g;             // B fragment: token 16mt+li, k 8g..8g+7
#ifndef KPFF_SKIP_GEMM
        // fragment-order packs: k-step ks of output tile ot lives at ((ot*KS + ks)*64 + lane)*8.  Every pass runs on a ring primed
        // earlier (kernel entry for the first output tile); behind each pass its ring is primed for the wave's NEXT output tile.
        const int obn = ob0 + OB_STEP;                             // the wave's next output tile (uniform)
        // the two short passes' rings are primed in front of the long one (they are consumed right behind it and would otherwise be
        // 32 more registers held across the epilogue), the long pass's ring for the NEXT output tile right behind it
        KRing<1, OT> ring_l, ring_p;
        kpff_prime<1, OT>(ring_l, wl_g(ob0), nullptr, (size_t)ksL * 512, ksL);
        kpff_prime<1, OT>(ring_p, wg_g(ob0), nullptr, (size_t)ksG * 512, ksG);
        kpff_stream<2, MT, OT, true>(xb, ld, 0, KSa, wa_g(ob0), wa_g(Cp + ob0), (size_t)KSa * 512, gl, gg, ring_g);                       // gates
        if (obn < Cp) kpff_prime<2, OT>(ring_g, wa_g(obn), wa_g(Cp + obn), (size_t)KSa * 512, KSa);
        KPFF_STAMP(ob0 < 16 * OT * (NTHR / 64) ? 4 : 8);
        kpff_stream<1, MT, OT, true>(xb, ld, ksP, ksL, wl_g(ob0), nullptr, (size_t)ksL * 512, lp, lp, ring_l);                             // L Wl^T
        kpff_stream<1, MT, OT, true>(xb, ld, ksP + ksL, ksG, wg_g(ob0), nullptr, (size_t)ksG * 512, gp, gp, ring_p);
#else
        gl[0][0][0] = xb[0]; (void)ksP; (void)ksL; (void)KSa;
#endif
        KPFF_STAMP(ob0 < 16 * OT * (NTHR / 64) ? 5 : 9);
#pragma unroll
        for (int o = 0; o < OT; ++o) {
        const int ob = ob0 + 16 * o;
        if (ob >= Cp) continue;
        // epilogue: this lane holds channels oc..oc+3 of token 16mt+li for every token tile mt
        const int oc = ob + 4 * g;
        const f32x4 bl4 = *reinterpret_cast<const f32x4*>(a.ba + oc), bg4 = *reinterpret_cast<const f32x4*>(a.ba + Cp + oc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int trow = 16 * mt + li, sb = (NT > 1 && trow >= SB) ? 1 : 0, tok = trow - SB * sb;
#ifdef KPFF_SKIP_EPI
            if (tok < t_ntok[sb] && gl[o][mt][0] == 123.f) {
#else
            if (tok < t_ntok[sb]) {
#endif
                const uint2 pu = *reinterpret_cast<const uint2*>(s_xb + (size_t)trow * ld + oc);
                const float pv[4] = {__uint_as_float(pu.x << 16), __uint_as_float(pu.x & 0xffff0000u),
                                     __uint_as_float(pu.y << 16), __uint_as_float(pu.y & 0xffff0000u)};
                float y[4], sl[4], sg[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sl[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (gl[o][mt][r] + bl4[r])));
                    sg[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (gg[o][mt][r] + bg4[r])));
                    y[r] = pv[r] + sl[r] * lp[o][mt][r] + sg[r] * gp[o][mt][r];
                }
                auto pack4 = [](const float (&v)[4]) {
                    return make_uint2((unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16),
                                      (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16));
                };
                const size_t grow = (size_t)t_f[sb] * N + gtok(sb, tok);
                *reinterpret_cast<uint2*>(a.out + grow * Cp + oc) = pack4(y);
                if (a.sv.gates) {
                    bf16_t* sg_ = static_cast<bf16_t*>(a.sv.gates);
                    const float lpv[4] = {lp[o][mt][0], lp[o][mt][1], lp[o][mt][2], lp[o][mt][3]}, gpv[4] = {gp[o][mt][0], gp[o][mt][1], gp[o][mt][2], gp[o][mt][3]};
                    *reinterpret_cast<uint2*>(sg_ + grow * 2 * Cp + oc) = pack4(sl);
                    *reinterpret_cast<uint2*>(sg_ + grow * 2 * Cp + Cp + oc) = pack4(sg);
                    *reinterpret_cast<uint2*>(static_cast<bf16_t*>(a.sv.lp) + grow * Cp + oc) = pack4(lpv);
                    *reinterpret_cast<uint2*>(static_cast<bf16_t*>(a.sv.gp) + grow * Cp + oc) = pack4(gpv);
                }
            }
        }
        }
        KPFF_STAMP(ob0 < 16 * OT * (NTHR / 64) ? 6 : 10);
    }
}


// ---------------------------------------------------------------------------------------------------------
// fp32 arm on 16-bit splits: fp32 I/O, every operand carried as two bf16 terms x = x_h + x_l (x_h = bf16(x), x_l = bf16(x - x_h):
// 16 significant bits, the full fp32 exponent range), every product as the three bf16 MFMAs x_h w_h + x_l w_h + x_h w_l with
// fp32 accumulation -- the dropped x_l w_l and the two truncations leave a relative error of about 2^-16 per product (measured
// against the fp64 oracle: see tests), at 3/16 of the exact arm's MFMA time (v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate).
// One workgroup of 8 waves owns one 64-token tile; LDS holds the tile's two bf16 images ([64][Cin + 8] each: 146 KB at
// Cin = 576), the pooled feature is produced straight from global memory in fp32 (one thread per 4x4 cell and channel pair, 16
// independent loads) and split afterwards, the residual P and the outputs stay fp32 end to end.
struct KpffSplitArgs {
    const float* L; const float* G; const float* P;
    const bf16_t* wpack; size_t lo_off;        // fragment-order packs (kpff_pack_weights_kernel): high terms, low terms lo_off elements on
    const float* ba;
    float* out;
    int Ck, Cv, Cp, h, w, rows_per_tile, tiles_per_frame;
    KpffSave sv;
    int cols_per_tile, col_tiles;
};

__device__ __forceinline__ void kpff_split2(float v, bf16_t& hi, bf16_t& lo)
{
    hi = f32_to_bf16(v);
    lo = f32_to_bf16(v - bf16_to_f32(hi));
}

// `n` k-steps of NS weight streams (each a high and a low pack) against the MT token tiles of the two LDS images; ring of WD
// register sets with static indices as in kpff_stream.
template <int NS, int MT>
__device__ __forceinline__ void kpff_stream_split(const bf16_t* xh, const bf16_t* xl, int ld, int ks0, int n,
                                                  const bf16_t* w0, const bf16_t* w1, size_t lo_off,
                                                  f32x4 (&acc0)[MT], f32x4 (&acc1)[MT])
{
    if (n <= 0) return;
    constexpr int WD = 3;
    bf16x8 bh0[WD], bl0[WD], bh1[WD], bl1[WD];
    auto wload = [&](int slot_ks, bf16x8& h0, bf16x8& l0, bf16x8& h1, bf16x8& l1) __attribute__((always_inline)) {
        const size_t off = (size_t)min(slot_ks, n - 1) * 512;
        h0 = *reinterpret_cast<const bf16x8*>(w0 + off);
        l0 = *reinterpret_cast<const bf16x8*>(w0 + lo_off + off);
        if constexpr (NS == 2) {
            h1 = *reinterpret_cast<const bf16x8*>(w1 + off);
            l1 = *reinterpret_cast<const bf16x8*>(w1 + lo_off + off);
        }
    };
#pragma unroll
    for (int d = 0; d < WD; ++d) wload(d, bh0[d], bl0[d], bh1[d], bl1[d]);
    bf16x8 ah[MT], al[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        ah[mt] = *reinterpret_cast<const bf16x8*>(xh + (size_t)mt * 16 * ld + 32 * ks0);
        al[mt] = *reinterpret_cast<const bf16x8*>(xl + (size_t)mt * 16 * ld + 32 * ks0);
    }
    auto body = [&](int i, auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value % WD;
        const int ksn = ks0 + min(i + 1, n - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc0[mt] = mfma_bf16(bh0[j], ah[mt], acc0[mt]);
            if constexpr (NS == 2) acc1[mt] = mfma_bf16(bh1[j], ah[mt], acc1[mt]);
            acc0[mt] = mfma_bf16(bl0[j], ah[mt], acc0[mt]);
            if constexpr (NS == 2) acc1[mt] = mfma_bf16(bl1[j], ah[mt], acc1[mt]);
            acc0[mt] = mfma_bf16(bh0[j], al[mt], acc0[mt]);
            if constexpr (NS == 2) acc1[mt] = mfma_bf16(bh1[j], al[mt], acc1[mt]);
            ah[mt] = *reinterpret_cast<const bf16x8*>(xh + (size_t)mt * 16 * ld + 32 * ksn);
            al[mt] = *reinterpret_cast<const bf16x8*>(xl + (size_t)mt * 16 * ld + 32 * ksn);
        }
        wload(i + WD, bh0[j], bl0[j], bh1[j], bl1[j]);
        __builtin_amdgcn_sched_barrier(0);
    };
    int i = 0;
    for (; i + WD <= n; i += WD) kpff_static_for<0, WD>([&](auto jc) { body(i + decltype(jc)::value, jc); });
    const int rem = n - i;
    kpff_static_for<0, WD - 1>([&](auto jc) { if (decltype(jc)::value < rem) body(i + decltype(jc)::value, jc); });
}

__global__ __launch_bounds__(512) void kpff_split_kernel(KpffSplitArgs a)
{
    constexpr int NTHR = 512, MT = 4;
    extern __shared__ __attribute__((aligned(16))) bf16_t s_sp[];     // [2][KPFF_TM][Cin + PAD16]: high image, low image
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ck = a.Ck, Cv = a.Cv, Cp = a.Cp, Cin = Cp + Ck + Cv, ld = Cin + KPFF_PAD16;
    bf16_t* s_hi = s_sp;
    bf16_t* s_lo = s_sp + (size_t)KPFF_TM * ld;
    const int N = a.h * a.w;
    const int f = blockIdx.x / a.tiles_per_frame, tf = blockIdx.x - f * a.tiles_per_frame;
    const int rt = tf / a.col_tiles, ct = tf - rt * a.col_tiles;
    const int row0 = rt * a.rows_per_tile, col0 = ct * a.cols_per_tile;
    const int nrows = min(a.rows_per_tile, a.h - row0), W = min(a.cols_per_tile, a.w - col0);
    const int ntok = nrows * W, n0 = row0 * a.w + col0;
    auto gtok = [&](int tok) { const int ty = tok / W; return n0 + ty * a.w + (tok - ty * W); };
    const size_t frow = (size_t)f * N;

    // ---- stage [P ; L]: 4 channels per thread (16-byte loads, 4 in flight), split, two 8-byte LDS stores ------------
    {
        const int cpl = Cp + Ck, q4 = cpl / 4, total = KPFF_TM * q4;
        for (int base = tid; base < total; base += 4 * NTHR) {
            f32x4 x[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTHR;
                const int tok = idx / q4, c = (idx - tok * q4) * 4;
                dst[u] = idx < total ? tok * ld + c : -1;
                x[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < total && tok < ntok) {
                    const size_t row = frow + gtok(tok);
                    x[u] = *reinterpret_cast<const f32x4*>(c < Cp ? a.P + row * Cp + c : a.L + row * Ck + (c - Cp));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) {
                    bf16_t hi[4], lo[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) kpff_split2(x[u][r], hi[r], lo[r]);
                    *reinterpret_cast<uint2*>(s_hi + dst[u]) = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
                    *reinterpret_cast<uint2*>(s_lo + dst[u]) = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
                }
        }
    }
    // ---- multi-scale pooling of G from global memory in fp32: one thread per (4x4 cell, channel pair) -------------------
    {
        const int cw = (W + 3) / 4, chh = (nrows + 3) / 4, cv2 = Cv / 2;
        float* gms = static_cast<float*>(a.sv.gms);
        for (int idx = tid; idx < cw * chh * cv2; idx += NTHR) {
            const int cell = idx / cv2, c = (idx - cell * cv2) * 2;
            const int cy = cell / cw, y0 = cy * 4, x0 = (cell - cy * cw) * 4;
            float2 u[16];
            bool ok[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int y = y0 + (p >> 2), x = x0 + (p & 3);
                ok[p] = y < nrows && x < W;
                u[p] = ok[p] ? *reinterpret_cast<const float2*>(a.G + (frow + n0 + y * a.w + x) * Cv + c) : make_float2(0.f, 0.f);
            }
            float s2[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, s4[2] = {0.f, 0.f};
            int n2[4] = {0, 0, 0, 0}, n4 = 0;
            // (row-major sums inside the cell, as the oracle and the exact arm: the pooled feature is bit-identical to theirs)
#pragma unroll
            for (int p = 0; p < 16; ++p)
                if (ok[p]) {
                    const int q = ((p >> 3) << 1) | ((p >> 1) & 1);
                    s2[q][0] += u[p].x; s2[q][1] += u[p].y; n2[q] += 1; s4[0] += u[p].x; s4[1] += u[p].y; n4 += 1;
                }
            const float m4[2] = {s4[0] / (float)n4, s4[1] / (float)n4};
#pragma unroll
            for (int p = 0; p < 16; ++p)
                if (ok[p]) {
                    const int q = ((p >> 3) << 1) | ((p >> 1) & 1);
                    const float r0 = (u[p].x + s2[q][0] / (float)n2[q] + m4[0]) * (1.0f / 3.0f);
                    const float r1 = (u[p].y + s2[q][1] / (float)n2[q] + m4[1]) * (1.0f / 3.0f);
                    const int y = y0 + (p >> 2), x = x0 + (p & 3);
                    bf16_t h0, l0, h1, l1;
                    kpff_split2(r0, h0, l0);
                    kpff_split2(r1, h1, l1);
                    const int o = (y * W + x) * ld + Cp + Ck + c;
                    *reinterpret_cast<unsigned*>(s_hi + o) = (unsigned)h0 | ((unsigned)h1 << 16);
                    *reinterpret_cast<unsigned*>(s_lo + o) = (unsigned)l0 | ((unsigned)l1 << 16);
                    if (gms) *reinterpret_cast<float2*>(gms + (frow + n0 + y * a.w + x) * Cv + c) = make_float2(r0, r1);
                }
        }
    }
    __syncthreads();

    // ---- fused channel mixes: wave w owns output channels 16*(8*chunk + w) .. +15 for the 64 tokens -------------------
    const int ksP = Cp / 32, ksL = Ck / 32, ksG = Cv / 32, KSa = Cin / 32;
    const bf16_t* wa = a.wpack;
    const bf16_t* wl = wa + (size_t)2 * Cp * Cin;
    const bf16_t* wg = wl + (size_t)Cp * Ck;
    const bf16_t* xh = s_hi + (size_t)li * ld + 8 * g;
    const bf16_t* xl = s_lo + (size_t)li * ld + 8 * g;
    for (int ob = 16 * w_id; ob < Cp; ob += 16 * (NTHR / 64)) {
        f32x4 gl[MT], gg[MT], lp[MT], gp[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) gl[mt] = gg[mt] = lp[mt] = gp[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        kpff_stream_split<2, MT>(xh, xl, ld, 0, KSa, wa + ((size_t)(ob / 16) * KSa * 64 + lane) * 8,
                                 wa + ((size_t)((Cp + ob) / 16) * KSa * 64 + lane) * 8, a.lo_off, gl, gg);
        kpff_stream_split<1, MT>(xh, xl, ld, ksP, ksL, wl + ((size_t)(ob / 16) * ksL * 64 + lane) * 8, nullptr, a.lo_off, lp, lp);
        kpff_stream_split<1, MT>(xh, xl, ld, ksP + ksL, ksG, wg + ((size_t)(ob / 16) * ksG * 64 + lane) * 8, nullptr, a.lo_off, gp, gp);
        // epilogue: this lane holds channels oc..oc+3 of token 16mt+li
        const int oc = ob + 4 * g;
        const f32x4 bl4 = *reinterpret_cast<const f32x4*>(a.ba + oc), bg4 = *reinterpret_cast<const f32x4*>(a.ba + Cp + oc);
        f32x4 pv[MT];
        size_t grow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int tok = 16 * mt + li;
            grow[mt] = frow + gtok(min(tok, ntok - 1));
            pv[mt] = *reinterpret_cast<const f32x4*>(a.P + grow[mt] * Cp + oc);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int tok = 16 * mt + li;
            if (tok < ntok) {
                f32x4 y, sl, sg;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sl[r] = 1.0f / (1.0f + expf(-(gl[mt][r] + bl4[r])));
                    sg[r] = 1.0f / (1.0f + expf(-(gg[mt][r] + bg4[r])));
                    y[r] = pv[mt][r] + sl[r] * lp[mt][r] + sg[r] * gp[mt][r];
                }
                *reinterpret_cast<f32x4*>(a.out + grow[mt] * Cp + oc) = y;
                if (a.sv.gates) {
                    float* sgt = static_cast<float*>(a.sv.gates);
                    *reinterpret_cast<f32x4*>(sgt + grow[mt] * 2 * Cp + oc) = sl;
                    *reinterpret_cast<f32x4*>(sgt + grow[mt] * 2 * Cp + Cp + oc) = sg;
                    *reinterpret_cast<f32x4*>(static_cast<float*>(a.sv.lp) + grow[mt] * Cp + oc) = lp[mt];
                    *reinterpret_cast<f32x4*>(static_cast<float*>(a.sv.gp) + grow[mt] * Cp + oc) = gp[mt];
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Backward (row a7) pieces that are not plain GEMMs.  With g = (g_l | g_g), Lp = L Wl^T, Gp = Gms Wg^T:
//   pre :  dz = (dF * Lp * g_l (1-g_l) | dF * Gp * g_g (1-g_g)),  dLp = dF * g_l,  dGp = dF * g_g     (elementwise)
//   ...    dX = dz Wa,  dL += dLp Wl,  dGms += dGp Wg,  dWa = dz^T [P;L;Gms], dWl = dLp^T L, dWg = dGp^T Gms   (library GEMMs)
//   post:  dP = dF + dX_P,  dL = dX_L + dLp Wl,  dG = pool(dX_G + dGp Wg)   (the multi-scale pooling is symmetric)
template <int IO>
__global__ void kpff_bwd_pre_kernel(const void* dF, const void* gates, const void* lp, const void* gp,
                                    void* dz, void* dlp, void* dgp, size_t M, int Cp)
{
    const size_t n = M * Cp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / Cp;
        const int o = (int)(i - row * Cp);
        const float d = load1<IO>(dF, i), gl = load1<IO>(gates, row * 2 * Cp + o), gg = load1<IO>(gates, row * 2 * Cp + Cp + o);
        store1<IO>(dz, row * 2 * Cp + o, d * load1<IO>(lp, i) * gl * (1.f - gl));
        store1<IO>(dz, row * 2 * Cp + Cp + o, d * load1<IO>(gp, i) * gg * (1.f - gg));
        store1<IO>(dlp, i, d * gl);
        store1<IO>(dgp, i, d * gg);
    }
}

// one workgroup per frame: dP, dL (elementwise sums) and dG = pool(dGms) over the h x w grid
template <int IO>
__global__ __launch_bounds__(256) void kpff_bwd_post_kernel(const void* dF, const void* dX, const void* dL_add, const void* dG_add,
                                                            void* dP, void* dL, void* dG, int Ck, int Cv, int Cp, int h, int w)
{
    const int f = blockIdx.x, N = h * w, Cin = Cp + Ck + Cv, tid = threadIdx.x;
    for (int idx = tid; idx < N * Cp; idx += 256) {
        const int tok = idx / Cp, c = idx - tok * Cp;
        const size_t row = (size_t)f * N + tok;
        store1<IO>(dP, row * Cp + c, load1<IO>(dF, row * Cp + c) + load1<IO>(dX, row * Cin + c));
    }
    for (int idx = tid; idx < N * Ck; idx += 256) {
        const int tok = idx / Ck, c = idx - tok * Ck;
        const size_t row = (size_t)f * N + tok;
        store1<IO>(dL, row * Ck + c, load1<IO>(dX, row * Cin + Cp + c) + load1<IO>(dL_add, row * Ck + c));
    }
    const int cw = (w + 3) / 4, chh = (h + 3) / 4;
    for (int idx = tid; idx < cw * chh * Cv; idx += 256) {
        const int c = idx % Cv, cell = idx / Cv;
        const int y0 = (cell / cw) * 4, x0 = (cell % cw) * 4;
        const int y1 = min(y0 + 4, h), x1 = min(x0 + 4, w);
        float s4 = 0.f, s2[4] = {0.f, 0.f, 0.f, 0.f}, v[16];
        int n2[4] = {0, 0, 0, 0};
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const size_t row = (size_t)f * N + y * w + x;
                const float val = load1<IO>(dX, row * Cin + Cp + Ck + c) + load1<IO>(dG_add, row * Cv + c);
                const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                v[(y - y0) * 4 + (x - x0)] = val;
                s2[q] += val; n2[q] += 1; s4 += val;
            }
        const float m4 = s4 / (float)((y1 - y0) * (x1 - x0));
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                const size_t row = (size_t)f * N + y * w + x;
                store1<IO>(dG, row * Cv + c, (v[(y - y0) * 4 + (x - x0)] + s2[q] / (float)n2[q] + m4) * (1.0f / 3.0f));
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Row n4: the key / query / value 1x1 projections of the stride-16 feature as ONE pass over the tokens (inference).
// As three library GEMMs (M = B*T*N tokens, K = Cp, N = 64 / 64 / 256) they cost 35 us of a 1.35 ms forward for 4.9 GFLOP;
// the shape is KPFF's own -- a token tile staged once in LDS as the MFMA B operand, the weights streamed from L2 in fragment
// order as the A operand (kpff_stream) -- so the same machinery serves: a workgroup stages 128 token rows, its 8 waves walk the
// 16-channel output tiles of all three projections, and the epilogue adds the bias and writes each tile into the tensor it
// belongs to (contiguous [tokens, width] rows: exactly what gdkvm_scan_prep and gdkvm_kpff_fwd read).
#ifndef PROJ_TM_SWITCH
#define PROJ_TM_SWITCH (128 * 512)
#endif
struct ProjArgs {
    const bf16_t* x; const bf16_t* wpack; const float* bias;
    bf16_t* out[3];
    int width[3];
    int M, K, ntile_out;
};

template <int TM>
__global__ __launch_bounds__(512) void proj_rows_kernel(ProjArgs a)
{
    constexpr int MT = TM / 16;
    extern __shared__ __attribute__((aligned(16))) bf16_t s_px[];       // [TM][K + PAD16]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, ld = K + KPFF_PAD16, KS = K / 32, q8 = K / 8;
    const size_t row0 = (size_t)blockIdx.x * TM;
    for (int base = tid; base < TM * q8; base += 4 * 512) {            // stage the token rows (zero beyond M), 4 loads in flight
        uint4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 512, r = idx / q8, c = (idx - r * q8) * 8;
            x[u] = (idx < TM * q8 && row0 + r < (size_t)a.M) ? *reinterpret_cast<const uint4*>(a.x + (row0 + r) * K + c) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 512, r = idx / q8, c = (idx - r * q8) * 8;
            if (idx < TM * q8) *reinterpret_cast<uint4*>(s_px + (size_t)r * ld + c) = x[u];
        }
    }
    __syncthreads();
    const bf16_t* xb = s_px + (size_t)li * ld + 8 * g;
    for (int ot = w_id; ot < a.ntile_out; ot += 8) {
        f32x4 acc[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        kpff_stream<1, MT, 1>(xb, ld, 0, KS, a.wpack + ((size_t)ot * KS * 64 + lane) * 8, nullptr, 0, acc, acc);
        const int oc = 16 * ot + 4 * g;                               // this lane: channels oc .. oc+3 of token 16mt + li
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + oc);
        const int seg = oc < a.width[0] ? 0 : (oc < a.width[0] + a.width[1] ? 1 : 2);
        const int cbase = oc - (seg == 0 ? 0 : (seg == 1 ? a.width[0] : a.width[0] + a.width[1]));
        bf16_t* dst = a.out[seg];
        const int wd = a.width[seg];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const size_t row = row0 + 16 * mt + li;
            if (row < (size_t)a.M) {
                const uint2 o = make_uint2((unsigned)f32_to_bf16(acc[0][mt][0] + b4[0]) | ((unsigned)f32_to_bf16(acc[0][mt][1] + b4[1]) << 16),
                                           (unsigned)f32_to_bf16(acc[0][mt][2] + b4[2]) | ((unsigned)f32_to_bf16(acc[0][mt][3] + b4[3]) << 16));
                *reinterpret_cast<uint2*>(dst + row * wd + cbase) = o;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Row n4, one step further: EVERYTHING the memory path derives from the stride-16 pixel feature in ONE launch (inference) -- the
// key / query / value projections (as proj_rows_kernel), the write-gate logit per token and the decay logit per frame (what
// gdkvm_gate_logits computed in a launch of its own, from a second read of the same rows), and the inverse L2 norms of the key
// and query rows (what gdkvm_scan_prep's first phase computed from a read of q it otherwise has no use for).
//   workgroups [0, nproj)      a 64- or 128-token tile: rows staged once in LDS, gate logits from the staged rows (fp32 weights,
//                              VALU), the projections' output tiles on the MFMA; the K and Q tiles leave per-token sums of squares
//                              of the ROUNDED outputs (the values prep would read back) in LDS, one slot per tile, added up in a
//                              fixed order after a barrier: deterministic, no atomics
//   workgroups [nproj, ...)    the decay logit: one wave per frame walks the frame's rows (a frame's tokens straddle tiles, and a
//                              sum over tiles would need atomics or a second launch); they run beside the tile workgroups
struct ProjGateArgs {
    ProjArgs p;
    const float* w_gate; const float* b_gate; const float* w_decay; const float* b_decay;
    float* beta; float* alpha; float* norms;
    int frames, N, Hh, Dk, nproj;
};

template <int TM>
__global__ __launch_bounds__(512) void proj_gates_kernel(ProjGateArgs ga)
{
    const ProjArgs& a = ga.p;
    constexpr int MT = TM / 16;
    extern __shared__ __attribute__((aligned(16))) bf16_t s_px[];       // [TM][K + PAD16] | per-tile sums of squares [ntile_norm][TM] fp32
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, ld = K + KPFF_PAD16, KS = K / 32, q8 = K / 8, Hh = ga.Hh;
    if ((int)blockIdx.x >= ga.nproj) {
        // ---- decay logits: alpha[f, h] = <mean_n x[f, n, :], w_decay[h, :]> + b_decay[h], one wave per frame
        const int G = q8, tpw = 64 / G, sub = lane / G, cg = lane % G;   // lanes per token row (K / 8 <= 64, a power of two)
        for (int f = ((int)blockIdx.x - ga.nproj) * 8 + w_id; f < ga.frames; f += ((int)gridDim.x - ga.nproj) * 8) {
            const uint4* pv = reinterpret_cast<const uint4*>(a.x) + (size_t)f * ga.N * G;
            for (int h = 0; h < Hh; ++h) {
                float wd[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wd[j] = ga.w_decay[(size_t)h * K + cg * 8 + j];
                float dsum = 0.f;
                constexpr int UNR = 8;
                for (int n0 = 0; n0 < ga.N; n0 += tpw * UNR) {
                    uint4 x[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) x[u] = pv[(size_t)min(n0 + tpw * u + sub, ga.N - 1) * G + cg];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const unsigned xw[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
                        float dd = 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            dd = fmaf(__uint_as_float(xw[j] << 16), wd[2 * j], dd);
                            dd = fmaf(__uint_as_float(xw[j] & 0xffff0000u), wd[2 * j + 1], dd);
                        }
                        dsum += (n0 + tpw * u + sub < ga.N) ? dd : 0.f;
                    }
                }
                for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o);
                if (lane == 0) ga.alpha[(size_t)f * Hh + h] = dsum / (float)ga.N + ga.b_decay[h];
            }
        }
        return;
    }
    float* s_part = reinterpret_cast<float*>(s_px + (size_t)TM * ld);
    const size_t row0 = (size_t)blockIdx.x * TM;
    KPFF_STAMP(0);
    for (int base = tid; base < TM * q8; base += 4 * 512) {            // stage the token rows (zero beyond M), 4 loads in flight
        uint4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 512, r = idx / q8, c = (idx - r * q8) * 8;
            x[u] = (idx < TM * q8 && row0 + r < (size_t)a.M) ? *reinterpret_cast<const uint4*>(a.x + (row0 + r) * K + c) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 512, r = idx / q8, c = (idx - r * q8) * 8;
            if (idx < TM * q8) *reinterpret_cast<uint4*>(s_px + (size_t)r * ld + c) = x[u];
        }
    }
    KPFF_STAMP(1);
    __syncthreads();
    KPFF_STAMP(2);
    {   // ---- write-gate logits from the staged rows: 512 / TM threads per token, 16-byte pieces interleaved over them
        constexpr int TPT = 512 / TM;
        const int tok = tid / TPT, part = tid % TPT;
        for (int h = 0; h < Hh; ++h) {
            float d = 0.f;
            for (int pc = part; pc < q8; pc += TPT) {
                const uint4 xv = *reinterpret_cast<const uint4*>(s_px + (size_t)tok * ld + pc * 8);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(ga.w_gate + (size_t)h * K + pc * 8);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(ga.w_gate + (size_t)h * K + pc * 8 + 4);
                d = fmaf(__uint_as_float(xv.x << 16), w0[0], d); d = fmaf(__uint_as_float(xv.x & 0xffff0000u), w0[1], d);
                d = fmaf(__uint_as_float(xv.y << 16), w0[2], d); d = fmaf(__uint_as_float(xv.y & 0xffff0000u), w0[3], d);
                d = fmaf(__uint_as_float(xv.z << 16), w1[0], d); d = fmaf(__uint_as_float(xv.z & 0xffff0000u), w1[1], d);
                d = fmaf(__uint_as_float(xv.w << 16), w1[2], d); d = fmaf(__uint_as_float(xv.w & 0xffff0000u), w1[3], d);
            }
#pragma unroll
            for (int o = TPT >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o);
            if (part == 0 && row0 + tok < (size_t)a.M) ga.beta[(row0 + tok) * Hh + h] = d + ga.b_gate[h];
        }
    }
    KPFF_STAMP(3);
    const bf16_t* xb = s_px + (size_t)li * ld + 8 * g;
    const int ntile_norm = (a.width[0] + a.width[1]) / 16;               // the key and query tiles
    for (int ot = w_id; ot < a.ntile_out; ot += 8) {
        f32x4 acc[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        kpff_stream<1, MT, 1>(xb, ld, 0, KS, a.wpack + ((size_t)ot * KS * 64 + lane) * 8, nullptr, 0, acc, acc);
        const int oc = 16 * ot + 4 * g;                               // this lane: channels oc .. oc+3 of token 16mt + li
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + oc);
        const int seg = oc < a.width[0] ? 0 : (oc < a.width[0] + a.width[1] ? 1 : 2);
        const int cbase = oc - (seg == 0 ? 0 : (seg == 1 ? a.width[0] : a.width[0] + a.width[1]));
        bf16_t* dst = a.out[seg];
        const int wd = a.width[seg];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const size_t row = row0 + 16 * mt + li;
            const uint2 o = make_uint2((unsigned)f32_to_bf16(acc[0][mt][0] + b4[0]) | ((unsigned)f32_to_bf16(acc[0][mt][1] + b4[1]) << 16),
                                       (unsigned)f32_to_bf16(acc[0][mt][2] + b4[2]) | ((unsigned)f32_to_bf16(acc[0][mt][3] + b4[3]) << 16));
            if (row < (size_t)a.M) *reinterpret_cast<uint2*>(dst + row * wd + cbase) = o;
            if (ot < ntile_norm) {                                    // sums of squares of the values as stored (bf16)
                const float v0 = __uint_as_float(o.x << 16), v1 = __uint_as_float(o.x & 0xffff0000u);
                const float v2 = __uint_as_float(o.y << 16), v3 = __uint_as_float(o.y & 0xffff0000u);
                float ss = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
                ss += __shfl_xor(ss, 16);
                ss += __shfl_xor(ss, 32);
                if (g == 0) s_part[ot * TM + 16 * mt + li] = ss;
            }
        }
    }
    KPFF_STAMP(4);
    __syncthreads();
    KPFF_STAMP(5);
    // inverse norms per (token, head): the Dk / 16 tile sums of a head in tile order; norms[row][head][0 = key, 1 = query]
    const int tph = ga.Dk / 16;
    for (int idx = tid; idx < TM * 2 * Hh; idx += 512) {
        const int tok = idx % TM, hk = idx / TM, which = hk / Hh, h = hk % Hh;
        const int t0 = (which ? a.width[0] / 16 : 0) + h * tph;
        float ss = 0.f;
        for (int t = 0; t < tph; ++t) ss += s_part[(t0 + t) * TM + tok];
        if (row0 + tok < (size_t)a.M) ga.norms[((row0 + tok) * Hh + h) * 2 + which] = 1.0f / sqrtf(ss + GDKVM_EPS_NORM);
    }
    KPFF_STAMP(6);
}

}  // namespace

// > 64 KiB of dynamic LDS needs an opt-in per kernel and device: done once (to the CU's 160 KiB), remembered in a lock-free mask per
// kernel (`slot`), as in gdr_scan.hip -- host threads may drive several devices concurrently.
static int kpff_lds_optin(const void* fn, int slot)
{
    static std::atomic<unsigned long long> done_mask[8];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "kpff_fwd: hipGetDevice");
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask[slot].load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "kpff_fwd: LDS attribute: %s", hipGetErrorString(e));
        done_mask[slot].fetch_or(bit, std::memory_order_relaxed);
    }
    return GDKVM_OK;
}

extern "C" size_t gdkvm_kpff_workspace_bytes(int Ck, int Cv, int Cp, int io_dtype)
{
    if ((io_dtype != GDKVM_BF16 && io_dtype != GDKVM_F32) || Ck <= 0 || Cv <= 0 || Cp <= 0) return 16;
    const size_t pack = ((size_t)2 * Cp * (Cp + Ck + Cv) + (size_t)Cp * Ck + (size_t)Cp * Cv) * sizeof(bf16_t);
    if (io_dtype == GDKVM_F32) return (Ck % 32 || Cv % 32 || Cp % 32) ? 16 : 2 * pack + 16;      // split arm: high and low packs
    return pack + 16;
}

static thread_local int g_kpff_skip_pack = 0;            // set by gdkvm_kpff_fwd_packed around its call

extern "C" int gdkvm_kpff_fwd_train(const void* local, const void* global, const void* pixel,
                                    const float* wa, const float* ba, const float* wl, const float* wg, void* out,
                                    void* save_gates, void* save_lp, void* save_gp, void* save_gms,
                                    void* workspace, size_t workspace_bytes,
                                    int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream)
{
    const bool any = save_gates || save_lp || save_gp || save_gms, all = save_gates && save_lp && save_gp && save_gms;
    if (any && !all) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: the four save buffers go together");
    const KpffSave sv{save_gates, save_lp, save_gp, save_gms};
    if (BT < 0 || Ck <= 0 || Cv <= 0 || Cp <= 0 || h <= 0 || w <= 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: bad shape BT=%d Ck=%d Cv=%d Cp=%d h=%d w=%d", BT, Ck, Cv, Cp, h, w);
    if (Ck % 16 || Cv % 16 || Cp % 16) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: channels must be multiples of 16");
    if ((long)h * w > GDKVM_MAX_N) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: grid %dx%d exceeds %d tokens", h, w, GDKVM_MAX_N);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "kpff_fwd: io_dtype=%d", io_dtype);
    if (BT == 0) return GDKVM_OK;
    const void* ptrs[] = {local, global, pixel, wa, ba, wl, wg, out};
    for (const void* p : ptrs) {
        if (!p) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: null pointer");
        if (!gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: pointer %p is not 16-byte aligned", p);
    }
    // tile = whole frame if it fits 64 tokens, else the largest multiple of 4 grid rows that does; a grid wider than 16 columns is
    // cut into 4-row x 16-column tiles (every 2x2 / 4x4 pooling cell stays inside one tile: both cuts are multiples of 4)
    int rows, cols = w;
    if (h * w <= KPFF_TM) rows = h;
    else if (w <= 16) rows = (KPFF_TM / w) & ~3;
    else { rows = 4; cols = 16; }
    const int col_tiles = (w + cols - 1) / cols;
    const int tiles = ((h + rows - 1) / rows) * col_tiles;
    const int Cin = Cp + Ck + Cv;
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(BT * tiles));

    // bf16 I/O with 32-aligned channel counts: bf16 MFMA arm (weights re-packed to bf16 in the workspace each call)
    if (io_dtype == GDKVM_BF16 && Ck % 32 == 0 && Cv % 32 == 0 && Cp % 32 == 0) {
        const size_t need = gdkvm_kpff_workspace_bytes(Ck, Cv, Cp, io_dtype) - 16;
        if (!workspace || !gdkvm_aligned16(workspace)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: workspace null or misaligned");
        if (workspace_bytes < need) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "kpff_fwd: workspace %zu < %zu bytes", workspace_bytes, need + 16);
        const size_t lds1 = (size_t)KPFF_TM * (Cin + KPFF_PAD16) * sizeof(bf16_t);
        if (lds1 > 160 * 1024) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: Cp+Ck+Cv=%d exceeds the LDS tile", Cin);
        const int total_tiles = BT * tiles;
#ifdef KPFF_FORCE_NT1                                           // ablation (tools/abl_kpff.py): one 64-token tile per four-wave workgroup, two workgroups per CU
        const bool pair = false;
#else
        // Up to one tile per CU (the per-frame step mode runs KPFF on ONE frame per clip; a group of clips of a forward split over streams): the
        // kernel's time is one workgroup's latency, and a four-wave workgroup with ONE tile is done sooner than an eight-wave one with two --
        // 24.3 against 25.7 us at 16 frames, 26.8 against 29.2 at 256, 34.1 against 30.2 at 320 (round 6; same bits).
        // GDKVM_KPFF_SINGLE_BELOW=n: single-tile workgroups for fewer than n tiles (0 = never).
        static const int single_below = [] {
            if (const char* e = getenv("GDKVM_KPFF_SINGLE_BELOW")) return atoi(e);
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n + 1;
            return 257;
        }();
        const bool pair = 2 * lds1 <= 160 * 1024 && total_tiles >= 2 && total_tiles >= single_below;     // two 64-token tiles per 8-wave workgroup
#endif
        bf16_t* wab = static_cast<bf16_t*>(workspace);
        const size_t na = (size_t)2 * Cp * Cin, nl = (size_t)Cp * Ck;
        if (!g_kpff_skip_pack) {
            hipLaunchKernelGGL(kpff_pack_weights_kernel, dim3(256), dim3(256), 0, st, wa, wl, wg, wab, static_cast<bf16_t*>(nullptr), Cp, Ck, Cv);
            GDKVM_LAUNCH_CHECK("kpff_pack_weights_kernel");
        }
        KpffBf16Args b{static_cast<const bf16_t*>(local), static_cast<const bf16_t*>(global), static_cast<const bf16_t*>(pixel),
                       wab, ba, wab + na, wab + na + nl, static_cast<bf16_t*>(out), Ck, Cv, Cp, h, w, rows, tiles, sv, cols, col_tiles};
        // two sub-tiles of at most 56 tokens each (two 7x7 frames): packed at 56 rows apiece, seven token tiles instead of eight
#ifdef KPFF_ABL_NOPACK56                                        // ablation: eight token tiles as in rounds 1-2
        const bool packed56 = false;
#else
        const bool packed56 = pair && rows * cols <= 56;
#endif
        const size_t lds = packed56 ? (size_t)112 * (Cin + KPFF_PAD16) * sizeof(bf16_t) : (pair ? 2 * lds1 : lds1);
        const void* fn = packed56 ? reinterpret_cast<const void*>(kpff_bf16_kernel<2, KPFF_OT, 56>)
                       : pair ? reinterpret_cast<const void*>(kpff_bf16_kernel<2, KPFF_OT>) : reinterpret_cast<const void*>(kpff_bf16_kernel<1, 1>);
        if (lds > 64 * 1024)
            if (int rc = kpff_lds_optin(fn, packed56 ? 7 : (pair ? 0 : 1))) return rc;
        if (packed56) hipLaunchKernelGGL((kpff_bf16_kernel<2, KPFF_OT, 56>), dim3((unsigned)((total_tiles + 1) / 2)), dim3(512 / KPFF_OT), lds, st, b, total_tiles);
        else if (pair) hipLaunchKernelGGL((kpff_bf16_kernel<2, KPFF_OT>), dim3((unsigned)((total_tiles + 1) / 2)), dim3(512 / KPFF_OT), lds, st, b, total_tiles);
        else hipLaunchKernelGGL((kpff_bf16_kernel<1, 1>), dim3((unsigned)total_tiles), dim3(256), lds, st, b, total_tiles);
        GDKVM_LAUNCH_CHECK("kpff_bf16_kernel");
        return GDKVM_OK;
    }

    // fp32 I/O with 32-aligned channel counts and a workspace for the two packs: the arm on bf16 splits (without a workspace the
    // exact fp32-MFMA arm below runs: that is also the way to ask for it)
    const size_t lds_split = (size_t)2 * KPFF_TM * (Cin + KPFF_PAD16) * sizeof(bf16_t);
    if (io_dtype == GDKVM_F32 && Ck % 32 == 0 && Cv % 32 == 0 && Cp % 32 == 0 && workspace && lds_split <= 160 * 1024) {
        const size_t need = gdkvm_kpff_workspace_bytes(Ck, Cv, Cp, io_dtype) - 16;
        if (!gdkvm_aligned16(workspace)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: workspace misaligned");
        if (workspace_bytes < need) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "kpff_fwd: workspace %zu < %zu bytes", workspace_bytes, need + 16);
        bf16_t* wpk = static_cast<bf16_t*>(workspace);
        const size_t lo_off = need / (2 * sizeof(bf16_t));
        if (!g_kpff_skip_pack) {
            hipLaunchKernelGGL(kpff_pack_weights_kernel, dim3(256), dim3(256), 0, st, wa, wl, wg, wpk, wpk + lo_off, Cp, Ck, Cv);
            GDKVM_LAUNCH_CHECK("kpff_pack_weights_kernel");
        }
        if (int rc = kpff_lds_optin(reinterpret_cast<const void*>(kpff_split_kernel), 2)) return rc;
        KpffSplitArgs b{static_cast<const float*>(local), static_cast<const float*>(global), static_cast<const float*>(pixel), wpk, lo_off, ba,
                        static_cast<float*>(out), Ck, Cv, Cp, h, w, rows, tiles, sv, cols, col_tiles};
        hipLaunchKernelGGL(kpff_split_kernel, grid, dim3(512), lds_split, st, b);
        GDKVM_LAUNCH_CHECK("kpff_split_kernel");
        return GDKVM_OK;
    }

    const size_t lds = (size_t)KPFF_TM * (Cin + KPFF_PAD) * sizeof(float);
    if (lds > 160 * 1024) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: Cp+Ck+Cv=%d exceeds the LDS tile", Cin);
    KpffArgs a{local, global, pixel, wa, ba, wl, wg, out, Ck, Cv, Cp, h, w, rows, tiles, sv, cols, col_tiles};
    const void* fn = io_dtype == GDKVM_F32 ? reinterpret_cast<const void*>(kpff_kernel<GDKVM_F32>)
                                           : reinterpret_cast<const void*>(kpff_kernel<GDKVM_BF16>);
    if (lds > 64 * 1024)
        if (int rc = kpff_lds_optin(fn, io_dtype == GDKVM_F32 ? 3 : 4)) return rc;
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((kpff_kernel<GDKVM_F32>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((kpff_kernel<GDKVM_BF16>), grid, dim3(256), lds, st, a);
    GDKVM_LAUNCH_CHECK("kpff_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_kpff_fwd(const void* local, const void* global, const void* pixel,
                              const float* wa, const float* ba, const float* wl, const float* wg, void* out,
                              void* workspace, size_t workspace_bytes,
                              int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream)
{
    return gdkvm_kpff_fwd_train(local, global, pixel, wa, ba, wl, wg, out, nullptr, nullptr, nullptr, nullptr,
                                workspace, workspace_bytes, BT, Ck, Cv, Cp, h, w, io_dtype, stream);
}

extern "C" int gdkvm_kpff_bwd_pre(const void* d_out, const void* gates, const void* lp, const void* gp,
                                  void* d_z, void* d_lp, void* d_gp, int BT, int N, int Cp, int io_dtype, void* stream)
{
    if (BT < 0 || N <= 0 || Cp <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_bwd_pre: bad shape");
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "kpff_bwd_pre: io_dtype=%d", io_dtype);
    if (BT == 0) return GDKVM_OK;
    const void* ptrs[] = {d_out, gates, lp, gp, d_z, d_lp, d_gp};
    for (const void* p : ptrs) if (!p || !gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_bwd_pre: null or misaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t M = (size_t)BT * N;
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((kpff_bwd_pre_kernel<GDKVM_F32>), dim3(2048), dim3(256), 0, st, d_out, gates, lp, gp, d_z, d_lp, d_gp, M, Cp);
    else hipLaunchKernelGGL((kpff_bwd_pre_kernel<GDKVM_BF16>), dim3(2048), dim3(256), 0, st, d_out, gates, lp, gp, d_z, d_lp, d_gp, M, Cp);
    GDKVM_LAUNCH_CHECK("kpff_bwd_pre_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_kpff_bwd_post(const void* d_out, const void* d_x, const void* d_l_add, const void* d_g_add,
                                   void* d_pixel, void* d_local, void* d_global,
                                   int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream)
{
    if (BT < 0 || Ck <= 0 || Cv <= 0 || Cp <= 0 || h <= 0 || w <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_bwd_post: bad shape");
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "kpff_bwd_post: io_dtype=%d", io_dtype);
    if (BT == 0) return GDKVM_OK;
    const void* ptrs[] = {d_out, d_x, d_l_add, d_g_add, d_pixel, d_local, d_global};
    for (const void* p : ptrs) if (!p || !gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_bwd_post: null or misaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((kpff_bwd_post_kernel<GDKVM_F32>), dim3(BT), dim3(256), 0, st, d_out, d_x, d_l_add, d_g_add, d_pixel, d_local, d_global, Ck, Cv, Cp, h, w);
    else hipLaunchKernelGGL((kpff_bwd_post_kernel<GDKVM_BF16>), dim3(BT), dim3(256), 0, st, d_out, d_x, d_l_add, d_g_add, d_pixel, d_local, d_global, Ck, Cv, Cp, h, w);
    GDKVM_LAUNCH_CHECK("kpff_bwd_post_kernel");
    return GDKVM_OK;
}

// Inference with constant weights: the caller keeps the workspace of an earlier gdkvm_kpff_fwd call with the SAME weight
// tensors alive and skips the re-pack (about 5 us per call at Cp = Cv = 256).  wa / wl / wg are still required for the
// fp32 arm and for argument checking.
extern "C" int gdkvm_kpff_fwd_packed(const void* local, const void* global, const void* pixel,
                                     const float* wa, const float* ba, const float* wl, const float* wg, void* out,
                                     void* packed_workspace, size_t workspace_bytes,
                                     int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream)
{
    g_kpff_skip_pack = 1;
    const int rc = gdkvm_kpff_fwd_train(local, global, pixel, wa, ba, wl, wg, out, nullptr, nullptr, nullptr, nullptr,
                                        packed_workspace, workspace_bytes, BT, Ck, Cv, Cp, h, w, io_dtype, stream);
    g_kpff_skip_pack = 0;
    return rc;
}

extern "C" int gdkvm_proj_rows(const void* x, const void* wpack, const float* bias, void* out0, void* out1, void* out2,
                               long long rows, int K, int w0, int w1, int w2, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "proj_rows: only bf16 is implemented");
    if (rows < 0 || rows > 0x7fffffffLL || K <= 0 || K % 32 || K > 512 || w0 <= 0 || w1 < 0 || w2 < 0 || w0 % 16 || w1 % 16 || w2 % 16)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "proj_rows: rows=%lld K=%d widths %d %d %d (K a multiple of 32 up to 512, widths multiples of 16)",
                          rows, K, w0, w1, w2);
    if (rows == 0) return GDKVM_OK;
    if (!x || !wpack || !bias || !out0 || (w1 && !out1) || (w2 && !out2)) return gdkvm_fail(GDKVM_ERR_ARG, "proj_rows: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(wpack) || !gdkvm_aligned16(bias) || !gdkvm_aligned16(out0) || (out1 && !gdkvm_aligned16(out1))
        || (out2 && !gdkvm_aligned16(out2)))
        return gdkvm_fail(GDKVM_ERR_ARG, "proj_rows: pointers must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    ProjArgs a;
    a.x = static_cast<const bf16_t*>(x); a.wpack = static_cast<const bf16_t*>(wpack); a.bias = bias;
    a.out[0] = static_cast<bf16_t*>(out0); a.out[1] = static_cast<bf16_t*>(out1); a.out[2] = static_cast<bf16_t*>(out2);
    a.width[0] = w0; a.width[1] = w1; a.width[2] = w2;
    a.M = (int)rows; a.K = K; a.ntile_out = (w0 + w1 + w2) / 16;
    // token rows per workgroup: 128 when that still gives every CU at least two workgroups, else 64 (cfg2: 25088 rows -> 392 x 64)
    const int TM = rows >= PROJ_TM_SWITCH ? 128 : 64;
    const size_t lds = (size_t)TM * (K + KPFF_PAD16) * sizeof(bf16_t);
    {   // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device; lock-free cache as in gdr_scan.hip
        static std::atomic<unsigned long long> done_mask{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "proj_rows: hipGetDevice");
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
            const int cap = (int)((size_t)128 * (512 + KPFF_PAD16) * sizeof(bf16_t));
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_rows_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_rows_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "proj_rows: %s", hipGetErrorString(e));
            done_mask.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    const unsigned grid = (unsigned)((rows + TM - 1) / TM);
    if (TM == 128) hipLaunchKernelGGL(proj_rows_kernel<128>, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(proj_rows_kernel<64>, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    GDKVM_LAUNCH_CHECK("proj_rows_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_proj_gates(const void* x, const void* wpack, const float* bias, void* out_k, void* out_q, void* out_v,
                                const float* w_gate, const float* b_gate, const float* w_decay, const float* b_decay,
                                float* beta, float* alpha, float* norms,
                                int frames, int N, int K, int Hh, int Dk, int Dv, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "proj_gates: only bf16 is implemented");
    const int q8 = K > 0 ? K / 8 : 0;
    if (frames < 0 || N <= 0 || K <= 0 || K % 32 || K > 512 || (q8 & (q8 - 1)) || Hh <= 0 || Dk <= 0 || Dk % 16 || Dv <= 0 || Dv % 16 ||
        (long long)frames * N > 0x7fffffffLL)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "proj_gates: frames=%d N=%d K=%d Hh=%d Dk=%d Dv=%d (K a power of two times 8 up to 512, Dk and Dv multiples of 16)",
                          frames, N, K, Hh, Dk, Dv);
    if (frames == 0) return GDKVM_OK;
    for (const void* p : {x, wpack, (const void*)bias, (const void*)out_k, (const void*)out_q, (const void*)out_v, (const void*)w_gate,
                          (const void*)b_gate, (const void*)w_decay, (const void*)b_decay, (const void*)beta, (const void*)alpha, (const void*)norms})
        if (!p) return gdkvm_fail(GDKVM_ERR_ARG, "proj_gates: null pointer");
    for (const void* p : {x, wpack, (const void*)bias, (const void*)out_k, (const void*)out_q, (const void*)out_v, (const void*)w_gate, (const void*)w_decay})
        if (!gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "proj_gates: pointers must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    const long long rows = (long long)frames * N;
    ProjGateArgs ga;
    ProjArgs& a = ga.p;
    a.x = static_cast<const bf16_t*>(x); a.wpack = static_cast<const bf16_t*>(wpack); a.bias = bias;
    a.out[0] = static_cast<bf16_t*>(out_k); a.out[1] = static_cast<bf16_t*>(out_q); a.out[2] = static_cast<bf16_t*>(out_v);
    a.width[0] = Hh * Dk; a.width[1] = Hh * Dk; a.width[2] = Hh * Dv;
    a.M = (int)rows; a.K = K; a.ntile_out = (2 * Hh * Dk + Hh * Dv) / 16;
    ga.w_gate = w_gate; ga.b_gate = b_gate; ga.w_decay = w_decay; ga.b_decay = b_decay;
    ga.beta = beta; ga.alpha = alpha; ga.norms = norms; ga.frames = frames; ga.N = N; ga.Hh = Hh; ga.Dk = Dk;
    const int TM = rows >= PROJ_TM_SWITCH ? 128 : 64;
    const size_t lds = (size_t)TM * (K + KPFF_PAD16) * sizeof(bf16_t) + (size_t)(2 * Hh * Dk / 16) * TM * sizeof(float);
    if (lds > 160 * 1024) return gdkvm_fail(GDKVM_ERR_SHAPE, "proj_gates: %zu bytes of LDS for Hh=%d Dk=%d", lds, Hh, Dk);
    {   // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device; lock-free cache as in gdr_scan.hip
        static std::atomic<unsigned long long> done_mask{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "proj_gates: hipGetDevice");
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
            const int cap = 160 * 1024;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_gates_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_gates_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "proj_gates: %s", hipGetErrorString(e));
            done_mask.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    ga.nproj = (int)((rows + TM - 1) / TM);
    const unsigned nalpha = (unsigned)((frames + 7) / 8 < 256 ? (frames + 7) / 8 : 256);
    const unsigned grid = (unsigned)ga.nproj + nalpha;
    if (TM == 128) hipLaunchKernelGGL(proj_gates_kernel<128>, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), ga);
    else hipLaunchKernelGGL(proj_gates_kernel<64>, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), ga);
    GDKVM_LAUNCH_CHECK("proj_gates_kernel");
    return GDKVM_OK;
}
