// gdr_ws.hpp -- workspace layout and argument checks shared by the forward and backward GDR scan entry points.
#pragma once
#include <atomic>
#include <initializer_list>

#include "gdkvm_common.hpp"

constexpr size_t GDKVM_WS_TAIL = 1024 + 1024;  // one zero G tile (1 KiB) + write-only slot for padded read-out rows (16 B per lane)
static inline int tiles_for(int N) { return N <= 64 ? 4 : 4 * ((N + 63) / 64); }   // 16-token tiles of the padded frame: whole 64-token chunks

// fp32 workspace per frame-head fh.  NP = 16*nb padded tokens (nb = 4 per 64-token chunk), NL = min(NP, 64):
//   legacy WY regions, written by the training-mode prep for frames of <= 64 tokens (the backward's operands), NL tokens wide:
//     wt [NL][64] | knT [64][NL] | ut [Dv/16][4][64][4] | kn [NL][64] | wtT [64][NL] | qnT [64][NL] | tii [4][16][16] | wti [4][4][64][4]
//     ppt = P^T as split3 images [4][3][2][64][8] bf16 (three bf16 terms: full fp32 range), the operator of the backward's
//           reverse recurrence
//   qinv [NP]
//   the folded per-frame affine map the forward scan consumes:
//     pp [4][NT][2][64][8] x 16 bit = P = I - Kn^T Wt as A-operand images of the 16x16x32 MFMA (row tile, term, 32-wide k step,
//        lane, 8 k values): NT = 2 fp16 terms (pair16, gdr_device.hpp: 16 KiB) by default, NT = 3 bf16 terms (split3: 24 KiB)
//        under GDKVM_FLAG_WIDE_RANGE; the slot is 24 KiB = 1.5 Dk Dk floats either way
//     gg [Dv/16][4][64][4] = G = Kn^T Ut as accumulator images (slice, row tile, lane)
//   frames of more than 64 tokens are folded in chunks of 64 tokens whose affine maps are then composed (nchunk = NP/64):
//     x0  [4 + Dv/16][4][64][4]  chunk 0 as accumulator images of [P | G]
//     ppc [nchunk-1] x pp, ggc [nchunk-1] x gg  for chunks 1..
//     simg [Dv/16][4][64][4]  the state before the frame as MFMA operand images (per slice: the h and m bf16 term images, or the
//          fp32 accumulator images), dumped by the serial kernel for the frame-parallel read-out kernel
//   range bookkeeping of the pair16 recurrence (gdr_scan.hip, "state exponent"):
//     gmax [Dv/16][4]  per frame-head and 16-column slice: max |G| of the frame's final map, one entry per row tile (a producer
//          that holds all four row tiles of a slice writes the maximum into entry 0 and zeros behind it; +inf = an intermediate of
//          the chunk composition left the fp16 pair's range)
//   and, per clip-head (behind everything per-frame): esc [Dv/16] = 2^e, the inverse of the scale the serial kernel carried that
//          slice's state at, for the frame-parallel read-out kernel
struct WsView {
    float* wt; float* knT; float* ut; float* qinv; float* kn; float* wtT; float* qnT; float* tii; float* wti; float* ppt;
    float* pp; float* gg; float* x0; float* ppc; float* ggc; float* simg; float* gmax; float* esc; float* zero; char* trash; int nb; int nchunk;
};

static inline size_t gdr_ws_floats_per_fh(int N, int Dk, int Dv)
{
    const size_t NP = 16 * (size_t)tiles_for(N), NL = NP < 64 ? NP : 64, C = (NP + 63) / 64;
    const size_t pg = (size_t)Dk * Dk * 3 / 2 + (size_t)Dk * Dv;
    return NL * (6 * (size_t)Dk + Dv + 16) + (size_t)Dk * Dk * 3 / 2 + NP + pg + (C > 1 ? (size_t)Dk * (Dk + Dv) + (C - 1) * pg + (size_t)Dk * Dv : 0)
           + (size_t)Dv / 4;                                                       // gmax: 4 floats per 16-column slice
}

static inline size_t gdr_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv)
{
    if (B <= 0 || T <= 0 || Hh <= 0 || N <= 0 || Dk <= 0 || Dv <= 0 || N > GDKVM_MAX_N) return GDKVM_WS_TAIL;
    return ((size_t)B * T * Hh * gdr_ws_floats_per_fh(N, Dk, Dv) + (size_t)B * Hh * (((size_t)Dv / 16 + 3) & ~(size_t)3)) * sizeof(float) + GDKVM_WS_TAIL;
}

// Key widths below the kernels' 64 (multiples of 8): gdkvm_scan_fwd runs the Dk = 64 kernels on zero-extended copies of q, k and the
// state kept behind the regular workspace -- norms, Gram matrices and read-outs are unchanged by zero channels, P stays the identity
// on the extra rows and the extra rows of the state stay zero, so the result on the real rows is that of the narrower problem.
static inline bool gdr_narrow_keys(int Dk) { return Dk >= 8 && Dk < GDKVM_DK && Dk % 8 == 0; }
static inline size_t gdr_up256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline size_t gdr_narrow_extra_bytes(int B, int T, int Hh, int N, int Dv)
{   // q and k rows at 64 channels (sized for fp32), state in and out at 64 rows
    return 2 * gdr_up256((size_t)B * T * N * Hh * GDKVM_DK * 4) + 2 * gdr_up256((size_t)B * Hh * GDKVM_DK * Dv * 4);
}
// 16-byte units: block b of nblk copies copy_q units from src + b*src_q to dst + b*dst_q and zero-fills up to fill_q (gdr_train.hip)
int gdr_block_copy(const void* src, void* dst, size_t nblk, size_t src_q, size_t dst_q, size_t copy_q, size_t fill_q, hipStream_t st);
// Key widths above 64 (multiples of 8 up to 256): gdkvm_scan_fwd runs the definitional kernel of gdr_general.hip (no workspace)
bool gdr_wide_keys(int Dk);
int gdr_general_scan_fwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta, const float* s_in, void* r_out,
                         float* s_out, int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, hipStream_t st);

static inline int carve(const char* fn, void* workspace, size_t workspace_bytes, int B, int T, int Hh, int N, int Dk, int Dv, WsView* v)
{
    const size_t need = gdr_workspace_bytes(B, T, Hh, N, Dk, Dv);
    if (workspace_bytes < need)
        return gdkvm_fail(GDKVM_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, workspace_bytes, need);
    v->nb = tiles_for(N);
    const size_t NP = 16 * (size_t)v->nb, NL = NP < 64 ? NP : 64, FH = (size_t)B * T * Hh;
    v->nchunk = N > 64 ? (N + 63) / 64 : 1;              // sized for NP/64, the upper bound
    // every P slot is sized for three bf16 terms (1.5 Dk Dk floats); the default pair16 images use the first Dk Dk of it
    const size_t ppf = (size_t)GDKVM_DK * GDKVM_DK * 3 / 2, ggf = (size_t)GDKVM_DK * Dv;
    float* p = static_cast<float*>(workspace);
    v->wt = p;    p += FH * NL * GDKVM_DK;
    v->knT = p;   p += FH * NL * GDKVM_DK;
    v->ut = p;    p += FH * NL * Dv;
    v->kn = p;    p += FH * NL * GDKVM_DK;
    v->wtT = p;   p += FH * NL * GDKVM_DK;
    v->qnT = p;   p += FH * NL * GDKVM_DK;
    v->tii = p;   p += FH * NL * 16;
    v->wti = p;   p += FH * NL * GDKVM_DK;
    v->ppt = p;   p += FH * ppf;
    v->qinv = p;  p += FH * NP;
    v->pp = p;    p += FH * ppf;
    v->gg = p;    p += FH * ggf;
    v->x0 = p;    p += NP > 64 ? FH * (size_t)GDKVM_DK * (GDKVM_DK + Dv) : 0;      // (sized for NP / 64 chunks, the upper bound)
    v->ppc = p;   p += NP > 64 ? FH * (NP / 64 - 1) * ppf : 0;
    v->ggc = p;   p += NP > 64 ? FH * (NP / 64 - 1) * ggf : 0;
    v->simg = p;  p += NP > 64 ? FH * ggf : 0;
    v->gmax = p;  p += FH * ((size_t)Dv / 4);
    v->esc = p;   p += (size_t)B * Hh * (((size_t)Dv / 16 + 3) & ~(size_t)3);
    v->zero = p;                                         // 256 floats, zeroed by gdkvm_scan_transition
    v->trash = reinterpret_cast<char*>(v->zero + 256);   // write-only slot for read-out rows of padding tokens
    return GDKVM_OK;
}

// > 64 KiB of dynamic LDS needs an opt-in per kernel and device: done once, remembered in the caller's lock-free mask (one
// static mask per kernel instantiation) -- host threads may drive several devices concurrently.
static inline int gdr_lds_optin(const void* fn, std::atomic<unsigned long long>& done_mask, size_t bytes, const char* who)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "%s: hipGetDevice", who);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "%s: LDS attribute: %s", who, hipGetErrorString(e));
        done_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    return GDKVM_OK;
}

static inline int check_common(const char* fn, int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int flags)
{
    if (B < 0 || T < 0 || Hh <= 0 || N < 0 || Dv <= 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: negative or zero dimension (B=%d T=%d Hh=%d N=%d Dv=%d)", fn, B, T, Hh, N, Dv);
    if (Dk != GDKVM_DK) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: Dk=%d unsupported (kernels are built for Dk=%d)", fn, Dk, GDKVM_DK);
    if (Dv % 16 != 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: Dv=%d must be a multiple of 16", fn, Dv);
    if (N > GDKVM_MAX_N) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: N=%d exceeds %d tokens per frame", fn, N, GDKVM_MAX_N);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: io_dtype=%d", fn, io_dtype);
    if (flags & ~15) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: unknown flags 0x%x", fn, flags);
    return GDKVM_OK;
}

static inline int check_ptrs(const char* fn, std::initializer_list<const void*> required, std::initializer_list<const void*> optional)
{
    for (const void* p : required) {
        if (!p) return gdkvm_fail(GDKVM_ERR_ARG, "%s: null pointer", fn);
        if (!gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "%s: pointer %p is not 16-byte aligned", fn, p);
    }
    for (const void* p : optional)
        if (p && !gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "%s: pointer %p is not 16-byte aligned", fn, p);
    return GDKVM_OK;
}
