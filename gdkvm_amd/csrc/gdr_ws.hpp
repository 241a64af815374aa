// gdr_ws.hpp -- workspace layout and argument checks shared by the forward and backward GDR scan entry points.
#pragma once
#include <initializer_list>

#include "gdkvm_common.hpp"

constexpr size_t GDKVM_WS_TAIL = 256 + 1024;   // one zero Ut tile (1 KiB) + trash slot for padded read-out rows
static inline int tiles_for(int N) { return N <= 64 ? 4 : (N <= 128 ? 8 : 16); }

// fp32 workspace per frame-head, NP = 16*nb padded tokens:  wt [NP][64] | knT [64][NP] | ut [Dv/16][nb][64][4] | qinv [NP]
// and, filled only by a training-mode prep (GDKVM_FLAG_TRAIN) for the backward:  kn [NP][64] | wtT [64][NP] | qnT [64][NP]
// | tii [nb][16][16] (the diagonal-block inverses T_II)
// wti [4][nb][64][4] = Wt as accumulator images (like ut), the fold kernel's B operand
// and the folded per-frame affine map the forward scan consumes (gdr_fold_kernel):  pp [4][3][2][64][8] bf16 = I - Kn^T Wt split into three bf16 terms (split3), as A-operand images of the
// bf16 MFMA (row tile, term, 32-wide k step, lane, 8 k values): 24 KiB, in units of float = 1.5 Dk Dk
// | gg [Dv/16][4][64][4] = Kn^T Ut as accumulator images (slice, row tile, lane)
struct WsView { float* wt; float* knT; float* ut; float* qinv; float* kn; float* wtT; float* qnT; float* tii; float* wti; float* pp; float* gg; float* zero; char* trash; int nb; };

static inline size_t gdr_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv)
{
    if (B <= 0 || T <= 0 || Hh <= 0 || N <= 0 || Dk <= 0 || Dv <= 0 || N > GDKVM_MAX_N) return GDKVM_WS_TAIL;
    const size_t NP = 16 * (size_t)tiles_for(N);
    return (size_t)B * T * Hh * (NP * (6 * (size_t)Dk + Dv + 1 + 16) + (size_t)Dk * (Dk + Dk / 2 + Dv)) * sizeof(float) + GDKVM_WS_TAIL;
}

static inline int carve(const char* fn, void* workspace, size_t workspace_bytes, int B, int T, int Hh, int N, int Dk, int Dv, WsView* v)
{
    const size_t need = gdr_workspace_bytes(B, T, Hh, N, Dk, Dv);
    if (workspace_bytes < need)
        return gdkvm_fail(GDKVM_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, workspace_bytes, need);
    v->nb = tiles_for(N);
    const size_t NP = 16 * (size_t)v->nb, FH = (size_t)B * T * Hh;
    v->wt = static_cast<float*>(workspace);
    v->knT = v->wt + FH * NP * GDKVM_DK;
    v->ut = v->knT + FH * NP * GDKVM_DK;
    v->qinv = v->ut + FH * NP * Dv;
    v->kn = v->qinv + FH * NP;
    v->wtT = v->kn + FH * NP * GDKVM_DK;
    v->qnT = v->wtT + FH * NP * GDKVM_DK;
    v->tii = v->qnT + FH * NP * GDKVM_DK;
    v->wti = v->tii + FH * NP * 16;
    v->pp = v->wti + FH * NP * GDKVM_DK;
    v->gg = v->pp + FH * (GDKVM_DK * GDKVM_DK * 3 / 2);
    v->zero = v->gg + FH * GDKVM_DK * Dv;                                 // 256 floats, zeroed by gdkvm_scan_transition
    v->trash = reinterpret_cast<char*>(v->zero + 256);   // write-only slot for read-out rows of padding tokens
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
    if (flags & ~7) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: unknown flags 0x%x", fn, flags);
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
