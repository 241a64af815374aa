// gdr_train.hip -- SURVEY.md §8 row a7 at the C ABI for frames of ANY token count: gdkvm_scan_train_fwd / gdkvm_scan_train_bwd.
//
// gdkvm_scan_bwd itself runs on the 64-token kernels.  The tokens of a frame act on the state in order, so a frame of N > 64 tokens is
// a sequence of C = ceil(N / 64) pseudo-frames of 64 tokens: the first carries the frame's gate, the others gate 1, padding tokens
// beta = 0.  The state recurrence and its backward then run over T * C steps (gdkvm_scan_fwd without a read-out, gdkvm_scan_state_bwd),
// and the read-out of ALL the frame's tokens uses the state before the frame's first pseudo-frame (gdkvm_readout_fwd / _bwd on the
// saved history, whose state gradients enter the reverse recurrence as its additive term).  Until round 2 only gdkvm_amd/ops.py
// composed these calls, with framework padding ops in between; here the padded operands are built by two copy kernels into ONE
// workspace that carries everything from the forward to the backward call.
#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

constexpr float TRAIN_BIG_LOGIT = 1.0e30f;        // sigmoid(+-1e30) is exactly 1 / 0 in the kernels' formulas

// 16-byte units: block b of `nblk` copies copy_q units from src + b*src_q to dst + b*dst_q and zero-fills up to fill_q units
// (padding: src_q = copy_q < fill_q = dst_q; un-padding: copy_q = fill_q = dst_q < src_q).
__global__ void gdr_block_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t nblk, size_t src_q, size_t dst_q,
                                      size_t copy_q, size_t fill_q)
{
    const size_t total = nblk * fill_q;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / fill_q, r = i - b * fill_q;
        dst[b * dst_q + r] = r < copy_q ? src[b * src_q + r] : make_uint4(0u, 0u, 0u, 0u);
    }
}

// alpha [BT,Hh] -> alpha_p [BT,C,Hh] (the frame's gate, then `one`);  beta [BT,N,Hh] -> beta_p [BT,C*64,Hh] (padding: `off`)
__global__ void gdr_pad_gates_kernel(const float* __restrict__ alpha, const float* __restrict__ beta, float* __restrict__ alpha_p,
                                     float* __restrict__ beta_p, size_t BT, int C, int N, int Hh, float one, float off)
{
    const size_t nb = (size_t)C * 64 * Hh, na = (size_t)C * Hh, total = BT * (nb + na);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        if (i < BT * nb) {
            const size_t f = i / nb, r = i - f * nb;
            beta_p[i] = r < (size_t)N * Hh ? beta[f * N * Hh + r] : off;
        } else {
            const size_t j = i - BT * nb, f = j / na, r = j - f * na;
            alpha_p[j] = r < (size_t)Hh ? alpha[f * Hh + r] : one;
        }
    }
}

// the way back: d_beta = d_beta_p of the real tokens, d_alpha = d_alpha_p of every frame's first pseudo-frame (the others' gates are constants)
__global__ void gdr_unpad_gates_kernel(const float* __restrict__ da_p, const float* __restrict__ db_p, float* __restrict__ d_alpha,
                                       float* __restrict__ d_beta, size_t BT, int C, int N, int Hh)
{
    const size_t nb = (size_t)N * Hh, total = BT * (nb + Hh);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        if (i < BT * nb) {
            const size_t f = i / nb, r = i - f * nb;
            d_beta[i] = db_p[f * C * 64 * Hh + r];
        } else {
            const size_t j = i - BT * nb, f = j / Hh, r = j - f * Hh;
            d_alpha[j] = da_p[f * C * Hh + r];
        }
    }
}

inline size_t up256(size_t x) { return gdr_up256(x); }

struct TrainView {
    int C, Np, Tc;
    float* hist; char* fws; size_t fws_bytes; char* bws; size_t bws_bytes;
    char* k_p; char* v_p; float* alpha_p; float* beta_p; float* d_hist; char* dk_p; char* dv_p; float* da_p; float* db_p;
    size_t total;
};

TrainView train_carve(void* base, int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype)
{
    TrainView v{};
    v.C = N > 64 ? (N + 63) / 64 : 1;
    v.Np = v.C > 1 ? 64 : N;
    v.Tc = T * v.C;
    const size_t es = io_dtype == GDKVM_F32 ? 4 : 2;
    const size_t FH = (size_t)B * v.Tc * Hh;
    size_t off = 0;                                   // (base == NULL: sizes only)
    auto take = [&](size_t bytes) { char* r = base ? static_cast<char*>(base) + off : nullptr; off += up256(bytes); return r; };
    v.hist = reinterpret_cast<float*>(take(FH * Dk * Dv * sizeof(float)));
    v.fws_bytes = gdkvm_scan_workspace_bytes(B, v.Tc, Hh, v.Np, Dk, Dv);
    v.fws = take(v.fws_bytes);
    v.bws_bytes = gdkvm_scan_bwd_workspace_bytes(B, v.Tc, Hh, v.Np, Dk, Dv);
    v.bws = take(v.bws_bytes);
    if (v.C > 1) {
        v.k_p = take(FH * 64 * Dk * es);
        v.v_p = take(FH * 64 * Dv * es);
        v.alpha_p = reinterpret_cast<float*>(take(FH * sizeof(float)));
        v.beta_p = reinterpret_cast<float*>(take(FH * 64 * sizeof(float)));
        v.d_hist = reinterpret_cast<float*>(take(FH * Dk * Dv * sizeof(float)));
        v.dk_p = take(FH * 64 * Dk * es);
        v.dv_p = take(FH * 64 * Dv * es);
        v.da_p = reinterpret_cast<float*>(take(FH * sizeof(float)));
        v.db_p = reinterpret_cast<float*>(take(FH * 64 * sizeof(float)));
    }
    v.total = off;
    return v;
}

int train_check(const char* fn, void* ws, size_t ws_bytes, int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags,
                TrainView* out)
{
    if (int rc = check_common(fn, B, T, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (rule < 0 || rule > 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: rule=%d", fn, rule);
    if (T <= 0 || N <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: T and N must be positive", fn);
    if (N > 64 && rule == GDKVM_RULE_DELTA_PARALLEL)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: training with rule delta_parallel is limited to 64 tokens per frame (its chunks combine additively)", fn);
    if (B == 0) return GDKVM_OK;
    if (!ws || !gdkvm_aligned16(ws)) return gdkvm_fail(GDKVM_ERR_ARG, "%s: workspace null or misaligned", fn);
    *out = train_carve(ws, B, T, Hh, N, Dk, Dv, io_dtype);
    if (ws_bytes < out->total) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, ws_bytes, out->total);
    return GDKVM_OK;
}

unsigned copy_grid(size_t n) { const size_t g = (n + 255) / 256; return (unsigned)(g > 16384 ? 16384 : (g ? g : 1)); }

}  // namespace

int gdr_block_copy(const void* src, void* dst, size_t nblk, size_t src_q, size_t dst_q, size_t copy_q, size_t fill_q, hipStream_t st)
{
    if (nblk == 0 || fill_q == 0) return GDKVM_OK;
    hipLaunchKernelGGL(gdr_block_copy_kernel, dim3(copy_grid(nblk * fill_q)), dim3(256), 0, st, static_cast<const uint4*>(src),
                       static_cast<uint4*>(dst), nblk, src_q, dst_q, copy_q, fill_q);
    GDKVM_LAUNCH_CHECK("gdr_block_copy_kernel");
    return GDKVM_OK;
}

extern "C" size_t gdkvm_scan_train_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype)
{
    if (B <= 0 || T <= 0 || Hh <= 0 || N <= 0 || Dk <= 0 || Dv <= 0) return 256;
    return train_carve(nullptr, B, T, Hh, N, Dk, Dv, io_dtype).total + 256;
}

extern "C" int gdkvm_scan_train_fwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta, const float* s_in,
                                    void* r_out, float* s_out, void* train_workspace, size_t train_workspace_bytes,
                                    int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    TrainView tv;
    if (int rc = train_check("scan_train_fwd", train_workspace, train_workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, &tv)) return rc;
    if (B == 0) return GDKVM_OK;
    if (int rc = check_ptrs("scan_train_fwd", {q, k, v, alpha, beta, r_out}, {s_in, s_out})) return rc;
    if (tv.C == 1)
        return gdkvm_scan_fwd(q, k, v, alpha, beta, s_in, r_out, s_out, tv.hist, tv.fws, tv.fws_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream);
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t es = io_dtype == GDKVM_F32 ? 4 : 2, BT = (size_t)B * T;
    const bool logits = flags & GDKVM_FLAG_GATE_LOGITS;
    {
        const size_t kq = (size_t)N * Hh * Dk * es / 16, kpq = (size_t)tv.C * 64 * Hh * Dk * es / 16;
        const size_t vq = (size_t)N * Hh * Dv * es / 16, vpq = (size_t)tv.C * 64 * Hh * Dv * es / 16;
        hipLaunchKernelGGL(gdr_block_copy_kernel, dim3(copy_grid(BT * kpq)), dim3(256), 0, st, static_cast<const uint4*>(k),
                           reinterpret_cast<uint4*>(tv.k_p), BT, kq, kpq, kq, kpq);
        hipLaunchKernelGGL(gdr_block_copy_kernel, dim3(copy_grid(BT * vpq)), dim3(256), 0, st, static_cast<const uint4*>(v),
                           reinterpret_cast<uint4*>(tv.v_p), BT, vq, vpq, vq, vpq);
        hipLaunchKernelGGL(gdr_pad_gates_kernel, dim3(copy_grid(BT * tv.C * 65 * Hh)), dim3(256), 0, st, alpha, beta, tv.alpha_p, tv.beta_p,
                           BT, tv.C, N, Hh, logits ? TRAIN_BIG_LOGIT : 1.0f, logits ? -TRAIN_BIG_LOGIT : 0.0f);
        GDKVM_LAUNCH_CHECK("gdr_pad kernels");
    }
    // the state recurrence over the pseudo-frames (no read-out; the padded keys stand in for the queries, whose norms nobody reads)
    if (int rc = gdkvm_scan_fwd(tv.k_p, tv.k_p, tv.v_p, tv.alpha_p, tv.beta_p, s_in, nullptr, s_out, tv.hist, tv.fws, tv.fws_bytes,
                                B, tv.Tc, Hh, 64, Dk, Dv, io_dtype, rule, flags, stream)) return rc;
    return gdkvm_readout_fwd(q, tv.hist, r_out, B, T, Hh, N, Dk, Dv, tv.C, io_dtype, flags & GDKVM_FLAG_NORMALIZE_QK, stream);
}

extern "C" int gdkvm_scan_train_bwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                                    const void* d_r, const float* d_s_out,
                                    void* d_q, void* d_k, void* d_v, float* d_alpha, float* d_beta, float* d_s_in,
                                    void* train_workspace, size_t train_workspace_bytes,
                                    int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream)
{
    TrainView tv;
    if (int rc = train_check("scan_train_bwd", train_workspace, train_workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, &tv)) return rc;
    if (B == 0) return GDKVM_OK;
    if (int rc = check_ptrs("scan_train_bwd", {q, k, v, alpha, beta, d_r, d_q, d_k, d_v, d_alpha, d_beta}, {d_s_out, d_s_in})) return rc;
    if (tv.C == 1)
        return gdkvm_scan_bwd(q, k, v, alpha, beta, tv.hist, tv.fws, tv.fws_bytes, d_r, d_s_out, d_q, d_k, d_v, d_alpha, d_beta, d_s_in,
                              tv.bws, tv.bws_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream);
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t es = io_dtype == GDKVM_F32 ? 4 : 2, BT = (size_t)B * T;
    // read-out backward: d_q, and the state gradients of the states the frames read (every C-th entry of d_hist; the rest is zero)
    if (int rc = gdkvm_zero_async(tv.d_hist, (size_t)B * tv.Tc * Hh * Dk * Dv * sizeof(float), st)) return rc;
    if (int rc = gdkvm_readout_bwd(q, tv.hist, d_r, d_q, tv.d_hist, B, T, Hh, N, Dk, Dv, tv.C, io_dtype, flags & GDKVM_FLAG_NORMALIZE_QK, stream))
        return rc;
    if (int rc = gdkvm_scan_state_bwd(tv.k_p, tv.v_p, tv.alpha_p, tv.beta_p, tv.hist, tv.fws, tv.fws_bytes, tv.d_hist, d_s_out,
                                      tv.dk_p, tv.dv_p, tv.da_p, tv.db_p, d_s_in, tv.bws, tv.bws_bytes, B, tv.Tc, Hh, 64, Dk, Dv,
                                      io_dtype, rule, flags, stream)) return rc;
    const size_t kq = (size_t)N * Hh * Dk * es / 16, kpq = (size_t)tv.C * 64 * Hh * Dk * es / 16;
    const size_t vq = (size_t)N * Hh * Dv * es / 16, vpq = (size_t)tv.C * 64 * Hh * Dv * es / 16;
    hipLaunchKernelGGL(gdr_block_copy_kernel, dim3(copy_grid(BT * kq)), dim3(256), 0, st, reinterpret_cast<const uint4*>(tv.dk_p),
                       static_cast<uint4*>(d_k), BT, kpq, kq, kq, kq);
    hipLaunchKernelGGL(gdr_block_copy_kernel, dim3(copy_grid(BT * vq)), dim3(256), 0, st, reinterpret_cast<const uint4*>(tv.dv_p),
                       static_cast<uint4*>(d_v), BT, vpq, vq, vq, vq);
    hipLaunchKernelGGL(gdr_unpad_gates_kernel, dim3(copy_grid(BT * (N + 1) * Hh)), dim3(256), 0, st, tv.da_p, tv.db_p, d_alpha, d_beta, BT, tv.C, N, Hh);
    GDKVM_LAUNCH_CHECK("gdr_unpad kernels");
    return GDKVM_OK;
}
