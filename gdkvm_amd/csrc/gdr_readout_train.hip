// gdr_readout_train.hip -- SURVEY.md §8 rows a1 / a7 for TRAINING on frames of more than 64 tokens (the guide's default
// recipe trains on 256x256 clips, N = 256: /root/reference/website/src/pages/[lang]/reprod/index.astro:246).  There the state
// recurrence runs over 64-token pseudo-frames (gdkvm_amd/ops.py::_scan_chunked) and the LKVA read-out of ALL the frame's tokens
// uses the state before the frame, taken from the saved state history:
//
//   gdr_readout_hist_kernel      R = diag(qinv) Q S                       (forward)
//   gdr_readout_hist_bwd_kernel  dQn = dR S^T,  dQ through the L2 normalisation;   dS = Qn^T dR   (the gradient with respect
//                                to the state before the frame, which gdkvm_scan_state_bwd takes as d_hist)
//
// Both are frame-parallel, exact fp32 (v_mfma_f32_16x16x4_f32) on operands read straight from global memory: the k index of
// an MFMA step is free to be permuted, and every product here is arranged so that a lane's k values are CONTIGUOUS in memory
// (lane group g of k step s takes k = g K/4 + s), i.e. plain 16-byte loads, no LDS staging, no transposes.
#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

struct ReadHistArgs {
    const void* q; const float* s_hist; const void* d_r; void* r_out; void* d_q; float* d_hist;
    int Hh, N, Dv, hist_stride, flags;
};

// 16 consecutive channels (starting at a multiple of 16) of one q row as fp32
template <int IO>
__device__ __forceinline__ void load_q16(const void* q, size_t off, float (&x)[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 v = load4<IO>(q, off + 4 * i);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[4 * i + r] = v[r];
    }
}

// Forward.  grid (B*T*Hh, ceil(Dv/64)); wave w owns column tile c = 4 blockIdx.y + w and keeps S[:, 16c .. 16c+15] in registers
// (lane (g, li): rows 16g .. 16g+15 of column 16c + li) for all token tiles.  The product is computed transposed,
// R^T = S^T Qn^T, so a lane ends with four consecutive columns of one token (one 8/16-byte store).
template <int IO>
__global__ __launch_bounds__(256) void gdr_readout_hist_kernel(ReadHistArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t fh = blockIdx.x;
    const int Hh = a.Hh, N = a.N, Dv = a.Dv, h = (int)(fh % Hh), nsl = Dv / 16;
    const size_t bt = fh / Hh;
    const int c = 4 * blockIdx.y + w;
    if (c >= nsl) return;
    const float* S = a.s_hist + ((bt * a.hist_stride) * Hh + h) * (size_t)(GDKVM_DK * Dv);   // the state before frame bt
    float sc[16];                                          // A operand of S^T: S[16g + s][16c + li], s = 0..15
#pragma unroll
    for (int s = 0; s < 16; ++s) sc[s] = S[(size_t)(16 * g + s) * Dv + 16 * c + li];
    const bool norm = a.flags & GDKVM_FLAG_NORMALIZE_QK;
    constexpr int ESZ = IO == GDKVM_F32 ? 4 : 2;
    for (int tt = 0; 16 * tt < N; ++tt) {
        const int n = min(16 * tt + li, N - 1);
        float qv[16];                                      // B operand of Qn^T: q[token][16g + s]
        load_q16<IO>(a.q, ((bt * N + n) * Hh + h) * (size_t)GDKVM_DK + 16 * g, qv);
        float qinv = 1.f;
        if (norm) {
            float ss = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) ss += qv[s] * qv[s];
            ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
            qinv = 1.0f / sqrtf(ss + GDKVM_EPS_NORM);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; s += 2) { acc0 = mfma4(sc[s], qv[s], acc0); acc1 = mfma4(sc[s + 1], qv[s + 1], acc1); }
        const f32x4 r = (acc0 + acc1) * qinv;              // rows = columns 16c + 4g .. +3 of S, column = token li
        if (16 * tt + li < N) {
            char* p = static_cast<char*>(a.r_out) + (((bt * N + 16 * tt + li) * Hh + h) * (size_t)Dv + 16 * c + 4 * g) * ESZ;
            if constexpr (IO == GDKVM_F32) *reinterpret_cast<f32x4*>(p) = r;
            else *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)f32_to_bf16(r[0]) | ((unsigned)f32_to_bf16(r[1]) << 16),
                                                           (unsigned)f32_to_bf16(r[2]) | ((unsigned)f32_to_bf16(r[3]) << 16));
        }
    }
}

// Backward.  grid (B*T*Hh, 1 + ceil(Dv/64)):
//   blockIdx.y == 0   dQ: wave w walks token tiles w, w+4, ...; per tile dQn[16 x 64] = dR[16 x Dv] S^T with the contraction
//                     index permuted to c = g Dv/4 + s (16-byte loads of dR rows and S rows), then
//                     dq = qinv (dQn - qn <qn, dQn>) through the L2 normalisation (the row dot product: 16 lanes x 4 k tiles).
//   blockIdx.y >= 1   dS for column tiles 4(y-1) .. +3, wave w one tile: dS[64 x 16] = Qn^T dR contracted over the tokens
//                     n = 4s + g (lane group g of k step s; rows of Qn and dR are read 64 bytes at a time), qinv from LDS.
template <int IO>
__global__ __launch_bounds__(256) void gdr_readout_hist_bwd_kernel(ReadHistArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_qinv[];     // [N]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t fh = blockIdx.x;
    const int Hh = a.Hh, N = a.N, Dv = a.Dv, h = (int)(fh % Hh), nsl = Dv / 16;
    const size_t bt = fh / Hh;
    const bool norm = a.flags & GDKVM_FLAG_NORMALIZE_QK;
    const size_t hist_off = ((bt * a.hist_stride) * Hh + h) * (size_t)(GDKVM_DK * Dv);
    const float* S = a.s_hist + hist_off;
    for (int n = tid; n < N; n += 256) {                   // inverse query norms of the frame
        float qi = 1.f;
        if (norm) {
            float ss = 0.f;
            for (int cq = 0; cq < GDKVM_DK; cq += 4) {
                const f32x4 x = load4<IO>(a.q, ((bt * N + n) * Hh + h) * (size_t)GDKVM_DK + cq);
                ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
            }
            qi = 1.0f / sqrtf(ss + GDKVM_EPS_NORM);
        }
        s_qinv[n] = qi;
    }
    __syncthreads();

    if (blockIdx.y == 0) {
        const int q4 = Dv / 4;                             // contraction slice of a lane group
        for (int tt = w; 16 * tt < N; tt += 4) {
            const int n = min(16 * tt + li, N - 1);
            f32x4 acc[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const size_t droff = ((bt * N + n) * Hh + h) * (size_t)Dv + g * q4;
            for (int s = 0; s < q4; s += 4) {
                const f32x4 av = load4<IO>(a.d_r, droff + s);                                // dR[token][g q4 + s .. +3]
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(S + (size_t)(16 * kt + li) * Dv + g * q4 + s);   // S[k][same c]
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[kt] = mfma4(av[r], bv[r], acc[kt]);
                }
            }
            // acc[kt][r] = dQn[token 16tt + 4g + r][k = 16kt + li]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = 16 * tt + 4 * g + r, tk = min(tok, N - 1);
                const float qi = s_qinv[tk];
                float qn[4], dot = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    qn[kt] = load1<IO>(a.q, ((bt * N + tk) * Hh + h) * (size_t)GDKVM_DK + 16 * kt + li) * qi;
                    dot += qn[kt] * acc[kt][r];
                }
                dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4); dot += __shfl_xor(dot, 8);
                if (tok < N) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) {
                        const float dq = norm ? qi * (acc[kt][r] - qn[kt] * dot) : acc[kt][r];
                        store1<IO>(a.d_q, ((bt * N + tok) * Hh + h) * (size_t)GDKVM_DK + 16 * kt + li, dq);
                    }
                }
            }
        }
        return;
    }

    const int c = 4 * ((int)blockIdx.y - 1) + w;
    if (c >= nsl) return;
    f32x4 acc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < N; n0 += 4) {                    // k step: tokens n0 + g
        const int n = min(n0 + g, N - 1);
        const float live = n0 + g < N ? s_qinv[n] : 0.f;   // padding tokens contribute nothing
        const float bv = load1<IO>(a.d_r, ((bt * N + n) * Hh + h) * (size_t)Dv + 16 * c + li);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float av = load1<IO>(a.q, ((bt * N + n) * Hh + h) * (size_t)GDKVM_DK + 16 * kt + li) * live;   // Qn[n][16kt + li]
            acc[kt] = mfma4(av, bv, acc[kt]);
        }
    }
    float* dS = a.d_hist + hist_off;                       // acc[kt][r] = dS[16kt + 4g + r][16c + li]
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dS[(size_t)(16 * kt + 4 * g + r) * Dv + 16 * c + li] = acc[kt][r];
}

int check_hist(const char* fn, int B, int T, int Hh, int N, int Dk, int Dv, int hist_stride, int io_dtype, int flags)
{
    if (B < 0 || T < 0 || Hh <= 0 || N < 0 || Dv <= 0 || hist_stride <= 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: bad dimension (B=%d T=%d Hh=%d N=%d Dv=%d hist_stride=%d)", fn, B, T, Hh, N, Dv, hist_stride);
    if (Dk != GDKVM_DK) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: Dk=%d unsupported (kernels are built for Dk=%d)", fn, Dk, GDKVM_DK);
    if (Dv % 16 != 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: Dv=%d must be a multiple of 16", fn, Dv);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: io_dtype=%d", fn, io_dtype);
    if (flags & ~15) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: unknown flags 0x%x", fn, flags);
    return GDKVM_OK;
}

}  // namespace

extern "C" int gdkvm_readout_fwd(const void* q, const float* s_hist, void* r_out, int B, int T, int Hh, int N, int Dk, int Dv,
                                 int hist_stride, int io_dtype, int flags, void* stream)
{
    if (int rc = check_hist("readout_fwd", B, T, Hh, N, Dk, Dv, hist_stride, io_dtype, flags)) return rc;
    if (B == 0 || T == 0 || N == 0) return GDKVM_OK;
    if (int rc = check_ptrs("readout_fwd", {q, s_hist, r_out}, {})) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    ReadHistArgs ra{q, s_hist, nullptr, r_out, nullptr, nullptr, Hh, N, Dv, hist_stride, flags};
    const dim3 grid((unsigned)(B * T * Hh), (unsigned)((Dv / 16 + 3) / 4));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gdr_readout_hist_kernel<GDKVM_F32>), grid, dim3(256), 0, st, ra);
    else hipLaunchKernelGGL((gdr_readout_hist_kernel<GDKVM_BF16>), grid, dim3(256), 0, st, ra);
    GDKVM_LAUNCH_CHECK("gdr_readout_hist_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_readout_bwd(const void* q, const float* s_hist, const void* d_r, void* d_q, float* d_hist,
                                 int B, int T, int Hh, int N, int Dk, int Dv, int hist_stride, int io_dtype, int flags, void* stream)
{
    if (int rc = check_hist("readout_bwd", B, T, Hh, N, Dk, Dv, hist_stride, io_dtype, flags)) return rc;
    if (B == 0 || T == 0 || N == 0) return GDKVM_OK;
    if (N > 16384) return gdkvm_fail(GDKVM_ERR_SHAPE, "readout_bwd: N=%d exceeds 16384 tokens per frame", N);
    if (int rc = check_ptrs("readout_bwd", {q, s_hist, d_r, d_q, d_hist}, {})) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    ReadHistArgs ra{q, s_hist, d_r, nullptr, d_q, d_hist, Hh, N, Dv, hist_stride, flags};
    const dim3 grid((unsigned)(B * T * Hh), (unsigned)(1 + (Dv / 16 + 3) / 4));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)N * sizeof(float);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gdr_readout_hist_bwd_kernel<GDKVM_F32>), grid, dim3(256), lds, st, ra);
    else hipLaunchKernelGGL((gdr_readout_hist_bwd_kernel<GDKVM_BF16>), grid, dim3(256), lds, st, ra);
    GDKVM_LAUNCH_CHECK("gdr_readout_hist_bwd_kernel");
    return GDKVM_OK;
}
