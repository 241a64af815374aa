// gdr_step.hip -- the two small kernels of the per-frame `step` mode with mask feedback (if the value written
// for frame t depends on the mask predicted for frame t, the time loop cannot be one scan launch -- it returns to the decoder every
// frame).  Per frame the module then runs: gdkvm_lkva_read (R_t = Qn_t S_{t-1}, below) -> KPFF -> decoder -> mask kernel ->
// gdkvm_mask_embed_add (v_t += w_embed * pooled mask_t, below) -> gdkvm_scan_fwd with T = 1 (the write; its read-out is not used).
//
// gdkvm_lkva_read: row a1 on its own, the state given as a tensor.  One workgroup per (clip, head, 64 value columns): the 64 x 64
// slice of S and 64 query rows at a time sit in LDS as fp32, a thread owns a 4-token x 4-column tile and runs the k-ordered fmaf chain
// over the 64 key channels -- plain fp32, deterministic, no operand splitting (a frame is 49 ... 256 tokens: ~3 us, launch-bound).
// Normalisation (GDKVM_FLAG_NORMALIZE_QK): the inverse norm of the row AS STORED multiplies the finished dot product; `norms`
// ([rows, Hh, 2] from gdkvm_proj_gates: entry 1 is the query's) is used when given.
#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

struct ReadArgs { const void* q; const float* norms; const float* s; void* r; int N, Hh, Dv, normalize; };

template <int IO>
__global__ __launch_bounds__(256) void lkva_read_kernel(ReadArgs a)
{
    constexpr int DK = GDKVM_DK, TN = 64, TC = 64, QS = DK + 1;     // q rows padded to 65 floats: the norm pass reads a column of rows
    __shared__ float s_s[DK * TC];                                   // S[d][c0 + c]
    __shared__ float s_q[TN * QS];
    __shared__ float s_inv[TN];
    const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.Hh, h = bh - b * a.Hh, c0 = blockIdx.y * TC;
    const float* S = a.s + (size_t)bh * DK * a.Dv;
    for (int i = tid; i < DK * TC / 4; i += 256) {
        const int d = i / (TC / 4), c4 = i - d * (TC / 4);
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (c0 + 4 * c4 < a.Dv) val = *reinterpret_cast<const f32x4*>(S + (size_t)d * a.Dv + c0 + 4 * c4);
        *reinterpret_cast<f32x4*>(s_s + d * TC + 4 * c4) = val;
    }
    const int tr = tid >> 4, tc = tid & 15;                          // this thread's tile: tokens 4 tr .. +3, columns 4 tc .. +3
    for (int n0 = 0; n0 < a.N; n0 += TN) {
        __syncthreads();                                             // (the previous chunk's reads of s_q are done; first pass: S has landed)
        for (int i = tid; i < TN * DK / 4; i += 256) {
            const int n = i / (DK / 4), d4 = i - n * (DK / 4);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (n0 + n < a.N) val = load4<IO>(a.q, (((size_t)b * a.N + n0 + n) * a.Hh + h) * DK + 4 * d4);
            float* dst = s_q + n * QS + 4 * d4;
            dst[0] = val[0]; dst[1] = val[1]; dst[2] = val[2]; dst[3] = val[3];
        }
        __syncthreads();
        if (tid < TN) {
            float inv = 1.f;
            if (a.normalize && n0 + tid < a.N) {
                if (a.norms) inv = a.norms[(((size_t)b * a.N + n0 + tid) * a.Hh + h) * 2 + 1];
                else {
                    float ss = 0.f;
                    for (int d = 0; d < DK; ++d) ss = fmaf(s_q[tid * QS + d], s_q[tid * QS + d], ss);
                    inv = 1.0f / sqrtf(ss + GDKVM_EPS_NORM);
                }
            }
            s_inv[tid] = inv;
        }
        __syncthreads();
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < DK; ++d) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s_s + d * TC + 4 * tc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float qv = s_q[(4 * tr + i) * QS + d];
                acc[i][0] = fmaf(qv, sv[0], acc[i][0]); acc[i][1] = fmaf(qv, sv[1], acc[i][1]);
                acc[i][2] = fmaf(qv, sv[2], acc[i][2]); acc[i][3] = fmaf(qv, sv[3], acc[i][3]);
            }
        }
        if (c0 + 4 * tc < a.Dv) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + 4 * tr + i;
                if (n >= a.N) continue;
                const float inv = s_inv[4 * tr + i];
                const size_t o = (((size_t)b * a.N + n) * a.Hh + h) * a.Dv + c0 + 4 * tc;
                if constexpr (IO == GDKVM_F32) {
                    *reinterpret_cast<f32x4*>(static_cast<float*>(a.r) + o) = f32x4{acc[i][0] * inv, acc[i][1] * inv, acc[i][2] * inv, acc[i][3] * inv};
                } else {
                    uint2 u;
                    u.x = (unsigned)f32_to_bf16(acc[i][0] * inv) | ((unsigned)f32_to_bf16(acc[i][1] * inv) << 16);
                    u.y = (unsigned)f32_to_bf16(acc[i][2] * inv) | ((unsigned)f32_to_bf16(acc[i][3] * inv) << 16);
                    *reinterpret_cast<uint2*>(static_cast<bf16_t*>(a.r) + o) = u;
                }
            }
        }
    }
}

// v[f, n, c] += w[c] * mean over token n's adaptive-average-pool cell of (mask[f] != 0 and != 255): one wave per token.
// Cell of token (i, j) on an h x w grid over an H x W mask: rows [floor(i H / h), ceil((i + 1) H / h)), columns likewise (PyTorch's
// adaptive_avg_pool2d).  The sum is an exact integer; the mean is one correctly rounded fp32 division.
struct EmbedArgs { const uint8_t* mask; const float* w; void* v; int H, W, h, wd, C; };

template <int IO>
__global__ __launch_bounds__(64) void mask_embed_add_kernel(EmbedArgs a)
{
    const int lane = threadIdx.x, N = a.h * a.wd;
    const size_t tok = blockIdx.x, f = tok / N;
    const int n = (int)(tok - f * N), i = n / a.wd, j = n - i * a.wd;
    const int y0 = (i * a.H) / a.h, y1 = ((i + 1) * a.H + a.h - 1) / a.h, x0 = (j * a.W) / a.wd, x1 = ((j + 1) * a.W + a.wd - 1) / a.wd;
    const int cw = x1 - x0, cnt = (y1 - y0) * cw;
    const uint8_t* m = a.mask + f * (size_t)a.H * a.W;
    int sum = 0;
    for (int p = lane; p < cnt; p += 64) {
        const int yy = y0 + p / cw, xx = x0 + p % cw;
        const uint8_t c = m[(size_t)yy * a.W + xx];
        sum += (c != 0 && c != 255) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float avg = (float)sum / (float)cnt;
    for (int c = lane; c < a.C; c += 64) {
        const size_t o = tok * (size_t)a.C + c;
        store1<IO>(a.v, o, fmaf(avg, a.w[c], load1<IO>(a.v, o)));
    }
}

}  // namespace

extern "C" int gdkvm_lkva_read(const void* q, const float* norms, const float* s, void* r_out,
                               int B, int N, int Hh, int Dk, int Dv, int io_dtype, int flags, void* stream)
{
    if (int rc = check_common("lkva_read", B, 1, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (B == 0 || N == 0) return GDKVM_OK;
    if (int rc = check_ptrs("lkva_read", {q, s, r_out}, {norms})) return rc;
    if (norms && !(flags & GDKVM_FLAG_NORMALIZE_QK)) return gdkvm_fail(GDKVM_ERR_ARG, "lkva_read: norms given without GDKVM_FLAG_NORMALIZE_QK");
    if (int rc = gdkvm_check_device()) return rc;
    if ((long long)B * Hh > 0x7fffffffLL) return gdkvm_fail(GDKVM_ERR_SHAPE, "lkva_read: B * Hh too large");
    ReadArgs a{q, norms, s, r_out, N, Hh, Dv, (flags & GDKVM_FLAG_NORMALIZE_QK) ? 1 : 0};
    const dim3 grid((unsigned)(B * Hh), (unsigned)((Dv + 63) / 64));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL(lkva_read_kernel<GDKVM_F32>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(lkva_read_kernel<GDKVM_BF16>, grid, dim3(256), 0, st, a);
    GDKVM_LAUNCH_CHECK("lkva_read");
    return GDKVM_OK;
}

extern "C" int gdkvm_mask_embed_add(const uint8_t* mask, const float* w_embed, void* v, int BT, int H, int W, int h, int w, int C,
                                    int io_dtype, void* stream)
{
    if (BT < 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || C <= 0 || h > H || w > W)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "mask_embed_add: BT=%d mask %dx%d tokens %dx%d C=%d", BT, H, W, h, w, C);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "mask_embed_add: io_dtype=%d", io_dtype);
    if (BT == 0) return GDKVM_OK;
    if (!mask) return gdkvm_fail(GDKVM_ERR_ARG, "mask_embed_add: null mask");      // (bytes: any alignment)
    if (int rc = check_ptrs("mask_embed_add", {w_embed, v}, {})) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    const long long blocks = (long long)BT * h * w;
    if (blocks > 0x7fffffffLL) return gdkvm_fail(GDKVM_ERR_SHAPE, "mask_embed_add: too many tokens");
    EmbedArgs a{mask, w_embed, v, H, W, h, w, C};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL(mask_embed_add_kernel<GDKVM_F32>, dim3((unsigned)blocks), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(mask_embed_add_kernel<GDKVM_BF16>, dim3((unsigned)blocks), dim3(64), 0, st, a);
    GDKVM_LAUNCH_CHECK("mask_embed_add");
    return GDKVM_OK;
}
