// gates.hip -- the two gate logits of the memory path from the stride-16 pixel feature in ONE pass:
//     beta_logit[f, n, h]  = <p[f, n, :], w_gate[h, :]>  + b_gate[h]                 (per token: write strength)
//     alpha_logit[f, h]    = <mean_n p[f, n, :], w_decay[h, :]> + b_decay[h]         (per frame: state decay)
// As framework ops this is a token-mean reduction, two N = 1 GEMMs, two bf16 -> fp32 casts and a bias add: six launches,
// ~34 us of a 1.4 ms forward, for 6.4 MB of input.  Here a workgroup reads its frame's tokens once (16-byte loads, fp32
// accumulation, no intermediate rounding) and writes fp32 logits -- the dtype gdkvm_scan_prep reads.
#include "gdkvm_common.hpp"

namespace {

template <int IO>
__global__ __launch_bounds__(256) void gate_logits_kernel(const void* p, const float* w_gate, const float* b_gate,
                                                          const float* w_decay, const float* b_decay, float* beta, float* alpha,
                                                          int N, int Cp, int Hh)
{
    constexpr int V = IO == GDKVM_F32 ? 4 : 8;
    __shared__ float s_part[8][8];                          // [wave][head]: partial sums of the per-token decay dots
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = Cp / V;                                   // lanes per token (a power of two <= 64)
    const int tpw = 64 / G, sub = lane / G, cg = lane % G;  // tokens per wave instruction
    const uint4* pv = static_cast<const uint4*>(p) + (size_t)f * N * G;
    for (int h = 0; h < Hh; ++h) {
        float wg[V], wd[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { wg[j] = w_gate[(size_t)h * Cp + cg * V + j]; wd[j] = w_decay[(size_t)h * Cp + cg * V + j]; }
        float dsum = 0.f;
        constexpr int UNR = 4;                              // independent loads in flight per lane (a frame is a few KB: latency-bound)
        for (int n0 = wv * tpw; n0 < N; n0 += 4 * tpw * UNR) {
            uint4 x[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) x[u] = pv[(size_t)min(n0 + 4 * tpw * u + sub, N - 1) * G + cg];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int n = n0 + 4 * tpw * u + sub;
                const unsigned xw[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
                float v[V];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (IO == GDKVM_F32) v[j] = __uint_as_float(xw[j]);
                    else { v[2 * j] = __uint_as_float(xw[j] << 16); v[2 * j + 1] = __uint_as_float(xw[j] & 0xffff0000u); }
                }
                float dg = 0.f, dd = 0.f;
#pragma unroll
                for (int j = 0; j < V; ++j) { dg = fmaf(v[j], wg[j], dg); dd = fmaf(v[j], wd[j], dd); }
                for (int o = G >> 1; o > 0; o >>= 1) { dg += __shfl_xor(dg, o); dd += __shfl_xor(dd, o); }
                if (n < N) {
                    if (cg == 0) beta[((size_t)f * N + n) * Hh + h] = dg + b_gate[h];
                    dsum += cg == 0 ? dd : 0.f;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o);
        if (lane == 0) s_part[wv][h & 7] = dsum;
        __syncthreads();
        if (tid == 0) alpha[(size_t)f * Hh + h] = (s_part[0][h & 7] + s_part[1][h & 7] + s_part[2][h & 7] + s_part[3][h & 7]) / (float)N + b_decay[h];
        __syncthreads();
    }
}

}  // namespace

extern "C" int gdkvm_gate_logits(const void* p, const float* w_gate, const float* b_gate, const float* w_decay, const float* b_decay,
                                 float* beta, float* alpha, int frames, int N, int Cp, int Hh, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "gate_logits: io_dtype=%d", io_dtype);
    const int V = io_dtype == GDKVM_F32 ? 4 : 8;
    const int G = Cp > 0 ? Cp / V : 0;
    if (frames < 0 || N <= 0 || Hh <= 0 || Cp <= 0 || Cp % V || G > 64 || (G & (G - 1)))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "gate_logits: frames=%d N=%d Cp=%d Hh=%d (Cp/%d a power of two <= 64)", frames, N, Cp, Hh, V);
    if (frames == 0) return GDKVM_OK;
    if (!p || !w_gate || !b_gate || !w_decay || !b_decay || !beta || !alpha) return gdkvm_fail(GDKVM_ERR_ARG, "gate_logits: null pointer");
    if (!gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "gate_logits: the feature must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gate_logits_kernel<GDKVM_F32>), dim3(frames), dim3(256), 0, st, p, w_gate, b_gate, w_decay, b_decay, beta, alpha, N, Cp, Hh);
    else hipLaunchKernelGGL((gate_logits_kernel<GDKVM_BF16>), dim3(frames), dim3(256), 0, st, p, w_gate, b_gate, w_decay, b_decay, beta, alpha, N, Cp, Hh);
    GDKVM_LAUNCH_CHECK("gate_logits_kernel");
    return GDKVM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The decoder's head: a 1x1 convolution from C channels to a few classes on the stride-4 map, written as NCHW planes -- the
// layout gdkvm_upsample_argmax_dice and gdkvm_seg_loss_fwd read.  As framework ops: library convolution + bias add + an
// NHWC -> NCHW copy (three launches, ~22 us); here one pass over the feature (the same lanes-per-pixel reduction as above).
namespace {

template <int IO>
__global__ __launch_bounds__(256) void head_logits_kernel(const void* x, const float* w, const float* b, void* out,
                                                          int HW, int C, int ncls, size_t npix)
{
    constexpr int V = IO == GDKVM_F32 ? 4 : 8;
    const int G = C / V, ppw = 64 / G;                       // lanes per pixel, pixels per wave instruction
    const int lane = threadIdx.x & 63, sub = lane / G, cg = lane % G;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (size_t)gridDim.x * 4;
    const uint4* xv = static_cast<const uint4*>(x);
    for (size_t p0 = wave * ppw; p0 < npix; p0 += nwave * ppw) {
        const size_t p = p0 + sub;
        const uint4 v4 = xv[min(p, npix - 1) * G + cg];
        const unsigned xw[4] = {v4.x, v4.y, v4.z, v4.w};
        float v[V];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (IO == GDKVM_F32) v[j] = __uint_as_float(xw[j]);
            else { v[2 * j] = __uint_as_float(xw[j] << 16); v[2 * j + 1] = __uint_as_float(xw[j] & 0xffff0000u); }
        }
        float mine = 0.f;                                    // lane cg keeps class cg's logit (ncls <= G)
        for (int c = 0; c < ncls; ++c) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) d = fmaf(v[j], w[(size_t)c * C + cg * V + j], d);
            for (int o = G >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o);
            if (cg == c) mine = d + b[c];
        }
        if (p < npix && cg < ncls) {
            const size_t n = p / HW, r = p - n * HW;
            store1<IO>(out, (n * ncls + cg) * HW + r, mine);
        }
    }
}

}  // namespace

// Training: the head's backward in one pass over the feature (round 4).  dz [N, classes, H*W] (io dtype, the NCHW planes the loss kernel
// writes), x [N, H*W, C] -> dx (NHWC, io dtype) = sum_c dz_c w[c][:], and per workgroup the partial sums of dW[c][:] = sum_p dz_c x[p][:] and
// db[c] = sum_p dz_c (fp32, fixed order; head_bwd_sum_kernel adds the workgroups' rows in index order).  The library's route -- data
// gradient, a batched GEMM for the weight gradient that accumulates bf16 atomically, a reduction for the bias, their zero-fills and casts --
// was ~100 us of a training step and differed by several bf16 ulps from run to run in this layer's weight gradient.
namespace {

constexpr int HB_MAXC = 8;               // classes (<= lanes per pixel)

template <int IO>
__global__ __launch_bounds__(256) void head_bwd_kernel(const void* x, const void* dz, const float* w, void* dx, float* part,
                                                       int HW, int C, int ncls, size_t npix)
{
    constexpr int V = IO == GDKVM_F32 ? 4 : 8;
    const int G = C / V, ppw = 64 / G;
    const int lane = threadIdx.x & 63, sub = lane / G, cg = lane % G, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + wv, nwave = (size_t)gridDim.x * 4;
    const uint4* xv = static_cast<const uint4*>(x);
    float wr[HB_MAXC][V], aw[HB_MAXC][V], ab = 0.f;        // this lane's weight columns, dW partials, and (lane cg == class) db partial
#pragma unroll
    for (int c = 0; c < HB_MAXC; ++c)
#pragma unroll
        for (int j = 0; j < V; ++j) { wr[c][j] = c < ncls ? w[(size_t)c * C + cg * V + j] : 0.f; aw[c][j] = 0.f; }
    for (size_t p0 = wave * ppw; p0 < npix; p0 += nwave * ppw) {
        const size_t p = p0 + sub;
        const bool live = p < npix;
        const size_t pc = live ? p : npix - 1;
        const uint4 v4 = xv[pc * G + cg];
        const unsigned xw[4] = {v4.x, v4.y, v4.z, v4.w};
        float v[V];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (IO == GDKVM_F32) v[j] = __uint_as_float(xw[j]);
            else { v[2 * j] = __uint_as_float(xw[j] << 16); v[2 * j + 1] = __uint_as_float(xw[j] & 0xffff0000u); }
        }
        const size_t n = pc / HW, r = pc - n * HW;
        float mine = (live && cg < ncls) ? load1<IO>(dz, (n * ncls + cg) * HW + r) : 0.f;       // lane cg fetches class cg's gradient
        ab += mine;
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = 0.f;
#pragma unroll
        for (int c = 0; c < HB_MAXC; ++c) {
            if (c >= ncls) break;
            const float d = __shfl(mine, sub * G + c);       // (0 for pixels past the end)
#pragma unroll
            for (int j = 0; j < V; ++j) { o[j] = fmaf(d, wr[c][j], o[j]); aw[c][j] = fmaf(d, v[j], aw[c][j]); }
        }
        if (live) {
            uint4 q;
            if constexpr (IO == GDKVM_F32) { q.x = __float_as_uint(o[0]); q.y = __float_as_uint(o[1]); q.z = __float_as_uint(o[2]); q.w = __float_as_uint(o[3]); }
            else {
                q.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16); q.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
                q.z = (unsigned)f32_to_bf16(o[4]) | ((unsigned)f32_to_bf16(o[5]) << 16); q.w = (unsigned)f32_to_bf16(o[6]) | ((unsigned)f32_to_bf16(o[7]) << 16);
            }
            static_cast<uint4*>(dx)[p * G + cg] = q;
        }
    }
    // the wave's pixel sub-groups (lanes with equal cg), then the four waves through LDS, in a fixed order
    __shared__ float s_w[4][HB_MAXC][64 * V];            // [wave][class][channel] (C = G V <= 64 V)
    __shared__ float s_b[4][HB_MAXC];
#pragma unroll
    for (int c = 0; c < HB_MAXC; ++c)
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float t = aw[c][j];
            for (int o2 = G; o2 < 64; o2 <<= 1) t += __shfl_xor(t, o2);
            aw[c][j] = t;
        }
    for (int o2 = G; o2 < 64; o2 <<= 1) ab += __shfl_xor(ab, o2);
    if (sub == 0) {
#pragma unroll
        for (int c = 0; c < HB_MAXC; ++c)
#pragma unroll
            for (int j = 0; j < V; ++j) s_w[wv][c][cg * V + j] = aw[c][j];
        if (cg < HB_MAXC) s_b[wv][cg] = cg < ncls ? ab : 0.f;
    }
    __syncthreads();
    float* row = part + (size_t)blockIdx.x * (ncls * C + ncls);
    for (int i = threadIdx.x; i < ncls * C; i += 256) {
        const int c = i / C, k = i - c * C;
        row[i] = (s_w[0][c][k] + s_w[1][c][k]) + (s_w[2][c][k] + s_w[3][c][k]);
    }
    if (threadIdx.x < ncls) row[ncls * C + threadIdx.x] = (s_b[0][threadIdx.x] + s_b[1][threadIdx.x]) + (s_b[2][threadIdx.x] + s_b[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void head_bwd_sum_kernel(const float* part, int nrows, int rowlen, int nw, float* dw, float* db)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rowlen) return;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nrows; b += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] += part[(size_t)(b + j) * rowlen + i];
    for (int j = 0; b < nrows; ++b, ++j) s8[j] += part[(size_t)b * rowlen + i];
    const float t = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    if (i < nw) dw[i] = t; else db[i - nw] = t;
}

}  // namespace

extern "C" size_t gdkvm_head_bwd_workspace_bytes(int C, int ncls)
{
    return C > 0 && ncls > 0 ? (size_t)512 * ((size_t)ncls * C + ncls) * sizeof(float) : 16;
}

extern "C" int gdkvm_head_bwd(const void* x, const void* dz, const float* w, void* dx, float* dw, float* db, void* workspace, size_t workspace_bytes,
                              int N, int H, int W, int C, int ncls, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "head_bwd: io_dtype=%d", io_dtype);
    const int V = io_dtype == GDKVM_F32 ? 4 : 8;
    const int G = C > 0 ? C / V : 0;
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || C % V || C > 512 || G > 64 || (G & (G - 1)) || ncls <= 0 || ncls > G || ncls > HB_MAXC)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "head_bwd: N=%d H=%d W=%d C=%d classes=%d (C/%d a power of two <= 64, C <= 512, classes <= min(C/%d, %d))", N, H, W, C, ncls, V, V, HB_MAXC);
    if (!dw || !db) return gdkvm_fail(GDKVM_ERR_ARG, "head_bwd: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (N == 0) {
        if (int rc = gdkvm_zero_async(dw, (size_t)ncls * C * sizeof(float), st)) return rc;
        return gdkvm_zero_async(db, (size_t)ncls * sizeof(float), st);
    }
    if (!x || !dz || !w || !dx || !workspace) return gdkvm_fail(GDKVM_ERR_ARG, "head_bwd: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(dx) || !gdkvm_aligned16(workspace)) return gdkvm_fail(GDKVM_ERR_ARG, "head_bwd: pointers must be 16-byte aligned");
    if (workspace_bytes < gdkvm_head_bwd_workspace_bytes(C, ncls)) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "head_bwd: workspace too small");
    if (int rc = gdkvm_check_device()) return rc;
    const size_t npix = (size_t)N * H * W;
    const int ppw = 64 / G;
    size_t blocks = (npix + 4 * (size_t)ppw - 1) / (4 * (size_t)ppw);
    if (blocks > 512) blocks = 512;
    float* part = static_cast<float*>(workspace);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((head_bwd_kernel<GDKVM_F32>), dim3((unsigned)blocks), dim3(256), 0, st, x, dz, w, dx, part, H * W, C, ncls, npix);
    else hipLaunchKernelGGL((head_bwd_kernel<GDKVM_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, x, dz, w, dx, part, H * W, C, ncls, npix);
    GDKVM_LAUNCH_CHECK("head_bwd_kernel");
    const int rowlen = ncls * C + ncls;
    hipLaunchKernelGGL(head_bwd_sum_kernel, dim3((rowlen + 255) / 256), dim3(256), 0, st, part, (int)blocks, rowlen, ncls * C, dw, db);
    GDKVM_LAUNCH_CHECK("head_bwd_sum_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_head_logits(const void* x, const float* w, const float* b, void* out, int N, int H, int W, int C, int ncls,
                                 int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "head_logits: io_dtype=%d", io_dtype);
    const int V = io_dtype == GDKVM_F32 ? 4 : 8;
    const int G = C > 0 ? C / V : 0;
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || C % V || G > 64 || (G & (G - 1)) || ncls <= 0 || ncls > G)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "head_logits: N=%d H=%d W=%d C=%d classes=%d (C/%d a power of two <= 64, classes <= C/%d)", N, H, W, C, ncls, V, V);
    if (N == 0) return GDKVM_OK;
    if (!x || !w || !b || !out) return gdkvm_fail(GDKVM_ERR_ARG, "head_logits: null pointer");
    if (!gdkvm_aligned16(x)) return gdkvm_fail(GDKVM_ERR_ARG, "head_logits: the feature must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    const size_t npix = (size_t)N * H * W;
    const int ppw = 64 / G;
    size_t blocks = (npix + 4 * (size_t)ppw - 1) / (4 * (size_t)ppw);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((head_logits_kernel<GDKVM_F32>), dim3((unsigned)blocks), dim3(256), 0, st, x, w, b, out, H * W, C, ncls, npix);
    else hipLaunchKernelGGL((head_logits_kernel<GDKVM_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, x, w, b, out, H * W, C, ncls, npix);
    GDKVM_LAUNCH_CHECK("head_logits_kernel");
    return GDKVM_OK;
}
