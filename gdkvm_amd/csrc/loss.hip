// loss.hip -- the training objective on the decoder's stride-4 logits without materialising full-resolution logits
// (the training twin of upsample_argmax_dice_kernel):
//     l = bilinear_upsample(z -> H x W, align_corners = false)          (fp32, PyTorch's source-index formula)
//     loss = mean_pixels CE(l, t) + dice_weight * (1 - mean_c (2 I_c + eps) / (P_c + O_c + eps))
//     with p = softmax(l),  I_c = sum p_c [t = c],  P_c = sum p_c,  O_c = sum [t = c]   (sums over the whole batch)
// In a cfg4 training step PyTorch spends ~0.75 ms on this chain (a 51 MB fp32 logit tensor written by an NCHW upsample kernel,
// log-softmax + softmax + one-hot + reductions and their backward).  Here: one pass over the pixels producing per-workgroup
// partial sums, a one-workgroup finalize (loss + the per-class coefficients of the backward), and a backward that is a
// GATHER per stride-4 pixel over its bilinear footprint, recomputing softmax on the fly -- deterministic, no atomics.
#include <type_traits>

#include "gdkvm_common.hpp"

namespace {

constexpr int LOSS_MAX_PART = 2048;
constexpr int LOSS_MAXC = 8;

struct Taps { int i0, i1; float l; };
__device__ __forceinline__ Taps taps_of(int o, float s, int n_in)
{
    const float f = fmaxf(s * ((float)o + 0.5f) - 0.5f, 0.f);
    Taps t;
    t.i0 = (int)f; t.i1 = min(t.i0 + 1, n_in - 1); t.l = f - (float)t.i0;
    return t;
}

// softmax of the upsampled logits of one pixel; returns log(sum exp) + max
template <int IO, int C>
__device__ __forceinline__ float pixel_softmax(const void* z, size_t img, int hw, int w, const Taps& ty, const Taps& tx, float (&p)[C],
                                               float (&lg)[C])
{
    const float hy = 1.f - ty.l, hx = 1.f - tx.l;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t pl = img + (size_t)c * hw;
        const float a = load1<IO>(z, pl + ty.i0 * w + tx.i0), b = load1<IO>(z, pl + ty.i0 * w + tx.i1);
        const float d = load1<IO>(z, pl + ty.i1 * w + tx.i0), e = load1<IO>(z, pl + ty.i1 * w + tx.i1);
        lg[c] = hy * (hx * a + tx.l * b) + ty.l * (hx * d + tx.l * e);
        m = fmaxf(m, lg[c]);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = expf(lg[c] - m); s += p[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
    return m + logf(s);
}

template <int TL>
__device__ __forceinline__ int target_at(const void* t, size_t i)
{
    if constexpr (TL == 8) return (int)static_cast<const long long*>(t)[i];
    else return (int)static_cast<const unsigned char*>(t)[i];
}

// part[block][1 + 3C]: ce sum, then I_c, P_c, O_c
template <int IO, int C, int TL>
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const void* z, const void* target, float* part,
                                                           int NI, int h, int w, int H, int W, float sy, float sx)
{
    constexpr int NV = 1 + 3 * C;
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    const size_t total = (size_t)NI * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const size_t r = i / W;
        const int y = (int)(r % H), n = (int)(r / H);
        const Taps ty = taps_of(y, sy, h), tx = taps_of(x, sx, w);
        float p[C], lg[C];
        const float lse = pixel_softmax<IO, C>(z, (size_t)n * C * h * w, h * w, w, ty, tx, p, lg);
        const int t = target_at<TL>(target, i);
        const float lab = (unsigned)t < (unsigned)C ? 1.f : 0.f;        // an unlabelled pixel (EchoNet: every untraced frame) adds to NO sum
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float is = t == c ? 1.f : 0.f;
            acc[0] += is * (lse - lg[c]);
            acc[1 + c] += is * p[c];
            acc[1 + C + c] += lab * p[c];
            acc[1 + 2 * C + c] += is;
        }
    }
    __shared__ float s[4][NV];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NV) part[(size_t)blockIdx.x * NV + threadIdx.x] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

// out[0] = loss, out[1] = CE, out[2] = Dice term;  coef[0..C) = dLoss/dI_c, coef[C..2C) = dLoss/dP_c, coef[2C] = 1 / pixels
__global__ __launch_bounds__(256) void seg_loss_finalize_kernel(const float* part, int nblk, int C, double npix, float dice_weight,
                                                                float eps, float* out, float* coef)
{
    __shared__ double s[256];
    __shared__ double tot[1 + 3 * LOSS_MAXC];
    const int NV = 1 + 3 * C;
    for (int k = 0; k < NV; ++k) {
        double a = 0.0;
        for (int b = threadIdx.x; b < nblk; b += 256) a += (double)part[(size_t)b * NV + k];
        s[threadIdx.x] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) tot[k] = s[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // labels outside [0, C) carry no class: the cross-entropy averages over the labelled pixels only (ignore_index semantics)
        double labelled = 0.0;
        for (int c = 0; c < C; ++c) labelled += tot[1 + 2 * C + c];
        npix = labelled > 0.0 ? labelled : 1.0;
        const double ce = tot[0] / npix;
        double dsum = 0.0;
        for (int c = 0; c < C; ++c) {
            const double I = tot[1 + c], D = tot[1 + C + c] + tot[1 + 2 * C + c] + (double)eps;
            dsum += (2.0 * I + (double)eps) / D;
            coef[c] = (float)(-(double)dice_weight / C * 2.0 / D);
            coef[C + c] = (float)((double)dice_weight / C * (2.0 * I + (double)eps) / (D * D));
        }
        const double dice = 1.0 - dsum / C;
        coef[2 * C] = (float)(1.0 / npix);
        out[0] = (float)(ce + (double)dice_weight * dice); out[1] = (float)ce; out[2] = (float)dice;
    }
}

// dz[n, c, i, j] = gout * sum over the pixels (y, x) whose taps touch (i, j) of  wy wx dL/dl_c(y, x),
//   dL/dl_c = (p_c - [t = c]) / pixels + p_c (u_c - sum_k p_k u_k),   u_c = coefI_c [t = c] + coefP_c.
template <int IO, int C, int TL>
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const void* z, const void* target, const float* coef, const float* gout,
                                                           void* dz, int NI, int h, int w, int H, int W, float sy, float sx)
{
    const size_t total = (size_t)NI * h * w;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % w);
    const size_t r = i / w;
    const int ii = (int)(r % h), n = (int)(r / h);
    float cI[C], cP[C], acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { cI[c] = coef[c]; cP[c] = coef[C + c]; acc[c] = 0.f; }
    const float inv_n = coef[2 * C];
    const int y0 = max((int)(((float)ii - 0.5f) / sy - 0.5f) - 1, 0), y1 = min((int)(((float)ii + 1.5f) / sy - 0.5f) + 1, H - 1);
    const int x0 = max((int)(((float)j - 0.5f) / sx - 0.5f) - 1, 0), x1 = min((int)(((float)j + 1.5f) / sx - 0.5f) + 1, W - 1);
    const size_t img = (size_t)n * C * h * w;
    auto pixel = [&](int y, int x, const Taps& ty, const Taps& tx, float wgt) __attribute__((always_inline)) {
        float p[C], lg[C];
        pixel_softmax<IO, C>(z, img, h * w, w, ty, tx, p, lg);
        const int t = target_at<TL>(target, ((size_t)n * H + y) * W + x);
        float u[C], dot = 0.f;
        const bool lab = (unsigned)t < (unsigned)C;
        const float ce_w = lab ? inv_n : 0.f;                         // unlabelled pixel: no cross-entropy term, and no Dice term either
#pragma unroll
        for (int c = 0; c < C; ++c) { u[c] = lab ? (t == c ? cI[c] : 0.f) + cP[c] : 0.f; dot = fmaf(p[c], u[c], dot); }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float dl = (p[c] - (t == c ? 1.f : 0.f)) * ce_w + p[c] * (u[c] - dot);
            acc[c] = fmaf(wgt, dl, acc[c]);
        }
    };
    if (H == 4 * h && W == 4 * w) {
        // exactly 4x (EchoNet 28 -> 112, CAMUS 64 -> 256): the footprint is rows 4i-2 .. 4i+5 and columns 4j-2 .. 4j+5; a fixed
        // 8-wide column loop puts the eight pixels' loads in flight together (the general loop below is one dependent chain
        // per pixel: 118 us per cfg4 step against 0.4 M threads x 64 pixels of arithmetic)
        for (int a = 0; a < 8; ++a) {
            const int y = 4 * ii - 2 + a;
            if (y < 0 || y >= H) continue;
            const Taps ty = taps_of(y, sy, h);
            const float wy = (ty.i0 == ii ? 1.f - ty.l : 0.f) + (ty.i1 == ii ? ty.l : 0.f);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int xr = 4 * j - 2 + b, x = min(max(xr, 0), W - 1);
                const Taps tx = taps_of(x, sx, w);
                const float wx = (tx.i0 == j ? 1.f - tx.l : 0.f) + (tx.i1 == j ? tx.l : 0.f);
                pixel(y, x, ty, tx, xr == x ? wy * wx : 0.f);
            }
        }
    } else
    for (int y = y0; y <= y1; ++y) {
        const Taps ty = taps_of(y, sy, h);
        const float wy = (ty.i0 == ii ? 1.f - ty.l : 0.f) + (ty.i1 == ii ? ty.l : 0.f);
        if (wy == 0.f) continue;
        for (int x = x0; x <= x1; ++x) {
            const Taps tx = taps_of(x, sx, w);
            const float wgt = wy * ((tx.i0 == j ? 1.f - tx.l : 0.f) + (tx.i1 == j ? tx.l : 0.f));
            if (wgt == 0.f) continue;
            pixel(y, x, ty, tx, wgt);
        }
    }
    const float g = gout ? gout[0] : 1.f;
#pragma unroll
    for (int c = 0; c < C; ++c) store1<IO>(dz, img + (size_t)c * h * w + (size_t)ii * w + j, g * acc[c]);
}

int loss_check(const char* who, int NI, int C, int h, int w, int H, int W, int io, int tl)
{
    if (NI <= 0 || C < 2 || C > LOSS_MAXC || h <= 0 || w <= 0 || H <= 0 || W <= 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: images=%d classes=%d (2..%d) %dx%d -> %dx%d", who, NI, C, LOSS_MAXC, h, w, H, W);
    if (io != GDKVM_F32 && io != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: io_dtype=%d", who, io);
    if (tl != 1 && tl != 8) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: target_bytes=%d (1 = uint8, 8 = int64)", who, tl);
    return gdkvm_check_device();
}

template <class F>
int dispatch_c(int C, F&& f)
{
    switch (C) {
        case 2: f(std::integral_constant<int, 2>{}); return 1;
        case 3: f(std::integral_constant<int, 3>{}); return 1;
        case 4: f(std::integral_constant<int, 4>{}); return 1;
        case 5: f(std::integral_constant<int, 5>{}); return 1;
        case 6: f(std::integral_constant<int, 6>{}); return 1;
        case 7: f(std::integral_constant<int, 7>{}); return 1;
        case 8: f(std::integral_constant<int, 8>{}); return 1;
    }
    return 0;
}

}  // namespace

extern "C" size_t gdkvm_seg_loss_workspace_bytes(int C)
{
    return C > 0 ? ((size_t)LOSS_MAX_PART * (1 + 3 * C) + 2 * (size_t)C + 4) * sizeof(float) : 0;
}

extern "C" int gdkvm_seg_loss_fwd(const void* z, const void* target, float* out, void* ws, size_t ws_bytes,
                                  int NI, int C, int h, int w, int H, int W, float dice_weight, float eps,
                                  int io_dtype, int target_bytes, void* stream)
{
    if (int rc = loss_check("seg_loss_fwd", NI, C, h, w, H, W, io_dtype, target_bytes)) return rc;
    if (!z || !target || !out || !ws) return gdkvm_fail(GDKVM_ERR_ARG, "seg_loss_fwd: null pointer");
    if (ws_bytes < gdkvm_seg_loss_workspace_bytes(C)) return gdkvm_fail(GDKVM_ERR_ARG, "seg_loss_fwd: workspace too small");
    const size_t total = (size_t)NI * H * W;
    size_t blocks = (total + 255) / 256;
    if (blocks > LOSS_MAX_PART) blocks = LOSS_MAX_PART;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    float* coef = part + (size_t)LOSS_MAX_PART * (1 + 3 * C);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    dispatch_c(C, [&](auto Cc) {
        constexpr int CC = decltype(Cc)::value;
#define GDKVM_LF(IO, TL) hipLaunchKernelGGL((seg_loss_fwd_kernel<IO, CC, TL>), dim3((unsigned)blocks), dim3(256), 0, st, z, target, part, NI, h, w, H, W, sy, sx)
        if (io_dtype == GDKVM_F32) { if (target_bytes == 8) GDKVM_LF(GDKVM_F32, 8); else GDKVM_LF(GDKVM_F32, 1); }
        else { if (target_bytes == 8) GDKVM_LF(GDKVM_BF16, 8); else GDKVM_LF(GDKVM_BF16, 1); }
#undef GDKVM_LF
    });
    GDKVM_LAUNCH_CHECK("seg_loss_fwd_kernel");
    hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(256), 0, st, part, (int)blocks, C, (double)total, dice_weight, eps, out, coef);
    GDKVM_LAUNCH_CHECK("seg_loss_finalize_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_seg_loss_bwd(const void* z, const void* target, const void* ws, size_t ws_bytes, const float* grad_out, void* dz,
                                  int NI, int C, int h, int w, int H, int W, int io_dtype, int target_bytes, void* stream)
{
    if (int rc = loss_check("seg_loss_bwd", NI, C, h, w, H, W, io_dtype, target_bytes)) return rc;
    if (!z || !target || !dz || !ws) return gdkvm_fail(GDKVM_ERR_ARG, "seg_loss_bwd: null pointer");
    if (ws_bytes < gdkvm_seg_loss_workspace_bytes(C)) return gdkvm_fail(GDKVM_ERR_ARG, "seg_loss_bwd: workspace too small");
    const size_t total = (size_t)NI * h * w;
    const size_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffu) return gdkvm_fail(GDKVM_ERR_SHAPE, "seg_loss_bwd: too many pixels");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* coef = static_cast<const float*>(ws) + (size_t)LOSS_MAX_PART * (1 + 3 * C);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    dispatch_c(C, [&](auto Cc) {
        constexpr int CC = decltype(Cc)::value;
#define GDKVM_LB(IO, TL) hipLaunchKernelGGL((seg_loss_bwd_kernel<IO, CC, TL>), dim3((unsigned)blocks), dim3(256), 0, st, z, target, coef, grad_out, dz, NI, h, w, H, W, sy, sx)
        if (io_dtype == GDKVM_F32) { if (target_bytes == 8) GDKVM_LB(GDKVM_F32, 8); else GDKVM_LB(GDKVM_F32, 1); }
        else { if (target_bytes == 8) GDKVM_LB(GDKVM_BF16, 8); else GDKVM_LB(GDKVM_BF16, 1); }
#undef GDKVM_LB
    });
    GDKVM_LAUNCH_CHECK("seg_loss_bwd_kernel");
    return GDKVM_OK;
}
