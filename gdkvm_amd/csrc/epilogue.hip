// epilogue.hip -- pointwise passes around the encoder/decoder convolutions: the fused epilogue
// y = act(x + bias[c] (+ residual)), NHWC, in place or out of place -- what a LIBRARY convolution is followed by (the training
// build's strided / 1x1 / stem layers, odd channel counts; the hand-written convolution kernels carry this epilogue inside) --
// plus upsample(+concat), the stem's bias + ReLU + max-pool and space-to-depth passes, and the max-pool pair of the training build.
// After BatchNorm folding every conv carries a bias; PyTorch then runs the bias add, the residual add and the ReLU as
// three separate full-tensor passes (1.1 ms of a 3.6 ms cfg2 forward).  This is one pass: 16-byte loads/stores,
// HBM-bound by construction.
#include <type_traits>

#include <stdlib.h>
#include "gdkvm_common.hpp"

namespace {

// Each thread owns UNR 16-byte vectors per trip, all loaded before any is used (a streaming pass needs ~60 KB in flight
// per CU to cover HBM latency; one vector per thread in flight left the first version at 37 % of the HBM rate).  When the
// grid stride is a multiple of the channel count the thread's channels never change and the bias lives in registers.
template <int IO, bool FIXED>
__global__ __launch_bounds__(256) void bias_act_kernel(const void* x, const float* bias, const void* res, void* y,
                                                       size_t nvec, int C, int relu)
{
    constexpr int V = IO == GDKVM_F32 ? 4 : 8;            // elements per 16-byte access
    constexpr int UNR = 4;
    typedef uint4 vec_t;
    const vec_t* xv = static_cast<const vec_t*>(x);
    const vec_t* rv = static_cast<const vec_t*>(res);
    vec_t* yv = static_cast<vec_t*>(y);
    const size_t stride = (size_t)gridDim.x * 256;
    float bfix[V];
    if constexpr (FIXED) {
        const int c0 = (int)((((size_t)blockIdx.x * 256 + threadIdx.x) * V) % (size_t)C);
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c0 + j);
            bfix[j] = b4[0]; bfix[j + 1] = b4[1]; bfix[j + 2] = b4[2]; bfix[j + 3] = b4[3];
        }
    }
    auto finish = [&](size_t i, const vec_t& a, const vec_t& r) __attribute__((always_inline)) {
        float v[V];
        const unsigned aw[4] = {a.x, a.y, a.z, a.w}, rw[4] = {r.x, r.y, r.z, r.w};
        if constexpr (IO == GDKVM_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(aw[j]) + (res ? __uint_as_float(rw[j]) : 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] = __uint_as_float(aw[j] << 16) + (res ? __uint_as_float(rw[j] << 16) : 0.f);
                v[2 * j + 1] = __uint_as_float(aw[j] & 0xffff0000u) + (res ? __uint_as_float(rw[j] & 0xffff0000u) : 0.f);
            }
        }
        float bb[V];
        if constexpr (FIXED) {
#pragma unroll
            for (int j = 0; j < V; ++j) bb[j] = bfix[j];
        } else {
            const int c0 = (int)((i * V) % (size_t)C);
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c0 + j);
                bb[j] = b4[0]; bb[j + 1] = b4[1]; bb[j + 2] = b4[2]; bb[j + 3] = b4[3];
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            v[j] += bb[j];
            if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        vec_t o;
        if constexpr (IO == GDKVM_F32) {
            o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]); o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
        } else {
            o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        }
        yv[i] = o;
    };
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNR - 1) * stride < nvec; i += UNR * stride) {
        vec_t a[UNR], r[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) a[u] = xv[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) r[u] = res ? rv[i + u * stride] : vec_t{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNR; ++u) finish(i + u * stride, a[u], r[u]);
    }
    for (; i < nvec; i += stride) finish(i, xv[i], res ? rv[i] : vec_t{0, 0, 0, 0});
}

// Stem tail: y = relu(max over the 3x3 / stride 2 / pad 1 window of x + bias).  One thread per (output pixel, 16-byte channel
// group): its nine window loads are independent and all in flight; neighbouring windows overlap in L1/L2, HBM sees x once.
template <int IO>
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(const void* x, const float* bias, void* y,
                                                                int N, int H, int W, int C, int Ho, int Wo, int Hp, int Wp)
{
    constexpr int V = IO == GDKVM_F32 ? 4 : 8;
    const int cg = C / V;
    const size_t total = (size_t)N * Ho * Wo * cg;
    const uint4* xv = static_cast<const uint4*>(x);
    uint4* yv = static_cast<uint4*>(y);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cg);
        size_t p = i / cg;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        uint4 win[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {                // out-of-range taps re-read the (always valid) centre: max unchanged
                const int ih = 2 * oh - 1 + dy, iw = 2 * ow - 1 + dx;
                const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
                const int hh = ok ? ih : 2 * oh, ww = ok ? iw : 2 * ow;
                win[3 * dy + dx] = xv[(((size_t)n * Hp + hh) * Wp + ww) * cg + c];     // Hp x Wp: the rows / columns x is stored with
            }
        float m[V];
#pragma unroll
        for (int j = 0; j < V; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned w4[4] = {win[t].x, win[t].y, win[t].z, win[t].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (IO == GDKVM_F32) m[j] = fmaxf(m[j], __uint_as_float(w4[j]));
                else {
                    m[2 * j] = fmaxf(m[2 * j], __uint_as_float(w4[j] << 16));
                    m[2 * j + 1] = fmaxf(m[2 * j + 1], __uint_as_float(w4[j] & 0xffff0000u));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c * V + j);
#pragma unroll
            for (int q = 0; q < 4; ++q) m[j + q] = fmaxf(m[j + q] + b4[q], 0.f);
        }
        uint4 o;
        if constexpr (IO == GDKVM_F32) {
            o.x = __float_as_uint(m[0]); o.y = __float_as_uint(m[1]); o.z = __float_as_uint(m[2]); o.w = __float_as_uint(m[3]);
        } else {
            o.x = (unsigned)f32_to_bf16(m[0]) | ((unsigned)f32_to_bf16(m[1]) << 16);
            o.y = (unsigned)f32_to_bf16(m[2]) | ((unsigned)f32_to_bf16(m[3]) << 16);
            o.z = (unsigned)f32_to_bf16(m[4]) | ((unsigned)f32_to_bf16(m[5]) << 16);
            o.w = (unsigned)f32_to_bf16(m[6]) | ((unsigned)f32_to_bf16(m[7]) << 16);
        }
        yv[i] = o;
    }
}

// Stem input in space-to-depth form: a 7x7 / stride 2 convolution on C channels is a 4x4 / stride 1 convolution on the
// 4C channels (c, row parity, column parity) of the half-resolution image.  MIOpen runs the 3-channel 7x7 stem at 220 us
// (implicit-GEMM K = 147, plus a zero-fill pass for its atomics) and the 16-channel 4x4 form at 100 us.  This kernel builds
// that input from the NCHW frames in the one pass that used to be the channels_last copy:
//     out[n, i, j, (c*2 + p)*2 + q] = x[n, c, 2i + p, 2j + q],  channels >= 4C zero;  out is NHWC with Cp channels.
template <int IO>
__global__ __launch_bounds__(256) void stem_s2d_kernel(const void* x, void* out, int N, int C, int H, int W, int Cp)
{
    typedef typename std::conditional<IO == GDKVM_F32, float, bf16_t>::type T;
    const T* xi = static_cast<const T*>(x);
    T* o = static_cast<T*>(out);
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i % Wo);
        const size_t r = i / Wo;
        const int ih = (int)(r % Ho), n = (int)(r / Ho);
        T* dst = o + i * Cp;
        if constexpr (IO == GDKVM_BF16) {
            if (Cp == 16 && C <= 4) {                      // the stem's case: six 4-byte loads, two 16-byte stores per pixel
                unsigned pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // channel pairs (c, p): [x(2i+p, 2j), x(2i+p, 2j+1)]
                for (int c = 0; c < C; ++c) {
                    const bf16_t* src = xi + (((size_t)n * C + c) * H + 2 * ih) * W + 2 * j;
                    pr[2 * c] = *reinterpret_cast<const unsigned*>(src);
                    pr[2 * c + 1] = *reinterpret_cast<const unsigned*>(src + W);
                }
                reinterpret_cast<uint4*>(dst)[0] = make_uint4(pr[0], pr[1], pr[2], pr[3]);
                reinterpret_cast<uint4*>(dst)[1] = make_uint4(pr[4], pr[5], pr[6], pr[7]);
                continue;
            }
        }
        for (int c = 0; c < C; ++c) {
            const T* src = xi + (((size_t)n * C + c) * H + 2 * ih) * W + 2 * j;
            dst[4 * c + 0] = src[0]; dst[4 * c + 1] = src[1]; dst[4 * c + 2] = src[W]; dst[4 * c + 3] = src[W + 1];
        }
        for (int c = 4 * C; c < Cp; ++c) dst[c] = T(0);
    }
}

}  // namespace

extern "C" int gdkvm_bias_act(const void* x, const float* bias, const void* residual, void* y,
                              size_t rows, int C, int relu, int io_dtype, void* stream)
{
    if (C <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "bias_act: C=%d", C);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "bias_act: io_dtype=%d", io_dtype);
    const int V = io_dtype == GDKVM_F32 ? 4 : 8;
    if (C % V) return gdkvm_fail(GDKVM_ERR_SHAPE, "bias_act: C=%d must be a multiple of %d", C, V);
    if (rows == 0) return GDKVM_OK;
    if (!x || !bias || !y) return gdkvm_fail(GDKVM_ERR_ARG, "bias_act: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(y) || !gdkvm_aligned16(bias) || (residual && !gdkvm_aligned16(residual)))
        return gdkvm_fail(GDKVM_ERR_ARG, "bias_act: pointers must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    const size_t nvec = rows * (size_t)C / V;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;               // ~8 blocks per CU, grid-stride the rest
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fixed = (blocks * 256) % (size_t)(C / V) == 0;     // every thread keeps its channels: bias in registers
#define GDKVM_BA(IO, FX) hipLaunchKernelGGL((bias_act_kernel<IO, FX>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias, residual, y, nvec, C, relu)
    if (io_dtype == GDKVM_F32) { if (fixed) GDKVM_BA(GDKVM_F32, true); else GDKVM_BA(GDKVM_F32, false); }
    else { if (fixed) GDKVM_BA(GDKVM_BF16, true); else GDKVM_BA(GDKVM_BF16, false); }
#undef GDKVM_BA
    GDKVM_LAUNCH_CHECK("bias_act_kernel");
    return GDKVM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Decoder glue of the inference build: out[n, y, x, :] = [ bilinear(lo)[n, y, x, :C1] ; skip[n, y, x, :C2] ]  (NHWC).
// PyTorch runs this as an upsample kernel that writes the enlarged map plus a concat kernel that re-reads and re-writes
// everything; here the enlarged map is never written on its own.  align_corners = false, PyTorch's source-index formula,
// blend in fp32.  One thread = 8 channels (16 bytes) of one output pixel.
namespace {

__device__ __forceinline__ void unpack8(const uint4& a, float (&v)[8])
{
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
}

// One workgroup per output row (image n, row y), grid-striding over rows: the vertical taps and weights are uniform per row
// and every index is 32-bit (the first version spent its time in 64-bit div/mod per vector and ran at 48 % of the HBM rate).
// Per row a thread handles its share of the W*C1/8 interpolated vectors (four independent tap loads each) and of the W*C2/8
// copied vectors; all loads of a row are issued before the first blend.
__global__ __launch_bounds__(256) void upsample_cat_bf16_kernel(const bf16_t* lo, const bf16_t* skip, bf16_t* out,
                                                                int Nimg, int hl, int wl, int H, int W, int C1, int C2,
                                                                float sy, float sx, float inv_c1n, float inv_c2n, float inv_h)
{
    // floor(n / d) for the small non-negative indices below (n < 2^20), inv = 1.0f / d: exact, 3 instructions instead of ~40 for
    // an integer division by a run-time divisor -- there were four to six of those per thread and row, as many as useful work
    auto qdiv = [](int n, float inv) { return (int)(((float)n + 0.5f) * inv); };
    const int C = C1 + C2, c1n = C1 / 8, c2n = C2 / 8, n1 = W * c1n, n2 = W * c2n, tid = threadIdx.x;
    constexpr int MAXV = 2;                                // interpolated vectors per thread per trip (W*C1/8 <= 512 in one trip)
    const bool exact2x = H == 2 * hl && W == 2 * wl;
    for (int row = blockIdx.x; row < Nimg * H; row += gridDim.x) {
        const int n = qdiv(row, inv_h), y = row - n * H;
        const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)fy, y1 = min(y0 + 1, hl - 1);
        const float ly = fy - (float)y0, hy = 1.f - ly;
        const bf16_t* lo0 = lo + ((size_t)n * hl + y0) * wl * C1;
        const bf16_t* lo1 = lo + ((size_t)n * hl + y1) * wl * C1;
        const bf16_t* sk = skip + (size_t)row * W * C2;
        bf16_t* orow = out + (size_t)row * W * C;
        uint4 cp[MAXV];                                    // this thread's first copy vectors, in flight behind the taps
#pragma unroll
        for (int u = 0; u < MAXV; ++u) cp[u] = n2 ? *reinterpret_cast<const uint4*>(sk + min(u * 256 + tid, n2 - 1) * 8) : make_uint4(0u, 0u, 0u, 0u);
        if (exact2x) {
            // exactly 2x: output columns 2jp+1 and 2jp+2 share the horizontal taps jp and jp+1 (weights 1/4 and 3/4), so a
            // thread owns one such pair per trip -- four tap loads for two outputs; jp = -1 and jp = wl-1 are the borders
            const int nu = (wl + 1) * c1n;
            for (int u0 = 0; u0 < nu; u0 += 256) {
                const int u = min(u0 + tid, nu - 1);
                const int jq = qdiv(u, inv_c1n), c = (u - jq * c1n) * 8, jp = jq - 1;
                const int x0 = max(jp, 0), x1 = min(jp + 1, wl - 1);
                const uint4 t00 = *reinterpret_cast<const uint4*>(lo0 + x0 * C1 + c), t01 = *reinterpret_cast<const uint4*>(lo0 + x1 * C1 + c);
                const uint4 t10 = *reinterpret_cast<const uint4*>(lo1 + x0 * C1 + c), t11 = *reinterpret_cast<const uint4*>(lo1 + x1 * C1 + c);
                if (u0 + tid >= nu) continue;
                float a[8], b[8], cc[8], d[8];
                unpack8(t00, a); unpack8(t01, b); unpack8(t10, cc); unpack8(t11, d);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int x = 2 * jp + 1 + e;
                    if (x < 0 || x >= W) continue;
                    const float lx = e ? 0.75f : 0.25f, hx = 1.f - lx;
                    unsigned r[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v0 = hy * (hx * a[2 * q] + lx * b[2 * q]) + ly * (hx * cc[2 * q] + lx * d[2 * q]);
                        const float v1 = hy * (hx * a[2 * q + 1] + lx * b[2 * q + 1]) + ly * (hx * cc[2 * q + 1] + lx * d[2 * q + 1]);
                        r[q] = (unsigned)f32_to_bf16(v0) | ((unsigned)f32_to_bf16(v1) << 16);
                    }
                    *reinterpret_cast<uint4*>(orow + x * C + c) = make_uint4(r[0], r[1], r[2], r[3]);
                }
            }
        } else
        for (int j0 = 0; j0 < n1; j0 += MAXV * 256) {
            uint4 t[MAXV][4];
            float lx[MAXV];
            int oidx[MAXV];
#pragma unroll
            for (int u = 0; u < MAXV; ++u) {
                const int j = min(j0 + u * 256 + tid, n1 - 1);
                const int x = qdiv(j, inv_c1n), c = (j - x * c1n) * 8;
                const float fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.f);
                const int x0 = (int)fx, x1 = min(x0 + 1, wl - 1);
                lx[u] = fx - (float)x0;
                oidx[u] = x * C + c;
                t[u][0] = *reinterpret_cast<const uint4*>(lo0 + x0 * C1 + c);
                t[u][1] = *reinterpret_cast<const uint4*>(lo0 + x1 * C1 + c);
                t[u][2] = *reinterpret_cast<const uint4*>(lo1 + x0 * C1 + c);
                t[u][3] = *reinterpret_cast<const uint4*>(lo1 + x1 * C1 + c);
            }
#pragma unroll
            for (int u = 0; u < MAXV; ++u) {
                if (j0 + u * 256 + tid >= n1) continue;
                const float hx = 1.f - lx[u];
                float a[8], b[8], cc[8], d[8];
                unpack8(t[u][0], a); unpack8(t[u][1], b); unpack8(t[u][2], cc); unpack8(t[u][3], d);
                unsigned r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v0 = hy * (hx * a[2 * q] + lx[u] * b[2 * q]) + ly * (hx * cc[2 * q] + lx[u] * d[2 * q]);
                    const float v1 = hy * (hx * a[2 * q + 1] + lx[u] * b[2 * q + 1]) + ly * (hx * cc[2 * q + 1] + lx[u] * d[2 * q + 1]);
                    r[q] = (unsigned)f32_to_bf16(v0) | ((unsigned)f32_to_bf16(v1) << 16);
                }
                *reinterpret_cast<uint4*>(orow + oidx[u]) = make_uint4(r[0], r[1], r[2], r[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            const int jc = u * 256 + tid;
            if (jc < n2) {
                const int x = qdiv(jc, inv_c2n);
                *reinterpret_cast<uint4*>(orow + x * C + C1 + (jc - x * c2n) * 8) = cp[u];
            }
        }
        for (int jc = MAXV * 256 + tid; jc < n2; jc += 256) {                 // wider rows: the rest of the copy half
            const int x = qdiv(jc, inv_c2n);
            *reinterpret_cast<uint4*>(orow + x * C + C1 + (jc - x * c2n) * 8) = *reinterpret_cast<const uint4*>(sk + jc * 8);
        }
    }
}

// Exactly 2x (the decoder's two enlargements): one workgroup iteration per PAIR of output rows 2ip+1, 2ip+2 -- they share the vertical taps
// ip, ip+1 (weights 1/4 and 3/4) as the column pairs of upsample_cat_bf16_kernel share the horizontal ones -- so a thread's four tap loads
// feed a 2 x 2 block of outputs (round 4: the row-at-a-time form issued four loads per two outputs and a workgroup iteration moved 7 KB;
// 49 us for the two enlargements of a cfg2 forward).  The arithmetic per output is that kernel's expression, so the bits are the same.
__global__ __launch_bounds__(256) void upsample_cat_bf16_2x_kernel(const bf16_t* lo, const bf16_t* skip, bf16_t* out,
                                                                   int Nimg, int hl, int wl, int C1, int C2, float inv_c1n, float inv_c2n, float inv_hp)
{
    auto qdiv = [](int n, float inv) { return (int)(((float)n + 0.5f) * inv); };
    const int H = 2 * hl, W = 2 * wl, C = C1 + C2, c1n = C1 / 8, c2n = C2 / 8, n2 = W * c2n, tid = threadIdx.x;
    const int nu = (wl + 1) * c1n, hp = hl + 1;           // column pairs x channel vectors; row pairs per image
    for (int rp = blockIdx.x; rp < Nimg * hp; rp += gridDim.x) {
        const int n = qdiv(rp, inv_hp), ip = rp - n * hp - 1;
        const int r0 = max(ip, 0), r1 = min(ip + 1, hl - 1);
        const bf16_t* lo0 = lo + ((size_t)n * hl + r0) * wl * C1;
        const bf16_t* lo1 = lo + ((size_t)n * hl + r1) * wl * C1;
        // the copy half of both rows, requested first (in flight behind the taps)
        constexpr int MAXV = 2;
        uint4 cp[2][MAXV];
#pragma unroll
        for (int ey = 0; ey < 2; ++ey) {
            const int y = min(max(2 * ip + 1 + ey, 0), H - 1);
            const bf16_t* sk = skip + ((size_t)n * H + y) * W * C2;
#pragma unroll
            for (int u = 0; u < MAXV; ++u) cp[ey][u] = n2 ? *reinterpret_cast<const uint4*>(sk + min(u * 256 + tid, n2 - 1) * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
        for (int u0 = 0; u0 < nu; u0 += 256) {
            const int u = min(u0 + tid, nu - 1);
            const int jq = qdiv(u, inv_c1n), c = (u - jq * c1n) * 8, jp = jq - 1;
            const int x0 = max(jp, 0), x1 = min(jp + 1, wl - 1);
            const uint4 t00 = *reinterpret_cast<const uint4*>(lo0 + x0 * C1 + c), t01 = *reinterpret_cast<const uint4*>(lo0 + x1 * C1 + c);
            const uint4 t10 = *reinterpret_cast<const uint4*>(lo1 + x0 * C1 + c), t11 = *reinterpret_cast<const uint4*>(lo1 + x1 * C1 + c);
            if (u0 + tid >= nu) continue;
            float a[8], b[8], cc[8], d[8];
            unpack8(t00, a); unpack8(t01, b); unpack8(t10, cc); unpack8(t11, d);
#pragma unroll
            for (int ey = 0; ey < 2; ++ey) {
                const int y = 2 * ip + 1 + ey;
                if (y < 0 || y >= H) continue;
                // the row-at-a-time kernel's weights: fy = max(y/2 - 1/4, 0), ly = fy - floor(fy): 0 on row 0, 1/4 / 3/4 inside
                const float fy = fmaxf(0.5f * ((float)y + 0.5f) - 0.5f, 0.f);
                const float ly = fy - (float)(int)fy, hy = 1.f - ly;
                bf16_t* orow = out + ((size_t)n * H + y) * W * C;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int x = 2 * jp + 1 + e;
                    if (x < 0 || x >= W) continue;
                    const float lx = e ? 0.75f : 0.25f, hx = 1.f - lx;
                    unsigned r[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v0 = hy * (hx * a[2 * q] + lx * b[2 * q]) + ly * (hx * cc[2 * q] + lx * d[2 * q]);
                        const float v1 = hy * (hx * a[2 * q + 1] + lx * b[2 * q + 1]) + ly * (hx * cc[2 * q + 1] + lx * d[2 * q + 1]);
                        r[q] = (unsigned)f32_to_bf16(v0) | ((unsigned)f32_to_bf16(v1) << 16);
                    }
                    *reinterpret_cast<uint4*>(orow + x * C + c) = make_uint4(r[0], r[1], r[2], r[3]);
                }
            }
        }
#pragma unroll
        for (int ey = 0; ey < 2; ++ey) {
            const int y = 2 * ip + 1 + ey;
            if (y < 0 || y >= H) continue;
            bf16_t* orow = out + ((size_t)n * H + y) * W * C;
            const bf16_t* sk = skip + ((size_t)n * H + y) * W * C2;
#pragma unroll
            for (int u = 0; u < MAXV; ++u) {
                const int jc = u * 256 + tid;
                if (jc < n2) {
                    const int x = qdiv(jc, inv_c2n);
                    *reinterpret_cast<uint4*>(orow + x * C + C1 + (jc - x * c2n) * 8) = cp[ey][u];
                }
            }
            for (int jc = MAXV * 256 + tid; jc < n2; jc += 256) {
                const int x = qdiv(jc, inv_c2n);
                *reinterpret_cast<uint4*>(orow + x * C + C1 + (jc - x * c2n) * 8) = *reinterpret_cast<const uint4*>(sk + jc * 8);
            }
        }
    }
}

// Training stem: 3x3 / stride 2 / pad 1 max-pool with the winning tap recorded (one byte per output element, 3*dy + dx in
// window coordinates; first maximum in row-major scan order, as PyTorch), and its backward as a GATHER: an input pixel
// belongs to at most 2 x 2 windows, so a thread reads their tap bytes and gradients and writes its 16 bytes of dx once
// (PyTorch's backward: 335 us per cfg4 step for a 205 MB dx; this is one streaming write plus cached reads).
template <int IO>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const void* x, void* y, unsigned char* idx,
                                                          int N, int H, int W, int C, int Ho, int Wo)
{
    constexpr int V = IO == GDKVM_F32 ? 4 : 8;
    const int cg = C / V;
    const size_t total = (size_t)N * Ho * Wo * cg;
    const uint4* xv = static_cast<const uint4*>(x);
    uint4* yv = static_cast<uint4*>(y);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cg);
        size_t p = i / cg;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        uint4 win[9];
        bool ok[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ih = 2 * oh - 1 + dy, iw = 2 * ow - 1 + dx;
                ok[3 * dy + dx] = ih >= 0 && ih < H && iw >= 0 && iw < W;
                const int hh = ok[3 * dy + dx] ? ih : 2 * oh, ww = ok[3 * dy + dx] ? iw : 2 * ow;
                win[3 * dy + dx] = xv[(((size_t)n * H + hh) * W + ww) * cg + c];
            }
        float m[V];
        unsigned char best[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { m[j] = -INFINITY; best[j] = 4; }      // (tap 4, the centre, is always inside the image)
        bool first = true;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (!ok[t]) continue;
            const unsigned w4[4] = {win[t].x, win[t].y, win[t].z, win[t].w};
            float v[V];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (IO == GDKVM_F32) v[j] = __uint_as_float(w4[j]);
                else { v[2 * j] = __uint_as_float(w4[j] << 16); v[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u); }
            }
#pragma unroll
            for (int j = 0; j < V; ++j)
                if (first || v[j] > m[j] || v[j] != v[j]) { m[j] = v[j]; best[j] = (unsigned char)t; }
            first = false;
        }
        uint4 o;
        if constexpr (IO == GDKVM_F32) {
            o.x = __float_as_uint(m[0]); o.y = __float_as_uint(m[1]); o.z = __float_as_uint(m[2]); o.w = __float_as_uint(m[3]);
            *reinterpret_cast<unsigned*>(idx + i * 4) = best[0] | (best[1] << 8) | (best[2] << 16) | ((unsigned)best[3] << 24);
        } else {
            o.x = (__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xffff0000u);       // exact: the values are bf16
            o.y = (__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xffff0000u);
            o.z = (__float_as_uint(m[4]) >> 16) | (__float_as_uint(m[5]) & 0xffff0000u);
            o.w = (__float_as_uint(m[6]) >> 16) | (__float_as_uint(m[7]) & 0xffff0000u);
            uint2 b;
            b.x = best[0] | (best[1] << 8) | (best[2] << 16) | ((unsigned)best[3] << 24);
            b.y = best[4] | (best[5] << 8) | (best[6] << 16) | ((unsigned)best[7] << 24);
            *reinterpret_cast<uint2*>(idx + i * 8) = b;
        }
        yv[i] = o;
    }
}

template <int IO>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const void* dy, const unsigned char* idx, void* dx,
                                                          int N, int H, int W, int C, int Ho, int Wo)
{
    constexpr int V = IO == GDKVM_F32 ? 4 : 8;
    const int cg = C / V;
    const size_t total = (size_t)N * H * W * cg;
    const uint4* dv = static_cast<const uint4*>(dy);
    uint4* xv = static_cast<uint4*>(dx);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cg);
        size_t p = i / cg;
        const int iw = (int)(p % W); p /= W;
        const int ih = (int)(p % H);
        const int n = (int)(p / H);
        const int oh0 = ih >> 1, ow0 = iw >> 1;            // windows oh0 (always) and oh0 + 1 (odd rows, if it exists); same for columns
        uint4 g[4];
        unsigned bt[4][2];
        int tap[4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int oh = oh0 + a, ow = ow0 + b;
                const bool ok = (a == 0 || ((ih & 1) && oh < Ho)) && (b == 0 || ((iw & 1) && ow < Wo)) && oh < Ho && ow < Wo;
                const size_t o = (((size_t)n * Ho + (ok ? oh : oh0 < Ho ? oh0 : Ho - 1)) * Wo + (ok ? ow : ow0 < Wo ? ow0 : Wo - 1)) * cg + c;
                g[2 * a + b] = dv[o];
                if constexpr (IO == GDKVM_F32) { bt[2 * a + b][0] = *reinterpret_cast<const unsigned*>(idx + o * 4); bt[2 * a + b][1] = 0; }
                else { const uint2 t = *reinterpret_cast<const uint2*>(idx + o * 8); bt[2 * a + b][0] = t.x; bt[2 * a + b][1] = t.y; }
                tap[2 * a + b] = ok ? 3 * (ih - (2 * oh - 1)) + (iw - (2 * ow - 1)) : 255;
            }
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned w4[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const unsigned wt = (bt[q][j >> 2] >> (8 * (j & 3))) & 0xffu;
                float d;
                if constexpr (IO == GDKVM_F32) d = __uint_as_float(w4[j]);
                else d = (j & 1) ? __uint_as_float(w4[j >> 1] & 0xffff0000u) : __uint_as_float(w4[j >> 1] << 16);
                acc[j] += (int)wt == tap[q] ? d : 0.f;
            }
        }
        uint4 o;
        if constexpr (IO == GDKVM_F32) {
            o.x = __float_as_uint(acc[0]); o.y = __float_as_uint(acc[1]); o.z = __float_as_uint(acc[2]); o.w = __float_as_uint(acc[3]);
        } else {
            o.x = (unsigned)f32_to_bf16(acc[0]) | ((unsigned)f32_to_bf16(acc[1]) << 16);
            o.y = (unsigned)f32_to_bf16(acc[2]) | ((unsigned)f32_to_bf16(acc[3]) << 16);
            o.z = (unsigned)f32_to_bf16(acc[4]) | ((unsigned)f32_to_bf16(acc[5]) << 16);
            o.w = (unsigned)f32_to_bf16(acc[6]) | ((unsigned)f32_to_bf16(acc[7]) << 16);
        }
        xv[i] = o;
    }
}

// Backward of the pass above (training): d_lo[n, i, j, :] = sum over the output pixels whose bilinear taps touch (i, j) of
// weight * d_out[.., :C1] -- a gather, so it is deterministic (PyTorch's backward scatters with float atomics into an fp32
// copy: 0.93 ms per cfg4 step) -- and d_skip = d_out[.., C1:].  One workgroup per low-resolution row; a thread owns one
// 16-byte channel group of one pixel, its taps (4 x 4 at exactly 2x) all in flight before the first add.  The weights come
// from the forward's own index formula evaluated per candidate row / column, so the borders need no special case.
__device__ __forceinline__ float tap_weight(int o, int i, float s, int n_in)
{
    const float f = fmaxf(s * ((float)o + 0.5f) - 0.5f, 0.f);
    const int i0 = (int)f, i1 = min(i0 + 1, n_in - 1);
    const float l = f - (float)i0;
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}

__global__ __launch_bounds__(256) void upsample_cat_bwd_bf16_kernel(const bf16_t* dout, bf16_t* dlo, bf16_t* dskip,
                                                                    int Nimg, int hl, int wl, int H, int W, int C1, int C2,
                                                                    float sy, float sx, int ry, int rx)
{
    const int C = C1 + C2, c1n = C1 / 8, c2n = C2 / 8, tid = threadIdx.x;
    const bool exact2x = H == 2 * hl && W == 2 * wl;
    for (int row = blockIdx.x; row < Nimg * hl; row += gridDim.x) {
        const int n = row / hl, i = row - n * hl;
        const bf16_t* dimg = dout + (size_t)n * H * W * C;
        for (int u = tid; u < wl * c1n; u += 256) {
            const int j = u / c1n, c = (u - j * c1n) * 8;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            if (exact2x) {
                uint4 t[4][4];
                float wy[4], wx[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int oy = 2 * i - 1 + a, ox = 2 * j - 1 + a;
                    wy[a] = (oy >= 0 && oy < H) ? tap_weight(oy, i, sy, hl) : 0.f;
                    wx[a] = (ox >= 0 && ox < W) ? tap_weight(ox, j, sx, wl) : 0.f;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int oy = min(max(2 * i - 1 + a, 0), H - 1), ox = min(max(2 * j - 1 + b, 0), W - 1);
                        t[a][b] = *reinterpret_cast<const uint4*>(dimg + ((size_t)oy * W + ox) * C + c);
                    }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        float v[8];
                        unpack8(t[a][b], v);
                        const float wgt = wy[a] * wx[b];
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, v[e], acc[e]);
                    }
            } else {
                const int oy0 = max((int)(((float)i - 0.5f) / sy - 0.5f) - ry, 0), oy1 = min((int)(((float)i + 1.5f) / sy - 0.5f) + ry, H - 1);
                const int ox0 = max((int)(((float)j - 0.5f) / sx - 0.5f) - rx, 0), ox1 = min((int)(((float)j + 1.5f) / sx - 0.5f) + rx, W - 1);
                for (int oy = oy0; oy <= oy1; ++oy) {
                    const float wy = tap_weight(oy, i, sy, hl);
                    if (wy == 0.f) continue;
                    for (int ox = ox0; ox <= ox1; ++ox) {
                        const float wgt = wy * tap_weight(ox, j, sx, wl);
                        if (wgt == 0.f) continue;
                        float v[8];
                        unpack8(*reinterpret_cast<const uint4*>(dimg + ((size_t)oy * W + ox) * C + c), v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, v[e], acc[e]);
                    }
                }
            }
            unsigned r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = (unsigned)f32_to_bf16(acc[2 * q]) | ((unsigned)f32_to_bf16(acc[2 * q + 1]) << 16);
            *reinterpret_cast<uint4*>(dlo + ((size_t)row * wl + j) * C1 + c) = make_uint4(r[0], r[1], r[2], r[3]);
        }
    }
    // the skip half: a strided copy, split evenly over the workgroups
    const size_t ncopy = (size_t)Nimg * H * W * c2n;
    for (size_t v = (size_t)blockIdx.x * 256 + tid; v < ncopy; v += (size_t)gridDim.x * 256) {
        const size_t pix = v / c2n;
        const int cv = (int)(v - pix * c2n) * 8;
        *reinterpret_cast<uint4*>(dskip + pix * C2 + cv) = *reinterpret_cast<const uint4*>(dout + pix * C + C1 + cv);
    }
}

}  // namespace

extern "C" int gdkvm_upsample_cat(const void* lo, const void* skip, void* out,
                                  int Nimg, int hl, int wl, int H, int W, int C1, int C2, int io_dtype, void* stream)
{
    if (Nimg < 0 || hl <= 0 || wl <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 < 0 || (C2 == 0) != (skip == nullptr))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "upsample_cat: bad shape (C2 == 0 goes with skip == NULL: the enlargement alone)");
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "upsample_cat: only bf16 (inference build) is implemented");
    if (C1 % 8 || C2 % 8) return gdkvm_fail(GDKVM_ERR_SHAPE, "upsample_cat: channel counts must be multiples of 8");
    if (Nimg == 0) return GDKVM_OK;
    if (!lo || !out || !gdkvm_aligned16(lo) || (skip && !gdkvm_aligned16(skip)) || !gdkvm_aligned16(out))
        return gdkvm_fail(GDKVM_ERR_ARG, "upsample_cat: null or misaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    if ((size_t)W * (C1 + C2) >= (1u << 20) || (size_t)H >= (1u << 20)) return gdkvm_fail(GDKVM_ERR_SHAPE, "upsample_cat: row too long or image too tall");
    // the kernel's float-reciprocal row -> (image, y) split is exact below 2^20 rows: larger batches go as several launches over
    // image ranges (16384 frames x 64 rows used to be refused)
    const int per = (int)(((1u << 20) - 1) / (unsigned)H);
    const char* env_pairs = getenv("GDKVM_UPSAMPLE_ROW_PAIRS");                     // ("0": A/B switch, the row-at-a-time kernel; read per call: tests toggle it)
    const bool pairs = !(env_pairs && env_pairs[0] == '0');
    for (int n0 = 0; n0 < Nimg; n0 += per) {
        const int nn = Nimg - n0 < per ? Nimg - n0 : per;
        if (pairs && H == 2 * hl && W == 2 * wl) {
            size_t blocks = (size_t)nn * (hl + 1);
            if (blocks > 256 * 16) blocks = 256 * 16;
            hipLaunchKernelGGL(upsample_cat_bf16_2x_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                               static_cast<const bf16_t*>(lo) + (size_t)n0 * hl * wl * C1,
                               skip ? static_cast<const bf16_t*>(skip) + (size_t)n0 * H * W * C2 : nullptr,
                               static_cast<bf16_t*>(out) + (size_t)n0 * H * W * (C1 + C2),
                               nn, hl, wl, C1, C2, 8.0f / (float)C1, 8.0f / (float)C2, 1.0f / (float)(hl + 1));
            continue;
        }
        size_t blocks = (size_t)nn * H;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(upsample_cat_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const bf16_t*>(lo) + (size_t)n0 * hl * wl * C1,
                           skip ? static_cast<const bf16_t*>(skip) + (size_t)n0 * H * W * C2 : nullptr,
                           static_cast<bf16_t*>(out) + (size_t)n0 * H * W * (C1 + C2),
                           nn, hl, wl, H, W, C1, C2, (float)hl / (float)H, (float)wl / (float)W, 8.0f / (float)C1, 8.0f / (float)C2, 1.0f / (float)H);
    }
    GDKVM_LAUNCH_CHECK("upsample_cat_bf16_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_upsample_cat_bwd(const void* dout, void* dlo, void* dskip,
                                      int Nimg, int hl, int wl, int H, int W, int C1, int C2, int io_dtype, void* stream)
{
    if (Nimg < 0 || hl <= 0 || wl <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 <= 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "upsample_cat_bwd: bad shape");
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "upsample_cat_bwd: only bf16 is implemented");
    if (C1 % 8 || C2 % 8) return gdkvm_fail(GDKVM_ERR_SHAPE, "upsample_cat_bwd: channel counts must be multiples of 8");
    if (Nimg == 0) return GDKVM_OK;
    if (!dout || !dlo || !dskip || !gdkvm_aligned16(dout) || !gdkvm_aligned16(dlo) || !gdkvm_aligned16(dskip))
        return gdkvm_fail(GDKVM_ERR_ARG, "upsample_cat_bwd: null or misaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    if ((size_t)H * W * (C1 + C2) >= (1u << 30)) return gdkvm_fail(GDKVM_ERR_SHAPE, "upsample_cat_bwd: image too large");
    size_t blocks = (size_t)Nimg * hl;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const float sy = (float)hl / (float)H, sx = (float)wl / (float)W;
    // candidate range slack (general path): one output row / column either side of the analytic bounds
    hipLaunchKernelGGL(upsample_cat_bwd_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(dout), static_cast<bf16_t*>(dlo), static_cast<bf16_t*>(dskip),
                       Nimg, hl, wl, H, W, C1, C2, sy, sx, 1, 1);
    GDKVM_LAUNCH_CHECK("upsample_cat_bwd_bf16_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_bias_relu_maxpool(const void* x, const float* bias, void* y, int N, int H, int W, int C,
                                       int x_rows, int x_cols, int io_dtype, void* stream)
{
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || x_rows < H || x_cols < W)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "bias_relu_maxpool: N=%d H=%d W=%d C=%d stored %dx%d", N, H, W, C, x_rows, x_cols);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "bias_relu_maxpool: io_dtype=%d", io_dtype);
    const int V = io_dtype == GDKVM_F32 ? 4 : 8;
    if (C % V) return gdkvm_fail(GDKVM_ERR_SHAPE, "bias_relu_maxpool: C=%d must be a multiple of %d", C, V);
    if (N == 0) return GDKVM_OK;
    if (!x || !bias || !y) return gdkvm_fail(GDKVM_ERR_ARG, "bias_relu_maxpool: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(y) || !gdkvm_aligned16(bias))
        return gdkvm_fail(GDKVM_ERR_ARG, "bias_relu_maxpool: pointers must be 16-byte aligned");
    if (x == y) return gdkvm_fail(GDKVM_ERR_ARG, "bias_relu_maxpool: cannot run in place");
    if (int rc = gdkvm_check_device()) return rc;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / V);
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((bias_relu_maxpool_kernel<GDKVM_F32>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias, y, N, H, W, C, Ho, Wo, x_rows, x_cols);
    else hipLaunchKernelGGL((bias_relu_maxpool_kernel<GDKVM_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias, y, N, H, W, C, Ho, Wo, x_rows, x_cols);
    GDKVM_LAUNCH_CHECK("bias_relu_maxpool_kernel");
    return GDKVM_OK;
}

namespace {
int maxpool_check(const char* who, int N, int H, int W, int C, int io_dtype, const void* a, const void* b, const void* c)
{
    if (N < 0 || H <= 0 || W <= 0 || C <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: N=%d H=%d W=%d C=%d", who, N, H, W, C);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: io_dtype=%d", who, io_dtype);
    const int V = io_dtype == GDKVM_F32 ? 4 : 8;
    if (C % V) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: C=%d must be a multiple of %d", who, C, V);
    if (N == 0) return GDKVM_OK;
    if (!a || !b || !c) return gdkvm_fail(GDKVM_ERR_ARG, "%s: null pointer", who);
    if (!gdkvm_aligned16(a) || !gdkvm_aligned16(b) || !gdkvm_aligned16(c)) return gdkvm_fail(GDKVM_ERR_ARG, "%s: pointers must be 16-byte aligned", who);
    return gdkvm_check_device();
}
}  // namespace

extern "C" int gdkvm_maxpool_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, int io_dtype, void* stream)
{
    if (int rc = maxpool_check("maxpool_fwd", N, H, W, C, io_dtype, x, y, idx)) return rc;
    if (N == 0) return GDKVM_OK;
    const int V = io_dtype == GDKVM_F32 ? 4 : 8, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / V);
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char* ix = static_cast<unsigned char*>(idx);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((maxpool_fwd_kernel<GDKVM_F32>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, ix, N, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL((maxpool_fwd_kernel<GDKVM_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, ix, N, H, W, C, Ho, Wo);
    GDKVM_LAUNCH_CHECK("maxpool_fwd_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_maxpool_bwd(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, int io_dtype, void* stream)
{
    if (int rc = maxpool_check("maxpool_bwd", N, H, W, C, io_dtype, dy, idx, dx)) return rc;
    if (N == 0) return GDKVM_OK;
    const int V = io_dtype == GDKVM_F32 ? 4 : 8, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)N * H * W * (C / V);
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned char* ix = static_cast<const unsigned char*>(idx);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((maxpool_bwd_kernel<GDKVM_F32>), dim3((unsigned)blocks), dim3(256), 0, st, dy, ix, dx, N, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL((maxpool_bwd_kernel<GDKVM_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, dy, ix, dx, N, H, W, C, Ho, Wo);
    GDKVM_LAUNCH_CHECK("maxpool_bwd_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_stem_s2d(const void* x, void* out, int N, int C, int H, int W, int Cp, int io_dtype, void* stream)
{
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cp < 4 * C)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "stem_s2d: N=%d C=%d H=%d W=%d Cp=%d (H, W even; Cp >= 4C)", N, C, H, W, Cp);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "stem_s2d: io_dtype=%d", io_dtype);
    if (N == 0) return GDKVM_OK;
    if (!x || !out) return gdkvm_fail(GDKVM_ERR_ARG, "stem_s2d: null pointer");
    if (int rc = gdkvm_check_device()) return rc;
    const size_t total = (size_t)N * (H / 2) * (W / 2);
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((stem_s2d_kernel<GDKVM_F32>), dim3((unsigned)blocks), dim3(256), 0, st, x, out, N, C, H, W, Cp);
    else hipLaunchKernelGGL((stem_s2d_kernel<GDKVM_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, x, out, N, C, H, W, Cp);
    GDKVM_LAUNCH_CHECK("stem_s2d_kernel");
    return GDKVM_OK;
}
