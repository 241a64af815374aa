// bn.hip -- training-mode BatchNorm fused with the residual add and the ReLU that follow it, forward and backward, for the
// NHWC conv outputs either side of the memory path (the training side).
//
// In a cfg4 training step the library BatchNorm kernels were 38 % of the GPU time (7.1 ms of 18.6 ms,
// profiles/r01_l_train_cfg4_steady_state.csv) for work that is three streaming passes forward and two backward.  Here:
//   forward   bn_stats (x once) -> bn_finalize (C channels) -> bn_apply  y = act(x*scale + shift (+ residual))
//   backward  bn_bwd_reduce (x, y, dy once) -> bn_bwd_finalize -> bn_bwd_dx  dx (and the masked gradient for the residual branch)
// All passes are HBM-bound by construction: 16-byte accesses, a thread keeps ONE channel group for the whole pass (its
// per-channel parameters and accumulators live in registers), every trip has UNR independent loads in flight.
// Reductions are deterministic (fixed partial layout, fixed summation order; no atomics).  The variance is accumulated on
// values shifted by the first row of the tensor, so it does not cancel when |mean| >> std.
#include "gdkvm_common.hpp"

namespace {

constexpr int BN_MAX_PART = 512;        // partial-sum rows (one per workgroup of the reduction passes)

template <int IO> struct VecOf { static constexpr int V = IO == GDKVM_F32 ? 4 : 8; };

template <int IO>
__device__ __forceinline__ void unpack(const uint4& a, float (&v)[VecOf<IO>::V])
{
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
    if constexpr (IO == GDKVM_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(w[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
    }
}

template <int IO>
__device__ __forceinline__ uint4 pack(const float (&v)[VecOf<IO>::V])
{
    uint4 o;
    if constexpr (IO == GDKVM_F32) {
        o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]); o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
    } else {
        o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    }
    return o;
}

template <int V>
__device__ __forceinline__ void load_param(const float* p, int c0, float (&v)[V])
{
#pragma unroll
    for (int j = 0; j < V; j += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + c0 + j);
        v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
    }
}

// Thread -> data map shared by every pass: the tensor is M rows of G = C/V 16-byte vectors; a workgroup owns the rows
// [r0, r1) and walks them RP = 256/G rows per trip, so thread `tid` (< RP*G) reads vector r0*G + tid + trip*RP*G --
// contiguous across the workgroup -- and its channel group is always tid % G.
struct Walk {
    size_t v, vend, step;
    bool active;
    int cg;
    __device__ Walk(int M, int G, int rows_per_block)
    {
        const int tid = threadIdx.x, nact = (256 / G) * G;
        const size_t r0 = (size_t)blockIdx.x * rows_per_block;
        const size_t r1 = min((size_t)M, r0 + rows_per_block);
        active = tid < nact && r0 < r1;
        cg = tid % G;
        v = r0 * G + tid; vend = r1 * G; step = nact;
    }
};

// Workgroup reduction of per-thread accumulators acc[2][V] into part[blockIdx.x][2*C] (layout: stat-major, channel minor).
template <int V>
__device__ __forceinline__ void block_reduce_store(const float (&a1)[V], const float (&a2)[V], float* part, int C, int G)
{
    __shared__ float s[256][2 * V + 1];
    const int tid = threadIdx.x, RP = 256 / G;
#pragma unroll
    for (int j = 0; j < V; ++j) { s[tid][j] = a1[j]; s[tid][V + j] = a2[j]; }
    __syncthreads();
    for (int o = tid; o < 2 * C; o += 256) {
        const int stat = o >= C, c = o - stat * C, cg = c / V, j = c - cg * V;
        float t = 0.f;
        for (int rp = 0; rp < RP; ++rp) t += s[rp * G + cg][stat * V + j];
        part[(size_t)blockIdx.x * 2 * C + o] = t;
    }
}

template <int IO>
__global__ __launch_bounds__(256) void bn_stats_kernel(const uint4* __restrict__ x, float* __restrict__ part, int M, int C, int G,
                                                       int rows_per_block)
{
    constexpr int V = VecOf<IO>::V, UNR = 8;
    Walk w(M, G, rows_per_block);
    float s1[V], s2[V], kk[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.f;
    if (w.active) {
        unpack<IO>(x[w.cg], kk);                                   // shift = row 0 of these channels
        auto eat = [&](const uint4& a) __attribute__((always_inline)) {
            float f[V];
            unpack<IO>(a, f);
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = f[j] - kk[j]; s1[j] += d; s2[j] = fmaf(d, d, s2[j]); }
        };
        size_t v = w.v;
        for (; v + (UNR - 1) * w.step < w.vend; v += UNR * w.step) {
            uint4 a[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) a[u] = x[v + u * w.step];
#pragma unroll
            for (int u = 0; u < UNR; ++u) eat(a[u]);
        }
        for (; v < w.vend; v += w.step) eat(x[v]);
    }
    block_reduce_store<V>(s1, s2, part, C, G);
}

// Sum of the partial rows for 32 channels x 2 statistics per workgroup of 1024 threads: 16 disjoint row subsets, every load
// of a thread independent.  Returns (to threads 0..31) the two sums of channel blockIdx.x*32 + tid.
__device__ __forceinline__ void sum_partials(const float* part, int nblk, int C, float& S1, float& S2)
{
    __shared__ float s[16][64];
    const int tid = threadIdx.x, c = blockIdx.x * 32 + (tid & 31), stat = (tid >> 5) & 1, sub = tid >> 6;
    float acc = 0.f;
    if (c < C)
        for (int b0 = sub; b0 < nblk; b0 += 16 * 8) {     // eight independent loads per trip (a dynamic-trip-count loop of
            float t[8];                                    // dependent adds was one L2 round trip per partial row: 11 us)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 16 * u;
                t[u] = b < nblk ? part[(size_t)b * 2 * C + stat * C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += t[u];
        }
    s[sub][tid & 63] = acc;
    __syncthreads();
    S1 = S2 = 0.f;
    if (tid < 32) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { S1 += s[k][tid]; S2 += s[k][32 + tid]; }
    }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int nblk, const void* x, int io,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* running_mean, float* running_var, float* stats, int M, int C,
                                                           float eps, float momentum)
{
    float d1, d2;
    sum_partials(part, nblk, C, d1, d2);
    const int c = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && c < C) {
        const float k = io == GDKVM_F32 ? static_cast<const float*>(x)[c] : bf16_to_f32(static_cast<const bf16_t*>(x)[c]);
        const double n = (double)M, m = (double)d1 / n;
        double var = (double)d2 / n - m * m;
        var = var > 0.0 ? var : 0.0;
        const float mean = (float)((double)k + m);
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd;
        stats[c] = mean; stats[C + c] = rstd;                                       // [4][C]: mean, rstd, scale, shift
        stats[2 * C + c] = sc; stats[3 * C + c] = beta[c] - mean * sc;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (M > 1 ? n / (n - 1.0) : 1.0));
    }
}

template <int IO, bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const uint4* __restrict__ x, const uint4* __restrict__ res,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       uint4* __restrict__ y, int M, int G, int rows_per_block)
{
    constexpr int V = VecOf<IO>::V, UNR = 4;
    Walk w(M, G, rows_per_block);
    if (!w.active) return;
    float sc[V], sh[V];
    load_param<V>(scale, w.cg * V, sc); load_param<V>(shift, w.cg * V, sh);
    auto finish = [&](size_t i, const uint4& a, const uint4& r) __attribute__((always_inline)) {
        float f[V], g[V];
        unpack<IO>(a, f);
        if constexpr (RES) unpack<IO>(r, g);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float t = fmaf(f[j], sc[j], sh[j]);
            if constexpr (RES) t += g[j];
            f[j] = RELU ? fmaxf(t, 0.f) : t;
        }
        y[i] = pack<IO>(f);
    };
    size_t v = w.v;
    for (; v + (UNR - 1) * w.step < w.vend; v += UNR * w.step) {
        uint4 a[UNR], r[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) a[u] = x[v + u * w.step];
#pragma unroll
        for (int u = 0; u < UNR; ++u) r[u] = RES ? res[v + u * w.step] : uint4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNR; ++u) finish(v + u * w.step, a[u], r[u]);
    }
    for (; v < w.vend; v += w.step) finish(v, x[v], RES ? res[v] : uint4{0, 0, 0, 0});
}

// ------------------------------------------------------------------------------------- stem: BatchNorm + ReLU + max-pool as one
// Round 4.  The training stem is conv -> BatchNorm(train) -> ReLU -> 3x3 / stride 2 / pad 1 max-pool on a 205 MB activation (cfg4); as
// bn_apply + gdkvm_maxpool_fwd the normalised tensor was written and read back, and backward the pre-pool gradient (205 MB) was written by
// gdkvm_maxpool_bwd and read twice by the BatchNorm backward.  Fused, the normalised activation and its gradient never exist:
//   forward   bn_stats -> bn_finalize -> bn_pool_fwd: every pooled element evaluates relu(x*scale + shift) of its 3x3 window itself,
//             rounded to the I/O type exactly as bn_apply would have stored it (same maxima, same winning taps as the two-kernel form);
//   backward  the gradient of a pre-pool element is GATHERED from its (at most 2 x 2) windows' gradients and winning taps
//             (maxpool_bwd_kernel's rule, rounded to the I/O type like its output) inside bn_bwd_reduce and bn_bwd_dx (POOLED = true).
struct PoolGeo { const uint4* dyp; const unsigned char* idx; int H, W, Ho, Wo; float inv_w, inv_hw; };

// the pre-pool gradient of vector v (= pixel m * G + channel group cg) in the bf16 layout, as maxpool_bwd_kernel<bf16> writes it
__device__ __forceinline__ uint4 pool_gather(const PoolGeo& q, size_t v, int G, int cg)
{
    const int m = (int)(v / G);                              // (M < 2^22: the float reciprocals below are exact)
    const int n = (int)(((float)m + 0.5f) * q.inv_hw), r = m - n * q.H * q.W;
    const int ih = (int)(((float)r + 0.5f) * q.inv_w), iw = r - ih * q.W;
    const int oh0 = ih >> 1, ow0 = iw >> 1;
    uint4 g[4];
    uint2 bt[4];
    int tap[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oh = oh0 + a, ow = ow0 + b;
            const bool ok = (a == 0 || ((ih & 1) && oh < q.Ho)) && (b == 0 || ((iw & 1) && ow < q.Wo)) && oh < q.Ho && ow < q.Wo;
            const size_t o = (((size_t)n * q.Ho + (ok ? oh : oh0 < q.Ho ? oh0 : q.Ho - 1)) * q.Wo + (ok ? ow : ow0 < q.Wo ? ow0 : q.Wo - 1)) * G + cg;
            g[2 * a + b] = q.dyp[o];
            bt[2 * a + b] = *reinterpret_cast<const uint2*>(q.idx + o * 8);
            tap[2 * a + b] = ok ? 3 * (ih - (2 * oh - 1)) + (iw - (2 * ow - 1)) : 255;
        }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned w4[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
        const unsigned tb[2] = {bt[k].x, bt[k].y};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned wt = (tb[j >> 2] >> (8 * (j & 3))) & 0xffu;
            const float d = (j & 1) ? __uint_as_float(w4[j >> 1] & 0xffff0000u) : __uint_as_float(w4[j >> 1] << 16);
            acc[j] += (int)wt == tap[k] ? d : 0.f;
        }
    }
    return pack<GDKVM_BF16>(acc);
}

// pooled y and winning taps from the raw convolution x [N, H, W, C] (bf16): item = (pooled pixel, 8-channel group)
__global__ __launch_bounds__(256) void bn_pool_fwd_kernel(const uint4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                          uint4* __restrict__ y, unsigned char* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo)
{
    const int G = C / 8;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % G);
        size_t p = i / G;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float sc[8], sh[8];
        load_param<8>(scale, cg * 8, sc); load_param<8>(shift, cg * 8, sh);
        uint4 win[9];
        bool ok[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ih = 2 * oh - 1 + dy, iw = 2 * ow - 1 + dx;
                ok[3 * dy + dx] = ih >= 0 && ih < H && iw >= 0 && iw < W;
                const int hh = ok[3 * dy + dx] ? ih : 2 * oh, ww = ok[3 * dy + dx] ? iw : 2 * ow;
                win[3 * dy + dx] = x[(((size_t)n * H + hh) * W + ww) * G + cg];
            }
        float m[8];
        unsigned char best[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; best[j] = 4; }
        bool first = true;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (!ok[t]) continue;
            float f[8];
            unpack<GDKVM_BF16>(win[t], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = bf16_to_f32(f32_to_bf16(fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f)));      // what bn_apply would have stored
                if (first || a > m[j] || a != a) { m[j] = a; best[j] = (unsigned char)t; }
            }
            first = false;
        }
        y[i] = pack<GDKVM_BF16>(m);
        uint2 b;
        b.x = best[0] | (best[1] << 8) | (best[2] << 16) | ((unsigned)best[3] << 24);
        b.y = best[4] | (best[5] << 8) | (best[6] << 16) | ((unsigned)best[7] << 24);
        *reinterpret_cast<uint2*>(idx + i * 8) = b;
    }
}

// ------------------------------------------------------------------------------------------------------------ backward
// g = dy masked by the ReLU;  S1 = sum g,  S2 = sum g (x - mean)  per channel.
// MASK: 0 = no ReLU, 1 = from the saved output (y > 0), 2 = recomputed from x with the forward's own fma (x*scale + shift > 0:
// the same expression bn_apply_kernel evaluated, so the same mask) -- the y tensor is then not read at all (no residual case).
template <int IO, int MASK, bool POOLED = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const uint4* __restrict__ x, const uint4* __restrict__ y,
                                                            const uint4* __restrict__ dy, const float* __restrict__ stats,
                                                            float* __restrict__ part, int M, int C, int G, int rows_per_block, PoolGeo pg = PoolGeo{})
{
    constexpr int V = VecOf<IO>::V, UNR = 4;
    constexpr bool RELU = MASK == 1;
    Walk w(M, G, rows_per_block);
    float s1[V], s2[V], mu[V], sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.f;
    if (w.active) {
        load_param<V>(stats, w.cg * V, mu);
        if constexpr (MASK == 2) { load_param<V>(stats + 2 * C, w.cg * V, sc); load_param<V>(stats + 3 * C, w.cg * V, sh); }
        auto eat = [&](const uint4& xa, const uint4& ya, const uint4& da) __attribute__((always_inline)) {
            float f[V], o[V], d[V];
            unpack<IO>(xa, f); unpack<IO>(da, d);
            if constexpr (RELU) unpack<IO>(ya, o);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if constexpr (MASK == 2) o[j] = fmaf(f[j], sc[j], sh[j]);
                const float g = MASK ? (o[j] > 0.f ? d[j] : 0.f) : d[j];
                s1[j] += g; s2[j] = fmaf(g, f[j] - mu[j], s2[j]);
            }
        };
        size_t v = w.v;
        for (; v + (UNR - 1) * w.step < w.vend; v += UNR * w.step) {
            uint4 xa[UNR], ya[UNR], da[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) xa[u] = x[v + u * w.step];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if constexpr (POOLED) da[u] = pool_gather(pg, v + u * w.step, G, w.cg);
                else da[u] = dy[v + u * w.step];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) ya[u] = RELU ? y[v + u * w.step] : uint4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < UNR; ++u) eat(xa[u], ya[u], da[u]);
        }
        for (; v < w.vend; v += w.step) {
            uint4 dd;
            if constexpr (POOLED) dd = pool_gather(pg, v, G, w.cg); else dd = dy[v];
            eat(x[v], RELU ? y[v] : uint4{0, 0, 0, 0}, dd);
        }
    }
    block_reduce_store<V>(s1, s2, part, C, G);
}

// d_beta = S1, d_gamma = rstd S2;  dx = cA g + c1 (x - mean) + c0  with  cA = gamma rstd, c1 = -cA rstd^2 S2 / n, c0 = -cA S1 / n.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk,
                                                               const float* __restrict__ gamma, const float* __restrict__ stats,
                                                               float* dgamma, float* dbeta, float* coef, int M, int C)
{
    float S1, S2;
    sum_partials(part, nblk, C, S1, S2);
    const int c = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && c < C) {
        const float rstd = stats[C + c], cA = gamma[c] * rstd, inv_n = 1.f / (float)M;
        dbeta[c] = S1; dgamma[c] = rstd * S2;
        coef[c] = cA; coef[C + c] = -cA * S1 * inv_n; coef[2 * C + c] = -cA * rstd * rstd * S2 * inv_n;
    }
}

template <int IO, int MASK, bool DRES, bool POOLED = false>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const uint4* __restrict__ x, const uint4* __restrict__ y,
                                                        const uint4* __restrict__ dy, const float* __restrict__ stats,
                                                        const float* __restrict__ coef, uint4* __restrict__ dx,
                                                        uint4* __restrict__ dres, int M, int C, int G, int rows_per_block, PoolGeo pg = PoolGeo{})
{
    constexpr int V = VecOf<IO>::V, UNR = 4;
    constexpr bool RELU = MASK == 1;
    Walk w(M, G, rows_per_block);
    if (!w.active) return;
    float mu[V], cA[V], c0[V], c1[V], sc[V], sh[V];
    load_param<V>(stats, w.cg * V, mu); load_param<V>(coef, w.cg * V, cA);
    load_param<V>(coef + C, w.cg * V, c0); load_param<V>(coef + 2 * C, w.cg * V, c1);
    if constexpr (MASK == 2) { load_param<V>(stats + 2 * C, w.cg * V, sc); load_param<V>(stats + 3 * C, w.cg * V, sh); }
    auto finish = [&](size_t i, const uint4& xa, const uint4& ya, const uint4& da) __attribute__((always_inline)) {
        float f[V], o[V], d[V];
        unpack<IO>(xa, f); unpack<IO>(da, d);
        if constexpr (RELU) unpack<IO>(ya, o);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if constexpr (MASK == 2) o[j] = fmaf(f[j], sc[j], sh[j]);
            const float g = MASK ? (o[j] > 0.f ? d[j] : 0.f) : d[j];
            d[j] = g;
            f[j] = fmaf(cA[j], g, fmaf(c1[j], f[j] - mu[j], c0[j]));
        }
        dx[i] = pack<IO>(f);
        if constexpr (DRES) dres[i] = pack<IO>(d);
    };
    size_t v = w.v;
    for (; v + (UNR - 1) * w.step < w.vend; v += UNR * w.step) {
        uint4 xa[UNR], ya[UNR], da[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) xa[u] = x[v + u * w.step];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if constexpr (POOLED) da[u] = pool_gather(pg, v + u * w.step, G, w.cg);
            else da[u] = dy[v + u * w.step];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) ya[u] = RELU ? y[v + u * w.step] : uint4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNR; ++u) finish(v + u * w.step, xa[u], ya[u], da[u]);
    }
    for (; v < w.vend; v += w.step) {
        uint4 dd;
        if constexpr (POOLED) dd = pool_gather(pg, v, G, w.cg); else dd = dy[v];
        finish(v, x[v], RELU ? y[v] : uint4{0, 0, 0, 0}, dd);
    }
}

// The pooled backward on 2 x 2 blocks of pre-pool pixels (round 4): the four pixels of a block share their (at most) 2 x 2 windows, so a
// thread loads four x vectors, four window gradients and four tap words for four results -- the per-pixel gather of pool_gather reads four
// windows for EVERY pixel (36 loads per block) and decodes sixteen (pixel, window) pairs where nine exist; the tap a pixel would have in a
// window is a compile-time constant of its position in the block.  DX = false: partial sums (S1, S2) per workgroup, as bn_bwd_reduce_kernel;
// DX = true: dx.  Same values per element as the gather form; the sums are added in another order (block-major), so dgamma / dbeta agree
// with the two-kernel form to fp32 rounding, not bit for bit.  bf16, channel-group counts that divide 256.
template <bool DX>
__global__ __launch_bounds__(256) void bn_pool_bwd2x2_kernel(const uint4* __restrict__ x, PoolGeo q, const float* __restrict__ stats,
                                                             const float* __restrict__ coef, float* __restrict__ part, uint4* __restrict__ dx,
                                                             int N, int C, int G)
{
    const int tid = threadIdx.x, cg = tid % G;
    const int Hb = (q.H + 1) / 2, Wb = (q.W + 1) / 2;
    const size_t total = (size_t)N * Hb * Wb * G;
    float mu[8], sc[8], sh[8], cA[8], c0[8], c1[8], s1[8], s2[8];
    load_param<8>(stats, cg * 8, mu); load_param<8>(stats + 2 * C, cg * 8, sc); load_param<8>(stats + 3 * C, cg * 8, sh);
    if constexpr (DX) { load_param<8>(coef, cg * 8, cA); load_param<8>(coef + C, cg * 8, c0); load_param<8>(coef + 2 * C, cg * 8, c1); }
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < total; i += (size_t)gridDim.x * 256) {      // (256 % G == 0: i % G == cg throughout)
        size_t p = i / G;
        const int bx = (int)(p % Wb); p /= Wb;
        const int by = (int)(p % Hb);
        const int n = (int)(p / Hb);
        uint4 xv[2][2], gw[2][2];
        uint2 tw[2][2];
        bool pok[2][2], wok[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ih = 2 * by + r, iw = 2 * bx + c, oh = by + r, ow = bx + c;
                pok[r][c] = ih < q.H && iw < q.W;
                wok[r][c] = oh < q.Ho && ow < q.Wo;
                xv[r][c] = x[(((size_t)n * q.H + (pok[r][c] ? ih : 2 * by)) * q.W + (pok[r][c] ? iw : 2 * bx)) * G + cg];
                const size_t o = (((size_t)n * q.Ho + (wok[r][c] ? oh : by)) * q.Wo + (wok[r][c] ? ow : bx)) * G + cg;      // (window (by, bx) always exists)
                gw[r][c] = q.dyp[o];
                tw[r][c] = *reinterpret_cast<const uint2*>(q.idx + o * 8);
            }
        float gwf[2][2][8];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) unpack<GDKVM_BF16>(gw[a][b], gwf[a][b]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = 0.f;
#pragma unroll
                for (int a = 0; a <= r; ++a)               // an even row lies in window `by` only, an odd one in `by` and `by + 1`
#pragma unroll
                    for (int b = 0; b <= c; ++b) {
                        const unsigned tap = (unsigned)(3 * (r - 2 * a + 1) + (c - 2 * b + 1));
                        const unsigned tb[2] = {tw[a][b].x, tw[a][b].y};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned wt = (tb[j >> 2] >> (8 * (j & 3))) & 0xffu;
                            g[j] += (wok[a][b] && wt == tap) ? gwf[a][b][j] : 0.f;
                        }
                    }
                float f[8];
                unpack<GDKVM_BF16>(xv[r][c], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gr = bf16_to_f32(f32_to_bf16(g[j]));          // (the gathered gradient as maxpool_bwd_kernel would have stored it)
                    const float gm = fmaf(f[j], sc[j], sh[j]) > 0.f ? gr : 0.f;
                    if constexpr (DX) f[j] = fmaf(cA[j], gm, fmaf(c1[j], f[j] - mu[j], c0[j]));
                    else if (pok[r][c]) { s1[j] += gm; s2[j] = fmaf(gm, f[j] - mu[j], s2[j]); }
                }
                if constexpr (DX) {
                    if (pok[r][c]) dx[(((size_t)n * q.H + 2 * by + r) * q.W + 2 * bx + c) * G + cg] = pack<GDKVM_BF16>(f);
                }
            }
    }
    if constexpr (!DX) block_reduce_store<8>(s1, s2, part, C, G);
}

struct BnPlan {
    int G, RP, nred, rpb_red, nmap, rpb_map;
};

// reduction passes: at most BN_MAX_PART workgroups (one partial row each); map passes: up to 8 workgroups per CU
BnPlan bn_plan(long long M, int C, int V, int unr_red)
{
    BnPlan p;
    p.G = C / V; p.RP = 256 / p.G;
    auto split = [&](long long max_blocks, int unr, int& nb, int& rpb) {
        const long long trip = (long long)p.RP * unr;
        long long blocks = (M + trip - 1) / trip;
        if (blocks > max_blocks) blocks = max_blocks;
        if (blocks < 1) blocks = 1;
        long long r = (M + blocks - 1) / blocks;
        r = (r + p.RP - 1) / p.RP * p.RP;
        rpb = (int)r; nb = (int)((M + r - 1) / r);
    };
    split(BN_MAX_PART, unr_red, p.nred, p.rpb_red);
    split(256 * 8, 4, p.nmap, p.rpb_map);
    return p;
}

int bn_check(const char* who, long long M, int C, int io)
{
    if (io != GDKVM_F32 && io != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: io_dtype=%d", who, io);
    const int V = io == GDKVM_F32 ? 4 : 8;
    if (M < 1 || M > 0x7fffffffLL || C <= 0 || C % V || C / V > 256)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: rows=%lld C=%d (C a multiple of %d, at most %d)", who, M, C, V, 256 * V);
    return GDKVM_OK;
}

}  // namespace

extern "C" size_t gdkvm_bn_workspace_bytes(int C)
{
    return C > 0 ? ((size_t)BN_MAX_PART * 2 * C + 4 * (size_t)C) * sizeof(float) : 0;      // partial rows + backward coefficients
}

extern "C" int gdkvm_bn_fwd_train(const void* x, const void* residual, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, void* y, float* save_stats,
                                  void* ws, size_t ws_bytes, long long rows, int C, float eps, float momentum, int relu,
                                  int io_dtype, void* stream)
{
    if (int rc = bn_check("bn_fwd_train", rows, C, io_dtype)) return rc;
    if (!x || !gamma || !beta || !y || !save_stats || !ws) return gdkvm_fail(GDKVM_ERR_ARG, "bn_fwd_train: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(y) || !gdkvm_aligned16(ws) || !gdkvm_aligned16(save_stats)
        || (residual && !gdkvm_aligned16(residual)))
        return gdkvm_fail(GDKVM_ERR_ARG, "bn_fwd_train: pointers must be 16-byte aligned");
    if (ws_bytes < gdkvm_bn_workspace_bytes(C)) return gdkvm_fail(GDKVM_ERR_ARG, "bn_fwd_train: workspace too small");
    if (int rc = gdkvm_check_device()) return rc;
    const int V = io_dtype == GDKVM_F32 ? 4 : 8, M = (int)rows;
    const BnPlan p = bn_plan(rows, C, V, 8);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const float* scale = save_stats + 2 * (size_t)C;
    const float* shift = save_stats + 3 * (size_t)C;
    const uint4* xv = static_cast<const uint4*>(x);
    const uint4* rv = static_cast<const uint4*>(residual);
    uint4* yv = static_cast<uint4*>(y);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((bn_stats_kernel<GDKVM_F32>), dim3(p.nred), dim3(256), 0, st, xv, part, M, C, p.G, p.rpb_red);
    else hipLaunchKernelGGL((bn_stats_kernel<GDKVM_BF16>), dim3(p.nred), dim3(256), 0, st, xv, part, M, C, p.G, p.rpb_red);
    GDKVM_LAUNCH_CHECK("bn_stats_kernel");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, part, p.nred, x, io_dtype, gamma, beta,
                       running_mean, running_var, save_stats, M, C, eps, momentum);
    GDKVM_LAUNCH_CHECK("bn_finalize_kernel");
#define GDKVM_BN_APPLY(IO, RL, RS) \
    hipLaunchKernelGGL((bn_apply_kernel<IO, RL, RS>), dim3(p.nmap), dim3(256), 0, st, xv, rv, scale, shift, yv, M, p.G, p.rpb_map)
#define GDKVM_BN_APPLY_IO(IO)                                                         \
    do {                                                                              \
        if (relu) { if (residual) GDKVM_BN_APPLY(IO, true, true); else GDKVM_BN_APPLY(IO, true, false); }   \
        else { if (residual) GDKVM_BN_APPLY(IO, false, true); else GDKVM_BN_APPLY(IO, false, false); }      \
    } while (0)
    if (io_dtype == GDKVM_F32) GDKVM_BN_APPLY_IO(GDKVM_F32); else GDKVM_BN_APPLY_IO(GDKVM_BF16);
#undef GDKVM_BN_APPLY_IO
#undef GDKVM_BN_APPLY
    GDKVM_LAUNCH_CHECK("bn_apply_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_bn_bwd(const void* x, const void* y, const void* dy, const float* gamma, const float* save_stats,
                            void* dx, void* dres, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                            long long rows, int C, int relu, int io_dtype, void* stream)
{
    if (int rc = bn_check("bn_bwd", rows, C, io_dtype)) return rc;
    if (relu < 0 || relu > 2) return gdkvm_fail(GDKVM_ERR_ARG, "bn_bwd: relu=%d (0 none, 1 mask from y, 2 mask recomputed from x)", relu);
    if (!x || !dy || !gamma || !save_stats || !dx || !dgamma || !dbeta || !ws || (relu == 1 && !y))
        return gdkvm_fail(GDKVM_ERR_ARG, "bn_bwd: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(dy) || !gdkvm_aligned16(dx) || !gdkvm_aligned16(ws) || !gdkvm_aligned16(save_stats)
        || (relu == 1 && !gdkvm_aligned16(y)) || (dres && !gdkvm_aligned16(dres)))
        return gdkvm_fail(GDKVM_ERR_ARG, "bn_bwd: pointers must be 16-byte aligned");
    if (ws_bytes < gdkvm_bn_workspace_bytes(C)) return gdkvm_fail(GDKVM_ERR_ARG, "bn_bwd: workspace too small");
    if (int rc = gdkvm_check_device()) return rc;
    const int V = io_dtype == GDKVM_F32 ? 4 : 8, M = (int)rows;
    const BnPlan p = bn_plan(rows, C, V, 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    float* coef = part + (size_t)BN_MAX_PART * 2 * C;
    const uint4* xv = static_cast<const uint4*>(x);
    const uint4* yv = static_cast<const uint4*>(y);
    const uint4* dv = static_cast<const uint4*>(dy);
    uint4* dxv = static_cast<uint4*>(dx);
    uint4* drv = static_cast<uint4*>(dres);
#define GDKVM_BN_RED(IO, MK) \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<IO, MK>), dim3(p.nred), dim3(256), 0, st, xv, yv, dv, save_stats, part, M, C, p.G, p.rpb_red)
#define GDKVM_BN_RED_IO(IO) do { if (relu == 2) GDKVM_BN_RED(IO, 2); else if (relu) GDKVM_BN_RED(IO, 1); else GDKVM_BN_RED(IO, 0); } while (0)
    if (io_dtype == GDKVM_F32) GDKVM_BN_RED_IO(GDKVM_F32); else GDKVM_BN_RED_IO(GDKVM_BF16);
#undef GDKVM_BN_RED_IO
#undef GDKVM_BN_RED
    GDKVM_LAUNCH_CHECK("bn_bwd_reduce_kernel");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, part, p.nred, gamma, save_stats, dgamma, dbeta, coef, M, C);
    GDKVM_LAUNCH_CHECK("bn_bwd_finalize_kernel");
#define GDKVM_BN_DX(IO, MK, DR) \
    hipLaunchKernelGGL((bn_bwd_dx_kernel<IO, MK, DR>), dim3(p.nmap), dim3(256), 0, st, xv, yv, dv, save_stats, coef, dxv, drv, M, C, p.G, p.rpb_map)
#define GDKVM_BN_DX_IO(IO)                                                            \
    do {                                                                              \
        if (relu == 2) { if (dres) GDKVM_BN_DX(IO, 2, true); else GDKVM_BN_DX(IO, 2, false); }      \
        else if (relu) { if (dres) GDKVM_BN_DX(IO, 1, true); else GDKVM_BN_DX(IO, 1, false); }      \
        else { if (dres) GDKVM_BN_DX(IO, 0, true); else GDKVM_BN_DX(IO, 0, false); }                \
    } while (0)
    if (io_dtype == GDKVM_F32) GDKVM_BN_DX_IO(GDKVM_F32); else GDKVM_BN_DX_IO(GDKVM_BF16);
#undef GDKVM_BN_DX_IO
#undef GDKVM_BN_DX
    GDKVM_LAUNCH_CHECK("bn_bwd_dx_kernel");
    return GDKVM_OK;
}


// ---- the training stem's BatchNorm + ReLU + 3x3 / stride 2 / pad 1 max-pool as one (bf16, NHWC): see the comment at PoolGeo ----------------
static int bn_pool_check(const char* who, int N, int H, int W, int C, int io)
{
    if (io != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: only bf16 is implemented", who);
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || C / 8 > 256 || (long long)N * H * W >= (1ll << 22))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: N=%d H=%d W=%d C=%d (C a multiple of 8 up to 2048, fewer than 2^22 pixels)", who, N, H, W, C);
    return GDKVM_OK;
}

extern "C" int gdkvm_bn_pool_fwd_train(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                       void* y_pool, void* idx, float* save_stats, void* ws, size_t ws_bytes,
                                       int N, int H, int W, int C, float eps, float momentum, int io_dtype, void* stream)
{
    if (int rc = bn_pool_check("bn_pool_fwd_train", N, H, W, C, io_dtype)) return rc;
    if (!x || !gamma || !beta || !y_pool || !idx || !save_stats || !ws) return gdkvm_fail(GDKVM_ERR_ARG, "bn_pool_fwd_train: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(y_pool) || !gdkvm_aligned16(idx) || !gdkvm_aligned16(ws) || !gdkvm_aligned16(save_stats))
        return gdkvm_fail(GDKVM_ERR_ARG, "bn_pool_fwd_train: pointers must be 16-byte aligned");
    if (ws_bytes < gdkvm_bn_workspace_bytes(C)) return gdkvm_fail(GDKVM_ERR_ARG, "bn_pool_fwd_train: workspace too small");
    if (int rc = gdkvm_check_device()) return rc;
    const int M = N * H * W, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const BnPlan p = bn_plan(M, C, 8, 8);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const uint4* xv = static_cast<const uint4*>(x);
    hipLaunchKernelGGL((bn_stats_kernel<GDKVM_BF16>), dim3(p.nred), dim3(256), 0, st, xv, part, M, C, p.G, p.rpb_red);
    GDKVM_LAUNCH_CHECK("bn_stats_kernel");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, part, p.nred, x, io_dtype, gamma, beta,
                       running_mean, running_var, save_stats, M, C, eps, momentum);
    GDKVM_LAUNCH_CHECK("bn_finalize_kernel");
    const size_t items = (size_t)N * Ho * Wo * (C / 8);
    const size_t blocks = (items + 255) / 256;
    hipLaunchKernelGGL(bn_pool_fwd_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, st, xv, save_stats + 2 * (size_t)C,
                       save_stats + 3 * (size_t)C, static_cast<uint4*>(y_pool), static_cast<unsigned char*>(idx), N, H, W, C, Ho, Wo);
    GDKVM_LAUNCH_CHECK("bn_pool_fwd_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_bn_pool_bwd(const void* x, const void* dy_pool, const void* idx, const float* gamma, const float* save_stats,
                                 void* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                 int N, int H, int W, int C, int io_dtype, void* stream)
{
    if (int rc = bn_pool_check("bn_pool_bwd", N, H, W, C, io_dtype)) return rc;
    if (!x || !dy_pool || !idx || !gamma || !save_stats || !dx || !dgamma || !dbeta || !ws) return gdkvm_fail(GDKVM_ERR_ARG, "bn_pool_bwd: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(dy_pool) || !gdkvm_aligned16(idx) || !gdkvm_aligned16(dx) || !gdkvm_aligned16(ws) || !gdkvm_aligned16(save_stats))
        return gdkvm_fail(GDKVM_ERR_ARG, "bn_pool_bwd: pointers must be 16-byte aligned");
    if (ws_bytes < gdkvm_bn_workspace_bytes(C)) return gdkvm_fail(GDKVM_ERR_ARG, "bn_pool_bwd: workspace too small");
    if (int rc = gdkvm_check_device()) return rc;
    const int M = N * H * W;
    const BnPlan p = bn_plan(M, C, 8, 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    float* coef = part + (size_t)BN_MAX_PART * 2 * C;
    PoolGeo pg{static_cast<const uint4*>(dy_pool), static_cast<const unsigned char*>(idx), H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1,
               1.0f / (float)W, 1.0f / (float)(H * W)};
    const uint4* xv = static_cast<const uint4*>(x);
    if (256 % p.G == 0) {                                  // 2 x 2 pixel blocks: a third of the gather's loads
        const size_t items = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * p.G;
        const size_t blocks = (items + 255) / 256;
        const int nred = (int)(blocks < (size_t)BN_MAX_PART ? blocks : (size_t)BN_MAX_PART);
        hipLaunchKernelGGL((bn_pool_bwd2x2_kernel<false>), dim3(nred), dim3(256), 0, st, xv, pg, save_stats, coef, part, static_cast<uint4*>(dx), N, C, p.G);
        GDKVM_LAUNCH_CHECK("bn_pool_bwd2x2_kernel");
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, part, nred, gamma, save_stats, dgamma, dbeta, coef, M, C);
        GDKVM_LAUNCH_CHECK("bn_bwd_finalize_kernel");
        hipLaunchKernelGGL((bn_pool_bwd2x2_kernel<true>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, xv, pg, save_stats, coef, part,
                           static_cast<uint4*>(dx), N, C, p.G);
        GDKVM_LAUNCH_CHECK("bn_pool_bwd2x2_kernel");
        return GDKVM_OK;
    }
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<GDKVM_BF16, 2, true>), dim3(p.nred), dim3(256), 0, st, xv, nullptr, nullptr, save_stats, part, M, C, p.G, p.rpb_red, pg);
    GDKVM_LAUNCH_CHECK("bn_bwd_reduce_kernel");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, part, p.nred, gamma, save_stats, dgamma, dbeta, coef, M, C);
    GDKVM_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    hipLaunchKernelGGL((bn_bwd_dx_kernel<GDKVM_BF16, 2, false, true>), dim3(p.nmap), dim3(256), 0, st, xv, nullptr, nullptr, save_stats, coef,
                       static_cast<uint4*>(dx), nullptr, M, C, p.G, p.rpb_map, pg);
    GDKVM_LAUNCH_CHECK("bn_bwd_dx_kernel");
    return GDKVM_OK;
}
