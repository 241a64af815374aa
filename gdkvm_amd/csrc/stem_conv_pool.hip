// stem_conv_pool.hip -- the inference stem in one kernel: 4x4 / stride 1 / pad 2 convolution on the 16-channel space-to-depth
// image (== the 7x7 / stride 2 / pad 3 convolution on the frames, model.FusedConvPool.enable_s2d), bias, ReLU and the
// 3x3 / stride 2 / pad 1 max-pool.  As two kernels (library convolution, then gdkvm_bias_relu_maxpool) the full-resolution
// activation -- 213 MB at cfg2, four times everything else the stem touches -- is written and read back: 95 + 50 us = 10 % of
// a forward, the convolution itself bound by that write.  Here it only ever exists as a 9-row LDS tile:
//   * a workgroup (8 waves) owns 4 x 28 POOLED pixels = a 9 x 57 patch of convolution outputs (one halo row / column);
//   * its 12 x 67 pixel input band (16 channels = 32 B per pixel) arrives by LDS-DMA, double-buffered; with pixels 32 B apart the
//     four horizontal taps of a kernel row are 128 contiguous bytes, so the K = 256 reduction is 8 k-steps of plain 16-byte
//     operand reads at base + immediate (conflict-free: 16 consecutive pixels, no padding needed);
//   * the 32 KB of weights stay in registers (64 per wave: 32 output channels x 8 k-steps as MFMA A operands; MFMA rows permuted
//     so a lane ends with 8 consecutive channels of its pixel);
//   * wave (wm, wn) computes column tile wm of every convolution row for channels 32wn.., three rows at a time, adds the bias,
//     applies the ReLU, zeroes positions outside the image (post-ReLU values are >= 0, so a zero is as good as -inf for the
//     max) and writes bf16 into the LDS tile; after one barrier all 512 threads pool 3 x 3 windows out of LDS and store the
//     pooled pixels -- 51 MB instead of 213 + 213 + 51.
#include <atomic>
#include "gdkvm_common.hpp"

namespace {

constexpr int SP_TPY = 4, SP_TPX = 28;           // pooled tile
constexpr int SP_CR = 2 * SP_TPY + 1;            // convolution rows of a tile (9)
constexpr int SP_CC = 64;                        // convolution columns computed per row (57 used)
constexpr int SP_BR = SP_CR + 3, SP_BC = SP_CC + 3;         // input band 12 x 67 pixels
constexpr int SP_SLOTS = SP_BR * SP_BC * 2;      // 16-byte slots
constexpr int SP_PIECES = (SP_SLOTS + 63) / 64;
constexpr int SP_BAND_BYTES = SP_PIECES * 1024;
constexpr int SP_CPIX = 144;                     // bytes between pixels of the convolution tile (128 + 16: two-way instead of
                                                 // sixteen-way conflicts on the epilogue's ds_write_b128)
constexpr int SP_CONV_BYTES = SP_CR * SP_CC * SP_CPIX;

__device__ const uint4 g_stem_zero16 = {0, 0, 0, 0};

struct StemArgs {
    const bf16_t* xs; const bf16_t* w; const float* bias; bf16_t* y;
    int N, Hs, Ws, Hp, Wp, tiles_x, tiles_y;
    const bf16_t* xf; int Cf;            // NCHW: the frames themselves [N, Cf <= 4, 2 Hs, 2 Ws] (xs unused)
};

// NCHW (round 4): the band is built from the NCHW frames by the workgroup itself -- a 16-byte slot of the space-to-depth image is four
// 4-byte column pairs (channel c, row parity p: x[c][2Y + p][2X .. 2X + 1]) -- so the space-to-depth pass (gdkvm_stem_s2d: 38 MB read,
// 51 MB written and read back, 20 us of a 1 ms forward) disappears.  The pieces the LDS-DMA fetched are fetched into registers a tile
// ahead (16 dwords per lane) and written to the other band buffer behind the tile's MFMAs; same bytes in LDS, hence the same results.
// POOL = false (round 4, the TRAINING stem's forward): the raw convolution -- no bias, no ReLU, no pooling -- written to memory as bf16
// [N, Hs, Ws, 64], in non-overlapping tiles of 9 x 57 outputs (BatchNorm needs the full-resolution activation; the library's implicit
// GEMM took 183 us for it behind a 46 us zero-fill of its output).
template <bool NCHW, bool POOL = true>
__global__ __launch_bounds__(512, 1) void stem_conv_pool_kernel(StemArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* band2 = smem;                            // two input bands
    unsigned char* ctile = smem + 2 * SP_BAND_BYTES;        // convolution tile [9][64] pixels x 144 B
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wn = w & 1, wm = w >> 1;      // wm = column tile 0..3
    const int ntiles = a.N * a.tiles_y * a.tiles_x;

    // weights W[cout][ky][kx][c] (K = 64 ky + 16 kx + c), k-step ks = 2 ky + half: A fragment = 8 consecutive K at 32 ks + 8 g.
    // MFMA row rho = 4 g' + r of n-tile nt stands for channel 32wn + 8g' + 4nt + r.
    bf16x8 wf[2][8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int co = 32 * wn + 8 * (li >> 2) + 4 * nt + (li & 3);
            wf[nt][ks] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)co * 256 + 32 * ks + 8 * g);
        }
    float bia[8];
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + 32 * wn + 8 * g), b1 = *reinterpret_cast<const f32x4*>(a.bias + 32 * wn + 8 * g + 4);
        bia[0] = b0[0]; bia[1] = b0[1]; bia[2] = b0[2]; bia[3] = b0[3]; bia[4] = b1[0]; bia[5] = b1[1]; bia[6] = b1[2]; bia[7] = b1[3];
    }

    // band fetch by LDS-DMA: slot d = 2 pix + c (c = 16-byte half of the pixel's 16 channels), piece j = w + 8u
    constexpr int PP = (SP_PIECES + 7) / 8;
    auto fetch = [&](int tile, int buf) __attribute__((always_inline)) {
        const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x, ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
        const int y0 = (POOL ? 2 * SP_TPY * ty - 1 : SP_CR * ty) - 2, x0 = (POOL ? 2 * SP_TPX * tx - 1 : (SP_CC - 7) * tx) - 2;       // input row / column of band pixel (0, 0)
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int j = w + 8 * u;
            if (j >= SP_PIECES) break;
            const int d = 64 * j + lane, pix = d >> 1, c = d & 1;
            const int by = pix / SP_BC, bx = pix - by * SP_BC, yy = y0 + by, xx = x0 + bx;
            const bool ok = pix < SP_BR * SP_BC && yy >= 0 && yy < a.Hs && xx >= 0 && xx < a.Ws;
            const bf16_t* src = ok ? a.xs + ((((size_t)n * a.Hs + yy) * a.Ws + xx) * 16 + c * 8) : reinterpret_cast<const bf16_t*>(&g_stem_zero16);
            __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                reinterpret_cast<uintptr_t>(band2 + buf * SP_BAND_BYTES + 1024 * j)), 16, 0, 0);
        }
    };

    // NCHW: the same slots from the frames: slot d = 2 pix + half holds channel pairs (2 half, 2 half + 1) x row parity (0, 1)
    unsigned stage[PP][4];
    auto fetch_nchw = [&](int tile) __attribute__((always_inline)) {
        const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x, ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
        const int y0 = (POOL ? 2 * SP_TPY * ty - 1 : SP_CR * ty) - 2, x0 = (POOL ? 2 * SP_TPX * tx - 1 : (SP_CC - 7) * tx) - 2;
        const int H = 2 * a.Hs, W = 2 * a.Ws;
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int j = w + 8 * u, d = 64 * j + lane, pix = d >> 1, half = d & 1;
            const int by = pix / SP_BC, bx = pix - by * SP_BC, yy = y0 + by, xx = x0 + bx;
            const bool ok = j < SP_PIECES && pix < SP_BR * SP_BC && yy >= 0 && yy < a.Hs && xx >= 0 && xx < a.Ws;
#pragma unroll
            for (int e = 0; e < 4; ++e) {                   // e = 2 (channel within the half) + row parity
                const int c = 2 * half + (e >> 1);
                const bool live = ok && c < a.Cf;
                const bf16_t* src = live ? a.xf + ((((size_t)n * a.Cf + c) * H + 2 * yy + (e & 1)) * W + 2 * xx) : reinterpret_cast<const bf16_t*>(&g_stem_zero16);
                stage[u][e] = *reinterpret_cast<const unsigned*>(src);
            }
        }
    };
    auto put_nchw = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int j = w + 8 * u;
            if (j < SP_PIECES)
                *reinterpret_cast<uint4*>(band2 + buf * SP_BAND_BYTES + 1024 * j + 16 * lane) = make_uint4(stage[u][0], stage[u][1], stage[u][2], stage[u][3]);
        }
    };

    const unsigned xoff = (unsigned)((16 * wm + li) * 32 + g * 16);          // this lane's pixel column / K group inside a band row
    int tile = blockIdx.x, cur = 0;
    if (tile < ntiles) {
        if constexpr (NCHW) { fetch_nchw(tile); put_nchw(0); }
        else fetch(tile, 0);
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        const bool more = tile + (int)gridDim.x < ntiles;
#ifndef STEM_ABL_NODMA
        if (more) {                                                                      // lands behind this tile's MFMAs
            if constexpr (NCHW) fetch_nchw(tile + gridDim.x);
            else fetch(tile + gridDim.x, cur ^ 1);
        }
#endif
        const unsigned char* band = band2 + cur * SP_BAND_BYTES;
        const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x, ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
        const int cy0 = POOL ? 2 * SP_TPY * ty - 1 : SP_CR * ty, cx0 = POOL ? 2 * SP_TPX * tx - 1 : (SP_CC - 7) * tx;   // convolution row / column of tile position (0, 0)
        const int cx = cx0 + 16 * wm + li;                                                // this lane's convolution column
        const bool col_ok = cx >= 0 && cx < a.Ws && (POOL || 16 * wm + li < SP_CC - 7);   // (conv only: 57 columns per tile, no overlap)

        // ---- convolution rows in three groups of three: acc[row in group][nt] ---------------------------------------------
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            f32x4 acc[3][2];
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[r][0] = acc[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the three rows of a group read band rows 3grp .. 3grp + 5 (row r + ky): each of the 12 operand fragments (6 rows x 2
            // K halves) is loaded ONCE and feeds every (row, ky) pair that meets it -- half the LDS reads of a per-k-step reload
            bf16x8 xf[6][2];
#pragma unroll
            for (int rho = 0; rho < 6; ++rho)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    xf[rho][half] = *reinterpret_cast<const bf16x8*>(band + xoff + ((3 * grp + rho) * SP_BC) * 32 + half * 64);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int ky = ks >> 1, half = ks & 1;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#ifdef STEM_ABL_NOMFMA                                      // (tools/abl_stem.py: timing ablations, wrong results by design)
                    acc[r][0][0] += __builtin_bit_cast(f32x4, xf[r + ky][half])[0];
#else
                    acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], xf[r + ky][half], acc[r][0], 0, 0, 0);
                    acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], xf[r + ky][half], acc[r][1], 0, 0, 0);
#endif
                }
            }
            if constexpr (!POOL) {                          // the raw convolution, straight to memory
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int cy = cy0 + 3 * grp + r;
                    if (col_ok && cy < a.Hs) {
                        unsigned ow[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            ow[q] = (unsigned)f32_to_bf16(acc[r][q >> 1][2 * (q & 1)]) | ((unsigned)f32_to_bf16(acc[r][q >> 1][2 * (q & 1) + 1]) << 16);
                        *reinterpret_cast<uint4*>(a.y + (((size_t)n * a.Hs + cy) * a.Ws + cx) * 64 + 32 * wn + 8 * g) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    }
                }
                continue;
            }
            // bias + ReLU (+ zero outside the image) -> bf16 -> LDS tile; lane (li, g): channels 32wn + 8g .. +7 of column 16wm + li
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int cr = 3 * grp + r, cy = cy0 + cr;
                const float keep = (col_ok && cy >= 0 && cy < a.Hs) ? 1.f : 0.f;        // (values are >= 0 after the ReLU: masking is a multiply)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                unsigned ow[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {                                            // packed fp32 pairs: add, max, mul
                    const f32x2 t = {acc[r][q >> 1][2 * (q & 1)], acc[r][q >> 1][2 * (q & 1) + 1]};
                    const f32x2 bb = {bia[2 * q], bia[2 * q + 1]};
                    f32x2 v = __builtin_elementwise_max(t + bb, f32x2{0.f, 0.f}) * f32x2{keep, keep};
                    ow[q] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                }
                *reinterpret_cast<uint4*>(ctile + (cr * SP_CC + 16 * wm + li) * SP_CPIX + (32 * wn + 8 * g) * 2) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
        }
        if constexpr (NCHW) { if (more) put_nchw(cur ^ 1); }   // (the other buffer was last read a tile ago, two barriers back)
        __syncthreads();                                   // the convolution tile is complete (and the next band has landed)

        // ---- 3x3 / stride 2 max-pool out of LDS: item = (pooled row q, pooled column px, 8-channel group) ------------------
#ifndef STEM_ABL_NOPOOL
        for (int it = tid; POOL && it < SP_TPY * SP_TPX * 8; it += 512) {
            const int cg = it & 7, pp = it >> 3, q = pp / SP_TPX, px = pp - q * SP_TPX;
            const int py_g = SP_TPY * ty + q, px_g = SP_TPX * tx + px;
            // post-ReLU bf16 values are non-negative, so their order is the order of their bit patterns as unsigned 16-bit
            // integers: the 3x3 maximum is nine packed u16 max per word, with no unpacking (a third of the fp32 form's VALU work)
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            u16x2 m[4] = {u16x2{0, 0}, u16x2{0, 0}, u16x2{0, 0}, u16x2{0, 0}};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const uint4 t = *reinterpret_cast<const uint4*>(ctile + ((2 * q + dy) * SP_CC + 2 * px + dx) * SP_CPIX + cg * 16);
                    m[0] = __builtin_elementwise_max(m[0], __builtin_bit_cast(u16x2, t.x));
                    m[1] = __builtin_elementwise_max(m[1], __builtin_bit_cast(u16x2, t.y));
                    m[2] = __builtin_elementwise_max(m[2], __builtin_bit_cast(u16x2, t.z));
                    m[3] = __builtin_elementwise_max(m[3], __builtin_bit_cast(u16x2, t.w));
                }
            if (py_g < a.Hp && px_g < a.Wp) {
                const uint4 o = {__builtin_bit_cast(unsigned, m[0]), __builtin_bit_cast(unsigned, m[1]), __builtin_bit_cast(unsigned, m[2]),
                                 __builtin_bit_cast(unsigned, m[3])};
                *reinterpret_cast<uint4*>(a.y + (((size_t)n * a.Hp + py_g) * a.Wp + px_g) * 64 + cg * 8) = o;
            }
        }
#endif
        if constexpr (POOL) __syncthreads();               // the tile may be overwritten
    }
}

// The 4 x 4 x 16 kernel of the space-to-depth form from the 7 x 7 kernel of the stride-2 stem (model.FusedConvPool.enable_s2d's
// arithmetic as one kernel, for weights that change every step): w4[k][a + 2][b + 2][4c + 2p + q] = w7[k][c][2a + p + 3][2b + q + 3]
// where that tap exists, 0 elsewhere; w7 fp32 [64, C, 7, 7] with element strides (sk, sc, sr, ss), w4 bf16 [64][4][4][16].
__global__ __launch_bounds__(256) void stem_pack_s2d_kernel(const float* w7, bf16_t* w4, int C, long long sk, long long sc, long long sr, long long ss)
{
    const int i = blockIdx.x * 256 + threadIdx.x;          // (k, a, b, ch)
    if (i >= 64 * 256) return;
    const int ch = i & 15, b = (i >> 4) & 3, a = (i >> 6) & 3, k = i >> 8;
    const int c = ch >> 2, p = (ch >> 1) & 1, q = ch & 1, u = 2 * (a - 2) + p + 3, v = 2 * (b - 2) + q + 3;
    float x = 0.f;
    if (c < C && u >= 0 && u <= 6 && v >= 0 && v <= 6) x = w7[k * sk + c * sc + u * sr + v * ss];
    w4[i] = f32_to_bf16(x);
}

}  // namespace

static int stem_conv_pool_impl(const char* who, const void* xs, const void* xf, int Cf, const void* w, const float* bias, void* y,
                               int N, int Hs, int Ws, int io_dtype, void* stream, bool pool = true)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: only bf16 is implemented", who);
    if (N < 0 || Hs <= 0 || Ws <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: N=%d Hs=%d Ws=%d", who, N, Hs, Ws);
    if (N == 0) return GDKVM_OK;
    const void* x = xf ? xf : xs;
    if (!x || !w || (pool && !bias) || !y) return gdkvm_fail(GDKVM_ERR_ARG, "%s: null pointer", who);
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(w) || !gdkvm_aligned16(y) || (bias && !gdkvm_aligned16(bias)))
        return gdkvm_fail(GDKVM_ERR_ARG, "%s: pointers must be 16-byte aligned", who);
    if (int rc = gdkvm_check_device()) return rc;
    StemArgs a;
    a.xs = static_cast<const bf16_t*>(xs); a.w = static_cast<const bf16_t*>(w); a.bias = bias; a.y = static_cast<bf16_t*>(y);
    a.xf = static_cast<const bf16_t*>(xf); a.Cf = Cf;
    a.N = N; a.Hs = Hs; a.Ws = Ws; a.Hp = (Hs - 1) / 2 + 1; a.Wp = (Ws - 1) / 2 + 1;
    a.tiles_x = (a.Wp + SP_TPX - 1) / SP_TPX; a.tiles_y = (a.Hp + SP_TPY - 1) / SP_TPY;
    if (!pool) { a.tiles_x = (Ws + SP_CC - 8) / (SP_CC - 7); a.tiles_y = (Hs + SP_CR - 1) / SP_CR; }       // 9 x 57 convolution outputs per tile
    const long long ntiles = (long long)N * a.tiles_x * a.tiles_y;
    if (ntiles > 0x7fffffffLL) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: too many tiles", who);
    const size_t lds = 2 * (size_t)SP_BAND_BYTES + SP_CONV_BYTES;
    {   // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device; lock-free cache as in gdr_scan.hip
        static std::atomic<unsigned long long> done_mask{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "%s: hipGetDevice", who);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_pool_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_pool_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_pool_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "%s: %s", who, hipGetErrorString(e));
            done_mask.fetch_or(bit, std::memory_order_relaxed);
        }
    }
#ifndef STEM_GRID
#define STEM_GRID 256                                      // persistent, one workgroup per CU
#endif
    const int grid = (int)(ntiles < STEM_GRID ? ntiles : STEM_GRID);
    if (!pool) hipLaunchKernelGGL((stem_conv_pool_kernel<true, false>), dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    else if (xf) hipLaunchKernelGGL(stem_conv_pool_kernel<true>, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(stem_conv_pool_kernel<false>, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    GDKVM_LAUNCH_CHECK("stem_conv_pool_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_stem_conv_pool(const void* xs, const void* w, const float* bias, void* y, int N, int Hs, int Ws,
                                    int io_dtype, void* stream)
{
    return stem_conv_pool_impl("stem_conv_pool", xs, nullptr, 0, w, bias, y, N, Hs, Ws, io_dtype, stream);
}

extern "C" int gdkvm_stem_conv_pool_nchw(const void* x, const void* w, const float* bias, void* y, int N, int C, int H, int W,
                                         int io_dtype, void* stream)
{
    if (C <= 0 || C > 4 || H <= 0 || W <= 0 || (H & 1) || (W & 1))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "stem_conv_pool_nchw: C=%d H=%d W=%d (at most 4 channels, H and W even)", C, H, W);
    return stem_conv_pool_impl("stem_conv_pool_nchw", nullptr, x, C, w, bias, y, N, H / 2, W / 2, io_dtype, stream);
}

// Training: the raw 7x7 / stride 2 / pad 3 convolution of NCHW frames x [N, C <= 4, H, W] (H, W even) -> y [N, H/2, W/2, 64] bf16 (NHWC),
// no bias, with the kernel given in the space-to-depth form w4 [64, 4, 4, 16] bf16 (gdkvm_stem_pack_s2d).
extern "C" int gdkvm_stem_conv_nchw(const void* x, const void* w4, void* y, int N, int C, int H, int W, int io_dtype, void* stream)
{
    if (C <= 0 || C > 4 || H <= 0 || W <= 0 || (H & 1) || (W & 1))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "stem_conv_nchw: C=%d H=%d W=%d (at most 4 channels, H and W even)", C, H, W);
    return stem_conv_pool_impl("stem_conv_nchw", nullptr, x, C, w4, nullptr, y, N, H / 2, W / 2, io_dtype, stream, false);
}

extern "C" int gdkvm_stem_pack_s2d(const float* w7, void* w4, int C, long long sk, long long sc, long long sr, long long ss, void* stream)
{
    if (C <= 0 || C > 4) return gdkvm_fail(GDKVM_ERR_SHAPE, "stem_pack_s2d: C=%d (at most 4 input channels)", C);
    if (!w7 || !w4 || !gdkvm_aligned16(w4)) return gdkvm_fail(GDKVM_ERR_ARG, "stem_pack_s2d: null or unaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    hipLaunchKernelGGL(stem_pack_s2d_kernel, dim3(64), dim3(256), 0, static_cast<hipStream_t>(stream), w7, static_cast<bf16_t*>(w4), C, sk, sc, sr, ss);
    GDKVM_LAUNCH_CHECK("stem_pack_s2d_kernel");
    return GDKVM_OK;
}
