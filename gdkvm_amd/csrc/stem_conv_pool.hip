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


// ---- training: the stem's WEIGHT GRADIENT (round 4), in the space-to-depth form the forward kernels use:
//   dW4[k][a][b][ch] = sum over output pixels  dy[n, cy, cx, k] * xs[n, cy + a - 2, cx + b - 2, ch]      (64 x 256, fp32)
// -- a product over the PIXEL axis, the slow axis of both operands, so both MFMA operands are read out of LDS transposed
// (ds_read_b64_tr_b16, as conv3x3_wgrad.hip does).  A workgroup (4 waves) walks tiles of 4 x 56 output pixels (7 k-steps of 32): the
// dy tile (64 channels, 144 B per pixel) arrives by LDS-DMA, the 7 x 59 pixel band of the space-to-depth image (32 B per pixel) is built
// from the NCHW frames through registers exactly like the forward's; wave w owns kernel row a = w: 4 output-channel tiles x 4 kernel
// columns (the 16 channels of band pixel (py + a, px + b) are one MFMA column tile) = 16 accumulator tiles, kept in registers over all
// the workgroup's tiles.  Three workgroups per CU cover each other's fetches.  Partials go to the workspace; a second kernel adds them
// in index order (deterministic, unlike the library's atomically accumulated gradient) and scatters the 7 x 7 taps that exist into the
// parameter's own layout.  The library kernel took 185 us (+ its casts and zero-fills) of a 7.0 ms training step.
constexpr int SW_TH = 4, SW_TW = 56;
constexpr int SW_BR = SW_TH + 3, SW_BC = SW_TW + 3;
constexpr int SW_BSLOTS = SW_BR * SW_BC * 2;                  // 16-byte slots of the band
constexpr int SW_BPT = (SW_BSLOTS + 255) / 256;               // slots per thread
constexpr int SW_BAND_BYTES = (SW_BSLOTS * 16 + 1023) / 1024 * 1024;
constexpr int SW_YPIX = 144;                                   // bytes between dy pixels in LDS (128 + 16: see conv3x3_wgrad.hip)
constexpr int SW_YPIECES = (SW_TH * SW_TW * 9 + 63) / 64;      // 1 KiB DMA pieces of the dy tile
constexpr int SW_KSTEPS = SW_TH * SW_TW / 32;
static_assert(SW_TH * SW_TW % 32 == 0 && SW_TW % 4 == 0, "tiles are whole k-steps; four consecutive pixels never straddle a row");

struct StemWgradArgs { const bf16_t* xf; const bf16_t* dy; float* part; int N, Cf, Hs, Ws, tiles_x, tiles_y; };

__global__ __launch_bounds__(256, 3) void stem_wgrad_kernel(StemWgradArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_band[SW_BAND_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char s_y[SW_YPIECES * 1024];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                     // wave = kernel row a
    const int q = li >> 2, p4 = (li & 3) * 4;                                    // transposed-read geometry (conv3x3_wgrad.hip)
    const int ntiles = a.N * a.tiles_y * a.tiles_x;
    const int H = 2 * a.Hs, W = 2 * a.Ws;

    f32x4 acc[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[kt][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x, ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
        const int cy0 = SW_TH * ty, cx0 = SW_TW * tx, y0 = cy0 - 2, x0 = cx0 - 2;
        // the band's slots from the frames, into registers (slot d = 2 pix + half: channels 2 half, 2 half + 1 x row parity x column pair).
        // (Requested a whole tile ahead instead -- asm loads with hand-written wait counts -- the kernel took 81 us against 76: the three
        // workgroups of a CU already cover each other's fetches.)
        unsigned stage[SW_BPT][4];
#pragma unroll
        for (int u = 0; u < SW_BPT; ++u) {
            const int d = 256 * u + tid, pix = d >> 1, half = d & 1;
            const int by = pix / SW_BC, bx = pix - by * SW_BC, yy = y0 + by, xx = x0 + bx;
            const bool ok = d < SW_BSLOTS && yy >= 0 && yy < a.Hs && xx >= 0 && xx < a.Ws;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 2 * half + (e >> 1);
                const bool live = ok && c < a.Cf;
                const bf16_t* src = live ? a.xf + ((((size_t)n * a.Cf + c) * H + 2 * yy + (e & 1)) * W + 2 * xx) : reinterpret_cast<const bf16_t*>(&g_stem_zero16);
                stage[u][e] = *reinterpret_cast<const unsigned*>(src);
            }
        }
        __syncthreads();                                   // everyone is done with the previous tile's buffers
        for (int j = w; j < SW_YPIECES; j += 4) {           // the dy tile by LDS-DMA (ninth slot of a pixel and pixels outside the image: zeros)
            const int d = 64 * j + lane, pix = d / 9, c = d - 9 * pix;
            const int py = pix / SW_TW, px = pix - py * SW_TW, cy = cy0 + py, cx = cx0 + px;
            const bool ok = c < 8 && pix < SW_TH * SW_TW && cy < a.Hs && cx < a.Ws;
            const bf16_t* src = ok ? a.dy + ((((size_t)n * a.Hs + cy) * a.Ws + cx) * 64 + c * 8) : reinterpret_cast<const bf16_t*>(&g_stem_zero16);
            __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(s_y + 1024 * j)), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < SW_BPT; ++u) {
            const int d = 256 * u + tid;
            if (d < SW_BSLOTS) *reinterpret_cast<uint4*>(s_band + 16 * d) = make_uint4(stage[u][0], stage[u][1], stage[u][2], stage[u][3]);
        }
        __syncthreads();                                   // (vmcnt(0) + barrier: band and dy tile are in LDS)
#pragma unroll 1
        for (int s = 0; s < SW_KSTEPS; ++s) {
            unsigned ya[2], xa[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int pp = 32 * s + 8 * g + 4 * hf + q, py = pp / SW_TW, px = pp - py * SW_TW;
                ya[hf] = (unsigned)(uintptr_t)(s_y + pp * SW_YPIX + p4 * 2);
                xa[hf] = (unsigned)(uintptr_t)(s_band + ((py + w) * SW_BC + px) * 32 + p4 * 2);
            }
            uint2 fa[4][2], fb[4][2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(fa[kt][hf]) : "v"(ya[hf] + 32 * kt) : "memory");
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(fb[b][hf]) : "v"(xa[hf] + 32 * b) : "memory");
            // (the fragments are operands of the wait so that no MFMA can be scheduled above it)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]),
                           "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[2][0]), "+v"(fb[2][1]), "+v"(fb[3][0]), "+v"(fb[3][1])
                         :: "memory");
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bf16x8 bf = __builtin_bit_cast(bf16x8, make_uint4(fb[b][0].x, fb[b][0].y, fb[b][1].x, fb[b][1].y));
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(fa[kt][0].x, fa[kt][0].y, fa[kt][1].x, fa[kt][1].y));
                    acc[kt][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[kt][b], 0, 0, 0);
                }
            }
        }
    }
    // partial block part[blockIdx.x][k 64][a 4][b 4][ch 16]: lane (li, g) holds rows 4g + r (output channel) of column li (channel)
    float* out = a.part + (size_t)blockIdx.x * (64 * 256);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[((16 * kt + 4 * g + r) * 4 + w) * 64 + 16 * b + li] = acc[kt][b][r];
}

// dW7[k][c][u][v] (element strides sk, sc, sr, ss) = sum over the gx partial blocks, in index order, of the space-to-depth element that
// stands for tap (u, v) of channel c: a = (u + 1) / 2, p = (u + 1) % 2 (stem_pack_s2d_kernel's map, inverted); the others are dropped.
__global__ __launch_bounds__(512) void stem_wgrad_reduce_kernel(const float* part, float* dw, int gx, int C, long long sk, long long sc, long long sr, long long ss)
{
    __shared__ float s_sum[8][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int per = (gx + 7) / 8, b0 = sl * per, b1 = min(gx, b0 + per);
    const int i = blockIdx.x * 64 + e;                      // (k, a, b, ch)
    // eight running sums (partials b0 + j, b0 + j + 8, ...), added up in index order: eight loads in flight instead of a dependent chain
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 8 <= b1; b += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] += part[(size_t)(b + j) * (64 * 256) + i];
    for (int j = 0; b < b1; ++b, ++j) s8[j] += part[(size_t)b * (64 * 256) + i];
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    s_sum[sl][e] = s;
    __syncthreads();
    if (sl == 0) {
        float t8 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t8 += s_sum[j][e];
        const int ch = i & 15, bb = (i >> 4) & 3, aa = (i >> 6) & 3, k = i >> 8;
        const int c = ch >> 2, p = (ch >> 1) & 1, qq = ch & 1, u = 2 * aa + p - 1, v = 2 * bb + qq - 1;
        if (c < C && u >= 0 && u <= 6 && v >= 0 && v <= 6) dw[k * sk + c * sc + u * sr + v * ss] = t8;
    }
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


// ---- training: weight gradient of the stem convolution ------------------------------------------------------------------------------
static int stem_wgrad_grid(int N, int Hs, int Ws, int* tiles_x, int* tiles_y)
{
    *tiles_x = (Ws + SW_TW - 1) / SW_TW; *tiles_y = (Hs + SW_TH - 1) / SW_TH;
    const long long ntiles = (long long)N * *tiles_x * *tiles_y;
    if (ntiles > 0x7fffffffLL) return -1;
    return (int)(ntiles < 768 ? ntiles : 768);             // persistent: three workgroups per CU
}

extern "C" size_t gdkvm_stem_wgrad_workspace_bytes(int N, int H, int W)
{
    int tx, ty;
    if (N <= 0 || H <= 0 || W <= 0) return 16;
    const int gx = stem_wgrad_grid(N, H / 2, W / 2, &tx, &ty);
    return gx <= 0 ? 16 : (size_t)gx * 64 * 256 * sizeof(float);
}

extern "C" int gdkvm_stem_wgrad_nchw(const void* x, const void* dy, float* dw, long long sk, long long sc, long long sr, long long ss,
                                     void* workspace, size_t workspace_bytes, int N, int C, int H, int W, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "stem_wgrad_nchw: only bf16 operands are implemented");
    if (C <= 0 || C > 4 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || N < 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "stem_wgrad_nchw: N=%d C=%d H=%d W=%d (at most 4 channels, H and W even)", N, C, H, W);
    if (N == 0) return GDKVM_OK;
    if (!x || !dy || !dw || !workspace || !gdkvm_aligned16(x) || !gdkvm_aligned16(dy) || !gdkvm_aligned16(workspace))
        return gdkvm_fail(GDKVM_ERR_ARG, "stem_wgrad_nchw: null or unaligned pointer");
    StemWgradArgs a;
    a.xf = static_cast<const bf16_t*>(x); a.dy = static_cast<const bf16_t*>(dy); a.part = static_cast<float*>(workspace);
    a.N = N; a.Cf = C; a.Hs = H / 2; a.Ws = W / 2;
    const int gx = stem_wgrad_grid(N, a.Hs, a.Ws, &a.tiles_x, &a.tiles_y);
    if (gx <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "stem_wgrad_nchw: too many tiles");
    const size_t need = (size_t)gx * 64 * 256 * sizeof(float);
    if (workspace_bytes < need) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "stem_wgrad_nchw: workspace %zu < %zu bytes", workspace_bytes, need);
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(gx), dim3(256), 0, st, a);
    GDKVM_LAUNCH_CHECK("stem_wgrad_kernel");
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64 * 256 / 64), dim3(512), 0, st, static_cast<const float*>(workspace), dw, gx, C, sk, sc, sr, ss);
    GDKVM_LAUNCH_CHECK("stem_wgrad_reduce_kernel");
    return GDKVM_OK;
}
