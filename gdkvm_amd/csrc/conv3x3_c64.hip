// conv3x3_c64.hip -- hand-written 3x3 / stride 1 / pad 1 convolution for the 64 -> 64 channel layers at stride 4 (five of the
// twenty convolutions of a forward, 21 % of its time), NHWC bf16, with bias (+ residual) (+ ReLU) in the epilogue.
//
// Why not the implicit-GEMM library kernel: as a GEMM this layer is M = N*H*W pixels x K = 576 with only 64 output channels, so
// the "A matrix" (every input pixel repeated for its nine taps) is 9x the input -- 462 MB through L2 and the vector-memory path
// per call for a 51 MB tensor -- and the library kernels sit at ~62 us = 20 % of the MFMA rate whichever tile is chosen
// (round-1 sweep).  Here the nine taps are nine SHIFTED READS of one LDS image:
//   * a workgroup owns a 4 x TW pixel tile; its (4+2) x (TW+2) input halo band (64 channels = 128 B per pixel) is staged in LDS
//     once, pixels 160 B apart (128 B of channels + 32 B of padding: a ds_read_b128 of 16 consecutive pixels is then
//     conflict-free in every one of the instruction's four 16-lane groups, and every operand address is base + immediate);
//   * the weights never touch LDS: wave (wm, wn) keeps the 32 output channels 32wn.. as MFMA A-operand fragments for all 18
//     k-steps (tap x channel half) in 144 registers for the lifetime of the (persistent) workgroup;
//   * per k-step a wave reads one 16-pixel B fragment per m-tile (4 reads) and issues 8 v_mfma_f32_16x16x32_bf16; with the
//     weights as the A operand a lane ends with 4 consecutive output channels of one pixel -> 8-byte stores;
//   * the next tile's band streams into the other LDS buffer by LDS-DMA (global_load_lds, no registers: the 144 weight
//     registers leave none to stage through) while the current tile computes; pixels outside the image are fetched from a
//     16-byte zero constant (the DMA writes lane-linear, so the padding slots are fetched from there too).
// HBM traffic: input once (+ halo rows from L2), output once.  Arithmetic: fp32 accumulation over the same 576 products as the
// library kernel, one rounding after the epilogue.
#include <type_traits>

#include "gdkvm_common.hpp"

namespace {

constexpr int CV_C = 64;                 // input = output channels
constexpr int CV_TH = 4;                 // tile rows
constexpr int CV_PIX = 160;              // bytes between LDS pixels: 8 data chunks + 2 padding chunks of 16 B

__device__ const uint4 g_conv_zero16 = {0, 0, 0, 0};          // source of the zero padding

#ifdef CV_DIAG
// diagnostic builds (tools/abl_conv_tile.py c64stamps): s_memtime per wave of workgroup 0 at the phase boundaries of its first tiles
__device__ unsigned long long* g_cv_diag = nullptr;
#define CV_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && nt_done < 8) { unsigned long long t__; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); g_cv_diag[(nt_done * 4 + w) * 8 + (slot)] = t__; } } while (0)
#else
#define CV_STAMP(slot) do {} while (0)
#endif

struct Conv64Args {
    const bf16_t* x; const bf16_t* w; const float* bias; const bf16_t* res; bf16_t* y;
    int N, H, W, tiles_x, tiles_y, relu;
    int packed;                          // w is the gdkvm_conv3x3_pack_weights copy: fragment (kt, ks) = 1 KiB contiguous, rows in this kernel's channel order
};

template <int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(Conv64Args a)
{
    constexpr int BW = TW + 2, NPIX = CV_TH * TW, BAND_PIX = (CV_TH + 2) * BW;
    constexpr int SLOTS = BAND_PIX * 10, NPIECES = (SLOTS + 63) / 64, BAND_BYTES = NPIECES * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char band2[2 * BAND_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int ntiles = a.N * a.tiles_y * a.tiles_x;
    int nt_done = 0;
    (void)nt_done;
    CV_STAMP(6);

    // band fetch by LDS-DMA: piece j = w + 4u (64 consecutive 16-byte LDS slots) is issued by wave w; slot d = 10 pix + c holds
    // channel chunk c of band pixel pix (c = 8, 9: padding).  The slot geometry does not depend on the tile: kept in registers.
    constexpr int PP = (NPIECES + 3) / 4;
    int g_rel[PP], g_yx[PP];                               // element offset from the band's (0, 0) pixel; (by << 8 | bx), -1 = no data
#pragma unroll
    for (int u = 0; u < PP; ++u) {
        const int j = w + 4 * u, d = 64 * j + lane, pix = d / 10, c = d - 10 * pix;
        const int by = pix / BW, bx = pix - by * BW;
        g_rel[u] = (by * a.W + bx) * CV_C + c * 8;
        bool live = j < NPIECES && c < 8 && pix < BAND_PIX;
        if (a.tiles_x == 1) live = live && bx >= 1 && bx <= a.W;     // one tile per row: the column test does not depend on the tile
        g_yx[u] = live ? (by << 8 | bx) : -1;
    }
    const bool one_col = a.tiles_x == 1;
    auto fetch = [&](int tile, int buf) __attribute__((always_inline)) {
        const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x, ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
        const int y0 = ty * CV_TH - 1, x0 = tx * TW - 1;
        const bf16_t* origin = a.x + (((long long)n * a.H + y0) * a.W + x0) * CV_C;
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int j = w + 4 * u;
            if (j >= NPIECES) break;                       // (wave-uniform)
            const int yy = y0 + (g_yx[u] >> 8), xx = x0 + (g_yx[u] & 255);
            const bool ok = g_yx[u] >= 0 && (unsigned)yy < (unsigned)a.H && (one_col || (unsigned)xx < (unsigned)a.W);
            const bf16_t* src = ok ? origin + g_rel[u] : reinterpret_cast<const bf16_t*>(&g_conv_zero16);
            __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                reinterpret_cast<uintptr_t>(band2 + buf * BAND_BYTES + 1024 * j)), 16, 0, 0);
        }
    };

    // this lane's pixel in each of the wave's m-tiles: LDS byte offset of tap (0, 0), channel chunk g
    unsigned pbase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int p = min(16 * (4 * wm + m) + li, NPIX - 1);
        const int ty = p / TW, tx = p - ty * TW;
        pbase[m] = (unsigned)((ty * BW + tx) * CV_PIX + g * 16);
    }

    // Per tile: MFMAs from buffer `cur` | barrier (+ vmcnt(0): the other buffer's band, issued a whole tile ago, has landed)
    // | this tile's epilogue | DMA of the tile after next into `cur`.  The stores and the DMA are never waited for right
    // after being issued: the next wait is a tile of MFMAs later.
    int tile = blockIdx.x, cur = 0;
    if (tile < ntiles) fetch(tile, 0);                     // (the first band is on its way while the weights load)

    // weights of this wave's 32 output channels, all 18 k-steps, as A-operand fragments.  Which channel an MFMA row stands for is
    // free: row rho = 4 g' + r of n-tile nt is channel 32wn + 8g' + 4nt + r, so that a lane's two accumulator tiles hold EIGHT
    // consecutive channels of its pixel (one 16-byte store / residual load instead of two 8-byte ones).
    // (packed weights -- conv3x3_tile.hip's ct_channel order is this one -- make each of the 36 loads one contiguous KiB: the
    // prologue was a quarter of the kernel's time with 16 half-used cache lines per load)
    bf16x8 wf[2][18];
    if (a.packed) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 18; ++ks)
                wf[nt][ks] = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)((2 * wn + nt) * 18 + ks) * 64 + lane) * 8);
    } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                const int tap = ks >> 1, kh = ks & 1, co = 32 * wn + 8 * (li >> 2) + 4 * nt + (li & 3);
                wf[nt][ks] = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)co * 9 + tap) * CV_C + 32 * kh + 8 * g);
            }
    }
    float bia[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + 32 * wn + 8 * g + 4 * nt);
        bia[nt][0] = b4[0]; bia[nt][1] = b4[1]; bia[nt][2] = b4[2]; bia[nt][3] = b4[3];
    }

    __syncthreads();                                       // (vmcnt(0) + barrier: the first band has landed)
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x, 1);
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        const unsigned char* band = band2 + cur * BAND_BYTES;
        CV_STAMP(0);

        f32x4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // NM = m-tiles of this wave that hold pixels (wave wm = 1 of a 4 x 28 tile has three): no MFMA is spent on padding
        auto compute = [&](auto nm_c) __attribute__((always_inline)) {
            constexpr int NM = decltype(nm_c)::value;
            auto load_x = [&](bf16x8 (&xb)[4], int ks) __attribute__((always_inline)) {
                const int tap = ks >> 1, kh = ks & 1, dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    xb[m] = *reinterpret_cast<const bf16x8*>(band + pbase[m] + (dy * BW + dx) * CV_PIX + kh * 64);
            };
            auto mfmas = [&](const bf16x8 (&xb)[4], int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], xb[m], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], xb[m], acc[m][1], 0, 0, 0);
                }
            };
            bf16x8 xa[4], xb[4];
            load_x(xa, 0);
#pragma unroll
            for (int ks = 0; ks < 18; ks += 2) {           // operands one k-step ahead of the MFMAs that use them
                load_x(xb, ks + 1);
                mfmas(xa, ks);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < 18) load_x(xa, ks + 2);
                mfmas(xb, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        constexpr int NMT = (NPIX + 15) / 16, NM1 = NMT - 4;            // m-tiles in all, and those of the wm = 1 waves
        if (wm == 0 || NM1 == 4) compute(std::integral_constant<int, (NMT < 4 ? NMT : 4)>{});
        else if constexpr (NM1 > 0 && NM1 < 4) compute(std::integral_constant<int, (NM1 > 0 ? NM1 : 1)>{});

        CV_STAMP(1);
        __syncthreads();                                   // everyone is done with this band; the next one has landed
        CV_STAMP(2);

        CV_STAMP(3);

        // epilogue: lane (li, g) holds channels 32wn + 8g .. +7 of pixel 16(4wm+m) + li (tile nt: the four channels 4nt ..)
        const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x, ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int p = 16 * (4 * wm + m) + li;
            const int py = p / TW, px = p - py * TW, yy = ty * CV_TH + py, xx = tx * TW + px;
            if (p >= NPIX || yy >= a.H || xx >= a.W) continue;
            const size_t o = (((size_t)n * a.H + yy) * a.W + xx) * CV_C + 32 * wn + 8 * g;
            typedef float f32x2 __attribute__((ext_vector_type(2)));       // packed fp32 pairs: v_pk_add_f32 / v_pk_max_f32
            f32x2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = f32x2{acc[m][q >> 1][2 * (q & 1)], acc[m][q >> 1][2 * (q & 1) + 1]} + f32x2{bia[q >> 1][2 * (q & 1)], bia[q >> 1][2 * (q & 1) + 1]};
            if (a.res) {
                const uint4 rr = *reinterpret_cast<const uint4*>(a.res + o);
                const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += f32x2{__uint_as_float(rw[q] << 16), __uint_as_float(rw[q] & 0xffff0000u)};
            }
            const float lo = a.relu ? 0.f : -INFINITY;
            unsigned ow[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = __builtin_elementwise_max(v[q], f32x2{lo, lo});
                ow[q] = (unsigned)f32_to_bf16(v[q][0]) | ((unsigned)f32_to_bf16(v[q][1]) << 16);
            }
            *reinterpret_cast<uint4*>(a.y + o) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
        // the band of the tile after next, into the buffer just consumed.  Issued BEHIND the epilogue: the compiler cannot order
        // LDS-DMA against register loads and waits with vmcnt(0) for the residual -- ahead of the epilogue that wait also covered
        // the band just requested from HBM (+9 us per residual layer); the band still has a whole tile of MFMAs to land.
        if (tile + 2 * (int)gridDim.x < ntiles) fetch(tile + 2 * gridDim.x, cur);
        CV_STAMP(4);
#ifdef CV_DIAG
        ++nt_done;
#endif
    }
}

}  // namespace

#ifdef CV_DIAG
extern "C" void gdkvm_cv_diag_buffer(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cv_diag), &p, sizeof(p)); }
#endif

// internal entry used by gdkvm_conv_bias_act (conv_dispatch.hip): returns 0 on launch
int gdkvm_conv3x3_c64_launch(const void* x, const void* w, const float* bias, const void* residual, void* y, int N, int H, int W,
                             int relu, int packed, hipStream_t st)
{
    Conv64Args a;
    a.packed = packed;
    a.x = static_cast<const bf16_t*>(x); a.w = static_cast<const bf16_t*>(w); a.bias = bias;
    a.res = static_cast<const bf16_t*>(residual); a.y = static_cast<bf16_t*>(y);
    a.N = N; a.H = H; a.W = W; a.relu = relu;
    // tile width: 28 for rows that are a multiple of 28 pixels (EchoNet's stride-4 map), 16 for narrow maps, else 32 (ragged edge masked)
    const int TW = W % 28 == 0 ? 28 : (W <= 16 ? 16 : 32);
    a.tiles_x = (W + TW - 1) / TW;
    a.tiles_y = (H + CV_TH - 1) / CV_TH;
    const long long ntiles = (long long)N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 1;
    const int grid = (int)(ntiles < 512 ? ntiles : 512);   // persistent: two workgroups per CU, weights loaded once each
    if (TW == 28) hipLaunchKernelGGL(conv3x3_c64_kernel<28>, dim3(grid), dim3(256), 0, st, a);
    else if (TW == 16) hipLaunchKernelGGL(conv3x3_c64_kernel<16>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv3x3_c64_kernel<32>, dim3(grid), dim3(256), 0, st, a);
    return 0;
}
