// conv3x3_c64.hip -- hand-written 3x3 / stride 1 / pad 1 convolution for the 64 -> 64 channel layers at stride 4 (five of the
// twenty convolutions of a forward, 21 % of its time), NHWC bf16, with bias (+ residual) (+ ReLU) in the epilogue.
//
// Why not the implicit-GEMM library kernel: as a GEMM this layer is M = N*H*W pixels x K = 576 with only 64 output channels, so
// the "A matrix" (every input pixel repeated for its nine taps) is 9x the input -- 462 MB through L2 and the vector-memory path
// per call for a 51 MB tensor -- and the library kernels sit at ~62 us = 20 % of the MFMA rate whichever tile is chosen
// (tools/ck_sweep).  Here the nine taps are nine SHIFTED READS of one LDS image:
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
#include "gdr_device.hpp"

namespace {

constexpr int CV_C = 64;                 // input = output channels
constexpr int CV_TH = 4;                 // tile rows
constexpr int CV_PIX = 160;              // bytes between LDS pixels: 8 data chunks + 2 padding chunks of 16 B

__device__ const uint4 g_conv_zero16 = {0, 0, 0, 0};          // source of the zero padding

struct Conv64Args {
    const bf16_t* x; const bf16_t* w; const float* bias; const bf16_t* res; bf16_t* y;
    int N, H, W, tiles_x, tiles_y, relu;
};

#ifdef CONV_DIAG                                           // tools/abl_conv.py: s_memtime stamps of workgroup 0's waves, 8 per tile
__device__ unsigned long long g_conv_diag[4 * 64 * 8];
#define CONV_STAMP(k)                                                                                              \
    do {                                                                                                           \
        if (blockIdx.x == 0 && lane == 0 && tile - tile0 < 8) g_conv_diag[(w * 8 + (tile - tile0)) * 8 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define CONV_STAMP(k) do {} while (0)
#endif

template <int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(Conv64Args a)
{
    constexpr int BW = TW + 2, NPIX = CV_TH * TW, BAND_PIX = (CV_TH + 2) * BW;
    constexpr int SLOTS = BAND_PIX * 10, NPIECES = (SLOTS + 63) / 64, BAND_BYTES = NPIECES * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char band2[2 * BAND_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int ntiles = a.N * a.tiles_y * a.tiles_x;

    // weights of this wave's 32 output channels, all 18 k-steps, as A-operand fragments.  Which channel an MFMA row stands for is
    // free: row rho = 4 g' + r of n-tile nt is channel 32wn + 8g' + 4nt + r, so that a lane's two accumulator tiles hold EIGHT
    // consecutive channels of its pixel (one 16-byte store / residual load instead of two 8-byte ones).
    bf16x8 wf[2][18];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            const int tap = ks >> 1, kh = ks & 1, co = 32 * wn + 8 * (li >> 2) + 4 * nt + (li & 3);
            wf[nt][ks] = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)co * 9 + tap) * CV_C + 32 * kh + 8 * g);
        }
    // (the 144 weight registers leave no room for anything else that is tile-invariant: the bias and the DMA slot geometry
    // live in LDS and are read back where they are used -- with them in registers the kernel spills and runs 20 % slower)
    __shared__ float s_bias[CV_C];
    if (tid < CV_C) s_bias[tid] = a.bias[tid];
    // band fetch by LDS-DMA: piece j = w + 4u (64 consecutive 16-byte LDS slots) is issued by wave w; slot d = 10 pix + c holds
    // channel chunk c of band pixel pix (c = 8, 9: padding).  The slot geometry does not depend on the tile: kept in registers.
    constexpr int PP = (NPIECES + 3) / 4;
    __shared__ int s_geo[PP][256];                         // (chunk << 16 | by << 8 | bx) of a lane's slot in piece w + 4u; -1 = no data
#pragma unroll
    for (int u = 0; u < PP; ++u) {
        const int j = w + 4 * u, d = 64 * j + lane, pix = d / 10, c = d - 10 * pix;
        const int by = pix / BW, bx = pix - by * BW;
        s_geo[u][tid] = (j < NPIECES && c < 8 && pix < BAND_PIX) ? (c << 16 | by << 8 | bx) : -1;
    }
    auto fetch = [&](int n, int ty, int tx, int buf) __attribute__((always_inline)) {
        const int y0 = ty * CV_TH - 1, x0 = tx * TW - 1;
        const bf16_t* origin = a.x + (((long long)n * a.H + y0) * a.W + x0) * CV_C;
        const unsigned long long pz = reinterpret_cast<unsigned long long>(&g_conv_zero16);
        int geo[PP];                                       // all slots' geometry in ONE batch of LDS reads: read one by one, every piece
#pragma unroll                                             // paid an LDS round trip (the other workgroup keeps the LDS busy) before its
        for (int u = 0; u < PP; ++u) geo[u] = s_geo[u][tid];        // DMA could issue: 1.7 us per tile in the stamps
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int j = w + 4 * u;
            if (j >= NPIECES) break;                       // (wave-uniform)
            const int by = (geo[u] >> 8) & 255, bx = geo[u] & 255, c = geo[u] >> 16, yy = y0 + by, xx = x0 + bx;
            const bool ok = geo[u] >= 0 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const unsigned long long pa = reinterpret_cast<unsigned long long>(origin + ((by * a.W + bx) * CV_C + c * 8));
            const unsigned long long src = pz + ((pa - pz) & (0ull - (unsigned long long)ok));      // branch-free select
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const bf16_t*>(src), reinterpret_cast<__attribute__((address_space(3))) void*>(
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

    // A workgroup owns a CONTIGUOUS range of tiles (neighbouring tiles share halo rows in L2, and the (image, tile row, tile
    // column) coordinates advance by carries instead of three runtime divisions per tile and use).
    // Per tile: MFMAs from buffer `cur` | barrier (+ vmcnt(0): the other buffer's band, issued a whole tile ago, has landed)
    // | DMA of the tile after next into `cur` | this tile's epilogue.  The stores and the DMA are never waited for right
    // after being issued: the next wait is a tile of MFMAs later.
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = blockIdx.x * per;
    const int tend = min(ntiles, tile + per), tile0 = tile;
    (void)tile0;
    struct Coord { int n, ty, tx; };
    auto advance = [&](Coord& c) __attribute__((always_inline)) {
        if (++c.tx == a.tiles_x) { c.tx = 0; if (++c.ty == a.tiles_y) { c.ty = 0; ++c.n; } }
    };
    Coord c0, c2;                                          // current tile, and the tile two ahead (the next DMA)
    c0.tx = tile % a.tiles_x; c0.ty = (tile / a.tiles_x) % a.tiles_y; c0.n = tile / (a.tiles_x * a.tiles_y);
    c2 = c0;
    int cur = 0;
    if (tile < tend) fetch(c2.n, c2.ty, c2.tx, 0);
    __syncthreads();                                       // (vmcnt(0) + barrier: the first band has landed)
    advance(c2);
    if (tile + 1 < tend) fetch(c2.n, c2.ty, c2.tx, 1);
    advance(c2);
    for (; tile < tend; ++tile, cur ^= 1) {
        const unsigned char* band = band2 + cur * BAND_BYTES;
        CONV_STAMP(0);

        f32x4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // NM = m-tiles of this wave that hold pixels (wave wm = 1 of a 4 x 28 tile has three): no MFMA is spent on padding
        // The operand reads are inline asm with hand-placed waits: the compiler's own insertion drains the LDS queue
        // (lgkmcnt(0)) in front of every MFMA group, i.e. it waits for the prefetch it has just issued.  Reads run TWO k-steps
        // ahead (three register stages): before the MFMAs of k-step s the 2 NM reads of the two younger stages may still be in
        // flight -> s_waitcnt lgkmcnt(2 NM) (LDS returns in order).
        unsigned lbase[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) lbase[m] = (unsigned)(uintptr_t)band + pbase[m];
        auto compute = [&](auto nm_c) __attribute__((always_inline)) {
            constexpr int NM = decltype(nm_c)::value;
            bf16x8 x3[3][4];
            auto load_x = [&](auto ks_c) __attribute__((always_inline)) {
                constexpr int ks = decltype(ks_c)::value, tap = ks >> 1, kh = ks & 1, dy = tap / 3, dx = tap - 3 * dy;
                constexpr int off = (dy * BW + dx) * CV_PIX + kh * 64;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
#ifdef CONV_ABL_NOLDS
                    x3[ks % 3][m] = wf[0][ks];
#else
                    bf16x8 t;                              // (locals: clang rejects captured variables as asm operands in a generic lambda)
                    const unsigned ad = lbase[m];
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(ad), "n"(off));
                    x3[ks % 3][m] = t;
#endif
                }
            };
            auto mfmas = [&](auto ks_c, auto pending_c) __attribute__((always_inline)) {
                constexpr int ks = decltype(ks_c)::value;
                {   // the wait names the stage's registers as in/out operands: nothing that reads them may be scheduled above it
                    constexpr int P = decltype(pending_c)::value, st = ks % 3;
                    bf16x8 t0 = x3[st][0], t1 = x3[st][NM > 1 ? 1 : 0], t2 = x3[st][NM > 2 ? 2 : 0], t3 = x3[st][NM > 3 ? 3 : 0];
                    if constexpr (NM == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "n"(P) : "memory");
                    else if constexpr (NM == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(t0), "+v"(t1), "+v"(t2) : "n"(P) : "memory");
                    else if constexpr (NM == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(t0), "+v"(t1) : "n"(P) : "memory");
                    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(t0) : "n"(P) : "memory");
                    x3[st][0] = t0;
                    if constexpr (NM > 1) x3[st][1] = t1;
                    if constexpr (NM > 2) x3[st][2] = t2;
                    if constexpr (NM > 3) x3[st][3] = t3;
                }
#pragma unroll
                for (int m = 0; m < NM; ++m) {
#ifdef CONV_ABL_NOMFMA                                      // (tools/abl_conv.py: timing ablations, wrong results by design)
                    const bf16x8 t = x3[ks % 3][m];
                    asm volatile("" ::"v"(t));
#else
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], x3[ks % 3][m], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], x3[ks % 3][m], acc[m][1], 0, 0, 0);
#endif
                }
            };
            load_x(std::integral_constant<int, 0>{});
            load_x(std::integral_constant<int, 1>{});
            static_for<0, 18>([&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                if constexpr (ks + 2 < 18) load_x(std::integral_constant<int, ks + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                mfmas(ks_c, std::integral_constant<int, (ks + 2 < 18 ? 2 * NM : (ks + 1 < 18 ? NM : 0))>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        constexpr int NMT = (NPIX + 15) / 16, NM1 = NMT - 4;            // m-tiles in all, and those of the wm = 1 waves
        if (wm == 0 || NM1 == 4) compute(std::integral_constant<int, (NMT < 4 ? NMT : 4)>{});
        else if constexpr (NM1 > 0 && NM1 < 4) compute(std::integral_constant<int, (NM1 > 0 ? NM1 : 1)>{});

        CONV_STAMP(1);
        __syncthreads();                                   // everyone is done with this band; the next one has landed
        CONV_STAMP(2);

#ifndef CONV_ABL_NODMA
        if (tile + 2 < tend) fetch(c2.n, c2.ty, c2.tx, cur);
#endif
        advance(c2);
        CONV_STAMP(3);

        // epilogue: lane (li, g) holds channels 32wn + 8g .. +7 of pixel 16(4wm+m) + li (tile nt: the four channels 4nt ..)
        // (hoisting the residual loads above the DMA issue was tried: the extra live registers spill, 41 -> 50 us)
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_bias + 32 * wn + 8 * g), b1 = *reinterpret_cast<const f32x4*>(s_bias + 32 * wn + 8 * g + 4);
        const int y_t = c0.ty * CV_TH, x_t = c0.tx * TW;
        const size_t tile_o = (((size_t)c0.n * a.H + y_t) * a.W + x_t) * CV_C;         // (wave-uniform)
        advance(c0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int p = 16 * (4 * wm + m) + li;
            const int py = p / TW, px = p - py * TW;
            if (p >= NPIX || y_t + py >= a.H || x_t + px >= a.W) continue;
            const size_t o = tile_o + (unsigned)((py * a.W + px) * CV_C + 32 * wn + 8 * g);
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[m][0][r] + b0[r]; v[4 + r] = acc[m][1][r] + b1[r]; }
            if (a.res) {
                const uint4 rr = *reinterpret_cast<const uint4*>(a.res + o);
                const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[2 * q] += __uint_as_float(rw[q] << 16); v[2 * q + 1] += __uint_as_float(rw[q] & 0xffff0000u); }
            }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            uint4 out;
            out.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            out.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            out.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            out.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
#ifdef CONV_ABL_NOSTORE
            asm volatile("" ::"v"(out.x), "v"(out.y), "v"(out.z), "v"(out.w));
#else
            *reinterpret_cast<uint4*>(a.y + o) = out;
#endif
        }
        CONV_STAMP(4);
    }
}

}  // namespace

#ifdef CONV_DIAG
extern "C" void conv_diag_read(unsigned long long* host) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_diag), sizeof(g_conv_diag)); }
#endif

// internal entry used by gdkvm_conv_bias_act (conv_ck.hip): returns 0 on launch
int gdkvm_conv3x3_c64_launch(const void* x, const void* w, const float* bias, const void* residual, void* y, int N, int H, int W,
                             int relu, hipStream_t st)
{
    Conv64Args a;
    a.x = static_cast<const bf16_t*>(x); a.w = static_cast<const bf16_t*>(w); a.bias = bias;
    a.res = static_cast<const bf16_t*>(residual); a.y = static_cast<bf16_t*>(y);
    a.N = N; a.H = H; a.W = W; a.relu = relu;
    // tile width: 28 for rows that are a multiple of 28 pixels (EchoNet's stride-4 map), 16 for narrow maps, else 32 (ragged edge masked)
    const int TW = W % 28 == 0 ? 28 : (W <= 16 ? 16 : 32);
    a.tiles_x = (W + TW - 1) / TW;
    a.tiles_y = (H + CV_TH - 1) / CV_TH;
    const long long ntiles = (long long)N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 1;
#ifndef CONV_GRID
#define CONV_GRID 512                                      // persistent: two workgroups per CU, weights loaded once each
#endif
    const int grid = (int)(ntiles < CONV_GRID ? ntiles : CONV_GRID);
    if (TW == 28) hipLaunchKernelGGL(conv3x3_c64_kernel<28>, dim3(grid), dim3(256), 0, st, a);
    else if (TW == 16) hipLaunchKernelGGL(conv3x3_c64_kernel<16>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv3x3_c64_kernel<32>, dim3(grid), dim3(256), 0, st, a);
    return 0;
}
