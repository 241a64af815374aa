// conv3x3_tile.hip -- hand-written 3x3 / stride 1 / pad 1 convolution for the encoder / decoder layers with 128 .. 384 input
// channels, NHWC bf16, bias (+ residual) (+ ReLU) in the epilogue.  It replaces the composable_kernel
// template instantiations of round 1: the layers it serves work on small maps (14x14, 7x7, 28x28 at the EchoNet shapes), where the
// implicit-GEMM library kernels re-fetch every input pixel nine times through L2 and sit near 20 % of the MFMA rate.
//
//   * a workgroup (8 waves) owns a tile of up to 208 output pixels -- a whole 14x14 frame, four 7x7 frames, or a 7-row band of a
//     28x28 frame -- and 64 or 128 output channels;
//   * the input is walked in chunks of 64 channels: the chunk's halo band ((rows + 2) x (W + 2) pixels per frame, 128 B of
//     channels + 16 B of padding per pixel: a ds_read_b128 of 16 consecutive pixels is conflict-free and every operand
//     address is base + immediate) is staged in LDS by LDS-DMA, double-buffered: chunk c+1 streams in while chunk c computes;
//     pixels outside the frame and the padding slots are fetched from a 16-byte zero constant;
//   * the nine taps are nine SHIFTED READS of that image; the weights never touch LDS: a wave owns one 16-channel output tile
//     and streams its weight fragments (one 16-byte load per k-step, four k-steps ahead, straight from L2) -- each is used for
//     all the wave's pixel tiles (up to 13 MFMAs per load);
//   * weights are the A operand, so a lane ends with 4 consecutive output channels of one pixel (8-byte stores / residual loads).
// Arithmetic: fp32 accumulation over the same 9 C products as a library convolution, one rounding after the epilogue.
#include <atomic>
#include <type_traits>

#include "gdkvm_common.hpp"

namespace {

constexpr int CT_CK = 64;                // input channels per LDS chunk
constexpr int CT_PIX = 160;              // bytes between LDS pixels: 8 data chunks + 2 padding chunks of 16 B (144 B is NOT
                                         // conflict-free for ds_read_b128: its 16-lane groups mix two lane quarters)
constexpr int CT_SLOTS = CT_PIX / 16;    // 16-byte LDS slots per pixel
constexpr int CT_MAXMT = 13;             // 16-pixel tiles per workgroup tile (208 pixels)

__device__ const uint4 g_ct_zero16 = {0, 0, 0, 0};

struct ConvTileArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; const bf16_t* res; bf16_t* y;
    int N, H, W, C, K;                   // frames, map, input / output channels
    int fpt, th, tiles_y, ntiles;        // frames per tile, rows per tile, row tiles per frame, tiles in all
    int bw, bh, band_px, npieces;        // band geometry: (th + 2) x (W + 2) pixels per frame; 1 KiB DMA pieces per chunk
    int relu;
};

// Waves form a 2 (pixel-tile parity) x 4 (output-channel group) grid; a wave owns NTW 16-channel output tiles (the workgroup
// 64 NTW output channels) and every other pixel tile (7 of 13).  Per 32-deep k-step and workgroup: 56 KB of LDS reads (about half
// the LDS rate), 8 NTW KB of weight fragments over the vector-memory path, 7 NTW x 8 MFMAs -- MFMA-bound by construction.
template <int NTW>
__global__ __launch_bounds__(512) void conv3x3_tile_kernel(ConvTileArgs a)
{
    constexpr int NWN = 4, MW = 2, MTW = (CT_MAXMT + MW - 1) / MW;     // channel groups, pixel-tile groups, pixel tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char ct_band[];    // [2][npieces * 1024] | 1 KiB dump slot
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wn = w % NWN, wm = w / NWN;
    const int H = a.H, W = a.W, C = a.C, K = a.K, BW = a.bw;
    const int band_bytes = a.npieces * 1024, nchunk = C / CT_CK;
    const int tpix = a.fpt * a.th * W;                     // output pixels of a full tile
    const int co0 = blockIdx.y * (64 * NTW) + 16 * NTW * wn;           // this wave's output channels co0 .. co0 + 16 NTW - 1

    // DMA piece geometry (tile-invariant): piece j = w + 8u, slot d = 64 j + lane = CT_SLOTS pix + c; c >= 8 is padding
    constexpr int PP = 7;                                  // pieces per wave: npieces <= 56
    int g_rel[PP], g_yx[PP], g_fr[PP];
#pragma unroll
    for (int u = 0; u < PP; ++u) {
        const int j = w + 8 * u, d = 64 * j + lane, pix = d / CT_SLOTS, c = d - CT_SLOTS * pix;
        const int f = pix / (a.bh * BW), r = pix - f * (a.bh * BW), by = r / BW, bx = r - by * BW;
        g_rel[u] = ((f * H + by) * W + bx) * C + c * 8;
        const bool live = j < a.npieces && c < 8 && pix < a.band_px && bx >= 1 && bx <= W;
        g_yx[u] = live ? by : -1;
        g_fr[u] = f;
    }
    auto fetch = [&](int tile, int chunk, int buf) __attribute__((always_inline)) {
        const int ty = tile % a.tiles_y, fg = tile / a.tiles_y;          // row tile, frame group
        const int y0 = ty * a.th - 1;
        const bf16_t* origin = a.x + (((long long)fg * a.fpt * H + y0) * W - 1) * C + chunk * CT_CK;
        const int nfr = min(a.fpt, a.N - fg * a.fpt);                   // frames that exist in the last group
        // straight-line on purpose (always PP pieces: surplus ones land in a dump slot): a branch would make the compiler's vmcnt
        // counting conservative for the weight fragments in flight around it
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int j = w + 8 * u;
            const int yy = y0 + g_yx[u];
            const bool ok = g_yx[u] >= 0 && (unsigned)yy < (unsigned)H && g_fr[u] < nfr;
            const bf16_t* src = ok ? origin + g_rel[u] : reinterpret_cast<const bf16_t*>(&g_ct_zero16);
            unsigned char* dst = j < a.npieces ? ct_band + buf * band_bytes + 1024 * j : ct_band + 2 * band_bytes;
            __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(dst)), 16, 0, 0);
        }
    };

    // this lane's pixel in each of the wave's pixel tiles: LDS byte offset of tap (0, 0), channel chunk g
    unsigned pbase[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int p = min(16 * (wm + MW * m) + li, tpix - 1);
        const int f = p / (a.th * W), r = p - f * (a.th * W), py = r / W, px = r - py * W;
        pbase[m] = (unsigned)(((f * a.bh + py) * BW + px) * CT_PIX + g * 16);
    }
    f32x4 bias4[NTW];
    const bf16_t* wrow[NTW];                               // weights [K][3][3][C]: A operand rows of output tile nt
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bias4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (co0 + 16 * nt < K) bias4[nt] = *reinterpret_cast<const f32x4*>(a.bias + co0 + 16 * nt + 4 * g);
        wrow[nt] = a.w + (size_t)min(co0 + 16 * nt + li, K - 1) * 9 * C + 8 * g;
    }

    // k-steps of one tile: chunk-major, then tap, then channel half: ks = (chunk * 9 + tap) * 2 + kh
    const int nks = nchunk * 18;
    struct WF { bf16x8 f[NTW]; };
    auto wload = [&](int ks) __attribute__((always_inline)) {           // (wraps: a tile's last loads fetch the next tile's first k-steps)
        ks = ks >= nks ? ks - nks : ks;
#ifdef CT_ABL_NOW
        ks = 0;
#endif
        const int chunk = ks / 18, r = ks - 18 * chunk, tap = r >> 1, kh = r & 1;
        const size_t off = (size_t)tap * C + chunk * CT_CK + 32 * kh;
        WF o;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) o.f[nt] = *reinterpret_cast<const bf16x8*>(wrow[nt] + off);
        return o;
    };
    int tile = blockIdx.x, gc = 0;                          // gc: chunks processed so far; chunk gc lives in LDS buffer gc & 1
    if (tile < a.ntiles) fetch(tile, 0, 0);
    WF wf0 = wload(0), wf1 = wload(1), wf2 = wload(2), wf3 = wload(3);           // weight fragments, four k-steps ahead
    for (; tile < a.ntiles; tile += gridDim.x) {
        f32x4 acc[MTW][NTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < nchunk; ++chunk, ++gc) {
            const int buf = gc & 1;
            __syncthreads();                               // vmcnt(0) + barrier: this chunk's band has landed, the other buffer is free
            // the next band streams in behind this chunk's MFMAs: the tile's next chunk, or chunk 0 of the workgroup's next tile
            // (past the last tile: one more band of the last tile, into the buffer nobody reads again)
            {
                const bool more = chunk + 1 < nchunk;
                const int nt_tile = more ? tile : min(tile + (int)gridDim.x, a.ntiles - 1);
                fetch(nt_tile, more ? chunk + 1 : 0, buf ^ 1);
            }
            const unsigned char* band = ct_band + buf * band_bytes;
            auto load_x = [&](bf16x8 (&xb)[MTW], int r) __attribute__((always_inline)) {
                const int tap = r >> 1, kh = r & 1, dy = tap / 3, dx = tap - 3 * dy;
                const unsigned off = (unsigned)((dy * BW + dx) * CT_PIX + kh * 64);
#pragma unroll
                for (int m = 0; m < MTW; ++m) xb[m] = *reinterpret_cast<const bf16x8*>(band + pbase[m] + off);
            };
            auto mfmas = [&](const bf16x8 (&xb)[MTW], const WF& wfr) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr.f[nt], xb[m], acc[m][nt], 0, 0, 0);
            };
            const int ksb = chunk * 18;
            bf16x8 xa[MTW], xb[MTW];
            load_x(xa, 0);
            // (a rolled loop on purpose: fully unrolled, the scheduler hoists the LDS reads of many k-steps and spills 200 registers)
#pragma unroll 1
            for (int r = 0; r < 18; r += 2) {              // operands one k-step ahead of the MFMAs that use them
                load_x(xb, r + 1);
                mfmas(xa, wf0);
                wf0 = wf1; wf1 = wf2; wf2 = wf3; wf3 = wload(ksb + r + 4);
                __builtin_amdgcn_sched_barrier(0);
                if (r + 2 < 18) load_x(xa, r + 2);
                mfmas(xb, wf0);
                wf0 = wf1; wf1 = wf2; wf2 = wf3; wf3 = wload(ksb + r + 5);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: lane (li, g) holds channels co0 + 16nt + 4g .. +3 of pixel 16 (wm + MW m) + li
#ifndef CT_ABL_NOEPI
        {
            const int ty = tile % a.tiles_y, fg = tile / a.tiles_y;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int p = 16 * (wm + MW * m) + li;
                if (p >= tpix) continue;
                const int f = p / (a.th * W), r = p - f * (a.th * W), py = r / W, px = r - py * W;
                const int n = fg * a.fpt + f, yy = ty * a.th + py;
                if (n >= a.N || yy >= H) continue;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    if (co0 + 16 * nt >= K) continue;
                    const size_t o = (((size_t)n * H + yy) * W + px) * K + co0 + 16 * nt + 4 * g;
                    f32x4 v = acc[m][nt] + bias4[nt];
                    if (a.res) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(a.res + o);
                        v += f32x4{__uint_as_float(rr.x << 16), __uint_as_float(rr.x & 0xffff0000u), __uint_as_float(rr.y << 16), __uint_as_float(rr.y & 0xffff0000u)};
                    }
                    if (a.relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
                    *reinterpret_cast<uint2*>(a.y + o) = make_uint2((unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16),
                                                                    (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16));
                }
            }
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing may land in LDS after the workgroup is gone
}

}  // namespace

// Shapes this kernel serves: 3x3 / stride 1 / pad 1, C a multiple of 64, K a multiple of 16, and a map that tiles into <= 208
// pixels: whole frames of <= 208 pixels (several small frames per tile), or row bands of a wider frame.  Returns 0 when launched, 1
// when the shape is not covered (the caller falls back to the framework convolution + epilogue pass).
int gdkvm_conv3x3_tile_launch(const void* x, const void* w, const float* bias, const void* residual, void* y,
                              int N, int C, int H, int W, int K, int relu, hipStream_t st)
{
    if (C % CT_CK || K % 16 || W > 64 || W < 1 || H < 1 || N < 1) return 1;
    ConvTileArgs a;
    a.x = static_cast<const bf16_t*>(x); a.w = static_cast<const bf16_t*>(w); a.bias = bias;
    a.res = static_cast<const bf16_t*>(residual); a.y = static_cast<bf16_t*>(y);
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.relu = relu;
    const int maxpix = 16 * CT_MAXMT;
    if (H * W <= maxpix) { a.th = H; a.fpt = maxpix / (H * W); a.tiles_y = 1; if (a.fpt > N) a.fpt = N; }
    else { a.fpt = 1; a.th = maxpix / W; if (a.th < 1) return 1; a.tiles_y = (H + a.th - 1) / a.th; }
    a.bw = W + 2; a.bh = a.th + 2;
    // the halo band of a tile must fit the 56 DMA pieces (8 waves x 7) of a chunk: tiny maps take fewer frames, wide ones fewer rows
    auto pieces = [&]() { a.band_px = a.fpt * a.bh * a.bw; return (a.band_px * CT_SLOTS + 63) / 64; };
    while (pieces() > 56 && a.fpt > 1) --a.fpt;
    while (pieces() > 56 && a.th > 1) { --a.th; a.bh = a.th + 2; a.tiles_y = (H + a.th - 1) / a.th; }
    a.npieces = pieces();
    if (a.npieces > 56) return 1;
    const long long groups = (N + a.fpt - 1) / a.fpt;
    const long long ntiles = groups * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return 1;
    a.ntiles = (int)ntiles;
    const int ntw = K % 128 == 0 ? 2 : 1;                  // output channels per workgroup: 128 or 64
    const int gy = (K + 64 * ntw - 1) / (64 * ntw);
    const size_t lds = (size_t)2 * a.npieces * 1024 + 1024;
    int per = 256 / gy; if (per < 1) per = 1;
    const int gx = (int)(ntiles < per ? ntiles : per);     // persistent: one workgroup per CU
    auto setattr = [&](const void* fn) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 113 * 1024); };
    static std::atomic<unsigned long long> done_mask{0};   // per device; a lost race only repeats the idempotent call
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
        if (setattr(reinterpret_cast<const void*>(conv3x3_tile_kernel<2>)) != hipSuccess || setattr(reinterpret_cast<const void*>(conv3x3_tile_kernel<1>)) != hipSuccess) return 1;
        done_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    if (ntw == 2) hipLaunchKernelGGL(conv3x3_tile_kernel<2>, dim3(gx, gy), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(conv3x3_tile_kernel<1>, dim3(gx, gy), dim3(512), lds, st, a);
    return 0;
}
