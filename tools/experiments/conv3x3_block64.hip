// conv3x3_block64.hip -- a whole residual block of the stride-4 stage (64 -> 64 -> 64 channels, no downsample branch) as ONE kernel:
//     out = relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2 + x)                    NHWC bf16, BatchNorm folded (inference build)
// with the intermediate activation resident in LDS -- it is never written to or read from memory.
//
// Why (round 6, profiles/r06_a_forward_cfg2_pmc_hbm.csv): the two layer1 blocks are four launches of conv3x3_c64_kernel at ~40 us, and
// each of them moves 146 MB (input + 0.8x halo re-reads + output) at 3.6 TB/s -- these 64-channel layers sit nearer their HBM bound
// than their MFMA bound.  A block as two launches moves x in (x1.8), y1 out, y1 in (x1.8), x in again (residual), out: 337 MB; fused
// it is x in (x1.6, the halo rows from L2) and out: 131 MB, one launch, one prologue, and the residual comes from the LDS band.
//
// Tiling.  A workgroup (4 waves, ONE per SIMD: the kernel owns the CU's LDS) walks tiles of 7 output rows x the full row (W <= 28):
//   x band   (7 + 4) x 30 pixels, 160 B apart (128 B of channels + 32 B padding: conflict-free 16-pixel b128 reads), by LDS-DMA,
//            double-buffered: the next tile's band lands behind this tile's second convolution;
//   y1 band  (7 + 2) x 30 pixels, same geometry: conv1 computes the 9 rows conv2 needs (1.29x its share; 1.14x in all), adds the
//            bias, applies the ReLU, ZEROES what lies outside the image (conv2's padding is zero, not conv1 of padding) and writes
//            bf16 -- the same rounding the two-launch form applies when it stores y1, so the results are bit-identical to it;
//   weights  of BOTH convolutions stay in registers for the workgroup's lifetime: wave (wm, wn) holds output channels 32wn .. +31
//            of both layers as MFMA A fragments, 2 x 144 registers (the one-wave-per-SIMD budget of 512 makes that possible);
//   conv1    16 m-tiles of 16 pixels (252 of 256 rows used), wave wm takes eight; conv2 13 m-tiles (196 of 208), 7 + 6;
//   epilogue of conv2: + bias + the block's input from the x band in LDS, ReLU, one 16-byte store per pixel and lane.
// Per tile: conv1 MFMAs | epilogue 1 -> y1 | barrier A (+ issue the next band's DMA into the buffer everyone has left) | conv2 MFMAs
// | barrier B (everyone is done with y1; the next band has landed) | epilogue 2.  Stores and DMA are never waited for where issued.
#include <atomic>
#include <type_traits>

#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

constexpr int CB_C = 64, CB_TH = 7, CB_TW = 28, CB_BW = CB_TW + 2, CB_PIX = 160;
constexpr int CB_XR = CB_TH + 4, CB_YR = CB_TH + 2;
constexpr int CB_XSLOTS = CB_XR * CB_BW * 10, CB_XPIECES = (CB_XSLOTS + 63) / 64, CB_XBYTES = CB_XPIECES * 1024;     // 52 KiB
constexpr int CB_YBYTES = ((CB_YR * CB_BW * CB_PIX + 1023) / 1024) * 1024;                                               // 43 KiB
constexpr int CB_LDS = 2 * CB_XBYTES + CB_YBYTES;
constexpr int CB_M1 = (CB_YR * CB_TW + 15) / 16, CB_M2 = (CB_TH * CB_TW + 15) / 16;                                      // 16, 13 m-tiles

__device__ const uint4 g_cb_zero16 = {0, 0, 0, 0};

#ifdef CB_DIAG
// diagnostic builds (tools/abl_block.py): s_memtime per wave of workgroup 0 at the phase boundaries of its first tiles
__device__ unsigned long long* g_cb_diag = nullptr;
#define CB_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && nt_done < 8) { unsigned long long t__; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); g_cb_diag[(nt_done * 4 + w) * 8 + (slot)] = t__; } } while (0)
#else
#define CB_STAMP(slot) do {} while (0)
#endif

struct Block64Args {
    const bf16_t* x; const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2; bf16_t* y;
    int N, H, W, tiles_y, ntiles;
};

__global__ __launch_bounds__(256, 1) void conv3x3_block64_kernel(Block64Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xband2 = smem;
    unsigned char* yband = smem + 2 * CB_XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;

    // tile order: virtual index v -> (frame, row tile) such that the row tiles of one frame run on ONE XCD (workgroup v lands on XCD
    // v % 8) at about the same time: their shared halo rows are then L2 hits.  Any bijection is correct; this one is the fast one.
    auto tile_of = [&](int v, int& n, int& ty) __attribute__((always_inline)) {
        const int per = 8 * a.tiles_y, blk = v / per, r = v - blk * per;
        n = blk * 8 + (r & 7);
        ty = r >> 3;
    };
    const int nvirt = ((a.N + 7) / 8) * 8 * a.tiles_y;                  // (frames padded to a multiple of 8: indices past N are skipped)

    // x-band fetch by LDS-DMA: piece j = w + 4u (64 consecutive 16-byte slots) is issued by wave w; slot d = 10 pix + c holds channel
    // chunk c of band pixel pix (c = 8, 9: padding).  The slot geometry does not depend on the tile: kept in registers.
    // (the slot geometry is recomputed per fetch -- a dozen integer operations per piece, once per tile: the weight fragments leave no
    // registers to keep it in)
    constexpr int PP = (CB_XPIECES + 3) / 4;
    auto fetch = [&](int n, int ty, int buf) __attribute__((always_inline)) {
        const int y0 = ty * CB_TH - 2;
        const bf16_t* origin = a.x + (((long long)n * a.H + y0) * a.W - 1) * CB_C;
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int j = w + 4 * u;
            if (j >= CB_XPIECES) break;
            const int d = 64 * j + lane, pix = d / 10, c = d - 10 * pix;
            const int by = pix / CB_BW, bx = pix - by * CB_BW, yy = y0 + by;
            const bool ok = c < 8 && pix < CB_XR * CB_BW && bx >= 1 && bx <= a.W && (unsigned)yy < (unsigned)a.H;
            const bf16_t* src = ok ? origin + ((by * a.W + bx) * CB_C + c * 8) : reinterpret_cast<const bf16_t*>(&g_cb_zero16);
            __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                reinterpret_cast<uintptr_t>(xband2 + buf * CB_XBYTES + 1024 * j)), 16, 0, 0);
        }
    };

    // first tile of this workgroup (skipping virtual indices past the last frame)
    int v = blockIdx.x, n = 0, ty = 0;
    auto advance = [&]() __attribute__((always_inline)) {               // -> the next valid virtual index at or after v, or nvirt
        while (v < nvirt) { tile_of(v, n, ty); if (n < a.N) break; v += gridDim.x; }
    };
    advance();
    if (v < nvirt) fetch(n, ty, 0);

    // y1 band: zero once -- the halo columns (0 and 29) and the alignment tail are never written again
    for (int i = tid; i < CB_YBYTES / 16; i += 256) reinterpret_cast<uint4*>(yband)[i] = make_uint4(0, 0, 0, 0);

    // weights of this wave's 32 output channels, both layers, all 18 k-steps, as A-operand fragments (packed copies: one contiguous KiB per load)
    bf16x8 wf1[2][18], wf2[2][18];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            wf1[nt][ks] = *reinterpret_cast<const bf16x8*>(a.w1 + ((size_t)((2 * wn + nt) * 18 + ks) * 64 + lane) * 8);
            wf2[nt][ks] = *reinterpret_cast<const bf16x8*>(a.w2 + ((size_t)((2 * wn + nt) * 18 + ks) * 64 + lane) * 8);
        }
    float bia1[8], bia2[8];
    {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(a.b1 + 32 * wn + 8 * g), p1 = *reinterpret_cast<const f32x4*>(a.b1 + 32 * wn + 8 * g + 4);
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(a.b2 + 32 * wn + 8 * g), q1 = *reinterpret_cast<const f32x4*>(a.b2 + 32 * wn + 8 * g + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { bia1[i] = p0[i]; bia1[4 + i] = p1[i]; bia2[i] = q0[i]; bia2[4 + i] = q1[i]; }
    }

    // this lane's pixel in each of the wave's m-tiles: LDS byte offset of tap (0, 0), channel chunk g.
    // conv1: pixel p of the 9 x 28 y1 patch -> x band pixel (py, px); conv2: pixel q of the 7 x 28 output tile -> y1 band pixel (qy, qx)
    // (conv1's lane writes its y1 pixel at band pixel (py, px + 1), channel bytes 64wn + 16g: pb1 + one pixel + 64wn)
    unsigned pb1[8], pb2[7];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int p = min(16 * (8 * wm + m) + li, CB_YR * CB_TW - 1);
        const int py = p / CB_TW, px = p - py * CB_TW;
        pb1[m] = (unsigned)((py * CB_BW + px) * CB_PIX + g * 16);
    }
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        const int q = min(16 * (7 * wm + m) + li, CB_TH * CB_TW - 1);
        const int qy = q / CB_TW, qx = q - qy * CB_TW;
        pb2[m] = (unsigned)((qy * CB_BW + qx) * CB_PIX + g * 16);
    }

    __syncthreads();                                                    // (vmcnt(0) + barrier: the first band has landed, y1 is zero)
    int cur = 0;
    int nt_done = 0;
    (void)nt_done;
    while (v < nvirt) {
        CB_STAMP(0);
        const unsigned char* xb = xband2 + cur * CB_XBYTES;
        const int y0 = ty * CB_TH;
        // the tile after this one (this workgroup's next): its band is requested at barrier A
        int v2 = v + gridDim.x, n2 = 0, ty2 = 0;
        while (v2 < nvirt) { tile_of(v2, n2, ty2); if (n2 < a.N) break; v2 += gridDim.x; }

        f32x4 acc[8][2];
        // One wave per SIMD and 288 registers of weights: the B operands are staged in TWO half-step buffers of four m-tiles (32 registers,
        // not 64).  Half-step h = 2 ks + (0: m-tiles 0..3, 1: m-tiles 4..): buffer P holds the first halves, Q the second; each is
        // refilled for the next k-step right after its MFMAs have been issued, so a read has the other half's eight MFMAs (128 cycles) to land.
        auto compute = [&](const unsigned char* band, const unsigned (&pb)[8], const bf16x8 (&wf)[2][18], auto nm_c) __attribute__((always_inline)) {
            constexpr int NM = decltype(nm_c)::value, NA = NM < 4 ? NM : 4, NB = NM - NA;
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto load_h = [&](bf16x8 (&o)[4], int ks, int m0, int cnt) __attribute__((always_inline)) {
                const int tap = ks >> 1, kh = ks & 1, dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (m < cnt) o[m] = *reinterpret_cast<const bf16x8*>(band + pb[m0 + m] + (dy * CB_BW + dx) * CB_PIX + kh * 64);
            };
            auto mfma_h = [&](const bf16x8 (&o)[4], int ks, int m0, int cnt) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (m < cnt) {
                        acc[m0 + m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], o[m], acc[m0 + m][0], 0, 0, 0);
                        acc[m0 + m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], o[m], acc[m0 + m][1], 0, 0, 0);
                    }
            };
            bf16x8 P[4], Q[4];
            load_h(P, 0, 0, NA);
            load_h(Q, 0, NA, NB);
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                mfma_h(P, ks, 0, NA);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < 18) load_h(P, ks + 1, 0, NA);
                mfma_h(Q, ks, NA, NB);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < 18) load_h(Q, ks + 1, NA, NB);
            }
        };

        // ---- conv1 on the x band -> y1 band (bias, ReLU, zero outside the image, bf16) ----
        compute(xb, pb1, wf1, std::integral_constant<int, 8>{});
        CB_STAMP(1);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int p = 16 * (8 * wm + m) + li, py = p / CB_TW, px = p - py * CB_TW, yy = y0 - 1 + py;
            const bool inside = p < CB_YR * CB_TW && (unsigned)yy < (unsigned)a.H && px < a.W;
            unsigned ow[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v0 = fmaxf(acc[m][q >> 1][2 * (q & 1)] + bia1[2 * q], 0.f), v1 = fmaxf(acc[m][q >> 1][2 * (q & 1) + 1] + bia1[2 * q + 1], 0.f);
                ow[q] = inside ? ((unsigned)f32_to_bf16(v0) | ((unsigned)f32_to_bf16(v1) << 16)) : 0u;
            }
            if (p < CB_YR * CB_TW) *reinterpret_cast<uint4*>(yband + pb1[m] + CB_PIX + 64 * wn) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
        CB_STAMP(2);
        __syncthreads();                                                // barrier A: y1 is complete; everyone has left the other x buffer
        if (v2 < nvirt) fetch(n2, ty2, cur ^ 1);                        // lands behind conv2
        CB_STAMP(3);

        // ---- conv2 on the y1 band ----
        {
            unsigned pb2w[8];
#pragma unroll
            for (int m = 0; m < 7; ++m) pb2w[m] = pb2[m];
            pb2w[7] = pb2[6];
            if (wm == 0) compute(yband, pb2w, wf2, std::integral_constant<int, 7>{});
            else compute(yband, pb2w, wf2, std::integral_constant<int, CB_M2 - 7>{});
        }
        CB_STAMP(4);
        __syncthreads();                                                // barrier B: everyone is done with y1; the next x band has landed
        CB_STAMP(5);

        // ---- epilogue 2: + bias + the block's input (x band, centre pixel) , ReLU, store ----
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            const int q = 16 * (7 * wm + m) + li;
            const int qy = q / CB_TW, qx = q - qy * CB_TW, yy = y0 + qy;
            if ((wm == 1 && m >= CB_M2 - 7) || q >= CB_TH * CB_TW || yy >= a.H || qx >= a.W) continue;
            const uint4 rr = *reinterpret_cast<const uint4*>(xb + ((qy + 2) * CB_BW + qx + 1) * CB_PIX + 64 * wn + 16 * g);
            const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
            unsigned ow[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v0 = acc[m][k >> 1][2 * (k & 1)] + bia2[2 * k] + __uint_as_float(rw[k] << 16);
                const float v1 = acc[m][k >> 1][2 * (k & 1) + 1] + bia2[2 * k + 1] + __uint_as_float(rw[k] & 0xffff0000u);
                ow[k] = (unsigned)f32_to_bf16(fmaxf(v0, 0.f)) | ((unsigned)f32_to_bf16(fmaxf(v1, 0.f)) << 16);
            }
            *reinterpret_cast<uint4*>(a.y + (((size_t)n * a.H + yy) * a.W + qx) * CB_C + 32 * wn + 8 * g) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
        CB_STAMP(6);
#ifdef CB_DIAG
        ++nt_done;
#endif
        v = v2; n = n2; ty = ty2; cur ^= 1;
    }
}

std::atomic<unsigned long long> g_cb_lds_done{0};

}  // namespace

#ifdef CB_DIAG
extern "C" void gdkvm_cb_diag_buffer(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cb_diag), &p, sizeof(p)); }
#endif

// internal entry used by gdkvm_conv_block_bias_act (below): 0 = launched, 1 = shape not served
static int conv3x3_block64_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int N, int H, int W, hipStream_t st)
{
    if (W > CB_TW || W < 1 || H < 1) return 1;
    Block64Args a;
    a.x = static_cast<const bf16_t*>(x); a.w1 = static_cast<const bf16_t*>(w1); a.b1 = b1; a.w2 = static_cast<const bf16_t*>(w2); a.b2 = b2;
    a.y = static_cast<bf16_t*>(y);
    a.N = N; a.H = H; a.W = W; a.tiles_y = (H + CB_TH - 1) / CB_TH;
    const long long nvirt = (long long)((N + 7) / 8) * 8 * a.tiles_y;
    if (nvirt > 0x7fffffffLL) return 1;
    a.ntiles = (int)nvirt;
    if (gdr_lds_optin(reinterpret_cast<const void*>(conv3x3_block64_kernel), g_cb_lds_done, CB_LDS, "conv_block_bias_act")) return 2;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0) cus = c;
    }
    const int grid = (int)(nvirt < cus ? nvirt : cus);                  // persistent: one workgroup per CU (it owns the CU's LDS), weights loaded once each
    hipLaunchKernelGGL(conv3x3_block64_kernel, dim3(grid), dim3(256), CB_LDS, st, a);
    return 0;
}

// C ABI (include/gdkvm.h): a residual block without a downsample branch, inference build
extern "C" int gdkvm_conv_block_bias_act(const void* x, const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2, void* y,
                                         int N, int C, int H, int W, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv_block_bias_act: only bf16 is implemented");
    if (N < 0 || H <= 0 || W <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_block_bias_act: N=%d H=%d W=%d", N, H, W);
    if (C != CB_C || W > CB_TW)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_block_bias_act: C=%d W=%d is not served (64 channels, rows of at most %d pixels): run the block as two "
                                           "gdkvm_conv_bias_act calls", C, W, CB_TW);
    if (N == 0) return GDKVM_OK;
    if (!x || !w1_packed || !bias1 || !w2_packed || !bias2 || !y) return gdkvm_fail(GDKVM_ERR_ARG, "conv_block_bias_act: null pointer");
    const void* ptrs[] = {x, w1_packed, bias1, w2_packed, bias2, y};
    for (const void* p : ptrs) if (!gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "conv_block_bias_act: pointers must be 16-byte aligned");
    if (x == y) return gdkvm_fail(GDKVM_ERR_ARG, "conv_block_bias_act: in-place operation is not supported (tiles read their neighbours' input rows)");
    if ((size_t)N * H * W * C >= (1ull << 31)) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_block_bias_act: tensor too large for 32-bit offsets");
    if (int rc = gdkvm_check_device()) return rc;
    const int rc = conv3x3_block64_launch(x, w1_packed, bias1, w2_packed, bias2, y, N, H, W, static_cast<hipStream_t>(stream));
    if (rc == 2) return GDKVM_ERR_LAUNCH;
    if (rc) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_block_bias_act: too many tiles");
    GDKVM_LAUNCH_CHECK("conv3x3_block64_kernel");
    return GDKVM_OK;
}
