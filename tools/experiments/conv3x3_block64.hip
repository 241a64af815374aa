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
constexpr int CB_GEO = CB_XPIECES * 64 * 4;                          // the band's slot geometry, one word per 16-byte slot (13 KiB)
constexpr int CB_LDS = CB_XBYTES + 2 * CB_YBYTES + 16 + CB_GEO;      // (+ the producers' arrival counter, + the geometry table)
constexpr int CB_M1 = (CB_YR * CB_TW + 15) / 16, CB_M2 = (CB_TH * CB_TW + 15) / 16;                                      // 16, 13 m-tiles

__device__ const uint4 g_cb_zero16 = {0, 0, 0, 0};

#ifdef CB_DIAG
// diagnostic builds (tools/abl_block.py): s_memtime per wave of workgroup 0 at the phase boundaries of its first tiles
__device__ unsigned long long* g_cb_diag = nullptr;
#define CB_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && nt_done < 8) { unsigned long long t__; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); g_cb_diag[(nt_done * 8 + w) * 8 + (slot)] = t__; } } while (0)
#else
#define CB_STAMP(slot) do {} while (0)
#endif

struct Block64Args {
    const bf16_t* x; const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2; bf16_t* y;
    int N, H, W, tiles_y, ntiles;
};

// v2 (round 6, second form): EIGHT waves, two per SIMD.  Waves 0-3 are the block's FIRST convolution (their 144 registers of weights are
// w1's), waves 4-7 the SECOND (w2's), one tile behind: in phase i the producers turn tile i's x band into y1[i & 1] while the consumers turn
// y1[(i - 1) & 1] into tile i - 1's output.  Wave w and wave w + 4 share a SIMD, so one role's epilogue (VALU, LDS / global stores) runs
// beside the other role's MFMAs -- what two workgroups per CU give conv3x3_c64_kernel, and what the one-wave-per-SIMD form above lacked.
//   LDS: x band 52 KiB (single: the residual is read from memory, L2-hot) + y1 band 2 x 43 KiB.
//   Per phase two s_barriers that all eight waves join (M behind the last MFMA pass, Y behind the epilogues: see the loop).
//   Each role works in two passes of at most four m-tiles (MFMAs, then that pass's epilogue): 32 accumulator registers, not 64 -- 144 weights
//   + 32 accumulators + 32 operand staging + addresses fit the 256-register budget of two waves per SIMD.
__global__ __launch_bounds__(512, 1) void conv3x3_block64_kernel(Block64Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xband = smem;
    unsigned char* yband2 = smem + CB_XBYTES;
    unsigned* arrivals = reinterpret_cast<unsigned*>(smem + CB_XBYTES + 2 * CB_YBYTES);     // producers done with the x band, counted up for ever
    unsigned* geo = reinterpret_cast<unsigned*>(smem + CB_XBYTES + 2 * CB_YBYTES + 16);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), role = w >> 2, wr = w & 3, wm = wr >> 1, wn = wr & 1;

    // tile order: virtual index v -> (frame, row tile) such that the row tiles of one frame run on ONE XCD (workgroup v lands on XCD
    // v % 8) at about the same time: their shared halo rows are then L2 hits.  Any bijection is correct; this one is the fast one.
    auto tile_of = [&](int v, int& n, int& ty) __attribute__((always_inline)) {
        const int per = 8 * a.tiles_y, blk = v / per, r = v - blk * per;
        n = blk * 8 + (r & 7);
        ty = r >> 3;
    };
    const int nvirt = a.ntiles;                                         // (frames padded to a multiple of 8: indices past N are skipped)
    auto next_valid = [&](int v, int& n, int& ty) __attribute__((always_inline)) {
        while (v < nvirt) { tile_of(v, n, ty); if (n < a.N) break; v += gridDim.x; }
        return v;
    };

    // x-band fetch by LDS-DMA (producer waves only): piece j = wr + 4u (64 consecutive 16-byte slots); slot d = 10 pix + c holds channel chunk
    // c of band pixel pix (c = 8, 9: padding).  The slot geometry does not depend on the tile; it is kept as a table in LDS (one word per slot:
    // element offset from the band's origin | band row << 16 | 1 << 24 if the slot carries data), built once -- recomputing it costs ~25
    // integer instructions per piece (4 k cycles per tile at the producers' priority), keeping it in registers costs 13 that do not exist.
    constexpr int PP = (CB_XPIECES + 3) / 4;
    for (int d = tid; d < CB_XPIECES * 64; d += 512) {
        const int pix = d / 10, c = d - 10 * pix, by = pix / CB_BW, bx = pix - by * CB_BW;
        const bool ok = c < 8 && pix < CB_XR * CB_BW && bx >= 1 && bx <= a.W;
        geo[d] = ok ? (unsigned)((by * a.W + bx) * CB_C + c * 8) | ((unsigned)by << 16) | (1u << 24) : 0u;
    }
    auto fetch = [&](int n, int ty) __attribute__((always_inline)) {
        const int y0 = ty * CB_TH - 2;
        const bf16_t* origin = a.x + (((long long)n * a.H + y0) * a.W - 1) * CB_C;
        unsigned e[PP];
#pragma unroll
        for (int u = 0; u < PP; ++u) e[u] = geo[64 * min(wr + 4 * u, CB_XPIECES - 1) + lane];
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int j = wr + 4 * u;
            if (j >= CB_XPIECES) break;
            const int yy = y0 + (int)((e[u] >> 16) & 0xff);
            const bool ok = (e[u] >> 24) && (unsigned)yy < (unsigned)a.H;
            const bf16_t* src = ok ? origin + (e[u] & 0xffff) : reinterpret_cast<const bf16_t*>(&g_cb_zero16);
            __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                reinterpret_cast<uintptr_t>(xband + 1024 * j)), 16, 0, 0);
        }
    };
    auto barrier_lds = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto barrier_all = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // this workgroup's tiles, in order: the producers are at tile (vP, nP, tyP), the consumers one behind
    int vP = blockIdx.x, nP = 0, tyP = 0;
    vP = next_valid(vP, nP, tyP);
    __syncthreads();                                                    // (the geometry table is complete)
    if (role == 0 && vP < nvirt) fetch(nP, tyP);

    // y1 bands: zero once -- the halo columns (0 and 29) and the alignment tails are never written again (and the arrival counter behind them)
    for (int i = tid; i < (2 * CB_YBYTES + 16) / 16; i += 512) reinterpret_cast<uint4*>(yband2)[i] = make_uint4(0, 0, 0, 0);

    // this wave's layer: weights of its 32 output channels, all 18 k-steps, as A-operand fragments (packed copies: one contiguous KiB per load)
    const bf16_t* wsrc = role == 0 ? a.w1 : a.w2;
    const float* bsrc = role == 0 ? a.b1 : a.b2;
    bf16x8 wf[2][18];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 18; ++ks)
            wf[nt][ks] = *reinterpret_cast<const bf16x8*>(wsrc + ((size_t)((2 * wn + nt) * 18 + ks) * 64 + lane) * 8);
    float bia[8];
    {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(bsrc + 32 * wn + 8 * g), p1 = *reinterpret_cast<const f32x4*>(bsrc + 32 * wn + 8 * g + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { bia[i] = p0[i]; bia[4 + i] = p1[i]; }
    }

    // this lane's pixel in each of the wave's m-tiles: LDS byte offset of tap (0, 0), channel chunk g.
    // producers: pixel p of the 9 x 28 y1 patch -> x band pixel (py, px) (and the lane WRITES y1 band pixel (py, px + 1): pb + one pixel + 64wn);
    // consumers: pixel q of the 7 x 28 output tile -> y1 band pixel (qy, qx)
    const int mt0 = role == 0 ? 8 * wm : 7 * wm, npix = role == 0 ? CB_YR * CB_TW : CB_TH * CB_TW;
    unsigned pb[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int p = min(16 * (mt0 + m) + li, npix - 1);
        const int py = p / CB_TW, px = p - py * CB_TW;
        pb[m] = (unsigned)((py * CB_BW + px) * CB_PIX + g * 16);
    }

    f32x4 acc[4][2];
    // MFMAs of one pass: `cnt` (<= 4) m-tiles starting at the wave's m-tile m0, all 18 k-steps.  B operands in two half buffers (two
    // m-tiles each), refilled for the next k-step right behind their MFMAs.
    // `early` runs behind the first half-step's MFMAs: the consumers request their residual rows there, NOT in front of the pass -- the
    // compiler's wait-count pass cannot tell the roles apart, assumes the producers' LDS-DMA may be pending at every LDS read that follows
    // a loop back-edge and puts vmcnt(0) on the pass's first ds_read; a global load issued before that read would be waited for in full.
    auto mfma_pass = [&](const unsigned char* band, int m0, auto cnt_c, auto early) __attribute__((always_inline)) {
        constexpr int CNT = decltype(cnt_c)::value, NA = CNT < 2 ? CNT : 2, NB = CNT - NA;
#pragma unroll
        for (int m = 0; m < CNT; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load_h = [&](bf16x8 (&o)[2], int ks, int mo, int cnt) __attribute__((always_inline)) {
            const int tap = ks >> 1, kh = ks & 1, dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (m < cnt) o[m] = *reinterpret_cast<const bf16x8*>(band + pb[m0 + mo + m] + (dy * CB_BW + dx) * CB_PIX + kh * 64);
        };
        auto mfma_h = [&](const bf16x8 (&o)[2], int ks, int mo, int cnt) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (m < cnt) {
                    acc[mo + m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], o[m], acc[mo + m][0], 0, 0, 0);
                    acc[mo + m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], o[m], acc[mo + m][1], 0, 0, 0);
                }
        };
        bf16x8 P0[2], Q0[2];                                            // (one half-step ahead: the SIMD's other wave covers the rest of the latency)
        load_h(P0, 0, 0, NA);
        load_h(Q0, 0, NA, NB);
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            mfma_h(P0, ks, 0, NA);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) { early(); __builtin_amdgcn_sched_barrier(0); }
            if (ks + 1 < 18) load_h(P0, ks + 1, 0, NA);
            mfma_h(Q0, ks, NA, NB);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < 18) load_h(Q0, ks + 1, NA, NB);
        }
    };

    barrier_all();                                                      // (the first band has landed -- its issuers waited --, y1 is zero)
    int vC = nvirt, nC = 0, tyC = 0;                                    // the consumers' tile: none yet
    int nt_done = 0;
    (void)nt_done;
    // ONE barrier per phase that all eight waves join -- Y, behind the epilogues: y1[i & 1] is complete (lgkmcnt(0)), the consumers are done
    // with y1[(i - 1) & 1], and x(i + 1) has landed (the producers' vmcnt(0)).  The x band being free is the PRODUCERS' business alone (the
    // consumers never read it): behind their last MFMA pass the four producer waves meet at a counter in LDS (one ds_add each, a short
    // s_sleep spin -- they run the same work and arrive together) and request the next band at once, ~6 k cycles ahead of Y, instead of
    // waiting at a full barrier for the slower consumers.  The consumers raise their wave priority for their MFMA passes (their chain --
    // MFMAs, residual loads, stores -- is the longer one; the producers fill in during the consumers' epilogues).
    unsigned p_phases = 0;                                               // phases in which the producers were active (the counter's target / 4)
    for (int phase = 0; vP < nvirt || vC < nvirt; ++phase) {
        // the producers' NEXT tile (requested behind barrier M)
        int vN = nvirt, nN = 0, tyN = 0;
        if (vP < nvirt) vN = next_valid(vP + gridDim.x, nN, tyN);
        if (role == 0) {
            // ================= producers: tile (nP, tyP): x band -> y1[phase & 1] =================
            CB_STAMP(0);
            if (vP < nvirt) {
                unsigned char* yb = yband2 + (phase & 1) * CB_YBYTES;
                const int y0 = tyP * CB_TH;
                auto epi1 = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int p = 16 * (mt0 + m0 + m) + li, py = p / CB_TW, px = p - py * CB_TW, yy = y0 - 1 + py;
                        const bool inside = p < CB_YR * CB_TW && (unsigned)yy < (unsigned)a.H && px < a.W;
                        unsigned ow[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float v0 = fmaxf(acc[m][q >> 1][2 * (q & 1)] + bia[2 * q], 0.f), v1 = fmaxf(acc[m][q >> 1][2 * (q & 1) + 1] + bia[2 * q + 1], 0.f);
                            ow[q] = inside ? ((unsigned)f32_to_bf16(v0) | ((unsigned)f32_to_bf16(v1) << 16)) : 0u;
                        }
                        if (p < CB_YR * CB_TW) *reinterpret_cast<uint4*>(yb + pb[m0 + m] + CB_PIX + 64 * wn) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    }
                };
                auto nothing = [] {};
                mfma_pass(xband, 0, std::integral_constant<int, 4>{}, nothing);
                CB_STAMP(1);
                epi1(0);
                CB_STAMP(2);
                mfma_pass(xband, 4, std::integral_constant<int, 4>{}, nothing);
                CB_STAMP(3);
                // producers-only meeting: every producer wave has issued its last read of x(i) -- and received it (the MFMAs that used it
                // are issued) -- before the band is overwritten
                ++p_phases;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * p_phases) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                if (vN < nvirt) fetch(nN, tyN);                          // lands behind the epilogue and the consumers' remaining work
                CB_STAMP(4);
                epi1(4);
                CB_STAMP(5);
            }
            barrier_all();                                               // Y (vmcnt(0): this wave's pieces of the next band have landed)
            CB_STAMP(6);
        } else {
            // ================= consumers: y1[(phase - 1) & 1] -> tile (nC, tyC) =================
            CB_STAMP(0);
            if (vC < nvirt) {
                const unsigned char* yb = yband2 + ((phase - 1) & 1) * CB_YBYTES;
                const int y0 = tyC * CB_TH;
                int li_o;                                                // (= li through an asm: pixel offsets are recomputed here, not hoisted and spilled)
                asm volatile("v_mov_b32 %0, %1" : "=v"(li_o) : "v"(li));
                // the block's input at this lane's output pixels (the residual): requested BEFORE a pass's MFMAs, used after them -- the
                // loads are L2 hits (this workgroup fetched the rows a tile ago) but still ~2 k cycles away
                uint4 rr[4];
                auto res_load = [&](int m0, int cnt) __attribute__((always_inline)) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (m >= cnt) continue;
                        const int q = min(16 * (mt0 + m0 + m) + li_o, CB_TH * CB_TW - 1), qy = q / CB_TW, qx = min(q - qy * CB_TW, a.W - 1), yy = min(y0 + qy, a.H - 1);
                        rr[m] = *reinterpret_cast<const uint4*>(a.x + (((size_t)nC * a.H + yy) * a.W + qx) * CB_C + 32 * wn + 8 * g);
                    }
                };
                auto epi2 = [&](int m0, int cnt) __attribute__((always_inline)) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (m >= cnt) continue;
                        const int q = 16 * (mt0 + m0 + m) + li_o, qy = q / CB_TW, qx = q - qy * CB_TW, yy = y0 + qy;
                        if (q >= CB_TH * CB_TW || yy >= a.H || qx >= a.W) continue;
                        const unsigned rw[4] = {rr[m].x, rr[m].y, rr[m].z, rr[m].w};
                        unsigned ow[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float v0 = acc[m][k >> 1][2 * (k & 1)] + bia[2 * k] + __uint_as_float(rw[k] << 16);
                            const float v1 = acc[m][k >> 1][2 * (k & 1) + 1] + bia[2 * k + 1] + __uint_as_float(rw[k] & 0xffff0000u);
                            ow[k] = (unsigned)f32_to_bf16(fmaxf(v0, 0.f)) | ((unsigned)f32_to_bf16(fmaxf(v1, 0.f)) << 16);
                        }
                        *reinterpret_cast<uint4*>(a.y + (((size_t)nC * a.H + yy) * a.W + qx) * CB_C + 32 * wn + 8 * g) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    }
                };
                __builtin_amdgcn_s_setprio(2);
                mfma_pass(yb, 0, std::integral_constant<int, 4>{}, [&] { res_load(0, 4); });
                __builtin_amdgcn_s_setprio(0);
                CB_STAMP(1);
                epi2(0, 4);
                CB_STAMP(2);
                __builtin_amdgcn_s_setprio(2);
                if (wm == 0) mfma_pass(yb, 4, std::integral_constant<int, 3>{}, [&] { res_load(4, 3); });
                else mfma_pass(yb, 4, std::integral_constant<int, CB_M2 - 7 - 4>{}, [&] { res_load(4, CB_M2 - 7 - 4); });
                __builtin_amdgcn_s_setprio(0);
                CB_STAMP(3);
                CB_STAMP(4);
                epi2(4, wm == 0 ? 3 : CB_M2 - 7 - 4);
                CB_STAMP(5);
            }
            barrier_lds();                                               // Y
            CB_STAMP(6);
        }
#ifdef CB_DIAG
        ++nt_done;
#endif
        // advance: the consumers take over the producers' tile, the producers move on
        vC = vP; nC = nP; tyC = tyP;
        vP = vN; nP = nN; tyP = tyN;
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
    hipLaunchKernelGGL(conv3x3_block64_kernel, dim3(grid), dim3(512), CB_LDS, st, a);
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
