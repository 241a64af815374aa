// conv3x3_wgrad.hip -- weight gradient of the 3x3 / stride 1 / pad 1 convolutions the hand-written kernels serve in training
// (SURVEY.md §8f row n1):   dW[k][ty][tx][c] = sum over pixels  dy[n, y, x, k] * x[n, y + ty - 1, x + tx - 1, c],
// NHWC bf16 operands, fp32 result.  It is a product over the PIXEL axis, which is the slow axis of both operands in memory, so
// both MFMA operands are read out of LDS transposed (ds_read_b64_tr_b16 with per-lane addresses):
//   * a workgroup owns one (64 output channels) x (9 taps x 64 input channels) block of dW and walks pixel tiles
//     (a whole 14x14 frame, four 7x7 frames, a 7-row band of a 28x28 frame: up to 224 pixels = 7 k-steps of 32), keeping
//     its 36 accumulator tiles per wave (4 output-channel tiles x 9 taps of input-channel tile w) in registers throughout;
//   * per tile the x halo band ((rows + 2) x (W + 2) pixels, 64 channels, zero borders) and the dy tile (64 channels, padding
//     pixels zero) arrive by LDS-DMA, pixels 144 B apart; a tap is a shifted window of the band, so a 32-pixel k-step costs a
//     wave 4 dy fragments + 9 x fragments (26 transposed 8-byte reads) for 36 MFMAs;
//   * (round 4) the workgroup is EIGHT waves, one per CU: waves 4 .. 7 own the same accumulator tiles as waves 0 .. 3, the two halves walk
//     separate tile streams with their own buffers, in step only with themselves (a barrier of four waves is a counter in LDS), and at the
//     end the halves' accumulators are added through LDS -- as two independent four-wave workgroups per CU (rounds 2-3) each half wrote a
//     147 KB partial block: 75 MB written and read back per layer;
//   * every workgroup writes its partial block (fp32) to the workspace; a second kernel adds the partials in a fixed order
//     (deterministic) into dW [K][C][3][3].
#include <atomic>
#include <type_traits>

#include "gdkvm_common.hpp"

namespace {

constexpr int WG_PIX = 144;                 // bytes between LDS pixels (128 of channels + 16 of padding: the transposed reads of
                                            // 4 pixel rows x 4 channel quads then fall on distinct banks)
constexpr int WG_SLOTS = WG_PIX / 16;       // 16-byte DMA slots per pixel (the ninth is padding)
constexpr int WG_MAXPIX = 224;              // pixels per tile (7 k-steps of 32)
constexpr int WG_KSTEPS = WG_MAXPIX / 32;

__device__ const uint4 g_wg_zero16 = {0, 0, 0, 0};

struct WgradArgs {
    const bf16_t* x; const bf16_t* dy; float* part;
    int N, H, W, C, K;
    int fpt, th, tiles_y, ntiles;           // frames per tile, rows per tile, row tiles per frame, tiles in all
    int bw, bh, band_px, tpix;              // band geometry, output pixels of a full tile
    int npx, npy;                           // 1 KiB DMA pieces of the band and of the dy tile
    float inv_band, inv_bw, inv_tw, inv_w;  // reciprocals for index arithmetic (see conv3x3_tile.hip)
    int cblocks;                            // C / 64: blockIdx.y = kb * cblocks + cb
};

__device__ __forceinline__ int wg_div(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_kernel(WgradArgs a)
{
    // 2 x (x band [npx KiB] | dy tile [npy KiB]); at the end the same memory carries one wave half's accumulators to the other (144 KiB)
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = w8 & 3, h = w8 >> 2;                       // wave = input-channel tile w of the block, in wave half h
    const int H = a.H, W = a.W, C = a.C, K = a.K, BW = a.bw;
    const int cb = blockIdx.y % a.cblocks, kb = blockIdx.y / a.cblocks;
    const int bufsz = (a.npx + a.npy) * 1024;

    f32x4 acc[4][9];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read lane geometry: lane 4q + p of 16-lane group g addresses pixel row (8g + 4 half + q), channels 4p .. 4p+3
    const int q = li >> 2, p4 = (li & 3) * 4;

    // DMA of a tile into this half's buffer: the x band (64 channels cb) and the dy tile (64 channels kb); geometry computed per piece (no tables)
    unsigned char* s_x = wg_lds + h * bufsz;
    unsigned char* s_y = s_x + a.npx * 1024;
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const int ty = tile % a.tiles_y, fg = tile / a.tiles_y;
        const int y0 = ty * a.th - 1;
        const int nfr = min(a.fpt, a.N - fg * a.fpt);
        const bf16_t* xo = a.x + (((long long)fg * a.fpt * H + y0) * W - 1) * C + cb * 64;
        for (int j = w; j < a.npx; j += 4) {
            const int d = 64 * j + lane, pix = d / WG_SLOTS, c = d - WG_SLOTS * pix;
            const int f = wg_div(pix, a.inv_band), r = pix - f * (a.bh * BW), by = wg_div(r, a.inv_bw), bx = r - by * BW;
            const int yy = y0 + by;
            const bool ok = c < 8 && pix < a.band_px && bx >= 1 && bx <= W && (unsigned)yy < (unsigned)H && f < nfr;
            const bf16_t* src = ok ? xo + ((f * H + by) * W + bx) * C + c * 8 : reinterpret_cast<const bf16_t*>(&g_wg_zero16);
            __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(s_x + 1024 * j)), 16, 0, 0);
        }
        const bf16_t* yo = a.dy + (((long long)fg * a.fpt * H + ty * a.th) * W) * K + kb * 64;
        for (int j = w; j < a.npy; j += 4) {
            const int d = 64 * j + lane, pix = d / WG_SLOTS, c = d - WG_SLOTS * pix;
            const int f = wg_div(pix, a.inv_tw), r = pix - f * (a.th * W), py = wg_div(r, a.inv_w);
            const bool ok = c < 8 && pix < a.tpix && f < nfr && ty * a.th + py < H;
            const bf16_t* src = ok ? yo + ((f * H + py) * W + (r - py * W)) * K + c * 8 : reinterpret_cast<const bf16_t*>(&g_wg_zero16);
            __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(s_y + 1024 * j)), 16, 0, 0);
        }
    };

    // The two wave halves are two independent four-wave "workgroups" -- own tile stream, own buffers, in step only with themselves -- as
    // the two workgroups per CU of rounds 2-3 were (in lockstep behind one s_barrier per tile and with the k-steps split between the
    // halves the kernel took 59 us per layer against 55: one half's fetch wait has to overlap the other's MFMAs).  A half's barrier is a
    // counter in LDS: arrive (after the wave's own DMA and LDS operations are done), then poll until all four waves of the round are in.
    unsigned* ctr = reinterpret_cast<unsigned*>(wg_lds + 2 * bufsz) + 16 * h;
    if (tid == 0) { reinterpret_cast<unsigned*>(wg_lds + 2 * bufsz)[0] = 0u; reinterpret_cast<unsigned*>(wg_lds + 2 * bufsz)[16] = 0u; }
    __syncthreads();
    unsigned round = 0;
    auto half_barrier = [&]() __attribute__((always_inline)) {
        round += 4;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < round) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    for (int tile = 2 * blockIdx.x + h; tile < a.ntiles; tile += 2 * gridDim.x) {
        half_barrier();                                    // the half is done with the previous tile's buffers
        fetch(tile);
        half_barrier();                                    // band and dy tile have landed
        // ---- 7 k-steps of 32 pixels: pixel rows 32 s + 8 g + 4 half + q of this lane
#pragma unroll 1
        for (int s = 0; s < WG_KSTEPS; ++s) {
            if (32 * s >= a.tpix) break;                   // (wave-uniform: the tile has fewer pixels)
            unsigned ya[2], xa[2];                         // this lane's LDS byte addresses in the dy tile / the x band (tap 0, 0)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int pp = 32 * s + 8 * g + 4 * hf + q, pix = min(pp, a.tpix - 1);
                ya[hf] = (unsigned)(uintptr_t)(s_y + min(pp, a.npy * 64 / WG_SLOTS - 1) * WG_PIX + p4 * 2);
                const int f = wg_div(pix, a.inv_tw), r = pix - f * (a.th * W), py = wg_div(r, a.inv_w), px = r - py * W;
                xa[hf] = (unsigned)(uintptr_t)(s_x + ((f * a.bh + py) * BW + px) * WG_PIX + (16 * w + p4) * 2);
            }
            // all 26 transposed reads of the k-step are issued, then ONE wait (LDS returns in order; the fragments are operands of
            // the wait so that no MFMA can be scheduled above it -- the compiler takes an asm output for ready at once)
            uint2 fa[4][2], fb[9][2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(fa[kt][hf]) : "v"(ya[hf] + 32 * kt) : "memory");
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dyy = t / 3, dxx = t - 3 * dyy;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(fb[t][hf]) : "v"(xa[hf] + (dyy * BW + dxx) * WG_PIX) : "memory");
            }
            // LDS returns in order: with the reads of taps 5 .. 8 (the last 8) still in flight everything before them has landed
            asm volatile("s_waitcnt lgkmcnt(8)"
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]),
                           "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[2][0]), "+v"(fb[2][1]), "+v"(fb[3][0]), "+v"(fb[3][1]),
                           "+v"(fb[4][0]), "+v"(fb[4][1])
                         :: "memory");
            bf16x8 af[4];                                  // dy^T: A operand, output-channel tile kt (rows), 32 pixels (k)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) af[kt] = __builtin_bit_cast(bf16x8, make_uint4(fa[kt][0].x, fa[kt][0].y, fa[kt][1].x, fa[kt][1].y));
            auto taps = [&](auto t0c, auto t1c) __attribute__((always_inline)) {
#pragma unroll
                for (int t = decltype(t0c)::value; t < decltype(t1c)::value; ++t) {
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, make_uint4(fb[t][0].x, fb[t][0].y, fb[t][1].x, fb[t][1].y));   // x^T window of tap t
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) acc[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kt], bf, acc[kt][t], 0, 0, 0);
                }
            };
            taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fb[5][0]), "+v"(fb[5][1]), "+v"(fb[6][0]), "+v"(fb[6][1]), "+v"(fb[7][0]), "+v"(fb[7][1]), "+v"(fb[8][0]), "+v"(fb[8][1])
                         :: "memory");
            taps(std::integral_constant<int, 5>{}, std::integral_constant<int, 9>{});
        }
    }
    // ---- the odd wave half hands its accumulators to the even one through LDS ([wave 4][tile 36][lane 64] x 16 B: 144 KiB, the tile buffers
    //      are free by now), which adds them and writes the workgroup's ONE partial block -- half the partial traffic of two independent
    //      four-wave workgroups per CU (75 MB written and read back per layer at the training shape)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4* ex = reinterpret_cast<f32x4*>(wg_lds);
    if (h == 1) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int t = 0; t < 9; ++t) ex[(w * 36 + kt * 9 + t) * 64 + lane] = acc[kt][t];
    }
    __syncthreads();
    if (h == 1) return;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[kt][t] += ex[(w * 36 + kt * 9 + t) * 64 + lane];
    // ---- partial block: part[blockIdx.x][blockIdx.y][k 64][tap 9][c 64]; lane (li, g) holds rows 4g + r (output channel) of
    //      column li (input channel 16 w + li)
    float* out = a.part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (64 * 9 * 64);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[((16 * kt + 4 * g + r) * 9 + t) * 64 + 16 * w + li] = acc[kt][t][r];
    (void)K;
}

// dW[k][c][ty][tx] (the framework's contiguous [K, C, 3, 3]) = sum over the gx partial blocks, in index order (deterministic).
// Threads walk the PARTIAL layout ([block][k 64][tap 9][c 64], c fastest: coalesced reads of gx x 147 KB); the transposition to
// [k][c][tap] happens on the (small) write.
// krsc: write dW as [k][ty][tx][c] instead -- the memory order of a channels_last [K, C, 3, 3] parameter, whose gradient the framework
// otherwise re-lays out with a copy per layer and step (and the order the partials come in: the write is coalesced too).
__global__ __launch_bounds__(512) void conv3x3_wgrad_reduce_kernel(const float* part, float* dw, int gx, int gy, int cblocks, int C, int K, int krsc)
{
    // a block = 64 consecutive elements x 8 slices of the partial index (thread = element e, slice ch): coalesced 256-byte reads,
    // eight times the threads of one-thread-per-element (which ran 119 us on 36864 threads for 512 partials), fixed summation order
    __shared__ float s_sum[8][64];
    const size_t blk = (size_t)64 * 9 * 64, total = (size_t)gy * blk;
    const int e = threadIdx.x & 63, ch = threadIdx.x >> 6;
    const int per = (gx + 7) / 8, b0 = ch * per, b1 = min(gx, b0 + per);
    for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
        const size_t i = base + e;                          // (total is a multiple of 64)
        // eight running sums (partials b0 + j, b0 + j + 8, ...) added up in index order: eight loads in flight per thread instead of a
        // dependent chain of up to 64 L2 round trips (round 4: 20 -> 8 us per layer; the same fixed order on every run)
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = b0;
        for (; b + 8 <= b1; b += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] += part[(size_t)(b + j) * total + i];
        for (int j = 0; b < b1; ++b, ++j) s8[j] += part[(size_t)b * total + i];
        s_sum[ch][e] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        __syncthreads();
        if (ch == 0) {
            float t8 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t8 += s_sum[j][e];
            const int by = (int)(i / blk), r = (int)(i % blk);
            const int c = r & 63, t = (r >> 6) % 9, kk = (r >> 6) / 9;
            const int k = (by / cblocks) * 64 + kk, cc = (by % cblocks) * 64 + c;
            dw[krsc ? ((size_t)k * 9 + t) * C + cc : ((size_t)k * C + cc) * 9 + t] = t8;
        }
        __syncthreads();
    }
    (void)K;
}

struct WgradPlan { WgradArgs a; int gx, gy; size_t lds; };

bool wgrad_plan(int N, int C, int H, int W, int K, WgradPlan* p)
{
    if (C % 64 || K % 64 || W > 64 || W < 1 || H < 1 || N < 1) return false;
    WgradArgs& a = p->a;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K;
    if (H * W <= WG_MAXPIX) { a.th = H; a.fpt = WG_MAXPIX / (H * W); a.tiles_y = 1; if (a.fpt > N) a.fpt = N; }
    else { a.fpt = 1; a.th = WG_MAXPIX / W; if (a.th < 1) return false; a.tiles_y = (H + a.th - 1) / a.th; }
    a.bw = W + 2;
    auto fits = [&]() {
        a.bh = a.th + 2; a.band_px = a.fpt * a.bh * a.bw; a.tpix = a.fpt * a.th * W;
        a.npx = (a.band_px * WG_SLOTS + 63) / 64;
        a.npy = (((a.tpix + 31) / 32 * 32) * WG_SLOTS + 63) / 64;
        return (size_t)(a.npx + a.npy) * 1024 <= 78 * 1024;               // two buffers in 160 KB
    };
    while (!fits() && a.fpt > 1) --a.fpt;
    while (!fits() && a.th > 1) { --a.th; a.tiles_y = (H + a.th - 1) / a.th; }
    if (!fits()) return false;
    a.inv_band = 1.0f / (float)(a.bh * a.bw); a.inv_bw = 1.0f / (float)a.bw;
    a.inv_tw = 1.0f / (float)(a.th * W); a.inv_w = 1.0f / (float)W;
    const long long ntiles = (long long)((N + a.fpt - 1) / a.fpt) * a.tiles_y;
    if (ntiles > 0x7fffffffLL) return false;
    a.ntiles = (int)ntiles;
    a.cblocks = C / 64;
    p->gy = (C / 64) * (K / 64);
    int gx = 256 / p->gy; if (gx < 1) gx = 1;              // one eight-wave workgroup per CU (more = more partial blocks to add up: 147 KB each)
    if (gx > (ntiles + 1) / 2) gx = (int)((ntiles + 1) / 2);   // (two tile streams per workgroup)
    p->gx = gx;
    p->lds = 2 * (size_t)(a.npx + a.npy) * 1024 + 128;   // two halves' buffers + their barrier counters
    if (p->lds < 4 * 36 * 64 * sizeof(f32x4)) p->lds = 4 * 36 * 64 * sizeof(f32x4);      // the accumulator exchange at the end
    return true;
}

}  // namespace

extern "C" size_t gdkvm_conv3x3_wgrad_workspace_bytes(int N, int C, int H, int W, int K)
{
    WgradPlan p;
    if (!wgrad_plan(N, C, H, W, K, &p)) return 16;
    return (size_t)p.gx * p.gy * (64 * 9 * 64) * sizeof(float);
}

static int conv3x3_wgrad_impl(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                              int N, int C, int H, int W, int K, int io_dtype, int krsc, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv3x3_wgrad: only bf16 operands are implemented");
    WgradPlan p;
    if (!wgrad_plan(N, C, H, W, K, &p))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv3x3_wgrad: N=%d C=%d H=%d W=%d K=%d (C, K multiples of 64, rows of at most 64 pixels)", N, C, H, W, K);
    if (!x || !dy || !dw || !workspace || !gdkvm_aligned16(x) || !gdkvm_aligned16(dy) || !gdkvm_aligned16(dw) || !gdkvm_aligned16(workspace))
        return gdkvm_fail(GDKVM_ERR_ARG, "conv3x3_wgrad: null or unaligned pointer");
    if ((size_t)N * H * W * C >= (1ull << 31) || (size_t)N * H * W * K >= (1ull << 31))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv3x3_wgrad: tensor too large for 32-bit offsets");
    const size_t need = (size_t)p.gx * p.gy * (64 * 9 * 64) * sizeof(float);
    if (workspace_bytes < need) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "conv3x3_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static std::atomic<unsigned long long> done_mask{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "conv3x3_wgrad: hipGetDevice");
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "conv3x3_wgrad: LDS attribute: %s", hipGetErrorString(e));
        done_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    p.a.x = static_cast<const bf16_t*>(x); p.a.dy = static_cast<const bf16_t*>(dy); p.a.part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3(p.gx, p.gy), dim3(512), p.lds, st, p.a);
    GDKVM_LAUNCH_CHECK("conv3x3_wgrad_kernel");
    const size_t n = (size_t)K * 9 * C;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)(n / 64 > 4096 ? 4096 : n / 64)), dim3(512), 0, st,
                       static_cast<const float*>(workspace), dw, p.gx, p.gy, p.a.cblocks, C, K, krsc);
    GDKVM_LAUNCH_CHECK("conv3x3_wgrad_reduce_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_conv3x3_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                                   int N, int C, int H, int W, int K, int io_dtype, void* stream)
{
    return conv3x3_wgrad_impl(x, dy, dw, workspace, workspace_bytes, N, C, H, W, K, io_dtype, 0, stream);
}

extern "C" int gdkvm_conv3x3_wgrad_krsc(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                                        int N, int C, int H, int W, int K, int io_dtype, void* stream)
{
    return conv3x3_wgrad_impl(x, dy, dw, workspace, workspace_bytes, N, C, H, W, K, io_dtype, 1, stream);
}
