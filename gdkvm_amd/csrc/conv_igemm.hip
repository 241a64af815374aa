// conv_igemm.hip -- SURVEY.md §8f row n1: the convolutions the two specialised kernels do not take -- strided layers, 1x1 layers,
// rows wider than 64 pixels -- as ONE hand-written implicit-GEMM kernel (NHWC bf16, bias (+ residual) (+ ReLU) in the epilogue).
// Until the end of round 2 these went to the framework convolution followed by a gdkvm_bias_act pass (four layers of an EchoNet
// forward: the two 3x3 / stride-2 convolutions and the two 1x1 / stride-2 downsamples, 99 us of 1.04 ms).
//
//   out[p][k] = act(bias[k] + sum_{r, s, c} x[pixel(p) * stride + (r, s) - pad][c] w[k][r][s][c]),   M = N Ho Wo pixels, Kd = R S C
//
//   * a workgroup of 4 waves owns 128 output pixels x 128 output channels; every wave takes ALL the pixels and 32 of the channels
//     (8 x 2 accumulator tiles): no weight fragment is fetched by two waves -- the kernel is bound by the CU's vector-memory path
//     (weights + gathers), and a 2 x 2 wave grid, where two waves stream the same weights, measured 7-10 % slower;
//   * per k-step (32 of the Kd products: one tap, 32 channels -- C is a multiple of 32, so a k-step never straddles taps) the 256
//     pixels' 64-byte channel runs are GATHERED into LDS, one pixel row 80 bytes apart (64 + 16: a ds_read_b128 of 16 consecutive
//     pixels is conflict-free), double-buffered: the gather of k-step s+1 is in flight as register loads behind the MFMAs of s;
//     out-of-image taps are zeros;
//   * the weights never touch LDS: packed in MFMA-fragment order (gdkvm_conv_igemm_pack_weights: fragment (n-tile, k-step) = one
//     contiguous KiB), streamed from L2 into a ring of register sets two k-steps ahead;
//   * weights are the A operand, pixels the B operand: a lane ends with 4 consecutive output channels of one pixel (8-byte stores).
// Arithmetic: fp32 accumulation over the same R S C products as a library convolution, one rounding after the epilogue.
#include <atomic>
#include <type_traits>

#include "gdkvm_common.hpp"

namespace {

constexpr int IG_TN = 128;                       // workgroup tile: channels (pixels: the template parameter TM); the data-gradient form also 64
#ifndef IG_WAVES_N
#define IG_WAVES_N 4
#endif
constexpr int IG_GD = 2;                         // gather distance, k-steps (two register sets)

#ifdef IG_DIAG
// diagnostic builds (tools/abl_igemm.py stamps): s_memtime of wave 0 of workgroup (0, 0) around the phases of its first steps
__device__ unsigned long long* g_ig_diag = nullptr;
#define IG_STAMP(row, slot) do { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && (row) < 16) { unsigned long long t__; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); g_ig_diag[(row) * 8 + (slot)] = t__; } } while (0)
#else
#define IG_STAMP(row, slot) do {} while (0)
#endif

struct IgArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; const bf16_t* res; bf16_t* y;
    int N, H, W, C, K, R, S, stride, pad, Ho, Wo, relu;
    int M, ksteps, cps;                          // output pixels; Kd / 32; k-steps per tap (C / 32)
    const bf16_t* w2; const float* bias2; bf16_t* y2; int centre;   // optional second output: the 1x1 convolution (same stride, K channels,
                                                 // bias2 may be NULL) of the window's centre pixel -- tap index `centre` -- with pack w2
    float inv_cps, inv_s;                        // per-k-step index arithmetic without integer division (see conv3x3_tile.hip: ct_div)
    // ---- data-gradient form (template parameter DG; training, stride 2): x = dy [N, H, W, C] is the gradient of the forward's OUTPUT
    // (C = the forward's output channels), y = dx [N, OH, OW, K] the gradient of its input.  Input pixel (2a + ph, 2b + pw) receives
    // the taps whose parity matches: blockIdx.z = 2 ph + pw is that class, a plain stride-1 convolution of dy with (1 + ph)(1 + pw)
    // taps at (a + r', b + s') -- no products with zeros -- and class (0, 0), whose only tap is the forward window's centre, takes the
    // 1x1 / stride-2 downsample branch's gradient x2 (same shape as x) as ksteps_all - ksteps_main more k-steps of the same sum.
    struct Cls { const bf16_t* w; int S, ksteps_main, ksteps_all, ksteps_row; float inv_s; } cls[4];   // ksteps_row: k-steps per row tile IN THE PACK (>= ksteps_all: a pack made with the branch read without it)
    const bf16_t* x2; int OH, OW;
};

__device__ __forceinline__ int ig_div(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

template <int I, int E, class F>
__device__ __forceinline__ void ig_static_for(F&& f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        ig_static_for<I + 1, E>(f);
    }
}

// TM = pixels per workgroup (128); SUB = 32-deep MFMA k-steps
// per barrier: with C a multiple of 64 a step gathers 128-byte channel runs (two MFMA k-steps: half the barriers, LDS round trips
// and index arithmetic per MFMA); SUB = 1 serves channel counts that are only multiples of 32.
template <int TM, int SUB, int TN = IG_TN, int WN = IG_WAVES_N, bool DG = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(IgArgs a)
{
    // WN = waves along the channels (4 / WN along the pixels)
    constexpr int NTW = TN / 16 / WN;            // 16-channel tiles per wave
    constexpr int MT = TM / 16 / (4 / WN);       // 16-pixel tiles per wave
    constexpr int PITCH = SUB == 2 ? 160 : 80;   // bytes between pixel rows of the LDS tile: 128 + 32 or 64 + 16 (conflict-free ds_read_b128)
    constexpr int PPP = 4 * SUB;                 // 16-byte pieces per pixel and step
    constexpr int GP = TM * PPP / 256;           // pieces per thread and step
    constexpr int WD = SUB == 2 ? 2 : 3;         // weight ring depth, steps
    constexpr int UF = SUB == 2 ? 2 : 6;         // unroll: lcm(weight ring, two gather sets) -> static register indices
    __shared__ __attribute__((aligned(16))) unsigned char s_a[2][TM * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w / WN, wn = w % WN;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int cls = DG ? (int)blockIdx.z : 0;
    const int S_ = DG ? a.cls[cls].S : a.S;
    const float inv_s_ = DG ? a.cls[cls].inv_s : a.inv_s;
    const int nsteps = (DG ? a.cls[cls].ksteps_all : a.ksteps) / SUB;           // steps of 32 SUB products
    const int main_steps = DG ? a.cls[cls].ksteps_main / SUB : nsteps;          // (DG: the steps after these gather from x2, tap (0, 0))
    const ptrdiff_t x2_diff = DG && a.x2 ? a.x2 - a.x : 0;

    // ---- gather geometry: piece q = 256 u + tid is 16-byte part q % PPP of the channel run of tile pixel q / PPP: PPP lanes share a
    //      run, a wave instruction touches 64 / PPP runs ----------------------------------------------------------------------------
    int g_iy0[GP], g_ix0[GP];
    const bf16_t* g_base[GP];
    unsigned g_lds[GP];
#pragma unroll
    for (int u = 0; u < GP; ++u) {
        const int q = 256 * u + tid, px = q / PPP, part = q % PPP;
        const int gp = min(m0 + px, a.M - 1);
        const int gn = gp / (a.Ho * a.Wo), grem = gp - gn * a.Ho * a.Wo, gyo = grem / a.Wo, gxo = grem - gyo * a.Wo;     // (once per thread)
        g_iy0[u] = m0 + px < a.M ? gyo * a.stride - a.pad : -0x40000000;
        g_ix0[u] = gxo * a.stride - a.pad;
        g_base[u] = a.x + (size_t)gn * a.H * a.W * a.C + 8 * part;
        g_lds[u] = (unsigned)(px * PITCH + 16 * part);
    }
    // (the loads are unconditional -- an out-of-image tap reads pixel (0, 0) of its frame and is zeroed when it goes to LDS: with
    // loads under a branch the compiler gives up counting and every wait becomes vmcnt(0), i.e. for the youngest prefetch too)
    uint4 ga[IG_GD][GP];
    unsigned gok[IG_GD];
    const int spt = a.cps / SUB;                 // steps per tap
    const float inv_spt = a.inv_cps * (float)SUB;
    auto gather = [&](int st, uint4 (&d)[GP], unsigned& okm) __attribute__((always_inline)) {
        const bool second = DG && st >= main_steps;
        const int stm = second ? st - main_steps : st;
        const int tap = ig_div(stm, inv_spt), c0 = (stm - tap * spt) * 32 * SUB, r = ig_div(tap, inv_s_), s = tap - r * S_;
        const ptrdiff_t src = second ? x2_diff : 0;
        okm = 0;
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int iy = g_iy0[u] + r, ix = g_ix0[u] + s;
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            okm |= ok ? 1u << u : 0u;
#ifdef IG_ABL_SAMEPIX                                          // ablation: every gather from one (cached) address
            d[u] = *reinterpret_cast<const uint4*>(a.x + 8 * (tid & 7));
#else
            d[u] = *reinterpret_cast<const uint4*>(g_base[u] + src + ((size_t)(ok ? iy : 0) * a.W + (ok ? ix : 0)) * a.C + c0);
#endif
        }
    };
    auto put = [&](int buf, const uint4 (&d)[GP], unsigned okm) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const unsigned m = (okm >> u) & 1u ? 0xffffffffu : 0u;
            *reinterpret_cast<uint4*>(&s_a[buf][g_lds[u]]) = make_uint4(d[u].x & m, d[u].y & m, d[u].z & m, d[u].w & m);
        }
    };

    const unsigned abase = (unsigned)((16 * MT * wm + li) * PITCH + 16 * g);      // this lane's B fragment of pixel tile 0: pixel li, k 8g..

    // One convolution over the gathered pixels: `L` steps starting at gather step g0, weights `wpk` packed for exactly those steps.
    // The layer itself is (0, nsteps); a second pass over the CENTRE tap alone with another weight pack is the 1x1 / same-stride
    // convolution of the same input (the residual block's downsample branch), computed here on the pixels' geometry already set up.
    auto run_pass = [&](const bf16_t* wpk, int g0, int L, const float* bias, const bf16_t* res, bf16_t* y, int relu, int Lrow) __attribute__((always_inline)) {
        // weights: fragment (n-tile, k-step) = 1 KiB contiguous; this wave's four n-tiles, SUB k-steps per step
        const bf16_t* wp = wpk + ((size_t)((n0 + 16 * NTW * wn) / 16) * (Lrow * SUB) * 64 + lane) * 8;      // (Lrow: steps per row tile in the pack)
        const size_t nt_stride = (size_t)(Lrow * SUB) * 512;
        bf16x8 wr[WD][SUB][NTW];
        auto wload = [&](int l, bf16x8 (&d)[SUB][NTW]) __attribute__((always_inline)) {
            const size_t off = (size_t)min(l, L - 1) * (512 * SUB);
#pragma unroll
            for (int sb = 0; sb < SUB; ++sb)
#pragma unroll
#ifdef IG_ABL_SAMEW                                            // ablation: every weight fragment from one (cached) KiB
                for (int nt = 0; nt < NTW; ++nt) d[sb][nt] = *reinterpret_cast<const bf16x8*>(wp + 0 * (nt * nt_stride + off + 512 * sb));
#else
                for (int nt = 0; nt < NTW; ++nt) d[sb][nt] = *reinterpret_cast<const bf16x8*>(wp + nt * nt_stride + off + 512 * sb);
#endif
        };
        const int glast = g0 + L - 1;
        f32x4 acc[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        IG_STAMP(15, 0);
        gather(min(g0, glast), ga[0], gok[0]);
        gather(min(g0 + 1, glast), ga[1], gok[1]);
#pragma unroll
        for (int d = 0; d < WD; ++d) wload(d, wr[d]);
        put(0, ga[0], gok[0]);
        gather(min(g0 + 2, glast), ga[0], gok[0]);
        __syncthreads();
        IG_STAMP(15, 1);

        // step l: MFMAs from LDS buffer l & 1; the pixels of step l + 1 (requested a step and a half ago, set (l + 1) & 1) go to the
        // other buffer behind them, and that set is refilled for step l + 3; weights of step l + WD refill ring set l % WD
        auto step = [&](int l, auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value % WD, gs = (decltype(jc)::value + 1) & 1;   // (l and jc agree modulo UF)
            const int buf = l & 1;
            IG_STAMP(l, 0);
#pragma unroll
            for (int sb = 0; sb < SUB; ++sb)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 xa = *reinterpret_cast<const bf16x8*>(&s_a[buf][abase + mt * 16 * PITCH + 64 * sb]);
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[j][sb][nt], xa, acc[mt][nt], 0, 0, 0);
                }
            IG_STAMP(l, 1);
            put(buf ^ 1, ga[gs], gok[gs]);                   // (everyone finished reading that buffer before the last barrier)
            IG_STAMP(l, 2);
            wload(l + WD, wr[j]);                            // refill the weight set just used
            gather(min(g0 + l + 1 + IG_GD, glast), ga[gs], gok[gs]);
            IG_STAMP(l, 3);
            __syncthreads();
            IG_STAMP(l, 4);
        };
        int l = 0;
        for (; l + UF <= L; l += UF)
            ig_static_for<0, UF>([&](auto jc) { step(l + decltype(jc)::value, jc); });
        ig_static_for<0, UF - 1>([&](auto jc) { if (l + decltype(jc)::value < L) step(l + decltype(jc)::value, jc); });

        // epilogue: lane (g, li) holds channels n + 4g .. +3 of pixel 16 mt + li
        IG_STAMP(15, 2);
        const float lo = relu ? 0.f : -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int ch = n0 + 16 * NTW * wn + 16 * nt + 4 * g;
            const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = m0 + 16 * MT * wm + 16 * mt + li;
                if (p >= a.M) continue;
                f32x4 v = acc[mt][nt] + b4;
                size_t o = (size_t)p * a.K + ch;
                if constexpr (DG) {                        // pixel (n, ya, xb) of the dy grid -> input pixel (2 ya + ph, 2 xb + pw)
                    const int pn = p / (a.Ho * a.Wo), prem = p - pn * a.Ho * a.Wo, ya = prem / a.Wo, xb = prem - ya * a.Wo;
                    const int oy = 2 * ya + (cls >> 1), ox = 2 * xb + (cls & 1);
                    if (oy >= a.OH || ox >= a.OW) continue; // (odd sizes: the last row / column of a class may not exist)
                    o = (((size_t)pn * a.OH + oy) * a.OW + ox) * a.K + ch;
                }
                if (res) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(res + o);
                    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lo);
                *reinterpret_cast<uint2*>(y + o) = make_uint2((unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16),
                                                              (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16));
            }
        }
    };
    if constexpr (DG) {
        run_pass(a.cls[cls].w, 0, nsteps, nullptr, nullptr, a.y, 0, a.cls[cls].ksteps_row / SUB);
    } else {
        run_pass(a.w, 0, nsteps, a.bias, a.res, a.y, a.relu, nsteps);
        IG_STAMP(15, 3);
        if (a.w2) run_pass(a.w2, a.centre * spt, spt, a.bias2, nullptr, a.y2, 0, spt);
    }
}

// weights [K][R][S][C] (= [K][Kd]) -> [K / 16][Kd / 32][lane = 16 g + li][8]:  w[16 nt + li][32 ks + 8 g ..]
__global__ __launch_bounds__(256) void conv_igemm_pack_kernel(const uint4* w, uint4* packed, int K, int Kd)
{
    const int nks = Kd / 32;
    const size_t total = (size_t)(K / 16) * nks * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), li = lane & 15, g = lane >> 4;
        const size_t f = i >> 6;
        const int ks = (int)(f % nks), nt = (int)(f / nks);
        packed[i] = w[((size_t)(16 * nt + li) * Kd + 32 * ks + 8 * g) / 8];
    }
}

}  // namespace

#ifdef IG_DIAG
extern "C" void gdkvm_ig_diag_buffer(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ig_diag), &p, sizeof(p)); }
#endif

extern "C" int gdkvm_conv_igemm_pack_weights(const void* w, void* packed, int K, int C, int R, int S, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv_igemm_pack_weights: only bf16 is implemented");
    if (K <= 0 || C <= 0 || R <= 0 || S <= 0 || K % 16 || C % 32)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_igemm_pack_weights: K=%d C=%d %dx%d (K a multiple of 16, C of 32)", K, C, R, S);
    if (!w || !packed || !gdkvm_aligned16(w) || !gdkvm_aligned16(packed)) return gdkvm_fail(GDKVM_ERR_ARG, "conv_igemm_pack_weights: null or unaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    const int Kd = R * S * C;
    const size_t total = (size_t)(K / 16) * (Kd / 32) * 64;
    hipLaunchKernelGGL(conv_igemm_pack_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint4*>(w), static_cast<uint4*>(packed), K, Kd);
    GDKVM_LAUNCH_CHECK("conv_igemm_pack_kernel");
    return GDKVM_OK;
}

// internal entry used by gdkvm_conv_bias_act (conv_dispatch.hip): 0 = launched, 1 = shape not covered
int gdkvm_conv_igemm_launch(const void* x, const void* wpacked, const float* bias, const void* residual, void* y,
                            int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu,
                            const void* w2packed, const float* bias2, void* y2, hipStream_t st)
{
    if (C % 32 || K % IG_TN || N < 1 || R < 1 || S < 1 || stride < 1 || pad < 0) return 1;
    if (w2packed && (!y2 || R != S || !(R & 1) || pad != R / 2)) return 1;        // (the 1x1's pixel must be the window's centre)
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho < 1 || Wo < 1) return 1;
    const long long M = (long long)N * Ho * Wo;
    if (M * (long long)(K > C ? K : C) > 0x7fffffffLL || (long long)N * H * W * C > 0x7fffffffLL || (long long)R * S * C / 32 > 65535) return 1;
    IgArgs a{};
    a.x = static_cast<const bf16_t*>(x); a.w = static_cast<const bf16_t*>(wpacked); a.bias = bias;
    a.res = static_cast<const bf16_t*>(residual); a.y = static_cast<bf16_t*>(y);
    a.w2 = static_cast<const bf16_t*>(w2packed); a.bias2 = bias2; a.y2 = static_cast<bf16_t*>(y2); a.centre = (R / 2) * S + S / 2;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.Ho = Ho; a.Wo = Wo; a.relu = relu;
    a.M = (int)M; a.cps = C / 32; a.ksteps = R * S * a.cps;
    a.inv_cps = 1.0f / (float)a.cps; a.inv_s = 1.0f / (float)S;
    const dim3 grid((unsigned)((M + 127) / 128), (unsigned)(K / IG_TN));
    if (C % 64 == 0) hipLaunchKernelGGL((conv_igemm_kernel<128, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<128, 1>), grid, dim3(256), 0, st, a);
    return 0;
}

// internal entry used by gdkvm_conv_s2_dgrad (conv_s2_train.hip): the data gradient of a 3x3 / stride-2 / pad-1 convolution (+ the 1x1 /
// stride-2 branch's, dy2 may be NULL) as the kernel's DG form.  dy, dy2 [N, Ho, Wo, Kf] bf16; dx [N, H, W, Cf]; packs: the four class packs
// of gdkvm_conv_s2_pack_train, consecutive.  with_down says how the PACK was built (class 0 carries one more tap when the branch rode
// along): the class offsets follow it, not dy2 -- dy2 == NULL on a pack with the branch is a zero branch gradient (the extra k-steps
// are skipped, the offsets stay).  0 = launched, 1 = shape not covered.
int gdkvm_conv_igemm_dgrad_launch(const void* dy, const void* dy2, const void* packs, int with_down, void* dx, int N, int Cf, int H, int W, int Kf, hipStream_t st)
{
    if (Cf % 64 || Kf % 64 || N < 1 || H < 1 || W < 1) return 1;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long M = (long long)N * Ho * Wo;
    if (M * Kf > 0x7fffffffLL || (long long)N * H * W * Cf > 0x7fffffffLL || 5LL * Kf / 32 > 65535) return 1;
    IgArgs a{};
    a.x = static_cast<const bf16_t*>(dy); a.x2 = static_cast<const bf16_t*>(dy2); a.y = static_cast<bf16_t*>(dx);
    a.N = N; a.H = Ho; a.W = Wo; a.C = Kf; a.K = Cf; a.R = 1; a.S = 1; a.stride = 1; a.pad = 0; a.Ho = Ho; a.Wo = Wo; a.relu = 0;
    a.OH = H; a.OW = W;
    a.M = (int)M; a.cps = Kf / 32; a.ksteps = a.cps;
    a.inv_cps = 1.0f / (float)a.cps; a.inv_s = 1.0f;
    const bf16_t* wp = static_cast<const bf16_t*>(packs);
    for (int c = 0; c < 4; ++c) {
        const int ph = c >> 1, pw = c & 1, taps = (1 + ph) * (1 + pw);
        a.cls[c].w = wp;
        a.cls[c].S = 1 + pw;
        a.cls[c].inv_s = 1.0f / (float)(1 + pw);
        a.cls[c].ksteps_main = taps * a.cps;
        a.cls[c].ksteps_all = (taps + (c == 0 && dy2 && with_down ? 1 : 0)) * a.cps;
        a.cls[c].ksteps_row = (taps + (c == 0 && with_down ? 1 : 0)) * a.cps;
        wp += gdkvm_conv_s2_dgrad_pack_elems(Cf, Kf, c, with_down);
    }
    if (Cf % 128 == 0) {
        const dim3 grid((unsigned)((M + 127) / 128), (unsigned)(Cf / 128), 4);
        hipLaunchKernelGGL((conv_igemm_kernel<128, 2, 128, 4, true>), grid, dim3(256), 0, st, a);
    } else {
        const dim3 grid((unsigned)((M + 127) / 128), (unsigned)(Cf / 64), 4);
        hipLaunchKernelGGL((conv_igemm_kernel<128, 2, 64, 2, true>), grid, dim3(256), 0, st, a);
    }
    return 0;
}
