// kpff.hip -- SURVEY.md §8 row a4 / Appendix A.5: Key-Pixel Feature Fusion, one fused kernel.
//
//   Gms = mean_{s in 1,2,4} cellmean_s(G)            (multi-scale "global key feature")
//   g   = sigmoid([P ; L ; Gms] Wa^T + ba) = (g_l | g_g)
//   F   = P + g_l * (L Wl^T) + g_g * (Gms Wg^T)
//
// One workgroup owns a tile of <= 64 tokens made of whole 4-row bands of the h x w grid (or the whole
// frame when it has <= 64 tokens), so every 2x2 and 4x4 pooling cell is tile-local.  The tile's [P;L;G]
// rows are staged once in LDS (fp32, padded rows -> conflict-free 16-byte operand reads), G is pooled in
// place, and the four channel mixes run as ONE pass of exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) over the
// concatenated input: each wave owns 16 output channels x 64 tokens x {g_l, g_g, L Wl, Gms Wg} = 16
// accumulator tiles, reads its weight rows straight from L2 with 16-byte loads (each weight element is
// fetched once per workgroup), and fuses bias, sigmoid, gating and the residual into the epilogue.
#include "gdkvm_common.hpp"

namespace {

struct KpffSave { void* gates; void* lp; void* gp; void* gms; };   // training: [M,2Cp] [M,Cp] [M,Cp] [M,Cv] (io dtype) or NULL

struct KpffArgs {
    const void* L; const void* G; const void* P;
    const float* wa; const float* ba; const float* wl; const float* wg;
    void* out;
    int Ck, Cv, Cp, h, w, rows_per_tile, tiles_per_frame;
    KpffSave sv;
};

constexpr int KPFF_TM = 64;       // tokens per workgroup tile
constexpr int KPFF_PAD = 4;       // row padding in floats: stride % 64 == 4 -> 16 rows cover all 64 banks

template <int IO>
__global__ __launch_bounds__(256) void kpff_kernel(KpffArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_x[];   // [KPFF_TM][Cin + PAD]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ck = a.Ck, Cv = a.Cv, Cp = a.Cp, Cin = Cp + Ck + Cv, ld = Cin + KPFF_PAD;
    const int f = blockIdx.x / a.tiles_per_frame, rt = blockIdx.x % a.tiles_per_frame;
    const int N = a.h * a.w, W = a.w;
    const int row0 = rt * a.rows_per_tile;
    const int nrows = min(a.rows_per_tile, a.h - row0);
    const int n0 = row0 * W, ntok = nrows * W;

    // ---- stage [P ; L ; G] rows (4 channels per thread, 16-byte LDS stores) -------------------------
    {
        const int q4 = Cin / 4;
        for (int idx = tid; idx < KPFF_TM * q4; idx += 256) {
            const int tok = idx / q4, c = (idx - tok * q4) * 4;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (tok < ntok) {
                const size_t row = (size_t)f * N + n0 + tok;
                if (c < Cp) x = load4<IO>(a.P, row * Cp + c);
                else if (c < Cp + Ck) x = load4<IO>(a.L, row * Ck + (c - Cp));
                else x = load4<IO>(a.G, row * Cv + (c - Cp - Ck));
            }
            *reinterpret_cast<f32x4*>(s_x + (size_t)tok * ld + c) = x;
        }
    }
    __syncthreads();
    // ---- multi-scale pooling of G in place: one thread per (4x4 cell, channel) -----------------------
    {
        const int cw = (W + 3) / 4, chh = (nrows + 3) / 4;
        float* gx = s_x + Cp + Ck;
        for (int idx = tid; idx < cw * chh * Cv; idx += 256) {
            const int c = idx % Cv, cell = idx / Cv;
            const int y0 = (cell / cw) * 4, x0 = (cell % cw) * 4;
            const int y1 = min(y0 + 4, nrows), x1 = min(x0 + 4, W);
            float s4 = 0.f, s2[4] = {0.f, 0.f, 0.f, 0.f};
            int n2[4] = {0, 0, 0, 0};
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const float v = gx[(size_t)(y * W + x) * ld + c];
                    const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                    s2[q] += v; n2[q] += 1; s4 += v;
                }
            const float m4 = s4 / (float)((y1 - y0) * (x1 - x0));
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                    float* p = gx + (size_t)(y * W + x) * ld + c;
                    *p = (*p + s2[q] / (float)n2[q] + m4) * (1.0f / 3.0f);
                }
        }
    }
    __syncthreads();
    if (a.sv.gms) {                                            // training: keep the pooled feature for the backward
        for (int idx = tid; idx < ntok * Cv; idx += 256) {
            const int tok = idx / Cv, c = idx - tok * Cv;
            store1<IO>(a.sv.gms, ((size_t)f * N + n0 + tok) * Cv + c, s_x[(size_t)tok * ld + Cp + Ck + c]);
        }
    }

    // ---- fused channel mixes: wave owns output channels o = 64*chunk + 16*wave + li -----------------
    const int kbP = Cp / 16, kbL = Ck / 16, kbG = Cv / 16;
    for (int chunk = 0; chunk * 64 < Cp; ++chunk) {
        const int ob = chunk * 64 + 16 * w_id;            // wave-uniform
        if (ob >= Cp) break;
        const int o = ob + li;
        f32x4 gl[4], gg[4], lp[4], gp[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gl[mt] = gg[mt] = lp[mt] = gp[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wal = a.wa + (size_t)o * Cin + 4 * g;
        const float* wag = a.wa + (size_t)(Cp + o) * Cin + 4 * g;
        const float* xa = s_x + (size_t)li * ld + 4 * g;

        auto step = [&](int kb, const float* wmix, f32x4* mix) {
            const f32x4 bl = *reinterpret_cast<const f32x4*>(wal + 16 * kb);
            const f32x4 bg = *reinterpret_cast<const f32x4*>(wag + 16 * kb);
            f32x4 bm = {0.f, 0.f, 0.f, 0.f};
            if (wmix) bm = *reinterpret_cast<const f32x4*>(wmix);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(xa + (size_t)mt * 16 * ld + 16 * kb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gl[mt] = mfma4(av[r], bl[r], gl[mt]);
                    gg[mt] = mfma4(av[r], bg[r], gg[mt]);
                    if (wmix) mix[mt] = mfma4(av[r], bm[r], mix[mt]);
                }
            }
        };
        for (int kb = 0; kb < kbP; ++kb) step(kb, nullptr, nullptr);
        for (int kb = 0; kb < kbL; ++kb) step(kbP + kb, a.wl + (size_t)o * Ck + 16 * kb + 4 * g, lp);
        for (int kb = 0; kb < kbG; ++kb) step(kbP + kbL + kb, a.wg + (size_t)o * Cv + 16 * kb + 4 * g, gp);

        const float bl = a.ba[o], bg = a.ba[Cp + o];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = 16 * mt + 4 * g + r;
                if (tok < ntok) {
                    const float sl = 1.0f / (1.0f + expf(-(gl[mt][r] + bl)));
                    const float sg = 1.0f / (1.0f + expf(-(gg[mt][r] + bg)));
                    const float y = s_x[(size_t)tok * ld + o] + sl * lp[mt][r] + sg * gp[mt][r];
                    const size_t grow = (size_t)f * N + n0 + tok;
                    store1<IO>(a.out, grow * Cp + o, y);
                    if (a.sv.gates) {
                        store1<IO>(a.sv.gates, grow * 2 * Cp + o, sl);
                        store1<IO>(a.sv.gates, grow * 2 * Cp + Cp + o, sg);
                        store1<IO>(a.sv.lp, grow * Cp + o, lp[mt][r]);
                        store1<IO>(a.sv.gp, grow * Cp + o, gp[mt][r]);
                    }
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------------
// bf16 arm: same tiling, operands in bf16 on v_mfma_f32_16x16x32_bf16 (fp32 accumulate), 16x the fp32 MFMA rate.
// The tile is staged as bf16 (72 KB -> two workgroups per CU), pooled in place (fp32 math, bf16 store), and the
// weights are read as bf16 from a workspace copy made by kpff_pack_weights_kernel (rows [out][in], so a B
// fragment = 8 consecutive k of one output channel = one 16-byte load).  Lane l = 16g + i:
//   A = X[row i][k 8g..8g+7],  B = W[col i][k 8g..8g+7],  C/D reg r = D[row 4g + r][col i].

struct KpffBf16Args {
    const bf16_t* L; const bf16_t* G; const bf16_t* P;
    const bf16_t* wa; const float* ba; const bf16_t* wl; const bf16_t* wg;
    bf16_t* out;
    int Ck, Cv, Cp, h, w, rows_per_tile, tiles_per_frame;
    KpffSave sv;
};

constexpr int KPFF_PAD16 = 8;     // bf16 elements (16 B) of row padding

__global__ void kpff_pack_weights_kernel(const float* wa, const float* wl, const float* wg, bf16_t* dst,
                                         size_t na, size_t nl, size_t ng)
{
    const size_t n = na + nl + ng;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = i < na ? wa[i] : (i < na + nl ? wl[i - na] : wg[i - na - nl]);
        dst[i] = f32_to_bf16(x);
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void kpff_bf16_kernel(KpffBf16Args a)
{
    extern __shared__ __attribute__((aligned(16))) bf16_t s_xb[];   // [KPFF_TM][Cin + PAD16]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ck = a.Ck, Cv = a.Cv, Cp = a.Cp, Cin = Cp + Ck + Cv, ld = Cin + KPFF_PAD16;
    const int f = blockIdx.x / a.tiles_per_frame, rt = blockIdx.x % a.tiles_per_frame;
    const int N = a.h * a.w, W = a.w;
    const int row0 = rt * a.rows_per_tile;
    const int nrows = min(a.rows_per_tile, a.h - row0);
    const int n0 = row0 * W, ntok = nrows * W;

    // ---- stage [P ; L ; G] rows as they are (8 channels = 16 bytes per thread) ------------------------
    {
        const int q8 = Cin / 8;
        for (int idx = tid; idx < KPFF_TM * q8; idx += 256) {
            const int tok = idx / q8, c = (idx - tok * q8) * 8;
            uint4 x = make_uint4(0u, 0u, 0u, 0u);
            if (tok < ntok) {
                const size_t row = (size_t)f * N + n0 + tok;
                if (c < Cp) x = *reinterpret_cast<const uint4*>(a.P + row * Cp + c);
                else if (c < Cp + Ck) x = *reinterpret_cast<const uint4*>(a.L + row * Ck + (c - Cp));
                else x = *reinterpret_cast<const uint4*>(a.G + row * Cv + (c - Cp - Ck));
            }
            *reinterpret_cast<uint4*>(s_xb + (size_t)tok * ld + c) = x;
        }
    }
    __syncthreads();
    // ---- multi-scale pooling of G in place: one thread per (4x4 cell, channel pair) -------------------
    {
        const int cw = (W + 3) / 4, chh = (nrows + 3) / 4, cv2 = Cv / 2;
        bf16_t* gx = s_xb + Cp + Ck;
        for (int idx = tid; idx < cw * chh * cv2; idx += 256) {
            const int c = (idx % cv2) * 2, cell = idx / cv2;
            const int y0 = (cell / cw) * 4, x0 = (cell % cw) * 4;
            const int y1 = min(y0 + 4, nrows), x1 = min(x0 + 4, W);
            float s4[2] = {0.f, 0.f}, s2[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            int n2[4] = {0, 0, 0, 0};
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const unsigned u = *reinterpret_cast<const unsigned*>(gx + (size_t)(y * W + x) * ld + c);
                    const float v0 = __uint_as_float(u << 16), v1 = __uint_as_float(u & 0xffff0000u);
                    const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                    s2[q][0] += v0; s2[q][1] += v1; n2[q] += 1; s4[0] += v0; s4[1] += v1;
                }
            const float i4 = 1.0f / (float)((y1 - y0) * (x1 - x0));
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                    unsigned* p = reinterpret_cast<unsigned*>(gx + (size_t)(y * W + x) * ld + c);
                    const unsigned u = *p;
                    const float i2 = 1.0f / (float)n2[q];
                    const float r0 = (__uint_as_float(u << 16) + s2[q][0] * i2 + s4[0] * i4) * (1.0f / 3.0f);
                    const float r1 = (__uint_as_float(u & 0xffff0000u) + s2[q][1] * i2 + s4[1] * i4) * (1.0f / 3.0f);
                    *p = (unsigned)f32_to_bf16(r0) | ((unsigned)f32_to_bf16(r1) << 16);
                }
        }
    }
    __syncthreads();
    if (a.sv.gms) {                                            // training: keep the pooled feature for the backward
        bf16_t* gms = static_cast<bf16_t*>(a.sv.gms);
        for (int idx = tid; idx < ntok * Cv; idx += 256) {
            const int tok = idx / Cv, c = idx - tok * Cv;
            gms[((size_t)f * N + n0 + tok) * Cv + c] = s_xb[(size_t)tok * ld + Cp + Ck + c];
        }
    }

    // ---- fused channel mixes: wave owns output channels o = 64*chunk + 16*wave + li -----------------
    const int ksP = Cp / 32, ksL = Ck / 32, ksG = Cv / 32;
    for (int chunk = 0; chunk * 64 < Cp; ++chunk) {
        const int ob = chunk * 64 + 16 * w_id;            // wave-uniform
        if (ob >= Cp) break;
        const int o = ob + li;
        f32x4 gl[4], gg[4], lp[4], gp[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gl[mt] = gg[mt] = lp[mt] = gp[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16_t* wal = a.wa + (size_t)o * Cin + 8 * g;
        const bf16_t* wag = a.wa + (size_t)(Cp + o) * Cin + 8 * g;
        const bf16_t* xa = s_xb + (size_t)li * ld + 8 * g;

        auto step = [&](int ks, const bf16_t* wmix, f32x4* mix) {
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wal + 32 * ks);
            const bf16x8 bg = *reinterpret_cast<const bf16x8*>(wag + 32 * ks);
            bf16x8 bm = {};
            if (wmix) bm = *reinterpret_cast<const bf16x8*>(wmix);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(xa + (size_t)mt * 16 * ld + 32 * ks);
                gl[mt] = mfma_bf16(av, bl, gl[mt]);
                gg[mt] = mfma_bf16(av, bg, gg[mt]);
                if (wmix) mix[mt] = mfma_bf16(av, bm, mix[mt]);
            }
        };
#pragma unroll 2
        for (int ks = 0; ks < ksP; ++ks) step(ks, nullptr, nullptr);
        for (int ks = 0; ks < ksL; ++ks) step(ksP + ks, a.wl + (size_t)o * Ck + 32 * ks + 8 * g, lp);
#pragma unroll 2
        for (int ks = 0; ks < ksG; ++ks) step(ksP + ksL + ks, a.wg + (size_t)o * Cv + 32 * ks + 8 * g, gp);

        const float bl = a.ba[o], bg = a.ba[Cp + o];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = 16 * mt + 4 * g + r;
                if (tok < ntok) {
                    const float sl = 1.0f / (1.0f + __expf(-(gl[mt][r] + bl)));
                    const float sg = 1.0f / (1.0f + __expf(-(gg[mt][r] + bg)));
                    const float y = bf16_to_f32(s_xb[(size_t)tok * ld + o]) + sl * lp[mt][r] + sg * gp[mt][r];
                    const size_t grow = (size_t)f * N + n0 + tok;
                    a.out[grow * Cp + o] = f32_to_bf16(y);
                    if (a.sv.gates) {
                        bf16_t* sg_ = static_cast<bf16_t*>(a.sv.gates);
                        sg_[grow * 2 * Cp + o] = f32_to_bf16(sl);
                        sg_[grow * 2 * Cp + Cp + o] = f32_to_bf16(sg);
                        static_cast<bf16_t*>(a.sv.lp)[grow * Cp + o] = f32_to_bf16(lp[mt][r]);
                        static_cast<bf16_t*>(a.sv.gp)[grow * Cp + o] = f32_to_bf16(gp[mt][r]);
                    }
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Backward (row a7) pieces that are not plain GEMMs.  With g = (g_l | g_g), Lp = L Wl^T, Gp = Gms Wg^T:
//   pre :  dz = (dF * Lp * g_l (1-g_l) | dF * Gp * g_g (1-g_g)),  dLp = dF * g_l,  dGp = dF * g_g     (elementwise)
//   ...    dX = dz Wa,  dL += dLp Wl,  dGms += dGp Wg,  dWa = dz^T [P;L;Gms], dWl = dLp^T L, dWg = dGp^T Gms   (library GEMMs)
//   post:  dP = dF + dX_P,  dL = dX_L + dLp Wl,  dG = pool(dX_G + dGp Wg)   (the multi-scale pooling is symmetric)
template <int IO>
__global__ void kpff_bwd_pre_kernel(const void* dF, const void* gates, const void* lp, const void* gp,
                                    void* dz, void* dlp, void* dgp, size_t M, int Cp)
{
    const size_t n = M * Cp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / Cp;
        const int o = (int)(i - row * Cp);
        const float d = load1<IO>(dF, i), gl = load1<IO>(gates, row * 2 * Cp + o), gg = load1<IO>(gates, row * 2 * Cp + Cp + o);
        store1<IO>(dz, row * 2 * Cp + o, d * load1<IO>(lp, i) * gl * (1.f - gl));
        store1<IO>(dz, row * 2 * Cp + Cp + o, d * load1<IO>(gp, i) * gg * (1.f - gg));
        store1<IO>(dlp, i, d * gl);
        store1<IO>(dgp, i, d * gg);
    }
}

// one workgroup per frame: dP, dL (elementwise sums) and dG = pool(dGms) over the h x w grid
template <int IO>
__global__ __launch_bounds__(256) void kpff_bwd_post_kernel(const void* dF, const void* dX, const void* dL_add, const void* dG_add,
                                                            void* dP, void* dL, void* dG, int Ck, int Cv, int Cp, int h, int w)
{
    const int f = blockIdx.x, N = h * w, Cin = Cp + Ck + Cv, tid = threadIdx.x;
    for (int idx = tid; idx < N * Cp; idx += 256) {
        const int tok = idx / Cp, c = idx - tok * Cp;
        const size_t row = (size_t)f * N + tok;
        store1<IO>(dP, row * Cp + c, load1<IO>(dF, row * Cp + c) + load1<IO>(dX, row * Cin + c));
    }
    for (int idx = tid; idx < N * Ck; idx += 256) {
        const int tok = idx / Ck, c = idx - tok * Ck;
        const size_t row = (size_t)f * N + tok;
        store1<IO>(dL, row * Ck + c, load1<IO>(dX, row * Cin + Cp + c) + load1<IO>(dL_add, row * Ck + c));
    }
    const int cw = (w + 3) / 4, chh = (h + 3) / 4;
    for (int idx = tid; idx < cw * chh * Cv; idx += 256) {
        const int c = idx % Cv, cell = idx / Cv;
        const int y0 = (cell / cw) * 4, x0 = (cell % cw) * 4;
        const int y1 = min(y0 + 4, h), x1 = min(x0 + 4, w);
        float s4 = 0.f, s2[4] = {0.f, 0.f, 0.f, 0.f}, v[16];
        int n2[4] = {0, 0, 0, 0};
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const size_t row = (size_t)f * N + y * w + x;
                const float val = load1<IO>(dX, row * Cin + Cp + Ck + c) + load1<IO>(dG_add, row * Cv + c);
                const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                v[(y - y0) * 4 + (x - x0)] = val;
                s2[q] += val; n2[q] += 1; s4 += val;
            }
        const float m4 = s4 / (float)((y1 - y0) * (x1 - x0));
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const int q = ((y - y0) >> 1) * 2 + ((x - x0) >> 1);
                const size_t row = (size_t)f * N + y * w + x;
                store1<IO>(dG, row * Cv + c, (v[(y - y0) * 4 + (x - x0)] + s2[q] / (float)n2[q] + m4) * (1.0f / 3.0f));
            }
    }
}

}  // namespace

extern "C" size_t gdkvm_kpff_workspace_bytes(int Ck, int Cv, int Cp, int io_dtype)
{
    if (io_dtype != GDKVM_BF16 || Ck <= 0 || Cv <= 0 || Cp <= 0) return 16;
    return ((size_t)2 * Cp * (Cp + Ck + Cv) + (size_t)Cp * Ck + (size_t)Cp * Cv) * sizeof(bf16_t) + 16;
}

extern "C" int gdkvm_kpff_fwd_train(const void* local, const void* global, const void* pixel,
                                    const float* wa, const float* ba, const float* wl, const float* wg, void* out,
                                    void* save_gates, void* save_lp, void* save_gp, void* save_gms,
                                    void* workspace, size_t workspace_bytes,
                                    int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream)
{
    const bool any = save_gates || save_lp || save_gp || save_gms, all = save_gates && save_lp && save_gp && save_gms;
    if (any && !all) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: the four save buffers go together");
    const KpffSave sv{save_gates, save_lp, save_gp, save_gms};
    if (BT < 0 || Ck <= 0 || Cv <= 0 || Cp <= 0 || h <= 0 || w <= 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: bad shape BT=%d Ck=%d Cv=%d Cp=%d h=%d w=%d", BT, Ck, Cv, Cp, h, w);
    if (Ck % 16 || Cv % 16 || Cp % 16) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: channels must be multiples of 16");
    if ((long)h * w > GDKVM_MAX_N || w > 64) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: grid %dx%d unsupported", h, w);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "kpff_fwd: io_dtype=%d", io_dtype);
    if (BT == 0) return GDKVM_OK;
    const void* ptrs[] = {local, global, pixel, wa, ba, wl, wg, out};
    for (const void* p : ptrs) {
        if (!p) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: null pointer");
        if (!gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: pointer %p is not 16-byte aligned", p);
    }
    // tile = whole frame if it fits 64 tokens, else the largest multiple of 4 grid rows that does
    int rows;
    if (h * w <= KPFF_TM) rows = h;
    else {
        rows = (KPFF_TM / w) & ~3;
        if (rows < 4) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: grid width %d too large for a 4-row tile", w);
    }
    const int tiles = (h + rows - 1) / rows;
    const int Cin = Cp + Ck + Cv;
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(BT * tiles));

    // bf16 I/O with 32-aligned channel counts: bf16 MFMA arm (weights re-packed to bf16 in the workspace each call)
    if (io_dtype == GDKVM_BF16 && Ck % 32 == 0 && Cv % 32 == 0 && Cp % 32 == 0) {
        const size_t need = gdkvm_kpff_workspace_bytes(Ck, Cv, Cp, io_dtype) - 16;
        if (!workspace || !gdkvm_aligned16(workspace)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_fwd: workspace null or misaligned");
        if (workspace_bytes < need) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "kpff_fwd: workspace %zu < %zu bytes", workspace_bytes, need + 16);
        const size_t lds = (size_t)KPFF_TM * (Cin + KPFF_PAD16) * sizeof(bf16_t);
        if (lds > 160 * 1024) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: Cp+Ck+Cv=%d exceeds the LDS tile", Cin);
        bf16_t* wab = static_cast<bf16_t*>(workspace);
        const size_t na = (size_t)2 * Cp * Cin, nl = (size_t)Cp * Ck, ng = (size_t)Cp * Cv;
        hipLaunchKernelGGL(kpff_pack_weights_kernel, dim3(256), dim3(256), 0, st, wa, wl, wg, wab, na, nl, ng);
        GDKVM_LAUNCH_CHECK("kpff_pack_weights_kernel");
        KpffBf16Args b{static_cast<const bf16_t*>(local), static_cast<const bf16_t*>(global), static_cast<const bf16_t*>(pixel),
                       wab, ba, wab + na, wab + na + nl, static_cast<bf16_t*>(out), Ck, Cv, Cp, h, w, rows, tiles, sv};
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kpff_bf16_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "kpff_fwd: LDS attribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kpff_bf16_kernel, grid, dim3(256), lds, st, b);
        GDKVM_LAUNCH_CHECK("kpff_bf16_kernel");
        return GDKVM_OK;
    }

    const size_t lds = (size_t)KPFF_TM * (Cin + KPFF_PAD) * sizeof(float);
    if (lds > 160 * 1024) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_fwd: Cp+Ck+Cv=%d exceeds the LDS tile", Cin);
    KpffArgs a{local, global, pixel, wa, ba, wl, wg, out, Ck, Cv, Cp, h, w, rows, tiles, sv};
    const void* fn = io_dtype == GDKVM_F32 ? reinterpret_cast<const void*>(kpff_kernel<GDKVM_F32>)
                                           : reinterpret_cast<const void*>(kpff_kernel<GDKVM_BF16>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "kpff_fwd: LDS attribute: %s", hipGetErrorString(e));
    }
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((kpff_kernel<GDKVM_F32>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((kpff_kernel<GDKVM_BF16>), grid, dim3(256), lds, st, a);
    GDKVM_LAUNCH_CHECK("kpff_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_kpff_fwd(const void* local, const void* global, const void* pixel,
                              const float* wa, const float* ba, const float* wl, const float* wg, void* out,
                              void* workspace, size_t workspace_bytes,
                              int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream)
{
    return gdkvm_kpff_fwd_train(local, global, pixel, wa, ba, wl, wg, out, nullptr, nullptr, nullptr, nullptr,
                                workspace, workspace_bytes, BT, Ck, Cv, Cp, h, w, io_dtype, stream);
}

extern "C" int gdkvm_kpff_bwd_pre(const void* d_out, const void* gates, const void* lp, const void* gp,
                                  void* d_z, void* d_lp, void* d_gp, int BT, int N, int Cp, int io_dtype, void* stream)
{
    if (BT < 0 || N <= 0 || Cp <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_bwd_pre: bad shape");
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "kpff_bwd_pre: io_dtype=%d", io_dtype);
    if (BT == 0) return GDKVM_OK;
    const void* ptrs[] = {d_out, gates, lp, gp, d_z, d_lp, d_gp};
    for (const void* p : ptrs) if (!p || !gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_bwd_pre: null or misaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t M = (size_t)BT * N;
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((kpff_bwd_pre_kernel<GDKVM_F32>), dim3(2048), dim3(256), 0, st, d_out, gates, lp, gp, d_z, d_lp, d_gp, M, Cp);
    else hipLaunchKernelGGL((kpff_bwd_pre_kernel<GDKVM_BF16>), dim3(2048), dim3(256), 0, st, d_out, gates, lp, gp, d_z, d_lp, d_gp, M, Cp);
    GDKVM_LAUNCH_CHECK("kpff_bwd_pre_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_kpff_bwd_post(const void* d_out, const void* d_x, const void* d_l_add, const void* d_g_add,
                                   void* d_pixel, void* d_local, void* d_global,
                                   int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream)
{
    if (BT < 0 || Ck <= 0 || Cv <= 0 || Cp <= 0 || h <= 0 || w <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "kpff_bwd_post: bad shape");
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "kpff_bwd_post: io_dtype=%d", io_dtype);
    if (BT == 0) return GDKVM_OK;
    const void* ptrs[] = {d_out, d_x, d_l_add, d_g_add, d_pixel, d_local, d_global};
    for (const void* p : ptrs) if (!p || !gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "kpff_bwd_post: null or misaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((kpff_bwd_post_kernel<GDKVM_F32>), dim3(BT), dim3(256), 0, st, d_out, d_x, d_l_add, d_g_add, d_pixel, d_local, d_global, Ck, Cv, Cp, h, w);
    else hipLaunchKernelGGL((kpff_bwd_post_kernel<GDKVM_BF16>), dim3(BT), dim3(256), 0, st, d_out, d_x, d_l_add, d_g_add, d_pixel, d_local, d_global, Ck, Cv, Cp, h, w);
    GDKVM_LAUNCH_CHECK("kpff_bwd_post_kernel");
    return GDKVM_OK;
}
