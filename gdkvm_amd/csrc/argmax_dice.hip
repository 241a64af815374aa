// argmax_dice.hip -- SURVEY.md §8 row a6: mask = argmax_c logits (ties -> lowest class index) and exact
// integer Dice counts.  HBM-bound: every logit is read once with 8/16-byte loads, the mask is written
// once; counts are reduced wave-wide with ballots, per block in LDS, then one integer atomic per
// (block, class, kind) -- integer adds commute, so the result is bit-reproducible.
#include <stdlib.h>

#include "gdkvm_common.hpp"

namespace {

struct AdArgs {
    const void* logits; const uint8_t* target; uint8_t* mask; int32_t* counts;
    int ncls, HW;
};

template <int IO, int VEC>
__global__ __launch_bounds__(256) void argmax_dice_kernel(AdArgs a)
{
    extern __shared__ int s_cnt[];                       // [ncls][3]
    const int f = blockIdx.y, ncls = a.ncls, HW = a.HW;
    const bool dice = a.target != nullptr;
    if (dice) {
        for (int i = threadIdx.x; i < ncls * 3; i += 256) s_cnt[i] = 0;
        __syncthreads();
    }
    const size_t base = (size_t)f * ncls * HW;
    const int nvec = (HW + VEC - 1) / VEC;
    for (int pv = blockIdx.x * 256 + threadIdx.x; pv < ((nvec + 255) / 256) * 256; pv += gridDim.x * 256) {
        const int p = pv * VEC;
        const bool act = pv < nvec;
        float best[VEC];
        int arg[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { best[e] = 0.f; arg[e] = 0; }
        if (act) {
            for (int c = 0; c < ncls; ++c) {
                float x[VEC];
                if constexpr (VEC == 4) {
                    const f32x4 v = load4<IO>(a.logits, base + (size_t)c * HW + p);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = v[e];
                } else {
                    x[0] = load1<IO>(a.logits, base + (size_t)c * HW + p);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (c == 0 || x[e] > best[e]) { best[e] = x[e]; arg[e] = c; }   // strict > keeps the lowest index
            }
            if constexpr (VEC == 4) {
                *reinterpret_cast<uchar4*>(a.mask + (size_t)f * HW + p) =
                    make_uchar4((unsigned char)arg[0], (unsigned char)arg[1], (unsigned char)arg[2], (unsigned char)arg[3]);
            } else {
                a.mask[(size_t)f * HW + p] = (uint8_t)arg[0];
            }
        }
        if (dice) {
            int tc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) tc[e] = -1;
            if (act) {
                if constexpr (VEC == 4) {
                    const uchar4 t4 = *reinterpret_cast<const uchar4*>(a.target + (size_t)f * HW + p);
                    tc[0] = t4.x; tc[1] = t4.y; tc[2] = t4.z; tc[3] = t4.w;
                } else {
                    tc[0] = a.target[(size_t)f * HW + p];
                }
            }
            for (int c = 0; c < ncls; ++c) {
                int ni = 0, np = 0, nt = 0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const unsigned long long mp = __ballot(act && arg[e] == c);
                    const unsigned long long mt = __ballot(tc[e] == c);
                    np += __popcll(mp); nt += __popcll(mt); ni += __popcll(mp & mt);
                }
                if ((threadIdx.x & 63) == 0 && (np | nt)) {
                    if (ni) atomicAdd(&s_cnt[c * 3 + 0], ni);
                    if (np) atomicAdd(&s_cnt[c * 3 + 1], np);
                    if (nt) atomicAdd(&s_cnt[c * 3 + 2], nt);
                }
            }
        }
    }
    if (dice) {
        __syncthreads();
        for (int i = threadIdx.x; i < ncls * 3; i += 256)
            if (s_cnt[i]) atomicAdd(&a.counts[(size_t)f * ncls * 3 + i], s_cnt[i]);
    }
}


// Fused form: bilinear upsampling (align_corners = false, PyTorch's formula) of low-resolution logits + argmax + Dice.
// The full-resolution logits are never written: per frame 2*hl*wl*ncls bytes are read (L2-resident, re-read per
// output row) and H*W mask bytes written.  Arithmetic is fp32 with explicitly un-fused multiplies and adds
// (__fmul_rn/__fadd_rn), so the result is bit-identical to the scalar oracle and near-ties resolve the same way.
struct UpArgs {
    const void* logits; const uint8_t* target; uint8_t* mask; int32_t* counts;
    int ncls, hl, wl, H, W;
    float sy, sx;
    int staged;                                          // the block's low-resolution logit rows go through LDS (fp32): taps are LDS reads
    // head fused in (gdkvm_head_upsample_argmax_dice): `logits` is the decoder feature [BT, hl, wl, C] (NHWC) and the class planes
    // are computed into LDS by the block itself, with gdkvm_head_logits' arithmetic and its rounding to the io dtype
    const float* hw_; const float* hb; int C;
    int lr_cap;                                          // low-resolution rows the LDS tile holds
};

__device__ __forceinline__ float up_src(float scale, int dst)
{
    const float s = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
    return s < 0.f ? 0.f : s;
}
// the same three roundings on the host (volatile: no contraction into an fma, no excess precision)
static float up_src_host(float scale, int dst)
{
    volatile float t = (float)dst + 0.5f;
    volatile float m = scale * t;
    volatile float s = m - 0.5f;
    return s < 0.f ? 0.f : s;
}

// NC = number of classes when it is 2..4 (loops unrolled, the Dice counters of a thread in registers and reduced once per
// block), 0 = any (per-pixel wave ballots).
template <int IO, int NC>
__global__ __launch_bounds__(256) void upsample_argmax_dice_kernel(UpArgs a)
{
    extern __shared__ int s_cnt[];                       // [ncls][3] counters, then (staged) [ncls][hl*wl] fp32 logits of the frame
    const int f = blockIdx.y, ncls = NC ? NC : a.ncls, HW = a.H * a.W, hw = a.hl * a.wl;
    const bool dice = a.target != nullptr;
    int cnt[NC ? NC : 1][3];
#pragma unroll
    for (int c = 0; c < (NC ? NC : 1); ++c) cnt[c][0] = cnt[c][1] = cnt[c][2] = 0;
    // Every output pixel blends four taps per class; as 2-byte global loads that is 32 vector-memory instructions per thread and the
    // kernel is bound by their issue (20 us at 512 frames of 112x112 for 14 MB of traffic).  The frame's low-resolution planes are a
    // few KB: staged once per block, widened to fp32, the taps become LDS reads -- same values, same arithmetic, same bits.
    float* s_log = reinterpret_cast<float*>(s_cnt + ((ncls * 3 + 3) & ~3));
    // PX consecutive pixels of a row per thread (4 when W % 4 == 0): one 4-byte mask store and one 4-byte target load
    // instead of four 1-byte ones, and the vertical taps / weights computed once per thread.  A block owns a contiguous range of
    // them, hence a contiguous band of low-resolution rows [lr0, lr1].
    const int PX = (a.W % 4 == 0) ? 4 : 1, nq = HW / PX;
    const int per = (((nq + (int)gridDim.x - 1) / (int)gridDim.x + 255) / 256) * 256;
    const int q_lo = blockIdx.x * per, q_hi = min(nq, q_lo + per);
    int lr0 = 0, lr1 = a.hl - 1;
    if (q_lo < q_hi) {
        lr0 = (int)up_src(a.sy, (q_lo * PX) / a.W);
        lr1 = min((int)up_src(a.sy, (q_hi * PX - 1) / a.W) + 1, a.hl - 1);
    }
    const int npl = (lr1 - lr0 + 1) * a.wl;              // low-resolution pixels of the band
    const bool staged = a.staged && lr1 - lr0 + 1 <= a.lr_cap;
    // Head fused in: `logits` is the NHWC decoder FEATURE, which the unstaged taps below would read as NCHW class planes (in bounds,
    // wrong values).  The host computes lr_cap with THIS arithmetic for every block of the launch (up_src_host: the same fp32
    // operations) and refuses the call with GDKVM_ERR_SHAPE when a band does not fit the LDS tile, so the condition cannot hold here;
    // a block that met it anyway fails LOUDLY in its output: every mask byte of its range becomes 255 (no class has that index, so the
    // mask is visibly wrong and its Dice counts stay zero) instead of whatever torch.empty held (GDKVM_DEBUG_TRAPS builds: it faults).
    if (a.hw_ && !staged) {
#ifdef GDKVM_DEBUG_TRAPS
        __builtin_trap();
#endif
        for (int p = q_lo * PX + (int)threadIdx.x; p < q_hi * PX; p += 256) a.mask[(size_t)f * HW + p] = 255;
        return;
    }
    if (staged && a.hw_) {
        // the head on the band: C/V lanes per pixel, lane cg keeps class cg (as head_logits_kernel; same sums, same rounding)
        constexpr int V = IO == GDKVM_F32 ? 4 : 8;
        const int G = a.C / V, ppw = 64 / G;
        const int lane = threadIdx.x & 63, sub = lane / G, cg = lane % G, wv = threadIdx.x >> 6;
        const uint4* xv = static_cast<const uint4*>(a.logits) + ((size_t)f * hw + (size_t)lr0 * a.wl) * G;
        // 2..4 classes: this lane's V weights of every class and the biases live in registers for the whole band (they were fetched
        // per pixel group: 8 ncls vector loads in front of 8 ncls FMAs), and eight lanes per pixel (bf16 I/O, C = 64) sum over DPP
        // lane permutations in the butterfly's order (xor 4, 2, 1: the same additions in the same order, hence the same bits) instead
        // of three LDS permutes per class
        float wreg[NC ? NC : 1][V], breg[NC ? NC : 1];
        if constexpr (NC > 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int j = 0; j < V; ++j) wreg[c][j] = a.hw_[(size_t)c * a.C + cg * V + j];
                breg[c] = a.hb[c];
            }
        }
        const bool dpp8 = NC > 0 && G == 8;               // (uniform)
        for (int p0 = wv * ppw; p0 < npl; p0 += 4 * ppw) {
            const int p = p0 + sub;
            const uint4 v4 = xv[(size_t)min(p, npl - 1) * G + cg];
            const unsigned xw[4] = {v4.x, v4.y, v4.z, v4.w};
            float v[V];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (IO == GDKVM_F32) v[j] = __uint_as_float(xw[j]);
                else { v[2 * j] = __uint_as_float(xw[j] << 16); v[2 * j + 1] = __uint_as_float(xw[j] & 0xffff0000u); }
            }
            float mine = 0.f;
            if constexpr (NC > 0) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < V; ++j) d = fmaf(v[j], wreg[c][j], d);
                    if (dpp8) {
                        // lane i ^ 4: the half-row mirror (i -> 7 - i) followed by the quad reversal; i ^ 2, i ^ 1: quad permutations
                        const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x141, 0xf, 0xf, false);        // row_half_mirror
                        d += __int_as_float(__builtin_amdgcn_update_dpp(0, m, 0x1b, 0xf, 0xf, false));                   // quad_perm [3,2,1,0]
                        d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
                        d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
                    } else {
                        for (int o = G >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o);
                    }
                    if (cg == c) mine = d + breg[c];
                }
            } else {
                for (int c = 0; c < ncls; ++c) {
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < V; ++j) d = fmaf(v[j], a.hw_[(size_t)c * a.C + cg * V + j], d);
                    for (int o = G >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o);
                    if (cg == c) mine = d + a.hb[c];
                }
            }
            if (p < npl && cg < ncls) s_log[cg * npl + p] = IO == GDKVM_F32 ? mine : bf16_to_f32(f32_to_bf16(mine));
        }
    } else if (staged) {
        for (int i = threadIdx.x; i < ncls * npl; i += 256) {
            const int c = i / npl, r = i - c * npl;
            s_log[i] = load1<IO>(a.logits, ((size_t)f * ncls + c) * hw + (size_t)lr0 * a.wl + r);
        }
    }
    if (dice)
        for (int i = threadIdx.x; i < ncls * 3; i += 256) s_cnt[i] = 0;
    if (dice || staged) __syncthreads();
    for (int qd = q_lo + threadIdx.x; qd < q_lo + per; qd += 256) {
        const bool act = qd < q_hi;
        int arg[4] = {0, 0, 0, 0};
        const int p0 = qd * PX;
        if (act) {
            const int y = p0 / a.W, xb = p0 - y * a.W;
            const float fy = up_src(a.sy, y);
            const int y0 = (int)fy, y1 = min(y0 + 1, a.hl - 1);
            const float ly = __fsub_rn(fy, (float)y0), hy = __fsub_rn(1.0f, ly);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= PX) break;
                const float fx = up_src(a.sx, xb + e);
                const int x0 = (int)fx, x1 = min(x0 + 1, a.wl - 1);
                const float lx = __fsub_rn(fx, (float)x0), hx = __fsub_rn(1.0f, lx);
                float best = 0.f;
#pragma unroll
                for (int c = 0; c < ncls; ++c) {
                    const size_t base = ((size_t)f * ncls + c) * hw;
                    float v00, v01, v10, v11;
                    if (staged) {
                        const float* pl = s_log + c * npl - lr0 * a.wl;
                        v00 = pl[y0 * a.wl + x0]; v01 = pl[y0 * a.wl + x1]; v10 = pl[y1 * a.wl + x0]; v11 = pl[y1 * a.wl + x1];
                    } else {
                        v00 = load1<IO>(a.logits, base + y0 * a.wl + x0); v01 = load1<IO>(a.logits, base + y0 * a.wl + x1);
                        v10 = load1<IO>(a.logits, base + y1 * a.wl + x0); v11 = load1<IO>(a.logits, base + y1 * a.wl + x1);
                    }
                    const float top = __fadd_rn(__fmul_rn(hx, v00), __fmul_rn(lx, v01));
                    const float bot = __fadd_rn(__fmul_rn(hx, v10), __fmul_rn(lx, v11));
                    const float v = __fadd_rn(__fmul_rn(hy, top), __fmul_rn(ly, bot));
                    if (c == 0 || v > best) { best = v; arg[e] = c; }
                }
            }
            if (PX == 4) *reinterpret_cast<unsigned*>(a.mask + (size_t)f * HW + p0) = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
            else a.mask[(size_t)f * HW + p0] = (uint8_t)arg[0];
        }
        if (dice) {
            unsigned tw = 0xffffffffu;
            if (act) tw = PX == 4 ? *reinterpret_cast<const unsigned*>(a.target + (size_t)f * HW + p0) : (0xffffff00u | a.target[(size_t)f * HW + p0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= PX) break;
                const int tc = act ? (int)((tw >> (8 * e)) & 0xff) : -1;
                if constexpr (NC > 0) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int ip = (act && arg[e] == c) ? 1 : 0, it = (tc == c) ? 1 : 0;
                        cnt[c][0] += ip & it; cnt[c][1] += ip; cnt[c][2] += it;
                    }
                    continue;
                }
                for (int c = 0; c < ncls; ++c) {
                    const unsigned long long mp = __ballot(act && arg[e] == c), mt = __ballot(tc == c);
                    if ((threadIdx.x & 63) == 0 && (mp | mt)) {
                        const int ni = __popcll(mp & mt), np = __popcll(mp), nt = __popcll(mt);
                        if (ni) atomicAdd(&s_cnt[c * 3 + 0], ni);
                        if (np) atomicAdd(&s_cnt[c * 3 + 1], np);
                        if (nt) atomicAdd(&s_cnt[c * 3 + 2], nt);
                    }
                }
            }
        }
    }
    if (dice) {
        if constexpr (NC > 0) {                            // one wave reduction per counter, one LDS atomic per wave
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    int v = cnt[c][k];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt[c * 3 + k], v);
                }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < ncls * 3; i += 256)
            if (s_cnt[i]) atomicAdd(&a.counts[(size_t)f * ncls * 3 + i], s_cnt[i]);
    }
}

}  // namespace

// The Dice counts are accumulated with integer atomics: they start from zero.  Zeroed by a KERNEL (gdkvm_zero_async), not hipMemsetAsync
// (round 6): as a memset node of a captured graph the 576-byte fill of a counts tensor that lives in the graph's own memory pool ran on the
// first replay only -- GraphedSegment(streams=1) with a target returned the first replay's counts plus whatever the pool block held
// afterwards (tests/test_model_gpu.py::test_forwards_in_flight...; the two-stream form, whose counts tensor is allocated outside the capture,
// was not affected).
static int zero_counts(int32_t* counts, size_t n, hipStream_t st) { return gdkvm_zero_async(counts, n * sizeof(int32_t), st); }

extern "C" int gdkvm_argmax_dice(const void* logits, const uint8_t* target, uint8_t* mask, int32_t* counts,
                                 int BT, int ncls, int H, int W, int io_dtype, void* stream)
{
    if (BT < 0 || ncls <= 0 || ncls > 255 || H <= 0 || W <= 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "argmax_dice: bad shape BT=%d ncls=%d H=%d W=%d", BT, ncls, H, W);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "argmax_dice: io_dtype=%d", io_dtype);
    if (BT == 0) return GDKVM_OK;
    if (!logits || !mask) return gdkvm_fail(GDKVM_ERR_ARG, "argmax_dice: null pointer");
    if (target && !counts) return gdkvm_fail(GDKVM_ERR_ARG, "argmax_dice: counts required with a target");
    if (!gdkvm_aligned16(logits) || !gdkvm_aligned16(mask) || (target && !gdkvm_aligned16(target)) ||
        (counts && !gdkvm_aligned16(counts)))
        return gdkvm_fail(GDKVM_ERR_ARG, "argmax_dice: pointers must be 16-byte aligned");
    if ((size_t)H * W > 0x7fffffffu / 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "argmax_dice: image too large");
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int HW = H * W;
    if (target) if (int rc = zero_counts(counts, (size_t)BT * ncls * 3, st)) return rc;
    AdArgs a{logits, target, mask, counts, ncls, HW};
    const bool vec = (HW % 4) == 0;
    const int nvec = vec ? HW / 4 : HW;
    int gx = (nvec + 255) / 256;
    if (gx > 64) gx = 64;
    const dim3 grid((unsigned)gx, (unsigned)BT);
    const size_t lds = sizeof(int) * (size_t)ncls * 3;
    if (io_dtype == GDKVM_F32) {
        if (vec) hipLaunchKernelGGL((argmax_dice_kernel<GDKVM_F32, 4>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((argmax_dice_kernel<GDKVM_F32, 1>), grid, dim3(256), lds, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((argmax_dice_kernel<GDKVM_BF16, 4>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((argmax_dice_kernel<GDKVM_BF16, 1>), grid, dim3(256), lds, st, a);
    }
    GDKVM_LAUNCH_CHECK("argmax_dice_kernel");
    return GDKVM_OK;
}

static int upsample_launch(const char* who, const void* src, const float* head_w, const float* head_b, int C,
                           const uint8_t* target, uint8_t* mask, int32_t* counts,
                           int BT, int ncls, int hl, int wl, int H, int W, int io_dtype, void* stream)
{
    if (BT < 0 || ncls <= 0 || ncls > 255 || hl <= 0 || wl <= 0 || H <= 0 || W <= 0)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: bad shape BT=%d ncls=%d %dx%d -> %dx%d", who, BT, ncls, hl, wl, H, W);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: io_dtype=%d", who, io_dtype);
    if (head_w) {
        const int V = io_dtype == GDKVM_F32 ? 4 : 8, G = C > 0 ? C / V : 0;
        if (C <= 0 || C % V || G > 64 || (G & (G - 1)) || ncls > G)
            return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: C=%d classes=%d (C/%d a power of two <= 64, classes <= C/%d)", who, C, ncls, V, V);
    }
    if (BT == 0) return GDKVM_OK;
    if (!src || !mask || (head_w && !head_b)) return gdkvm_fail(GDKVM_ERR_ARG, "%s: null pointer", who);
    if (target && !counts) return gdkvm_fail(GDKVM_ERR_ARG, "%s: counts required with a target", who);
    if (!gdkvm_aligned16(src) || !gdkvm_aligned16(mask) || (target && !gdkvm_aligned16(target)) || (counts && !gdkvm_aligned16(counts)))
        return gdkvm_fail(GDKVM_ERR_ARG, "%s: pointers must be 16-byte aligned", who);
    if ((size_t)H * W > 0x7fffffffu / 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: image too large", who);
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (target) if (int rc = zero_counts(counts, (size_t)BT * ncls * 3, st)) return rc;
    const int nq = (W % 4 == 0) ? H * W / 4 : H * W;       // work items per frame (pixel quads when rows allow)
    int gx = (nq + 255) / 256;
    static const int cap_env = [] { const char* e = getenv("GDKVM_ARGMAX_BLOCKS"); return e ? atoi(e) : 0; }();   // (A/B runs)
    // enough blocks to fill the chip, few enough that launch and the per-block count reduction do not dominate.  256 .. 1023 frames: THREE
    // (round 5, 512 frames of 112 x 112: 20.7 us against 23.8 with four, 22.0 with two; the ranges are multiples of 256 quads, so three
    // blocks own 1280 / 1280 / 576 of a frame's 3136 -- four owned 1024 / 1024 / 1024 / 64, the last staging a band for 64 quads -- and
    // blocks of unequal length fall out of step: one's memory-bound band phase overlaps another's arithmetic; evenly split ranges
    // measured 21.3 - 22.3)
    const int cap = cap_env > 0 ? cap_env : BT >= 1024 ? 2 : (BT >= 256 ? 3 : 16);
    if (gx > cap) gx = cap;
    // low-resolution rows a block's contiguous output range touches: the kernel's own arithmetic (band of block bx = rows
    // up_src(q_lo) .. up_src(q_hi - 1) + 1), evaluated here for every block of the launch -- lr_cap is the largest band, exactly
    const int per = (((nq + gx - 1) / gx + 255) / 256) * 256, PX = (W % 4 == 0) ? 4 : 1;
    const float sy_f = (float)hl / (float)H;
    int lr_cap = 1;
    for (int bx = 0; bx < gx; ++bx) {
        const int q_lo = bx * per, q_hi = nq < q_lo + per ? nq : q_lo + per;
        if (q_lo >= q_hi) continue;
        const int lr0 = (int)up_src_host(sy_f, (q_lo * PX) / W);
        int lr1 = (int)up_src_host(sy_f, (q_hi * PX - 1) / W) + 1;
        if (lr1 > hl - 1) lr1 = hl - 1;
        if (lr1 - lr0 + 1 > lr_cap) lr_cap = lr1 - lr0 + 1;
    }
    const size_t cnt_bytes = sizeof(int) * (size_t)((ncls * 3 + 3) & ~3), log_bytes = sizeof(float) * (size_t)ncls * lr_cap * wl;
    const int staged = cnt_bytes + log_bytes <= 48 * 1024;
    if (head_w && !staged) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: %d classes x %d x %d low-resolution rows do not fit the LDS tile", who, ncls, lr_cap, wl);
    UpArgs a{src, target, mask, counts, ncls, hl, wl, H, W, (float)hl / (float)H, (float)wl / (float)W, staged, head_w, head_b, C, lr_cap};
    const dim3 grid((unsigned)gx, (unsigned)BT);
    const size_t lds = cnt_bytes + (staged ? log_bytes : 0);
#define GDKVM_UP_LAUNCH(IO)                                                                                                   \
    switch (ncls) {                                                                                                           \
        case 2: hipLaunchKernelGGL((upsample_argmax_dice_kernel<IO, 2>), grid, dim3(256), lds, st, a); break;                 \
        case 3: hipLaunchKernelGGL((upsample_argmax_dice_kernel<IO, 3>), grid, dim3(256), lds, st, a); break;                 \
        case 4: hipLaunchKernelGGL((upsample_argmax_dice_kernel<IO, 4>), grid, dim3(256), lds, st, a); break;                 \
        default: hipLaunchKernelGGL((upsample_argmax_dice_kernel<IO, 0>), grid, dim3(256), lds, st, a);                       \
    }
    if (io_dtype == GDKVM_F32) { GDKVM_UP_LAUNCH(GDKVM_F32) } else { GDKVM_UP_LAUNCH(GDKVM_BF16) }
#undef GDKVM_UP_LAUNCH
    GDKVM_LAUNCH_CHECK("upsample_argmax_dice_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_upsample_argmax_dice(const void* logits, const uint8_t* target, uint8_t* mask, int32_t* counts,
                                          int BT, int ncls, int hl, int wl, int H, int W, int io_dtype, void* stream)
{
    return upsample_launch("upsample_argmax_dice", logits, nullptr, nullptr, 0, target, mask, counts, BT, ncls, hl, wl, H, W, io_dtype, stream);
}

// The decoder's head folded in: x [BT, hl, wl, C] is the stride-4 feature (NHWC), w [ncls, C] and b [ncls] the 1x1 head; the class
// planes exist only as the LDS bands of the blocks -- bit-identical to gdkvm_head_logits followed by gdkvm_upsample_argmax_dice.
extern "C" int gdkvm_head_upsample_argmax_dice(const void* x, const float* w, const float* b, const uint8_t* target, uint8_t* mask,
                                               int32_t* counts, int BT, int C, int ncls, int hl, int wl, int H, int W, int io_dtype,
                                               void* stream)
{
    if (!w) return gdkvm_fail(GDKVM_ERR_ARG, "head_upsample_argmax_dice: null pointer");
    return upsample_launch("head_upsample_argmax_dice", x, w, b, C, target, mask, counts, BT, ncls, hl, wl, H, W, io_dtype, stream);
}
