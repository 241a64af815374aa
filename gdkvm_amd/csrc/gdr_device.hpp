// gdr_device.hpp -- device helpers shared by the GDR kernels (gdr_prep.hip, gdr_scan.hip): in-kernel diagnostic stamps,
// compile-time loops, the three-term bf16 operand format (split3) and its MFMA image layout.  Not part of the C ABI.
#pragma once
#include <type_traits>

#include "gdkvm_common.hpp"

#ifdef GDKVM_DIAG
// Diagnostic build only (libgdkvm_hip_diag.so, built by tools/diag_scan.py): wave 0 of block 0 stamps s_memtime at
// five points per frame into a buffer of its own.  Never compiled into the product library.
extern unsigned long long* g_gdkvm_diag_buf;     // defined in gdr_scan.hip (gdkvm_diag_set_buffer)
#define DIAG_STAMP(slot)                                                                      \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t__;                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (a.diag && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && tid == 0) a.diag[(size_t)t * 8 + (slot)] = t__;      \
    } while (0)
// a second row of stamps (row T - 1) for points inside a phase
#define DIAG_STAMP2(slot)                                                                     \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t__;                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (a.diag && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && tid == 0) a.diag[(size_t)(t - 1) * 8 + (slot)] = t__; \
    } while (0)
#else
#define DIAG_STAMP(slot) do {} while (0)
#define DIAG_STAMP2(slot) do {} while (0)
#endif

template <int I, int E, class F>
static __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// x = h + m + l with h, m, l bfloat16: 24 significant bits, every step exact in fp32.  A product of such a triple with an
// exact bf16 operand on the bf16 MFMA (fp32 accumulate) is as accurate as the fp32 MFMA at 3/16 of its issue cycles; a
// product of two triples needs the six terms down to 2^-16 (hh, hm, mh, hl, lh, mm): 12 MFMA of 16 cycles for K = 64
// instead of 16 of 32.
static __device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l)
{
    h = static_cast<__bf16>(x);
    const float r1 = x - static_cast<float>(h);
    m = static_cast<__bf16>(r1);
    l = static_cast<__bf16>(r1 - static_cast<float>(m));
}
// split3 of four values at once on the packed converter: the three terms as 4 x bf16 each
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
static __device__ __forceinline__ void split3x4(const f32x4& x, uint2& h, uint2& m, uint2& l)
{
    h = make_uint2(cvt_pk_bf16(x[0], x[1]), cvt_pk_bf16(x[2], x[3]));
    const float r0 = x[0] - __uint_as_float(h.x << 16), r1 = x[1] - __uint_as_float(h.x & 0xffff0000u);
    const float r2 = x[2] - __uint_as_float(h.y << 16), r3 = x[3] - __uint_as_float(h.y & 0xffff0000u);
    m = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
    l = make_uint2(cvt_pk_bf16(r0 - __uint_as_float(m.x << 16), r1 - __uint_as_float(m.x & 0xffff0000u)),
                   cvt_pk_bf16(r2 - __uint_as_float(m.y << 16), r3 - __uint_as_float(m.y & 0xffff0000u)));
}
static __device__ __forceinline__ uint2 pack_bf16x4(const __bf16 (&x)[4])
{
    return make_uint2((unsigned)__builtin_bit_cast(unsigned short, x[0]) | ((unsigned)__builtin_bit_cast(unsigned short, x[1]) << 16),
                      (unsigned)__builtin_bit_cast(unsigned short, x[2]) | ((unsigned)__builtin_bit_cast(unsigned short, x[3]) << 16));
}
// Four consecutive k values (k0 = 16c + 4g .. +3, accumulator layout of k tile c) of row `row16` of a 16-row tile go to the
// 16-bit A/B-operand image of that tile (natural k order: element j of lane group G of k step s is k = 32s + 8G + j) as half
// (g & 1) of lane (2(c&1) + (g>>1), row16) of k step c >> 1.  Returns the index in 8-byte units inside a [2 ksteps][64 lanes]
// [2 halves] image.
static __device__ __forceinline__ int split_slot(int c, int g, int row16)
{
    return ((((c >> 1) * 64) + (2 * (c & 1) + (g >> 1)) * 16 + row16) << 1) + (g & 1);
}
static constexpr int SPLIT_IMG = 2 * 64 * 2;                    // uint2 per term image of one 16-row tile (K = 64)

// ---- pair16: the forward recurrence's operand format --------------------------------------------------------------------
// x * scale = h + 2^-11 l  with h, l IEEE half:  h = fp16(x scale),  l = fp16((x scale - h) 2^11).  22 significant bits like
// a plain fp16 pair, but the low term is carried at the high term's exponent (its 2^-11 is applied to a separate accumulator
// chain after the MFMAs), so it does not go subnormal until |x scale| < 2^-24 / 2^11: full precision over ~13 decades.  A
// product of two such pairs needs three f16 MFMAs per k step (hh on the main chain; hl, lh on the cross chain; ll = 2^-22 is
// dropped) against six for two bf16 triples, and two operand images instead of three.  |x scale| saturates at 65504.
// The state is published with scale = 2^-e, e = 4 by default (|S| up to 1e6) and larger when the serial kernel's bound on the
// state asks for it (gdr_scan.hip, "the state's exponent"); P and the composed maps use scale 1.  The backward's reverse
// recurrence keeps split3: gradients have no natural magnitude.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
static constexpr float PAIR_LO = 2048.0f, PAIR_LO_INV = 1.0f / 2048.0f;
static constexpr float STATE_SCALE = 0.0625f, STATE_SCALE_INV = 16.0f;
static __device__ __forceinline__ unsigned cvt_pk_f16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, f16x2_t));
}
// Conversions round toward zero (v_cvt_pkrtz_f16_f32): beyond the half range the result is the largest finite half, not an
// infinity, so a state that outgrows the format saturates instead of turning into NaNs; the low term absorbs the (exactly
// representable) truncation error of the high one, so the pair still carries 21-22 bits.
static __device__ __forceinline__ unsigned cvt_pkrtz_f16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// four values (already multiplied by the format's scale) -> the two term images' 8-byte pieces
static __device__ __forceinline__ void pair16x4(const f32x4& x, uint2& h, uint2& l)
{
    h = make_uint2(cvt_pkrtz_f16(x[0], x[1]), cvt_pkrtz_f16(x[2], x[3]));
    // l = (x - h) 2^11, formed as fma(h, -2^11, x 2^11): the same bits (x 2^11 and h 2^11 are exact, and so is x - h), but the half is
    // read by the fma itself (v_fma_mix_f32) and x 2^11 is a packed multiply -- six instructions for four values instead of twelve
    // (convert, subtract, multiply each), on the serial scan's dependent chain
    const f16x2_t h01 = __builtin_bit_cast(f16x2_t, h.x), h23 = __builtin_bit_cast(f16x2_t, h.y);
    const f32x2_t s01 = (f32x2_t){x[0], x[1]} * (f32x2_t){PAIR_LO, PAIR_LO}, s23 = (f32x2_t){x[2], x[3]} * (f32x2_t){PAIR_LO, PAIR_LO};
    l = make_uint2(cvt_pkrtz_f16(__builtin_fmaf(static_cast<float>(h01[0]), -PAIR_LO, s01[0]), __builtin_fmaf(static_cast<float>(h01[1]), -PAIR_LO, s01[1])),
                   cvt_pkrtz_f16(__builtin_fmaf(static_cast<float>(h23[0]), -PAIR_LO, s23[0]), __builtin_fmaf(static_cast<float>(h23[1]), -PAIR_LO, s23[1])));
}
static __device__ __forceinline__ void pair16(float x, _Float16& h, _Float16& l)
{
    h = __builtin_bit_cast(f16x2_t, cvt_pkrtz_f16(x, 0.f))[0];
    l = __builtin_bit_cast(f16x2_t, cvt_pkrtz_f16((x - static_cast<float>(h)) * PAIR_LO, 0.f))[0];
}
static __device__ __forceinline__ uint2 pack_f16x4(const _Float16 (&x)[4])
{
    return make_uint2((unsigned)__builtin_bit_cast(unsigned short, x[0]) | ((unsigned)__builtin_bit_cast(unsigned short, x[1]) << 16),
                      (unsigned)__builtin_bit_cast(unsigned short, x[2]) | ((unsigned)__builtin_bit_cast(unsigned short, x[3]) << 16));
}
// Eight bf16 values times an exact power of two -> eight halves (the read-out's q operand beside pair16 images of S).  With
// sc = pow2_floor(1 / |q|) the scaled row has norm in [1/2, 1]: every element above 2^-14 of it converts exactly.
static __device__ __forceinline__ float pow2_floor(float x) { return __uint_as_float(__float_as_uint(x) & 0x7f800000u); }
static __device__ __forceinline__ float pow2_inv(float p) { return __uint_as_float(0x7f000000u - (__float_as_uint(p) & 0x7f800000u)); }
static __device__ __forceinline__ f16x8 bf16x8_to_f16(const bf16x8& q, float sc)
{
    const uint4 u = __builtin_bit_cast(uint4, q);
    return __builtin_bit_cast(f16x8, make_uint4(cvt_pk_f16(__uint_as_float(u.x << 16) * sc, __uint_as_float(u.x & 0xffff0000u) * sc),
                                                cvt_pk_f16(__uint_as_float(u.y << 16) * sc, __uint_as_float(u.y & 0xffff0000u) * sc),
                                                cvt_pk_f16(__uint_as_float(u.z << 16) * sc, __uint_as_float(u.z & 0xffff0000u) * sc),
                                                cvt_pk_f16(__uint_as_float(u.w << 16) * sc, __uint_as_float(u.w & 0xffff0000u) * sc)));
}
// Operand formats of the recurrence kernels (scan, chunk composition).  FMT_PAIR16 is the forward's default; FMT_SPLIT3 (three
// bf16 terms: the whole fp32 range at 24 bits, twice the MFMAs) serves GDKVM_FLAG_WIDE_RANGE -- rule delta_parallel is not
// contractive and may outgrow pair16's 1e6 -- and the backward's reverse recurrence.  Images are [term][k step][lane] x 16 bytes.
constexpr int FMT_SPLIT3 = 0, FMT_PAIR16 = 1;
__host__ __device__ constexpr int fmt_terms(int fmt) { return fmt == FMT_PAIR16 ? 2 : 3; }
template <int FMT> struct OpFmt {
    static constexpr int NT = fmt_terms(FMT);
    static constexpr bool PAIR = FMT == FMT_PAIR16;
    // scale of the state inside the recurrence kernels: they carry S' = S * STATE (prep hands over G * STATE), so that nothing
    // is rescaled per frame; only s_in / s_out / s_hist and the read-out's final factor convert
    static constexpr float STATE = PAIR ? STATE_SCALE : 1.0f, STATE_INV = PAIR ? STATE_SCALE_INV : 1.0f;
    // four values (in the format's scale) -> the 8-byte pieces of their term images
    static __device__ __forceinline__ void split4(const f32x4& x, uint2 (&t)[3])
    {
        if constexpr (PAIR) pair16x4(x, t[0], t[1]);
        else split3x4(x, t[0], t[1], t[2]);
    }
    static __device__ __forceinline__ void split1(float x, unsigned short (&t)[3])
    {
        if constexpr (PAIR) {
            _Float16 h, l;
            pair16(x, h, l);
            t[0] = __builtin_bit_cast(unsigned short, h); t[1] = __builtin_bit_cast(unsigned short, l);
        } else {
            __bf16 h, m, l;
            split3(x, h, m, l);
            t[0] = __builtin_bit_cast(unsigned short, h); t[1] = __builtin_bit_cast(unsigned short, m); t[2] = __builtin_bit_cast(unsigned short, l);
        }
    }
    // one 32-deep k step of that product on two accumulator chains (pair16: acc0 = hh, acc1 = hl + lh carried at 2^11)
    static __device__ __forceinline__ void mma_ks(f32x4& acc0, f32x4& acc1, const uint4 (&a)[NT][2], const uint4 (&b)[NT][2], int ks)
    {
        if constexpr (PAIR) {
#define GDKVM_MM(ACC, AT, BT) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[AT][ks]), __builtin_bit_cast(f16x8, b[BT][ks]), ACC, 0, 0, 0)
            GDKVM_MM(acc1, 1, 0); GDKVM_MM(acc0, 0, 0); GDKVM_MM(acc1, 0, 1);
#undef GDKVM_MM
        } else {
#define GDKVM_MM(ACC, AT, BT) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[AT][ks]), __builtin_bit_cast(bf16x8, b[BT][ks]), ACC, 0, 0, 0)
            GDKVM_MM(acc0, 2, 0); GDKVM_MM(acc1, 0, 2); GDKVM_MM(acc0, 1, 1); GDKVM_MM(acc1, 1, 0); GDKVM_MM(acc0, 0, 1); GDKVM_MM(acc1, 0, 0);
#undef GDKVM_MM
        }
    }
    static __device__ __forceinline__ f32x4 combine(const f32x4& acc0, const f32x4& acc1)
    {
        if constexpr (PAIR) return acc0 + PAIR_LO_INV * acc1;
        else return acc0 + acc1;
    }
    // [16 x 64] x [64 x 16]: A term images a[term][k step], B term images b[term][k step]; every product term above 2^-22 (pair16:
    // hh on one chain, hl + lh on the chain carried at 2^11) or 2^-16 of split3 (six terms, smallest first, two chains)
    // pair16: the two chains before they are combined (acc0 + 2^-11 acc1) -- the serial scan folds the combination into packed FMAs
    static __device__ __forceinline__ void product_pair(const uint4 (&a)[NT][2], const uint4 (&b)[NT][2], f32x4& acc0, f32x4& acc1)
    {
        static_assert(PAIR, "pair16 only");
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#define GDKVM_MM(ACC, AT, BT, KS) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[AT][KS]), __builtin_bit_cast(f16x8, b[BT][KS]), ACC, 0, 0, 0)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { GDKVM_MM(acc1, 1, 0, ks); GDKVM_MM(acc0, 0, 0, ks); GDKVM_MM(acc1, 0, 1, ks); }
#undef GDKVM_MM
    }
    static __device__ __forceinline__ f32x4 product(const uint4 (&a)[NT][2], const uint4 (&b)[NT][2])
    {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PAIR) {
#define GDKVM_MM(ACC, AT, BT, KS) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[AT][KS]), __builtin_bit_cast(f16x8, b[BT][KS]), ACC, 0, 0, 0)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { GDKVM_MM(acc1, 1, 0, ks); GDKVM_MM(acc0, 0, 0, ks); GDKVM_MM(acc1, 0, 1, ks); }
#undef GDKVM_MM
            return acc0 + PAIR_LO_INV * acc1;
        } else {
#define GDKVM_MM(ACC, AT, BT, KS) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[AT][KS]), __builtin_bit_cast(bf16x8, b[BT][KS]), ACC, 0, 0, 0)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { GDKVM_MM(acc0, 2, 0, ks); GDKVM_MM(acc1, 0, 2, ks); GDKVM_MM(acc0, 1, 1, ks); }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { GDKVM_MM(acc1, 1, 0, ks); GDKVM_MM(acc0, 0, 1, ks); }
            GDKVM_MM(acc1, 0, 0, 0); GDKVM_MM(acc0, 0, 0, 1);
#undef GDKVM_MM
            return acc0 + acc1;
        }
    }
};

// ---- wave-wide reductions on DPP row rotations (all 64 lanes active; the result is uniform) -------------------------------
// row_ror:n (dpp_ctrl 0x120 + n) rotates within each row of 16 lanes: four steps leave every lane with its row's result, the
// four rows are combined through readlane.  Used once per tile / per launch, never inside the serial chain.
template <int CTRL> static __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
static __device__ __forceinline__ float wave_max_nonneg(float x)          // x >= 0 (or NaN / +inf, which win): compares as integers
{
    int v = __float_as_int(x);
    v = max(v, dpp_i32<0x121>(v));
    v = max(v, dpp_i32<0x122>(v));
    v = max(v, dpp_i32<0x124>(v));
    v = max(v, dpp_i32<0x128>(v));
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
              d = __builtin_amdgcn_readlane(v, 48);
    return __int_as_float(max(max(a, b), max(c, d)));
}
static __device__ __forceinline__ float wave_sum(float x)
{
    x += __int_as_float(dpp_i32<0x121>(__float_as_int(x)));
    x += __int_as_float(dpp_i32<0x122>(__float_as_int(x)));
    x += __int_as_float(dpp_i32<0x124>(__float_as_int(x)));
    x += __int_as_float(dpp_i32<0x128>(__float_as_int(x)));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16)),
                c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return (a + b) + (c + d);
}
static __device__ __forceinline__ float absmax4(const f32x4& t)
{
    return fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fabsf(t[2]), fabsf(t[3])));
}
// m = max(m, |t0|, .., |t3|) in two instructions (as fmaxf, a NaN in t is ignored; written out because the compiler quiets every
// MFMA result with a max against itself before an fmaxf: five instructions per tile in a loop whose MFMAs leave few issue slots)
static __device__ __forceinline__ void absmax4_into(float& m, const f32x4& t)
{
    asm("v_max3_f32 %0, |%1|, |%2|, %0\n\tv_max3_f32 %0, |%3|, |%4|, %0" : "+v"(m) : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
}
// The fp16 pair format saturates at 65504 (after the format's scale): what a producer of pair16 images checks its inputs against
static constexpr float PAIR_SAT = 65504.0f;

static __device__ __forceinline__ float fast_sigmoid(float x)
{   // v_exp_f32 + v_rcp_f32 (1 ulp each): relative error < 1e-6 for |x| < 16, far inside the 1e-4 budget
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

