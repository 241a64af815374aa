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
        if (a.diag && blockIdx.x == 0 && tid == 0) a.diag[(size_t)t * 8 + (slot)] = t__;      \
    } while (0)
#else
#define DIAG_STAMP(slot) do {} while (0)
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
// bf16 A/B-operand image of that tile as half (g & 1) of lane (2(c&1) + (g>>1), row16) of k step c >> 1.  Returns the index in
// 8-byte units inside a [2 ksteps][64 lanes][2 halves] image.
static __device__ __forceinline__ int split_slot(int c, int g, int row16)
{
    return ((((c >> 1) * 64) + (2 * (c & 1) + (g >> 1)) * 16 + row16) << 1) + (g & 1);
}
static constexpr int SPLIT_IMG = 2 * 64 * 2;                    // uint2 per term image of one 16-row tile (K = 64)

static __device__ __forceinline__ float fast_sigmoid(float x)
{   // v_exp_f32 + v_rcp_f32 (1 ulp each): relative error < 1e-6 for |x| < 16, far inside the 1e-4 budget
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

