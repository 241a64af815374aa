// gdkvm_common.hpp -- shared device/host helpers for the gfx950 GDKVM kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gdkvm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define GDKVM_DK 64             // per-head key dim the kernels are specialised for (SURVEY.md §8 defaults)
#define GDKVM_MAX_N 4096         // tokens per frame (frames of more than 64 are folded per 64-token chunk and composed)
#define GDKVM_EPS_NORM 1e-12f

// ---- host side -------------------------------------------------------------------------------------
int gdkvm_fail(int code, const char* fmt, ...);      // records the thread-local message, returns `code`
int gdkvm_check_device(void);                         // GDKVM_OK iff the current device is gfx950
// Zero `bytes` (a multiple of 4) at p (4-byte aligned) on `st` with a KERNEL.  Not hipMemsetAsync: as a memset node of a captured graph a
// small fill of memory inside the graph's own pool ran on the first replay only (round 6, argmax_dice.hip); a kernel node has no such mode.
int gdkvm_zero_async(void* p, size_t bytes, hipStream_t st);
static inline bool gdkvm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define GDKVM_LAUNCH_CHECK(name)                                                        \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// elements of the data-gradient weight pack of parity class cls = 2 ph + pw of a 3x3 / stride-2 layer (conv_s2_train.hip): (1 + ph)(1 + pw)
// taps of Kf x Cf weights, class 0 one more when the block's 1x1 downsample branch rides along
__host__ __device__ static inline size_t gdkvm_conv_s2_dgrad_pack_elems(int Cf, int Kf, int cls, int with_down)
{
    const int taps = (1 + (cls >> 1)) * (1 + (cls & 1)) + (cls == 0 && with_down ? 1 : 0);
    return (size_t)taps * Kf * Cf;
}

// ---- device side -----------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));   // v_cvt_pk_bf16_f32: RNE, NaN-safe
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// 4 consecutive channels starting at element offset `off` (a multiple of 4) of a row, widened to fp32
template <int IO>
__device__ __forceinline__ f32x4 load4(const void* base, size_t off)
{
    if constexpr (IO == GDKVM_F32) {
        return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + off);
    } else {
        const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(base) + off);
        f32x4 r;
        r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
        r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
        return r;
    }
}
template <int IO>
__device__ __forceinline__ float load1(const void* base, size_t off)
{
    if constexpr (IO == GDKVM_F32) return static_cast<const float*>(base)[off];
    else return bf16_to_f32(static_cast<const bf16_t*>(base)[off]);
}
template <int IO>
__device__ __forceinline__ void store1(void* base, size_t off, float x)
{
    if constexpr (IO == GDKVM_F32) static_cast<float*>(base)[off] = x;
    else static_cast<bf16_t*>(base)[off] = f32_to_bf16(x);
}

// v_mfma_f32_16x16x4_f32: exact fp32 (a k-ordered fmaf chain).  Lane l = 16*g + i:
//   A operand = A[row i][k g],  B operand = B[k g][col i],  C/D reg r = D[row 4g + r][col i].
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
