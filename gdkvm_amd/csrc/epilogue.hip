// epilogue.hip -- fused pointwise epilogue for the (unchanged, MIOpen) encoder/decoder convolutions in the
// inference build: y = act(x + bias[c] (+ residual)), NHWC, in place or out of place.
// After BatchNorm folding every conv carries a bias; PyTorch then runs the bias add, the residual add and the ReLU as
// three separate full-tensor passes (1.1 ms of a 3.6 ms cfg2 forward).  This is one pass: 16-byte loads/stores,
// HBM-bound by construction.
#include "gdkvm_common.hpp"

namespace {

template <int IO>
__global__ __launch_bounds__(256) void bias_act_kernel(const void* x, const float* bias, const void* res, void* y,
                                                       size_t nvec, int C, int relu)
{
    constexpr int V = IO == GDKVM_F32 ? 4 : 8;            // elements per 16-byte access
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const size_t e0 = i * V;
        const int c0 = (int)(e0 % (size_t)C);
        float v[V];
        if constexpr (IO == GDKVM_F32) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(static_cast<const float*>(x) + e0);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a[j];
            if (res) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(static_cast<const float*>(res) + e0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += r[j];
            }
        } else {
            const uint4 a = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(x) + e0);
            const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
            if (res) {
                const uint4 r = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(res) + e0);
                const unsigned rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[2 * j] += __uint_as_float(rw[j] << 16); v[2 * j + 1] += __uint_as_float(rw[j] & 0xffff0000u); }
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            v[j] += bias[c0 + j];
            if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        if constexpr (IO == GDKVM_F32) {
            *reinterpret_cast<f32x4*>(static_cast<float*>(y) + e0) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            uint4 o;
            o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
            *reinterpret_cast<uint4*>(static_cast<bf16_t*>(y) + e0) = o;
        }
    }
}

}  // namespace

extern "C" int gdkvm_bias_act(const void* x, const float* bias, const void* residual, void* y,
                              size_t rows, int C, int relu, int io_dtype, void* stream)
{
    if (C <= 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "bias_act: C=%d", C);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "bias_act: io_dtype=%d", io_dtype);
    const int V = io_dtype == GDKVM_F32 ? 4 : 8;
    if (C % V) return gdkvm_fail(GDKVM_ERR_SHAPE, "bias_act: C=%d must be a multiple of %d", C, V);
    if (rows == 0) return GDKVM_OK;
    if (!x || !bias || !y) return gdkvm_fail(GDKVM_ERR_ARG, "bias_act: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(y) || !gdkvm_aligned16(bias) || (residual && !gdkvm_aligned16(residual)))
        return gdkvm_fail(GDKVM_ERR_ARG, "bias_act: pointers must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    const size_t nvec = rows * (size_t)C / V;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;               // ~8 blocks per CU, grid-stride the rest
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((bias_act_kernel<GDKVM_F32>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias, residual, y, nvec, C, relu);
    else hipLaunchKernelGGL((bias_act_kernel<GDKVM_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias, residual, y, nvec, C, relu);
    GDKVM_LAUNCH_CHECK("bias_act_kernel");
    return GDKVM_OK;
}
