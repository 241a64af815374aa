// conv_ck_common.hpp -- the inference build's 3x3 convolutions with the folded-BatchNorm bias, the residual add and the ReLU in
// the GEMM epilogue.  MIOpen offers no usable fused convolution at these shapes (its fusion plan
// resolves to the naive kernel), so every convolution was followed by a separate in-place pass over its output
// (gdkvm_bias_act: 16 launches, 0.21 ms = 12 % of a cfg2 forward, all of it HBM traffic).  The implicit-GEMM kernels themselves
// are composable_kernel's gfx950 templates -- the same tile configurations MIOpen's solver search picks for these layers
// (read from the kernel names in profiles/r01_n_bench_cfg2_steady_state.csv) -- instantiated here with a CDE functor that
// applies bias (+ residual) (+ ReLU) to the fp32 accumulator before the single rounding to bf16.
// One translation unit per tile configuration (conv_ck_t*.hip) so that they compile in parallel.
#pragma once
#include <array>

#include "ck/ck.hpp"
#include "ck/stream_config.hpp"
#include "ck/tensor_operation/gpu/device/convolution_forward_specialization.hpp"
#include "ck/tensor_operation/gpu/device/gemm_specialization.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"

namespace gdkvm_ck {

using BF16 = ck::bhalf_t;
using F32 = float;
template <ck::index_t... Is>
using S = ck::Sequence<Is...>;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
namespace L = ck::tensor_layout::convolution;
constexpr auto ConvDefault = ck::tensor_operation::device::ConvolutionForwardSpecialization::Default;
constexpr auto GemmMNKPadding = ck::tensor_operation::device::GemmSpecialization::MNKPadding;

struct BiasAct {                 // e = act(c + bias[k])
    int relu;
    template <typename E, typename C, typename D0>
    __host__ __device__ constexpr void operator()(E& e, const C& c, const D0& d0) const
    {
        const float x = ck::type_convert<float>(c) + ck::type_convert<float>(d0);
        e = ck::type_convert<E>(relu && x < 0.f ? 0.f : x);
    }
};

struct BiasResAct {              // e = act(c + bias[k] + residual)
    int relu;
    template <typename E, typename C, typename D0, typename D1>
    __host__ __device__ constexpr void operator()(E& e, const C& c, const D0& d0, const D1& d1) const
    {
        const float x = ck::type_convert<float>(c) + ck::type_convert<float>(d0) + ck::type_convert<float>(d1);
        e = ck::type_convert<E>(relu && x < 0.f ? 0.f : x);
    }
};

struct ConvShape { int N, C, H, W, K, R, S, stride, pad; };

// NHWC activations (G = 1), KYXC weights, NHWK output; bias broadcast over N, H, W through zero strides.
template <class Conv, class Op, int ND>
int run_conv(const void* x, const void* w, const std::array<const void*, ND>& ds, void* y, const ConvShape& s, const Op& op, hipStream_t st)
{
    const int Ho = (s.H + 2 * s.pad - s.R) / s.stride + 1, Wo = (s.W + 2 * s.pad - s.S) / s.stride + 1;
    using I5 = std::array<ck::index_t, 5>;
    const I5 a_len{1, s.N, s.C, s.H, s.W}, a_str{s.C, s.H * s.W * s.C, 1, s.W * s.C, s.C};
    const I5 b_len{1, s.K, s.C, s.R, s.S}, b_str{s.K * s.R * s.S * s.C, s.R * s.S * s.C, 1, s.S * s.C, s.C};
    const I5 e_len{1, s.N, s.K, Ho, Wo}, e_str{s.K, Ho * Wo * s.K, 1, Wo * s.K, s.K};
    std::array<I5, ND> d_len, d_str;
    d_len[0] = e_len; d_str[0] = I5{s.K, 0, 1, 0, 0};
    if constexpr (ND == 2) { d_len[1] = e_len; d_str[1] = e_str; }
    const std::array<ck::index_t, 2> cs{s.stride, s.stride}, cd{1, 1}, lp{s.pad, s.pad}, rp{s.pad, s.pad};
    Conv conv;
    auto arg = conv.MakeArgument(x, w, ds, y, a_len, a_str, b_len, b_str, d_len, d_str, e_len, e_str, cs, cd, lp, rp,
                                 PassThrough{}, PassThrough{}, op);
    if (!conv.IsSupportedArgument(arg)) return 1;
    conv.MakeInvoker().Run(arg, StreamConfig{st, false});
    return 0;
}

// every tile configuration exports one function of this type (residual may be NULL)
using ConvFn = int (*)(const void* x, const void* w, const float* bias, const void* residual, void* y, const ConvShape& s, int relu,
                       hipStream_t st);

template <template <class, class, class> class K>
int conv_entry(const void* x, const void* w, const float* bias, const void* residual, void* y, const ConvShape& s, int relu, hipStream_t st)
{
    if (residual) {
        using Conv = K<ck::Tuple<L::G_K, L::NHWGK>, ck::Tuple<F32, BF16>, BiasResAct>;
        return run_conv<Conv, BiasResAct, 2>(x, w, {bias, residual}, y, s, BiasResAct{relu}, st);
    }
    using Conv = K<ck::Tuple<L::G_K>, ck::Tuple<F32>, BiasAct>;
    return run_conv<Conv, BiasAct, 1>(x, w, {bias}, y, s, BiasAct{relu}, st);
}

int conv_t0(const void*, const void*, const float*, const void*, void*, const ConvShape&, int, hipStream_t);
int conv_t1(const void*, const void*, const float*, const void*, void*, const ConvShape&, int, hipStream_t);
int conv_t2(const void*, const void*, const float*, const void*, void*, const ConvShape&, int, hipStream_t);
int conv_t3(const void*, const void*, const float*, const void*, void*, const ConvShape&, int, hipStream_t);

}  // namespace gdkvm_ck
