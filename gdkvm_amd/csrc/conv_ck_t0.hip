// tile 0: 128 x 64 x 32, 256 threads (MIOpen's pick for the 64-channel 28 x 28 layers)
#include "conv_ck_common.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle.hpp"
namespace gdkvm_ck {
template <class DsLayout, class DsTypes, class Op>
using Kernel = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<2, L::NHWGC, L::GKYXC, DsLayout, L::NHWGK, BF16, BF16,
    F32, F32, DsTypes, BF16, PassThrough, PassThrough, Op, ConvDefault, GemmMNKPadding, 1, 256, 128, 64, 32, 8, 8, 32, 32, 2, 1, S<4, 64, 1>, S<1, 0, 2>, S<1, 0, 2>, 2, 8, 8, 1, S<4, 64, 1>, S<1, 0, 2>,
    S<1, 0, 2>, 2, 8, 8, 1, 1, 1, S<1, 32, 1, 8>, 8>;

int conv_t0(const void* x, const void* w, const float* bias, const void* residual, void* y, const ConvShape& s, int relu, hipStream_t st)
{
    return conv_entry<Kernel>(x, w, bias, residual, y, s, relu, st);
}
}  // namespace gdkvm_ck
