// tile 2: 128 x 128 x 64, v3 kernel, pipeline v3 intrawave (256-channel 7 x 7 layers and the 384 -> 128 decoder convolution;
// MIOpen's pick is the same tile on the interwave v1 pipeline, 6 us slower here: tools/ck_sweep)
#include "conv_ck_common.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle_v3.hpp"
namespace gdkvm_ck {
template <class DsLayout, class DsTypes, class Op>
using Kernel = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle_V3<2, L::NHWGC, L::GKYXC, DsLayout, L::NHWGK, BF16, BF16,
    F32, F32, DsTypes, BF16, PassThrough, PassThrough, Op, ConvDefault, GemmMNKPadding, 256, 128, 128, 64, 8, 8, 32, 32, 2, 2, S<8, 32, 1>, S<1, 0, 2>, S<1, 0, 2>, 2, 8, 8, 0, S<8, 32, 1>, S<1, 0, 2>,
    S<1, 0, 2>, 2, 8, 8, 0, 1, 1, S<1, 32, 1, 8>, 8, ck::BlockGemmPipelineScheduler::Intrawave, ck::BlockGemmPipelineVersion::v3>;

int conv_t2(const void* x, const void* w, const float* bias, const void* residual, void* y, const ConvShape& s, int relu, hipStream_t st)
{
    return conv_entry<Kernel>(x, w, bias, residual, y, s, relu, st);
}
}  // namespace gdkvm_ck
