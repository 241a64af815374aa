// conv_ck.hip -- C entry of the fused-epilogue convolutions (see conv_ck_common.hpp; the kernels live in conv_ck_t*.hip).
#include "gdkvm_common.hpp"

namespace gdkvm_ck {
struct ConvShape { int N, C, H, W, K, R, S, stride, pad; };
int conv_t0(const void*, const void*, const float*, const void*, void*, const ConvShape&, int, hipStream_t);
int conv_t1(const void*, const void*, const float*, const void*, void*, const ConvShape&, int, hipStream_t);
int conv_t2(const void*, const void*, const float*, const void*, void*, const ConvShape&, int, hipStream_t);
int conv_t3(const void*, const void*, const float*, const void*, void*, const ConvShape&, int, hipStream_t);
}  // namespace gdkvm_ck

int gdkvm_conv3x3_c64_launch(const void* x, const void* w, const float* bias, const void* residual, void* y, int N, int H, int W,
                             int relu, hipStream_t st);          // conv3x3_c64.hip (hand-written, LDS halo band)

extern "C" int gdkvm_conv_bias_act(const void* x, const void* w, const float* bias, const void* residual, void* y,
                                   int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu, int tile,
                                   int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv_bias_act: only bf16 is implemented");
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0 || C % 8 || K % 8
        || H + 2 * pad < R || W + 2 * pad < S)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: N=%d C=%d H=%d W=%d K=%d %dx%d stride %d pad %d (C, K multiples of 8)",
                          N, C, H, W, K, R, S, stride, pad);
    if (tile < 0 || tile > 4) return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: tile=%d (0..4)", tile);
    if (tile == 4 && !(C == 64 && K == 64 && R == 3 && S == 3 && stride == 1 && pad == 1)) tile = 3;      // the hand-written kernel's one shape
    if (N == 0) return GDKVM_OK;
    if (!x || !w || !bias || !y) return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(w) || !gdkvm_aligned16(y) || !gdkvm_aligned16(bias) || (residual && !gdkvm_aligned16(residual)))
        return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: pointers must be 16-byte aligned");
    if ((size_t)N * H * W * C >= (1ull << 31) || (size_t)N * H * W * K >= (1ull << 31))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: tensor too large for 32-bit offsets");
    if (int rc = gdkvm_check_device()) return rc;
    const gdkvm_ck::ConvShape s{N, C, H, W, K, R, S, stride, pad};
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto run = [&](int t) {
        switch (t) {
            case 0: return gdkvm_ck::conv_t0(x, w, bias, residual, y, s, relu, st);
            case 1: return gdkvm_ck::conv_t1(x, w, bias, residual, y, s, relu, st);
            case 2: return gdkvm_ck::conv_t2(x, w, bias, residual, y, s, relu, st);
            default: return gdkvm_ck::conv_t3(x, w, bias, residual, y, s, relu, st);
        }
    };
    if (tile == 4) {
        if (gdkvm_conv3x3_c64_launch(x, w, bias, residual, y, N, H, W, relu, st)) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: too many tiles");
        GDKVM_LAUNCH_CHECK("conv3x3_c64_kernel");
        return GDKVM_OK;
    }
    int rc = run(tile);
    if (rc && tile != 0) rc = run(0);                     // a configuration that cannot address the problem (very few channels): tile 0
    if (rc) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: no tile configuration supports this problem");
    GDKVM_LAUNCH_CHECK("conv_bias_act");
    return GDKVM_OK;
}
