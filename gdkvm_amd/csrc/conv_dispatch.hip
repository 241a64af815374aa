// conv_dispatch.hip -- C entry of the convolutions that carry their epilogue.  The three kernels behind it are
// hand-written for gfx950: conv3x3_c64.hip (3x3 / 1 / 1, 64 -> 64 channels: weights resident in registers), conv3x3_tile.hip
// (3x3 / 1 / 1, input channels in multiples of 64 walked in LDS chunks, weights streamed, rows of <= 64 pixels) and conv_igemm.hip
// (kernel 9: any window / stride / padding / row width as an implicit GEMM; C a multiple of 32, K of 128).  Shapes none of them
// covers (odd channel counts) are NOT computed here: the caller keeps them on the framework convolution followed by the
// gdkvm_bias_act epilogue pass.
#include "gdkvm_common.hpp"

int gdkvm_conv3x3_c64_launch(const void* x, const void* w, const float* bias, const void* residual, void* y, int N, int H, int W,
                             int relu, int packed, hipStream_t st);   // conv3x3_c64.hip
int gdkvm_conv3x3_tile_launch(const void* x, const void* x2, int C1, const void* w, const float* bias, const void* residual, void* y,
                              int N, int C, int H, int W, int K, int relu, int variant, int packed, hipStream_t st);   // conv3x3_tile.hip
int gdkvm_conv_igemm_launch(const void* x, const void* wpacked, const float* bias, const void* residual, void* y,
                            int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu,
                            const void* w2packed, const float* bias2, void* y2, hipStream_t st);   // conv_igemm.hip

extern "C" int gdkvm_conv_bias_act(const void* x, const void* w, const float* bias, const void* residual, void* y,
                                   int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu, int kernel,
                                   int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv_bias_act: only bf16 is implemented");
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0 || C % 8 || K % 8
        || H + 2 * pad < R || W + 2 * pad < S)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: N=%d C=%d H=%d W=%d K=%d %dx%d stride %d pad %d (C, K multiples of 8)",
                          N, C, H, W, K, R, S, stride, pad);
    const bool packed = kernel & GDKVM_CONV_PACKED_WEIGHTS;
    kernel &= ~GDKVM_CONV_PACKED_WEIGHTS;
    if (kernel != 0 && (kernel < 4 || kernel > 11)) return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: kernel=%d (0 = by shape, 4, 5, 6..8, 9, 10, 11)", kernel);
    if (kernel == 9) {
        // the general implicit-GEMM kernel: any R x S / stride / pad, weights as gdkvm_conv_igemm_pack_weights wrote them
        if (!packed) return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: kernel 9 reads the gdkvm_conv_igemm_pack_weights copy of the weights (flag %d)", GDKVM_CONV_PACKED_WEIGHTS);
        if (C % 32 || K % 128) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: kernel 9 needs C a multiple of 32 and K of 128 (C=%d K=%d)", C, K);
        if (N == 0) return GDKVM_OK;
        if (!x || !w || !bias || !y) return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: null pointer");
        if (!gdkvm_aligned16(x) || !gdkvm_aligned16(w) || !gdkvm_aligned16(y) || !gdkvm_aligned16(bias) || (residual && !gdkvm_aligned16(residual)))
            return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: pointers must be 16-byte aligned");
        if (int rc = gdkvm_check_device()) return rc;
        if (gdkvm_conv_igemm_launch(x, w, bias, residual, y, N, C, H, W, K, R, S, stride, pad, relu, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream)))
            return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: N=%d C=%d H=%d W=%d K=%d %dx%d stride %d pad %d is too large for 32-bit offsets", N, C, H, W, K, R, S, stride, pad);
        GDKVM_LAUNCH_CHECK("conv_igemm_kernel");
        return GDKVM_OK;
    }
    if (packed && (C % 64 || K % 16))
        return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: packed weights need C a multiple of 64 and K of 16 (C=%d K=%d)", C, K);
    if (!(R == 3 && S == 3 && stride == 1 && pad == 1))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: %dx%d stride %d pad %d is not served by the 3x3 / 1 / 1 kernels: kernel 9 (the "
                                           "implicit-GEMM kernel, packed weights) or the framework convolution + gdkvm_bias_act", R, S, stride, pad);
    if (N == 0) return GDKVM_OK;
    if (!x || !w || !bias || !y) return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: null pointer");
    if (!gdkvm_aligned16(x) || !gdkvm_aligned16(w) || !gdkvm_aligned16(y) || !gdkvm_aligned16(bias) || (residual && !gdkvm_aligned16(residual)))
        return gdkvm_fail(GDKVM_ERR_ARG, "conv_bias_act: pointers must be 16-byte aligned");
    if ((size_t)N * H * W * C >= (1ull << 31) || (size_t)N * H * W * K >= (1ull << 31))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: tensor too large for 32-bit offsets");
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool c64 = C == 64 && K == 64;
    if (kernel == 4 && !c64) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: kernel 4 serves 64 -> 64 channels only (C=%d K=%d)", C, K);
    if (kernel == 4 || (kernel == 0 && c64)) {
        if (gdkvm_conv3x3_c64_launch(x, w, bias, residual, y, N, H, W, relu, packed ? 1 : 0, st)) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: too many tiles");
        GDKVM_LAUNCH_CHECK("conv3x3_c64_kernel");
        return GDKVM_OK;
    }
    if (gdkvm_conv3x3_tile_launch(x, nullptr, 0, w, bias, residual, y, N, C, H, W, K, relu, kernel >= 10 ? kernel - 6 : (kernel >= 6 ? kernel - 5 : 0), packed ? 1 : 0, st))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_bias_act: C=%d K=%d %dx%d is not served by the hand-written kernels (C a multiple of 64, K of "
                                           "16, rows of at most 64 pixels): use the framework convolution + gdkvm_bias_act", C, K, H, W);
    GDKVM_LAUNCH_CHECK("conv3x3_tile_kernel");
    return GDKVM_OK;
}

// The same convolution over the channel concatenation [x1 (C1 channels) ; x2 (C2)] of two NHWC tensors, which is never
// materialised (the decoder's "upsampled feature ; skip feature" input): the chunked kernel fetches every 64-channel chunk from
// the tensor it lies in.  C1 and C2 multiples of 64; w [K, 3, 3, C1 + C2] (or its packed copy); kernel 0 / 5..8 as above.
extern "C" int gdkvm_conv_cat_bias_act(const void* x1, const void* x2, const void* w, const float* bias, const void* residual, void* y,
                                       int N, int C1, int C2, int H, int W, int K, int relu, int kernel, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv_cat_bias_act: only bf16 is implemented");
    const bool packed = kernel & GDKVM_CONV_PACKED_WEIGHTS;
    kernel &= ~GDKVM_CONV_PACKED_WEIGHTS;
    if (kernel != 0 && (kernel < 5 || kernel > 11 || kernel == 9)) return gdkvm_fail(GDKVM_ERR_ARG, "conv_cat_bias_act: kernel=%d (0, 5..8, 10, 11)", kernel);
    if (N < 0 || C1 <= 0 || C2 <= 0 || C1 % 64 || C2 % 64 || H <= 0 || W <= 0 || K <= 0 || K % 16)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_cat_bias_act: N=%d C1=%d C2=%d H=%d W=%d K=%d (C1, C2 multiples of 64, K of 16)", N, C1, C2, H, W, K);
    if (N == 0) return GDKVM_OK;
    if (!x1 || !x2 || !w || !bias || !y) return gdkvm_fail(GDKVM_ERR_ARG, "conv_cat_bias_act: null pointer");
    if (!gdkvm_aligned16(x1) || !gdkvm_aligned16(x2) || !gdkvm_aligned16(w) || !gdkvm_aligned16(y) || !gdkvm_aligned16(bias) || (residual && !gdkvm_aligned16(residual)))
        return gdkvm_fail(GDKVM_ERR_ARG, "conv_cat_bias_act: pointers must be 16-byte aligned");
    if ((size_t)N * H * W * (C1 + C2) >= (1ull << 31) || (size_t)N * H * W * K >= (1ull << 31))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_cat_bias_act: tensor too large for 32-bit offsets");
    if (int rc = gdkvm_check_device()) return rc;
    if (gdkvm_conv3x3_tile_launch(x1, x2, C1, w, bias, residual, y, N, C1 + C2, H, W, K, relu, kernel >= 10 ? kernel - 6 : (kernel >= 6 ? kernel - 5 : 0), packed ? 1 : 0,
                                  static_cast<hipStream_t>(stream)))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_cat_bias_act: %dx%d is not served (rows of at most 64 pixels)", H, W);
    GDKVM_LAUNCH_CHECK("conv3x3_tile_kernel");
    return GDKVM_OK;
}

// A residual block's first convolution AND its downsample branch in one launch: y = act(conv_RxS(x, w) + bias) as kernel 9 above, and
// y_down = conv_1x1(x, w_down) (+ bias_down, may be NULL) with the same stride and K output channels -- the 1x1's input pixel is the
// centre of the R x S window (R = S odd, pad = R / 2), so the second result is one more pass over pixels the kernel has already
// set up.  Both weight tensors as gdkvm_conv_igemm_pack_weights copies ([K, R, S, C] and [K, 1, 1, C]).
extern "C" int gdkvm_conv_down_bias_act(const void* x, const void* w, const float* bias, void* y, int relu,
                                        const void* w_down, const float* bias_down, void* y_down,
                                        int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv_down_bias_act: only bf16 is implemented");
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || R <= 0 || S != R || !(R & 1) || stride <= 0 || pad != R / 2 || C % 32 || K % 128)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_down_bias_act: N=%d C=%d H=%d W=%d K=%d %dx%d stride %d pad %d (odd square window, pad = R/2, "
                                           "C a multiple of 32, K of 128)", N, C, H, W, K, R, S, stride, pad);
    if (N == 0) return GDKVM_OK;
    if (!x || !w || !bias || !y || !w_down || !y_down) return gdkvm_fail(GDKVM_ERR_ARG, "conv_down_bias_act: null pointer");
    const void* ptrs[] = {x, w, bias, y, w_down, y_down};
    for (const void* p : ptrs) if (!gdkvm_aligned16(p)) return gdkvm_fail(GDKVM_ERR_ARG, "conv_down_bias_act: pointers must be 16-byte aligned");
    if (bias_down && !gdkvm_aligned16(bias_down)) return gdkvm_fail(GDKVM_ERR_ARG, "conv_down_bias_act: pointers must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    if (gdkvm_conv_igemm_launch(x, w, bias, nullptr, y, N, C, H, W, K, R, S, stride, pad, relu, w_down, bias_down, y_down, static_cast<hipStream_t>(stream)))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_down_bias_act: tensor too large for 32-bit offsets");
    GDKVM_LAUNCH_CHECK("conv_igemm_kernel");
    return GDKVM_OK;
}
