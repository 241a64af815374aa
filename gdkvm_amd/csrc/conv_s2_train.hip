// conv_s2_train.hip -- training: a residual block's 3x3 / stride-2 / pad-1 convolution together with its 1x1 / stride-2 downsample
// branch (encoder layer2.0 / layer3.0), forward and backward on hand-written kernels.  Until round 5 these four layers were the last
// library convolutions of a training step (MIOpen / CK: weight and data gradients accumulated with atomics -- the one non-deterministic
// part of the step, and the reason a DDP-wrapped step at world size 1 was not bit-equal to the bare one).
//
//   forward      gdkvm_conv_down_bias_act (conv_igemm.hip) on the packs made here: both outputs from one launch;
//   data grad    dx[n, h, w, c] = sum_{k, r, s} dy[n, (h+1-r)/2, (w+1-s)/2, k] w[k, c, r, s]  over the taps whose parity fits, + the
//                downsample branch's dyd[n, h/2, w/2, k] wd[k, c] on even (h, w): conv_igemm_kernel's DG form, one launch, four
//                parity classes of input pixels, each a stride-1 convolution of dy with 1 / 2 / 2 / 4 taps (no products with zeros);
//   weight grad  gdkvm_conv_wgrad_strided (gemm.hip): dy^T im2col(x) without the im2col, fixed-order split sums.
// Everything is deterministic: same inputs, same bits.
#include "gdkvm_common.hpp"

int gdkvm_conv_igemm_dgrad_launch(const void* dy, const void* dy2, const void* packs, int with_down, void* dx, int N, int Cf, int H, int W, int Kf, hipStream_t st);

namespace {

struct S2PackArgs {
    const float* w; long long sk, sc, sr, ss;            // fp32 [K, C, 3, 3], element strides
    const float* wd; long long dk, dc;                   // fp32 [K, C] (1x1), element strides; may be NULL
    uint4* fwd; uint4* fwd_down; uint4* dgrad;
    int K, C;
};

__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16); }

// One thread = one 16-byte piece (8 consecutive reduction indices of one fragment row) of one of the packs:
//   forward  [K/16][9C/32][lane][8]:  w[16 nt + li][kd = 32 ks + 8 g ..],  kd = (3 r + s) C + c          (gdkvm_conv_igemm_pack_weights' order)
//   down     [K/16][ C/32][lane][8]:  wd[16 nt + li][c = 32 ks + 8 g ..]
//   dgrad, class (ph, pw): [C/16][taps K/32][lane][8]:  row = input channel 16 nt + li,  kd = tap' K + k,  tap' = (1 + pw) r' + s',
//            r' = 0 <-> forward tap r = (ph ? 2 : 1),  r' = 1 <-> r = 0  (same for s);  class (0, 0) ends with one more "tap": wd[k][c]
__global__ __launch_bounds__(256) void conv_s2_pack_kernel(S2PackArgs a)
{
    const int K = a.K, C = a.C;
    const size_t n_fwd = (size_t)K * 9 * C / 8, n_down = a.wd ? (size_t)K * C / 8 : 0, n_dg = (size_t)(9 + (a.wd ? 1 : 0)) * K * C / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_fwd + n_down + n_dg; i += (size_t)gridDim.x * 256) {
        float v[8];
        uint4* dst;
        if (i < n_fwd + n_down) {
            const bool down = i >= n_fwd;
            const size_t j = down ? i - n_fwd : i;
            const int Kd = down ? C : 9 * C, nks = Kd / 32;
            const int lane = (int)(j & 63), li = lane & 15, g = lane >> 4;
            const size_t f = j >> 6;
            const int ks = (int)(f % nks), nt = (int)(f / nks);
            const int k = 16 * nt + li, kd = 32 * ks + 8 * g, tap = kd / C, c = kd - tap * C;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = down ? a.wd[k * a.dk + (c + e) * a.dc] : a.w[k * a.sk + (c + e) * a.sc + (tap / 3) * a.sr + (tap % 3) * a.ss];
            dst = (down ? a.fwd_down : a.fwd) + j;
        } else {
            size_t j = i - n_fwd - n_down;
            int cls = 0;
            for (; cls < 3; ++cls) {
                const size_t n_c = gdkvm_conv_s2_dgrad_pack_elems(C, K, cls, a.wd != nullptr) / 8;   // (host-device inline below)
                if (j < n_c) break;
                j -= n_c;
            }
            size_t base = 0;
            for (int q = 0; q < cls; ++q) base += gdkvm_conv_s2_dgrad_pack_elems(C, K, q, a.wd != nullptr) / 8;
            const int ph = cls >> 1, pw = cls & 1, taps_main = (1 + ph) * (1 + pw), taps = taps_main + (cls == 0 && a.wd ? 1 : 0);
            const int nks = taps * K / 32;
            const int lane = (int)(j & 63), li = lane & 15, g = lane >> 4;
            const size_t f = j >> 6;
            const int ks = (int)(f % nks), nt = (int)(f / nks);
            const int c = 16 * nt + li, kd = 32 * ks + 8 * g, tp = kd / K, k = kd - tp * K;
            if (tp >= taps_main) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = a.wd[(k + e) * a.dk + c * a.dc];
            } else {
                const int rp = tp / (1 + pw), sp = tp - rp * (1 + pw);
                const int r = ph ? (rp ? 0 : 2) : 1, s = pw ? (sp ? 0 : 2) : 1;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = a.w[(k + e) * a.sk + c * a.sc + r * a.sr + s * a.ss];
            }
            dst = a.dgrad + base + j;
        }
        *dst = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
}

}  // namespace

extern "C" size_t gdkvm_conv_s2_dgrad_pack_bytes(int C, int K, int with_down)
{
    if (C <= 0 || K <= 0) return 16;
    return (size_t)(9 + (with_down ? 1 : 0)) * K * C * sizeof(bf16_t);
}

extern "C" int gdkvm_conv_s2_pack_train(const float* w, const long long* w_strides, const float* w_down, const long long* w_down_strides,
                                        void* packed_fwd, void* packed_fwd_down, void* packed_dgrad, int K, int C, void* stream)
{
    if (K <= 0 || C <= 0 || K % 128 || C % 64) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_s2_pack_train: K=%d C=%d (K a multiple of 128, C of 64)", K, C);
    if (!w || !w_strides || !packed_fwd || !packed_dgrad || (w_down && (!w_down_strides || !packed_fwd_down)))
        return gdkvm_fail(GDKVM_ERR_ARG, "conv_s2_pack_train: null pointer");
    if (!gdkvm_aligned16(packed_fwd) || !gdkvm_aligned16(packed_dgrad) || (w_down && !gdkvm_aligned16(packed_fwd_down)))
        return gdkvm_fail(GDKVM_ERR_ARG, "conv_s2_pack_train: packs must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    S2PackArgs a{w, w_strides[0], w_strides[1], w_strides[2], w_strides[3], w_down, w_down ? w_down_strides[0] : 0, w_down ? w_down_strides[1] : 0,
                 static_cast<uint4*>(packed_fwd), static_cast<uint4*>(packed_fwd_down), static_cast<uint4*>(packed_dgrad), K, C};
    const size_t total = ((size_t)K * 9 * C + (w_down ? (size_t)K * C : 0) + (size_t)(9 + (w_down ? 1 : 0)) * K * C) / 8;
    hipLaunchKernelGGL(conv_s2_pack_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    GDKVM_LAUNCH_CHECK("conv_s2_pack_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_conv_s2_dgrad(const void* dy, const void* dy_down, const void* packed_dgrad, void* dx,
                                   int N, int C, int H, int W, int K, int with_down, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv_s2_dgrad: only bf16 is implemented");
    if (dy_down && !with_down) return gdkvm_fail(GDKVM_ERR_ARG, "conv_s2_dgrad: dy_down given for a pack made without the 1x1 branch (with_down = 0)");
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || C % 64 || K % 64)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_s2_dgrad: N=%d C=%d H=%d W=%d K=%d (C, K multiples of 64)", N, C, H, W, K);
    if (N == 0) return GDKVM_OK;
    if (!dy || !packed_dgrad || !dx) return gdkvm_fail(GDKVM_ERR_ARG, "conv_s2_dgrad: null pointer");
    if (!gdkvm_aligned16(dy) || !gdkvm_aligned16(packed_dgrad) || !gdkvm_aligned16(dx) || (dy_down && !gdkvm_aligned16(dy_down)))
        return gdkvm_fail(GDKVM_ERR_ARG, "conv_s2_dgrad: pointers must be 16-byte aligned");
    if (int rc = gdkvm_check_device()) return rc;
    if (gdkvm_conv_igemm_dgrad_launch(dy, dy_down, packed_dgrad, with_down != 0, dx, N, C, H, W, K, static_cast<hipStream_t>(stream)))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_s2_dgrad: tensor too large for 32-bit offsets");
    GDKVM_LAUNCH_CHECK("conv_igemm_kernel<dgrad>");
    return GDKVM_OK;
}
