// gdr_normalizer.hip -- the `normalizer` flag of the LKVA read: gdkvm_scan_fwd_normalizer.
//
// SPEC: besides S the memory carries z in R^Dk per (clip, head), "the same recurrence on v == 1", and the read-out of token n is
// R_t[n, :] / (|q_n . z_{t-1}| + eps).  z IS a column of the state: one more value channel whose value is 1 for every token.  So the
// call runs the library's own scan on Dv + 16 value channels -- the 16-column slice granularity of the recurrence kernel, channel Dv
// holding the ones, the other fifteen zeros (they stay zero: columns of S never mix) -- and divides on the way out:
//   norm_pack_kernel     v [rows, Dv] -> v_aug [rows, Dv + 16] = [v | 1 0 ... 0]            (16-byte units, one pass)
//   norm_state_kernel    S_in [Dk, Dv], z_in [Dk] -> S_aug [Dk, Dv + 16]  (and back)
//   gdkvm_scan_fwd       on the augmented problem: every contract of the scan carries over -- chunked calls with (S, z) carried are
//                        bit-identical to one call, the state exponent of the z slice is sized like any other slice's
//   norm_divide_kernel   r_out[row, c] = r_aug[row, c] / (|r_aug[row, Dv]| + eps), fp32 division, one rounding on store
// bf16 I/O: numerator and denominator reach the division rounded to bf16 (they are the scan's stored read-outs), so the quotient is
// within ~3 x 2^-9 relative of the fp32 one; fp32 I/O: within 1e-4 of the fp64 oracle wherever |q . z| is not itself ~eps.
// Not a fast path (the flag is off by default in SPEC-v0): two extra passes over v and r; the recurrence itself is the shipped one.
#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

constexpr int NZ = 16;                   // extra value channels: one 16-column slice of the recurrence kernel

inline size_t nrm_up256(size_t x) { return (x + 255) & ~(size_t)255; }

// one thread per 16-byte unit of the augmented rows
template <int IO>
__global__ __launch_bounds__(256) void norm_pack_kernel(const void* __restrict__ v, void* __restrict__ v_aug, size_t rows, int Dv)
{
    constexpr int E = IO == GDKVM_F32 ? 4 : 8;                        // elements per unit
    const int ua = (Dv + NZ) / E, uv = Dv / E;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ua) return;
    const size_t row = i / ua;
    const int u = (int)(i - row * ua);
    uint4 val = make_uint4(0, 0, 0, 0);
    if (u < uv) val = static_cast<const uint4*>(v)[row * uv + u];
    else if (u == uv) val.x = IO == GDKVM_F32 ? 0x3f800000u : 0x00003f80u;    // 1.0f, or bf16 1.0 in the low half (element Dv)
    static_cast<uint4*>(v_aug)[i] = val;
}

// S [BH, Dk, Dv] (+ z [BH, Dk]) -> S_aug [BH, Dk, Dv + 16]; one thread per float4 of the augmented state
__global__ __launch_bounds__(256) void norm_state_pack_kernel(const float* __restrict__ s, const float* __restrict__ z, float* __restrict__ s_aug,
                                                              size_t rows /* BH * Dk */, int Dv)
{
    const int ua = (Dv + NZ) / 4, uv = Dv / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ua) return;
    const size_t row = i / ua;
    const int u = (int)(i - row * ua);
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if (u < uv) { if (s) val = reinterpret_cast<const f32x4*>(s)[row * uv + u]; }
    else if (u == uv && z) val[0] = z[row];
    reinterpret_cast<f32x4*>(s_aug)[i] = val;
}

__global__ __launch_bounds__(256) void norm_state_unpack_kernel(const float* __restrict__ s_aug, float* __restrict__ s, float* __restrict__ z,
                                                                size_t rows, int Dv)
{
    const int ua = (Dv + NZ) / 4, uv = Dv / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ua) return;
    const size_t row = i / ua;
    const int u = (int)(i - row * ua);
    const f32x4 val = reinterpret_cast<const f32x4*>(s_aug)[i];
    if (u < uv) { if (s) reinterpret_cast<f32x4*>(s)[row * uv + u] = val; }
    else if (u == uv && z) z[row] = val[0];
}

template <int IO>
__global__ __launch_bounds__(256) void norm_divide_kernel(const void* __restrict__ r_aug, void* __restrict__ r_out, size_t rows, int Dv, float eps)
{
    constexpr int E = IO == GDKVM_F32 ? 4 : 8;
    const int uv = Dv / E;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * uv) return;
    const size_t row = i / uv;
    const int u = (int)(i - row * uv);
    const size_t base = row * (size_t)(Dv + NZ);
    const float den = fabsf(load1<IO>(r_aug, base + Dv)) + eps;
    if constexpr (IO == GDKVM_F32) {
        f32x4 x = reinterpret_cast<const f32x4*>(static_cast<const float*>(r_aug) + base)[u];
        x[0] = x[0] / den; x[1] = x[1] / den; x[2] = x[2] / den; x[3] = x[3] / den;
        reinterpret_cast<f32x4*>(r_out)[i] = x;
    } else {
        const f32x4 a = load4<IO>(r_aug, base + 8 * (size_t)u), b = load4<IO>(r_aug, base + 8 * (size_t)u + 4);
        uint4 o;
        o.x = (unsigned)f32_to_bf16(a[0] / den) | ((unsigned)f32_to_bf16(a[1] / den) << 16);
        o.y = (unsigned)f32_to_bf16(a[2] / den) | ((unsigned)f32_to_bf16(a[3] / den) << 16);
        o.z = (unsigned)f32_to_bf16(b[0] / den) | ((unsigned)f32_to_bf16(b[1] / den) << 16);
        o.w = (unsigned)f32_to_bf16(b[2] / den) | ((unsigned)f32_to_bf16(b[3] / den) << 16);
        static_cast<uint4*>(r_out)[i] = o;
    }
}

struct NormView { char* v_aug; char* r_aug; float* s_in; float* s_out; char* ws; size_t ws_bytes; size_t total; };

NormView norm_carve(void* base, int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype)
{
    NormView nv{};
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = base ? static_cast<char*>(base) + off : nullptr; off += nrm_up256(bytes); return r; };
    const size_t es = io_dtype == GDKVM_F32 ? 4 : 2, rows = (size_t)B * T * N * Hh;
    nv.v_aug = take(rows * (Dv + NZ) * es);
    nv.r_aug = take(rows * (Dv + NZ) * es);
    nv.s_in = reinterpret_cast<float*>(take((size_t)B * Hh * Dk * (Dv + NZ) * sizeof(float)));
    nv.s_out = reinterpret_cast<float*>(take((size_t)B * Hh * Dk * (Dv + NZ) * sizeof(float)));
    nv.ws_bytes = gdkvm_scan_workspace_bytes(B, T, Hh, N, Dk, Dv + NZ);
    nv.ws = take(nv.ws_bytes);
    nv.total = off;
    return nv;
}

inline unsigned nblocks(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" size_t gdkvm_scan_normalizer_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype)
{
    if (B < 0 || T < 0 || Hh <= 0 || N < 0 || Dk <= 0 || Dv <= 0 || (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16)) return 256;
    return norm_carve(nullptr, B, T, Hh, N, Dk, Dv, io_dtype).total + 256;
}

extern "C" int gdkvm_scan_fwd_normalizer(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                                         const float* s_in, const float* z_in, void* r_out, float* s_out, float* z_out,
                                         void* workspace, size_t workspace_bytes,
                                         int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, float eps, void* stream)
{
    if (int rc = check_common("scan_fwd_normalizer", B, T, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (!(eps > 0.f) || !(eps < 1.f)) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd_normalizer: eps=%g must lie in (0, 1)", (double)eps);
    if (flags & GDKVM_FLAG_TRAIN) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd_normalizer: inference only (no s_hist / GDKVM_FLAG_TRAIN)");
    if (rule < 0 || rule > 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd_normalizer: rule=%d", rule);
    if (int rc = check_ptrs("scan_fwd_normalizer", {q, k, v, alpha, beta, r_out, workspace}, {s_in, z_in, s_out, z_out})) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t rows = (size_t)B * T * N * Hh, srows = (size_t)B * Hh * Dk;
    if (B == 0) return GDKVM_OK;
    const NormView nv = norm_carve(workspace, B, T, Hh, N, Dk, Dv, io_dtype);
    if (workspace_bytes < nv.total) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "scan_fwd_normalizer: workspace %zu < %zu bytes", workspace_bytes, nv.total);
    const bool carry_in = s_in || z_in;
    if (carry_in)
        hipLaunchKernelGGL(norm_state_pack_kernel, dim3(nblocks(srows * ((Dv + NZ) / 4))), dim3(256), 0, st, s_in, z_in, nv.s_in, srows, Dv);
    if (rows) {
        const size_t units = rows * ((Dv + NZ) / (io_dtype == GDKVM_F32 ? 4 : 8));
        if (io_dtype == GDKVM_F32) hipLaunchKernelGGL(norm_pack_kernel<GDKVM_F32>, dim3(nblocks(units)), dim3(256), 0, st, v, nv.v_aug, rows, Dv);
        else hipLaunchKernelGGL(norm_pack_kernel<GDKVM_BF16>, dim3(nblocks(units)), dim3(256), 0, st, v, nv.v_aug, rows, Dv);
    }
    GDKVM_LAUNCH_CHECK("scan_fwd_normalizer (pack)");
    const bool carry_out = s_out || z_out;
    if (int rc = gdkvm_scan_fwd(q, k, nv.v_aug, alpha, beta, carry_in ? nv.s_in : nullptr, nv.r_aug, carry_out ? nv.s_out : nullptr, nullptr,
                                nv.ws, nv.ws_bytes, B, T, Hh, N, Dk, Dv + NZ, io_dtype, rule, flags, stream))
        return rc;
    if (rows) {
        const size_t units = rows * (Dv / (io_dtype == GDKVM_F32 ? 4 : 8));
        if (io_dtype == GDKVM_F32) hipLaunchKernelGGL(norm_divide_kernel<GDKVM_F32>, dim3(nblocks(units)), dim3(256), 0, st, nv.r_aug, r_out, rows, Dv, eps);
        else hipLaunchKernelGGL(norm_divide_kernel<GDKVM_BF16>, dim3(nblocks(units)), dim3(256), 0, st, nv.r_aug, r_out, rows, Dv, eps);
    }
    if (carry_out)
        hipLaunchKernelGGL(norm_state_unpack_kernel, dim3(nblocks(srows * ((Dv + NZ) / 4))), dim3(256), 0, st, nv.s_out, s_out, z_out, srows, Dv);
    GDKVM_LAUNCH_CHECK("scan_fwd_normalizer (divide)");
    return GDKVM_OK;
}
