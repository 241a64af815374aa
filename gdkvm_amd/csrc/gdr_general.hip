// gdr_general.hip -- gdkvm_scan_fwd for per-head key widths ABOVE the 64 the fast kernels are built for (72 .. 256 in multiples of 8;
// SURVEY.md §8 rows a1-a3, a5; the reference's real key width is unknown -- SURVEY A.7 -- and round 3's ABI returned GDKVM_ERR_SHAPE here).
//
// This is the DEFINITION of the recurrence (read, then the token loop of the write) on the device, not the affine-map form of
// gdr_prep.hip / gdr_scan.hip: one workgroup per (clip, head, 16-column slice of the state) keeps its [Dk][16] fp32 slice in LDS and walks
// the frames; per frame it reads every token against the state BEFORE the frame's write (R_t = Qn S), decays the state, then applies the
// frame's tokens in raster order -- rule 0: S += b k v^T; rule 2: e = v - S^T k against the state as updated so far, S += b k e^T; rule 1:
// the same with e taken against the decayed state of the frame's start (kept in a second LDS slice).  Everything is fp32 FMAs in a fixed
// order (deterministic; bit-identical when a clip is cut into calls with the state carried); q / k normalisation and the gate sigmoids as
// in the fast path's prologue (flags).  Inference only (no s_hist).  It is serial over tokens by construction -- two workgroup barriers per
// token -- and is NOT the measured path: cfg2-sized problems run ~40x slower than the Dk = 64 kernels; it exists so that a model with wider
// keys gets the same results from the same entry point instead of an error.
#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

constexpr int GG_MAXDK = 256;
constexpr int GG_CHUNK = 64;             // tokens whose inverse key norms are staged at a time

struct GeneralArgs {
    const void* q; const void* k; const void* v; const float* alpha; const float* beta; const float* s_in; void* r; float* s_out;
    int B, T, Hh, N, Dk, Dv, rule, flags;
};

template <int IO>
__global__ __launch_bounds__(256) void gdr_general_scan_kernel(GeneralArgs a)
{
    __shared__ float s_S[GG_MAXDK][16];                   // this slice of the state
    __shared__ float s_S0[GG_MAXDK][16];                  // rule 1: the decayed state at the frame's start
    __shared__ float s_red[16][17];                       // partial dots [row group][column]
    __shared__ float s_e[16];
    __shared__ float s_kinv[GG_CHUNK];
    const int tid = threadIdx.x, c = tid & 15, dg = tid >> 4;
    const int sl = blockIdx.x, bh = blockIdx.y, b = bh / a.Hh, h = bh % a.Hh;
    const int N = a.N, Dk = a.Dk, Dv = a.Dv, Hh = a.Hh, c0 = 16 * sl;
    const bool norm = a.flags & GDKVM_FLAG_NORMALIZE_QK, logits = a.flags & GDKVM_FLAG_GATE_LOGITS;
    const int nj = (Dk + 15) / 16;                        // rows dg + 16 j of this thread

    for (int j = 0; j < nj; ++j) {
        const int d = dg + 16 * j;
        if (d < Dk) s_S[d][c] = a.s_in ? a.s_in[((size_t)bh * Dk + d) * Dv + c0 + c] : 0.f;
    }
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
        const size_t bt = (size_t)b * a.T + t;
        // ---- read: token n = 16 pass + dg, column c; a thread walks the whole key axis of its token (q rows are L1 hits across the 16 columns)
        for (int n0 = 0; a.r && n0 < N; n0 += 16) {
            const int n = n0 + dg;
            if (n < N) {
                const size_t row = ((bt * N + n) * Hh + h) * Dk;
                float dot = 0.f, ss = 0.f;
                for (int d = 0; d < Dk; ++d) {
                    const float x = load1<IO>(a.q, row + d);
                    dot = fmaf(x, s_S[d][c], dot);
                    ss = fmaf(x, x, ss);
                }
                if (norm) dot *= rsqrtf(ss + 1e-12f);
                store1<IO>(a.r, ((bt * N + n) * Hh + h) * Dv + c0 + c, dot);      // (the launcher skips this phase without r_out)
            }
        }
        __syncthreads();
        // ---- write: decay, then the tokens in raster order
        float al = a.alpha[bt * Hh + h];
        if (logits) al = 1.f / (1.f + __expf(-al));
        for (int j = 0; j < nj; ++j) {
            const int d = dg + 16 * j;
            if (d < Dk) { const float s = al * s_S[d][c]; s_S[d][c] = s; if (a.rule == 1) s_S0[d][c] = s; }
        }
        __syncthreads();
        for (int nc = 0; nc < N; nc += GG_CHUNK) {
            const int cnt = min(GG_CHUNK, N - nc);
            if (tid < cnt) {                               // inverse key norms of the chunk's tokens
                float inv = 1.f;
                if (norm) {
                    const size_t row = ((bt * N + nc + tid) * Hh + h) * Dk;
                    float ss = 0.f;
                    for (int d = 0; d < Dk; ++d) { const float x = load1<IO>(a.k, row + d); ss = fmaf(x, x, ss); }
                    inv = rsqrtf(ss + 1e-12f);
                }
                s_kinv[tid] = inv;
            }
            __syncthreads();
            for (int i = 0; i < cnt; ++i) {
                const int n = nc + i;
                const size_t krow = ((bt * N + n) * Hh + h) * Dk;
                const float kinv = s_kinv[i];
                float be = a.beta[(bt * N + n) * Hh + h];
                if (logits) be = 1.f / (1.f + __expf(-be));
                float kn[GG_MAXDK / 16];
                float part = 0.f;
                for (int j = 0; j < nj; ++j) {
                    const int d = dg + 16 * j;
                    kn[j] = d < Dk ? load1<IO>(a.k, krow + d) * kinv : 0.f;
                    if (a.rule != 0 && d < Dk) part = fmaf(kn[j], a.rule == 1 ? s_S0[d][c] : s_S[d][c], part);
                }
                float e = load1<IO>(a.v, ((bt * N + n) * Hh + h) * Dv + c0 + c);
                if (a.rule != 0) {                         // e = v - S^T k: sixteen partial dots per column, added in row-group order
                    s_red[dg][c] = part;
                    __syncthreads();
                    if (dg == 0) {
                        float dot = 0.f;
#pragma unroll
                        for (int g = 0; g < 16; ++g) dot += s_red[g][c];
                        s_e[c] = e - dot;
                    }
                    __syncthreads();
                    e = s_e[c];
                }
                const float bev = be * e;
                for (int j = 0; j < nj; ++j) {
                    const int d = dg + 16 * j;
                    if (d < Dk) s_S[d][c] = fmaf(kn[j], bev, s_S[d][c]);
                }
                // (the next token's partial dots read s_S rows of this thread only; s_red / s_e are rewritten behind the barriers above)
            }
            __syncthreads();
        }
    }
    if (a.s_out) {
        for (int j = 0; j < nj; ++j) {
            const int d = dg + 16 * j;
            if (d < Dk) a.s_out[((size_t)bh * Dk + d) * Dv + c0 + c] = s_S[d][c];
        }
    }
}

}  // namespace

bool gdr_wide_keys(int Dk) { return Dk > GDKVM_DK && Dk <= GG_MAXDK && Dk % 8 == 0; }

// gdkvm_scan_fwd for wide keys (called from gdr_scan.hip)
int gdr_general_scan_fwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta, const float* s_in, void* r_out,
                         float* s_out, int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, hipStream_t st)
{
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "scan_fwd: io_dtype=%d", io_dtype);
    if (B < 0 || T < 0 || Hh <= 0 || N < 0 || N > GDKVM_MAX_N || Dv <= 0 || Dv % 16 || !gdr_wide_keys(Dk))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd: B=%d T=%d Hh=%d N=%d Dk=%d Dv=%d", B, T, Hh, N, Dk, Dv);
    if (rule < 0 || rule > 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd: rule=%d", rule);
    if ((long long)B * Hh > 65535) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd: B*Hh=%lld clip-heads (at most 65535 per call with Dk > %d)", (long long)B * Hh, GDKVM_DK);
    if (B == 0) return GDKVM_OK;
    if (!r_out && !s_out) return GDKVM_OK;
    if (T > 0 && N > 0 && (!q || !k || !v || !alpha || !beta)) return gdkvm_fail(GDKVM_ERR_ARG, "scan_fwd: null pointer");
    if (int rc = gdkvm_check_device()) return rc;
    GeneralArgs a{q, k, v, alpha, beta, s_in, r_out, s_out, B, T, Hh, N, Dk, Dv, rule, flags};
    const dim3 grid((unsigned)(Dv / 16), (unsigned)(B * Hh));
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gdr_general_scan_kernel<GDKVM_F32>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gdr_general_scan_kernel<GDKVM_BF16>), grid, dim3(256), 0, st, a);
    GDKVM_LAUNCH_CHECK("gdr_general_scan_kernel");
    return GDKVM_OK;
}
