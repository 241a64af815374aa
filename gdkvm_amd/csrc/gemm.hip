// gemm.hip -- the plain products of the training backward (KPFF's dX / dW products and the 1x1
// projections' gradients), hand-written for gfx950 instead of library GEMMs.  Token-major operands, fp32 accumulation.
//
//   gdkvm_gemm_nt   C[M,N]  = A[M,K] B[N,K]^T (+ bias[N])   both operands K-contiguous (token rows x weight rows): every MFMA fragment
//                                                   is one 16-byte global load, no LDS.  (dX = dY W^T-form products: the caller
//                                                   passes the weight with its input index leading, i.e. transposed once.)
//   gdkvm_gemm_tn   C[K1,N] = A[M,K1]^T B[M,N]     the reduction runs over the ROWS (tokens of the batch: M ~ 25k >> K1, N): split
//                                                   over workgroups into fp32 partial tiles, summed by a second kernel in a fixed
//                                                   order (deterministic: no atomics).  Row-major slabs are staged in LDS and read
//                                                   back transposed (ds_read_b64_tr_b16) as MFMA fragments.
// bf16 operands run on v_mfma_f32_16x16x32_bf16, fp32 operands on the exact v_mfma_f32_16x16x4_f32.
#include <stdlib.h>

#include "gdkvm_common.hpp"

namespace {

struct GemmArgs { const void* a; const void* b; void* c; float* part; const float* bias; int M, N, K, K1, rows_per_split, splits;
                  float* part_cs;      // TN: optional partial column sums of A, [splits][K1] (the bias gradient of a token-major product)
                  // TN, CONV form (the weight gradient of a strided convolution): row m of B is not stored -- it is pixel
                  // (n, yo * stride - pad + r, xo * stride - pad + s) of the NHWC tensor b [cN, cH, cW, cC] for m = (n, yo, xo) on the
                  // cHo x cWo output grid and column tile n0 = tap * cC + c0 (zeros outside the image): C = dy^T im2col(x) with no im2col
                  int cH, cW, cC, cHo, cWo, cS, cstride, cpad; };

// ---- NT: workgroup tile 128 (M) x 64 (N); wave w owns rows 32w .. 32w+31 and all 64 columns: 2 x 4 accumulator tiles.  The
// product is computed transposed (C^T = B A^T: the weight rows are the A operand) so that a lane ends with four consecutive
// columns of one output row (8/16-byte stores).
template <int IO>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs p)
{
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N, K = p.K;
    const int m0 = blockIdx.x * 128 + 32 * w, n0 = blockIdx.y * 64;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (IO == GDKVM_BF16) {
        const bf16_t* A = static_cast<const bf16_t*>(p.a);
        const bf16_t* B = static_cast<const bf16_t*>(p.b);
        const bf16_t* ar[2]; const bf16_t* br[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) ar[i] = A + (size_t)min(m0 + 16 * i + li, M - 1) * K + 8 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) br[j] = B + (size_t)min(n0 + 16 * j + li, N - 1) * K + 8 * g;
        bf16x8 av[2], bv[4], an[2], bn[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const bf16x8*>(ar[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(br[j]);
        for (int k = 0; k < K; k += 32) {
            const int kn = min(k + 32, K - 32);            // next step's fragments in flight behind this step's MFMAs
#pragma unroll
            for (int i = 0; i < 2; ++i) an[i] = *reinterpret_cast<const bf16x8*>(ar[i] + kn);
#pragma unroll
            for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const bf16x8*>(br[j] + kn);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[j], av[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = an[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = bn[j];
        }
    } else {
        // exact fp32: lane group g of k step s takes k = g K/4 + s, so four consecutive steps are one 16-byte load
        const float* A = static_cast<const float*>(p.a);
        const float* B = static_cast<const float*>(p.b);
        const int q4 = K / 4;
        const float* ar[2]; const float* br[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) ar[i] = A + (size_t)min(m0 + 16 * i + li, M - 1) * K + g * q4;
#pragma unroll
        for (int j = 0; j < 4; ++j) br[j] = B + (size_t)min(n0 + 16 * j + li, N - 1) * K + g * q4;
        for (int s = 0; s < q4; s += 4) {
            f32x4 av[2], bv[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const f32x4*>(ar[i] + s);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(br[j] + s);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma4(bv[j][r], av[i][r], acc[i][j]);
        }
    }
    // acc[i][j][r] = C[m0 + 16i + li][n0 + 16j + 4g + r]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 16 * i + li;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 16 * j + 4 * g;
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += p.bias[min(n + r, N - 1)];
            }
            if (n + 3 < N && (N & 3) == 0) {
                if constexpr (IO == GDKVM_F32) *reinterpret_cast<f32x4*>(static_cast<float*>(p.c) + (size_t)m * N + n) = acc[i][j];
                else *reinterpret_cast<uint2*>(static_cast<bf16_t*>(p.c) + (size_t)m * N + n) =
                         make_uint2((unsigned)f32_to_bf16(acc[i][j][0]) | ((unsigned)f32_to_bf16(acc[i][j][1]) << 16),
                                    (unsigned)f32_to_bf16(acc[i][j][2]) | ((unsigned)f32_to_bf16(acc[i][j][3]) << 16));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) store1<IO>(p.c, (size_t)m * N + n + r, acc[i][j][r]);
            }
        }
    }
}

// ---- TN: workgroup tile 64 (K1) x 64 (N) of one row split; wave w owns K1 tile w and all four N tiles.  Per step 32 rows of A
// (64 columns of this tile) and of B are staged as 16-column sub-tiles [32 rows][16] (1 KiB each, two LDS buffers) and come
// back through ds_read_b64_tr_b16 with the row index as the MFMA k index.  fp32: 16-row steps, plain LDS reads, exact MFMA.
constexpr int TN_ROWS = 32;

// KT = 16-row K1 tiles per wave (1: a 64 x 64 workgroup tile; 2, bf16 only: 128 x 64 -- eight MFMAs per six fragment reads instead of four
// per five: the strided convolutions' weight gradients, whose K1 = output channels is 128 / 256).
template <int IO, bool CONV = false, int KT = 1>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs p)
{
    constexpr int ESZ = IO == GDKVM_F32 ? 4 : 2;
    static_assert(KT == 1 || IO == GDKVM_BF16, "the 128-row tile is built for bf16 operands");
    __shared__ __attribute__((aligned(16))) char s_a[2][TN_ROWS * 64 * KT * ESZ];   // [buffer][sub-tile][row][16]
    __shared__ __attribute__((aligned(16))) char s_b[2][TN_ROWS * 64 * ESZ];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K1 = p.K1, N = p.N, M = p.M;
    const int k0 = blockIdx.x * 64 * KT, n0 = blockIdx.y * 64, z = blockIdx.z;
    const int r_lo = z * p.rows_per_split, r_hi = min(M, r_lo + p.rows_per_split);
    f32x4 acc[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[kt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // column sums of A (sum over the rows m of A[m][k1] -- the bias gradient when A = dY): one more MFMA per step against a fragment of
    // ones, in the workgroups of the first N tile only (the framework's reduction for it was a zero-fill and 10 - 24 us per projection)
    const bool colsum = KT == 1 && p.part_cs != nullptr && blockIdx.y == 0;
    f32x4 accs = {0.f, 0.f, 0.f, 0.f};
    // staging: thread t moves 16-byte pieces; piece = (row, 16-byte chunk c of the slab row)
    constexpr int EPC = 16 / ESZ;                          // elements per chunk
    constexpr int CPR_B = 64 * ESZ / 16, CPR_A = CPR_B * KT;           // chunks per slab row
    constexpr int PIECES_B = TN_ROWS * CPR_B, PIECES_A = TN_ROWS * CPR_A;   // B: 256 (bf16) or 512 (fp32)
    constexpr int PER_B = (PIECES_B + 255) / 256, PER_A = (PIECES_A + 255) / 256;
    uint4 ra[PER_A], rb[PER_B];
    // CONV: this tile's tap (r, s) and first channel -- N = taps * cC and cC is a multiple of 64, so a 64-column tile lies in one tap
    const int tap = CONV ? n0 / p.cC : 0, tap_r = CONV ? tap / p.cS : 0, tap_s = CONV ? tap - tap_r * p.cS : 0, cc0 = CONV ? n0 - tap * p.cC : 0;
    auto fetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < PER_A; ++u) {
            const int pc = tid + 256 * u, row = pc / CPR_A, c = pc % CPR_A;
            const int m = r0 + row, ka = k0 + c * EPC;
            ra[u] = make_uint4(0u, 0u, 0u, 0u);
            if (pc < PIECES_A && m < r_hi && ka + EPC <= K1)
                ra[u] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.a) + ((size_t)m * K1 + ka) * ESZ);
        }
#pragma unroll
        for (int u = 0; u < PER_B; ++u) {
            const int pc = tid + 256 * u, row = pc / CPR_B, c = pc % CPR_B;
            const int m = r0 + row;
            rb[u] = make_uint4(0u, 0u, 0u, 0u);
            if (pc < PIECES_B && m < r_hi) {
                const int nb = n0 + c * EPC;
                if constexpr (CONV) {
                    const int hw = p.cHo * p.cWo, gn = m / hw, rem = m - gn * hw, yo = rem / p.cWo, xo = rem - yo * p.cWo;
                    const int iy = yo * p.cstride - p.cpad + tap_r, ix = xo * p.cstride - p.cpad + tap_s;
                    if ((unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW)
                        rb[u] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.b) +
                                                                ((((size_t)gn * p.cH + iy) * p.cW + ix) * p.cC + cc0 + c * EPC) * ESZ);
                } else {
                    if (nb + EPC <= N) rb[u] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.b) + ((size_t)m * N + nb) * ESZ);
                }
            }
        }
    };
    auto stage = [&](int buf) {
        // sub-tile = 16 columns: chunk c covers columns c*EPC .. ; sub = column / 16, offset inside the sub-tile row
#pragma unroll
        for (int u = 0; u < PER_A; ++u) {
            const int pc = tid + 256 * u, row = pc / CPR_A, c = pc % CPR_A;
            if (pc < PIECES_A) {
                const int col = c * EPC, sub = col >> 4, off = ((row * 16) + (col & 15)) * ESZ + sub * (TN_ROWS * 16 * ESZ);
                *reinterpret_cast<uint4*>(&s_a[buf][off]) = ra[u];
            }
        }
#pragma unroll
        for (int u = 0; u < PER_B; ++u) {
            const int pc = tid + 256 * u, row = pc / CPR_B, c = pc % CPR_B;
            if (pc < PIECES_B) {
                const int col = c * EPC, sub = col >> 4, off = ((row * 16) + (col & 15)) * ESZ + sub * (TN_ROWS * 16 * ESZ);
                *reinterpret_cast<uint4*>(&s_b[buf][off]) = rb[u];
            }
        }
    };
    fetch(r_lo);
    int buf = 0;
    for (int r0 = r_lo; r0 < r_hi; r0 += TN_ROWS) {
        stage(buf);
        __syncthreads();
        if (r0 + TN_ROWS < r_hi) fetch(r0 + TN_ROWS);      // next slab in flight behind this step's MFMAs
        if constexpr (IO == GDKVM_BF16) {
            // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the group's 4-row block of a [32][16] sub-tile
            const unsigned la = (unsigned)((8 * g + (li >> 2)) * 16 + 4 * (li & 3)) * 2;
            auto frag = [&](const char* base) {
                const unsigned addr = (unsigned)(uintptr_t)base + la;
                uint2 x0, x1;
                asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(x0), "=&v"(x1) : "v"(addr) : "memory");
                return __builtin_bit_cast(bf16x8, make_uint4(x0.x, x0.y, x1.x, x1.y));
            };
            bf16x8 af[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) af[kt] = frag(&s_a[buf][(w * KT + kt) * (TN_ROWS * 16 * 2)]);
            __builtin_amdgcn_sched_barrier(0);
            if (colsum) {
                const unsigned one2 = 0x3f803f80u;          // bf16 1.0 twice
                accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], __builtin_bit_cast(bf16x8, make_uint4(one2, one2, one2, one2)), accs, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 bf = frag(&s_b[buf][j * (TN_ROWS * 16 * 2)]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) acc[kt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kt], bf, acc[kt][j], 0, 0, 0);
            }
        } else {
            const float* sa = reinterpret_cast<const float*>(&s_a[buf][w * (TN_ROWS * 16 * 4)]);
#pragma unroll
            for (int s = 0; s < TN_ROWS / 4; ++s) {        // k step: rows 4s + g
                const float av = sa[(4 * s + g) * 16 + li];
                if (colsum) accs = mfma4(av, 1.0f, accs);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float bv = reinterpret_cast<const float*>(&s_b[buf][j * (TN_ROWS * 16 * 4)])[(4 * s + g) * 16 + li];
                    acc[0][j] = mfma4(av, bv, acc[0][j]);
                }
            }
        }
        buf ^= 1;                                          // (the other buffer was last read two barriers ago)
    }
    // acc[kt][j][r] = C[k0 + 16 (w KT + kt) + 4g + r][n0 + 16j + li] of this split
    float* P = p.part + (size_t)z * K1 * N;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = k0 + 16 * (w * KT + kt) + 4 * g + r, n = n0 + 16 * j + li;
                if (kk < K1 && n < N) P[(size_t)kk * N + n] = acc[kt][j][r];
            }
    if (colsum && li == 0) {                               // every column of accs holds the same sums: column 0 writes them
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = k0 + 16 * w + 4 * g + r;
            if (kk < K1) p.part_cs[(size_t)z * K1 + kk] = accs[r];
        }
    }
}

// (n2 > 0: a second, short array -- the partial column sums -- is summed by the same launch: elements n .. n + n2 - 1)
__global__ void gemm_reduce_kernel(const float* part, float* c, size_t n, int splits, const float* part2 = nullptr, float* c2 = nullptr, size_t n2 = 0)
{
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n + n2; i0 += (size_t)gridDim.x * blockDim.x) {
        const bool second = i0 >= n;
        const size_t i = second ? i0 - n : i0;
        if (second) {
            float s = 0.f;
            for (int z = 0; z < splits; ++z) s += part2[(size_t)z * n2 + i];
            c2[i] = s;
            continue;
        }
        // fixed order (deterministic): eight running sums over splits z = j (mod 8), eight loads in flight instead of a dependent chain
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= splits; z += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] += part[(size_t)(z + j) * n + i];
        for (int j = 0; z < splits; ++z, ++j) s8[j] += part[(size_t)z * n + i];
        c[i] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
}

// The split sums of the CONV form, written where the parameter's gradient lives: part [splits][K][taps][C] -> dw[k sk + c sc + tap_r sr + tap_s ss]
// (element strides: any memory format of a [K, C, R, S] parameter), same fixed summation order as gemm_reduce_kernel.
__global__ void conv_wgrad_reduce_kernel(const float* part, float* dw, int K, int C, int R, int S, int splits,
                                         long long sk, long long sc, long long sr, long long ss)
{
    const size_t n = (size_t)K * R * S * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= splits; z += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] += part[(size_t)(z + j) * n + i];
        for (int j = 0; z < splits; ++z, ++j) s8[j] += part[(size_t)z * n + i];
        const int c = (int)(i % C), tap = (int)((i / C) % (R * S)), k = (int)(i / ((size_t)C * R * S));
        dw[(long long)k * sk + (long long)c * sc + (long long)(tap / S) * sr + (long long)(tap % S) * ss] =
            ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
}

int check_gemm(const char* fn, int M, int N, int K, int io_dtype)
{
    if (M < 0 || N < 0 || K < 0) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: negative dimension (M=%d N=%d K=%d)", fn, M, N, K);
    if (io_dtype != GDKVM_F32 && io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "%s: io_dtype=%d", fn, io_dtype);
    return GDKVM_OK;
}

}  // namespace

extern "C" int gdkvm_gemm_nt(const void* a, const void* b, const float* bias, void* c, int M, int N, int K, int io_dtype, void* stream)
{
    if (int rc = check_gemm("gemm_nt", M, N, K, io_dtype)) return rc;
    if (M == 0 || N == 0) return GDKVM_OK;
    const int kq = io_dtype == GDKVM_BF16 ? 32 : 16;
    if (K == 0 || K % kq) return gdkvm_fail(GDKVM_ERR_SHAPE, "gemm_nt: K=%d must be a positive multiple of %d", K, kq);
    if (!a || !b || !c || !gdkvm_aligned16(a) || !gdkvm_aligned16(b) || !gdkvm_aligned16(c))
        return gdkvm_fail(GDKVM_ERR_ARG, "gemm_nt: null or unaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    GemmArgs ga{a, b, c, nullptr, bias, M, N, K, 0, 0, 0};
    const dim3 grid((unsigned)((M + 127) / 128), (unsigned)((N + 63) / 64));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gemm_nt_kernel<GDKVM_F32>), grid, dim3(256), 0, st, ga);
    else hipLaunchKernelGGL((gemm_nt_kernel<GDKVM_BF16>), grid, dim3(256), 0, st, ga);
    GDKVM_LAUNCH_CHECK("gemm_nt_kernel");
    return GDKVM_OK;
}

static int tn_splits(int M) { const int s = (M + 1023) / 1024; return s < 1 ? 1 : s; }

extern "C" size_t gdkvm_gemm_tn_workspace_bytes(int M, int K1, int N)
{
    if (M <= 0 || K1 <= 0 || N <= 0) return 16;
    return (size_t)tn_splits(M) * K1 * N * sizeof(float);
}

static int gemm_tn_impl(const char* who, const void* a, const void* b, float* c, float* colsum, void* workspace, size_t workspace_bytes,
                        int M, int K1, int N, int io_dtype, void* stream)
{
    if (int rc = check_gemm(who, M, N, K1, io_dtype)) return rc;
    if (K1 == 0 || N == 0) return GDKVM_OK;
    const int eq = io_dtype == GDKVM_BF16 ? 8 : 4;
    if (K1 % eq || N % eq) return gdkvm_fail(GDKVM_ERR_SHAPE, "%s: K1=%d and N=%d must be multiples of %d", who, K1, N, eq);
    if (!c || !gdkvm_aligned16(c) || (M > 0 && (!a || !b || !gdkvm_aligned16(a) || !gdkvm_aligned16(b))))
        return gdkvm_fail(GDKVM_ERR_ARG, "%s: null or unaligned pointer", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = gdkvm_check_device()) return rc;
    if (M == 0) {
        if (int rc = gdkvm_zero_async(c, (size_t)K1 * N * sizeof(float), st)) return rc;
        return colsum ? gdkvm_zero_async(colsum, (size_t)K1 * sizeof(float), st) : GDKVM_OK;
    }
    const size_t need = gdkvm_gemm_tn_workspace_bytes(M, K1, N) + (colsum ? (size_t)tn_splits(M) * K1 * sizeof(float) : 0);
    if (!workspace || workspace_bytes < need) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, need);
    const int splits = tn_splits(M);
    const int rows = ((M + splits - 1) / splits + TN_ROWS - 1) / TN_ROWS * TN_ROWS;
    float* part_cs = colsum ? static_cast<float*>(workspace) + (size_t)splits * K1 * N : nullptr;
    GemmArgs ga{a, b, nullptr, static_cast<float*>(workspace), nullptr, M, N, 0, K1, rows, splits, part_cs};
    const dim3 grid((unsigned)((K1 + 63) / 64), (unsigned)((N + 63) / 64), (unsigned)splits);
    if (io_dtype == GDKVM_F32) hipLaunchKernelGGL((gemm_tn_kernel<GDKVM_F32>), grid, dim3(256), 0, st, ga);
    else hipLaunchKernelGGL((gemm_tn_kernel<GDKVM_BF16>), grid, dim3(256), 0, st, ga);
    GDKVM_LAUNCH_CHECK("gemm_tn_kernel");
    const size_t n = (size_t)K1 * N, n2 = colsum ? (size_t)K1 : 0;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((n + n2 + 255) / 256 > 1024 ? 1024 : (n + n2 + 255) / 256)), dim3(256), 0, st,
                       static_cast<const float*>(workspace), c, n, splits, static_cast<const float*>(part_cs), colsum, n2);
    GDKVM_LAUNCH_CHECK("gemm_reduce_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_gemm_tn(const void* a, const void* b, float* c, void* workspace, size_t workspace_bytes,
                             int M, int K1, int N, int io_dtype, void* stream)
{
    return gemm_tn_impl("gemm_tn", a, b, c, nullptr, workspace, workspace_bytes, M, K1, N, io_dtype, stream);
}

extern "C" size_t gdkvm_gemm_tn_colsum_workspace_bytes(int M, int K1, int N)
{
    return gdkvm_gemm_tn_workspace_bytes(M, K1, N) + (size_t)tn_splits(M) * (K1 > 0 ? K1 : 0) * sizeof(float);
}

extern "C" int gdkvm_gemm_tn_colsum(const void* a, const void* b, float* c, float* colsum, void* workspace, size_t workspace_bytes,
                                    int M, int K1, int N, int io_dtype, void* stream)
{
    if (!colsum) return gdkvm_fail(GDKVM_ERR_ARG, "gemm_tn_colsum: null pointer");
    return gemm_tn_impl("gemm_tn_colsum", a, b, c, colsum, workspace, workspace_bytes, M, K1, N, io_dtype, stream);
}

// ---- weight gradient of a strided convolution (training; SURVEY.md §8 row a7's neighbours: the two 3x3 / stride-2 layers and the two
// 1x1 / stride-2 downsample branches of the encoder, the last layers whose gradients were library kernels with atomic sums):
//   dw[k][c][r][s] = sum over (n, yo, xo) dy[n, yo, xo, k] * x[n, yo stride - pad + r, xo stride - pad + s, c]
// as gemm_tn's CONV form: the rows of the batch are split over workgroups (2048 per split) into fp32 partial tiles, summed in a fixed order.
// Row splits: enough workgroups to keep every CU several deep (the kernel is a chain of staged 32-row slabs: latency-bound per workgroup, so
// what fills the chip is the NUMBER of resident workgroups) without drowning the reduction in partial tiles: ~1024 workgroups in all, at
// least 256 rows each.  (A 1x1 / stride-2 branch has ONE column tile: at 2048 rows per split it ran on 49 workgroups and took as long as
// its block's 3x3 layer with nine times the work.)  GDKVM_CW_WGS overrides the target (A/B runs).
static int cw_splits(long long M, int K, int taps, int C)
{
    static const int target = [] { const char* e = getenv("GDKVM_CW_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    const long long tiles = (long long)((K + 127) / 128) * (taps * C / 64);
    long long s = (target + tiles - 1) / tiles;
    const long long smax = (M + 255) / 256;
    if (s > smax) s = smax;
    return (int)(s < 1 ? 1 : s);
}

extern "C" size_t gdkvm_conv_wgrad_strided_workspace_bytes(int N, int C, int H, int W, int K, int R, int S, int stride, int pad)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return 16;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho < 1 || Wo < 1) return 16;
    return (size_t)cw_splits((long long)N * Ho * Wo, K, R * S, C) * K * R * S * C * sizeof(float);
}

extern "C" int gdkvm_conv_wgrad_strided(const void* x, const void* dy, float* dw, long long sk, long long sc, long long sr, long long ss,
                                        void* workspace, size_t workspace_bytes,
                                        int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int io_dtype, void* stream)
{
    if (io_dtype != GDKVM_BF16) return gdkvm_fail(GDKVM_ERR_DTYPE, "conv_wgrad_strided: only bf16 is implemented");
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0 || C % 64 || K % 8)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_wgrad_strided: N=%d C=%d H=%d W=%d K=%d %dx%d stride %d pad %d (C a multiple of 64, K of 8)",
                          N, C, H, W, K, R, S, stride, pad);
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho < 1 || Wo < 1) return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_wgrad_strided: empty output grid");
    const long long M = (long long)N * Ho * Wo;
    if (M > 0x7fffffffLL || (long long)N * H * W * C > 0x7fffffffLL || M * K > 0x7fffffffLL)
        return gdkvm_fail(GDKVM_ERR_SHAPE, "conv_wgrad_strided: tensor too large for 32-bit offsets");
    if (!dw || (N > 0 && (!x || !dy || !gdkvm_aligned16(x) || !gdkvm_aligned16(dy)))) return gdkvm_fail(GDKVM_ERR_ARG, "conv_wgrad_strided: null or unaligned pointer");
    if (int rc = gdkvm_check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int taps = R * S;
    const size_t n = (size_t)K * taps * C;
    const int splits = N == 0 ? 0 : cw_splits(M, K, taps, C);
    if (N > 0) {
        const size_t need = (size_t)splits * n * sizeof(float);
        if (!workspace || !gdkvm_aligned16(workspace) || workspace_bytes < need)
            return gdkvm_fail(GDKVM_ERR_WORKSPACE, "conv_wgrad_strided: workspace %zu < %zu bytes", workspace_bytes, need);
        const int rows = (int)(((M + splits - 1) / splits + TN_ROWS - 1) / TN_ROWS * TN_ROWS);
        GemmArgs ga{dy, x, nullptr, static_cast<float*>(workspace), nullptr, (int)M, taps * C, 0, K, rows, splits, nullptr, H, W, C, Ho, Wo, S, stride, pad};
        if (K % 128 == 0) {                                 // 128 x 64 workgroup tiles: twice the MFMAs per staged row slab
            const dim3 grid((unsigned)(K / 128), (unsigned)(taps * C / 64), (unsigned)splits);
            hipLaunchKernelGGL((gemm_tn_kernel<GDKVM_BF16, true, 2>), grid, dim3(256), 0, st, ga);
        } else {
            const dim3 grid((unsigned)((K + 63) / 64), (unsigned)(taps * C / 64), (unsigned)splits);
            hipLaunchKernelGGL((gemm_tn_kernel<GDKVM_BF16, true>), grid, dim3(256), 0, st, ga);
        }
        GDKVM_LAUNCH_CHECK("gemm_tn_kernel<conv>");
    }
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, st,
                       static_cast<const float*>(workspace), dw, K, C, R, S, splits, sk, sc, sr, ss);
    GDKVM_LAUNCH_CHECK("conv_wgrad_reduce_kernel");
    return GDKVM_OK;
}
