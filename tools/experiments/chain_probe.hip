// chain_probe.hip -- feasibility probe for a register-resident GDR chain (no product code): ONE wave carries a 64 x 16 slice of
// the state through S <- a P S + G with P as three-term bf16 A images read from LDS and S never leaving the wave's registers
// (the accumulator tiles of step t are the B operand of step t+1 under the k permutation k = 32s + 16(j>>2) + 4G + (j&3)).
// Prints cycles and ns per frame for NCH chain waves per workgroup.   hipcc --offload-arch=gfx950 -O3 chain_probe.hip -o chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
static __device__ __forceinline__ void split3x4(const f32x4& x, uint2& h, uint2& m, uint2& l)
{
    h = make_uint2(cvt_pk_bf16(x[0], x[1]), cvt_pk_bf16(x[2], x[3]));
    const float r0 = x[0] - __uint_as_float(h.x << 16), r1 = x[1] - __uint_as_float(h.x & 0xffff0000u);
    const float r2 = x[2] - __uint_as_float(h.y << 16), r3 = x[3] - __uint_as_float(h.y & 0xffff0000u);
    m = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
    l = make_uint2(cvt_pk_bf16(r0 - __uint_as_float(m.x << 16), r1 - __uint_as_float(m.x & 0xffff0000u)),
                   cvt_pk_bf16(r2 - __uint_as_float(m.y << 16), r3 - __uint_as_float(m.y & 0xffff0000u)));
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// x = h + l with h, l fp16: 22 significant bits (absolute floor 2^-25 once l is subnormal)
static __device__ __forceinline__ void split2x4(const f32x4& x, uint2& h, uint2& l)
{
    const f16x2_t h01 = __builtin_convertvector((f32x2_t){x[0], x[1]}, f16x2_t), h23 = __builtin_convertvector((f32x2_t){x[2], x[3]}, f16x2_t);
    const f32x2_t b01 = __builtin_convertvector(h01, f32x2_t), b23 = __builtin_convertvector(h23, f32x2_t);
    const f16x2_t l01 = __builtin_convertvector((f32x2_t){x[0] - b01[0], x[1] - b01[1]}, f16x2_t);
    const f16x2_t l23 = __builtin_convertvector((f32x2_t){x[2] - b23[0], x[3] - b23[1]}, f16x2_t);
    h = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    l = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

constexpr int P_F4 = 4 * 3 * 2 * 64;     // P images of one frame (uint4 units): [m][term][s][lane]
constexpr int SLOTS = 2;

template <int NCH>
__global__ __launch_bounds__(64 * NCH) void chain_kernel(const uint4* pimg, const f32x4* gimg, float* out, unsigned long long* cyc, int T)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4* s_P = smem;                                  // [SLOTS][P_F4]
    f32x4* s_G = reinterpret_cast<f32x4*>(smem + SLOTS * P_F4);   // [SLOTS][NCH][4][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < SLOTS * P_F4; i += 64 * NCH) s_P[i] = pimg[i];
    for (int i = tid; i < SLOTS * NCH * 4 * 64; i += 64 * NCH) s_G[i] = gimg[i % (SLOTS * 4 * 64)];
    __syncthreads();
    f32x4 S[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = f32x4{0.01f * lane, 0.02f, -0.01f * m, 0.03f};
    bf16x8 sb[3][2];
    auto resplit = [&](int s) __attribute__((always_inline)) {
        uint2 a0, a1, a2, b0, b1, b2;
        split3x4(S[2 * s], a0, a1, a2);
        split3x4(S[2 * s + 1], b0, b1, b2);
        sb[0][s] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
        sb[1][s] = __builtin_bit_cast(bf16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
        sb[2][s] = __builtin_bit_cast(bf16x8, make_uint4(a2.x, a2.y, b2.x, b2.y));
    };
    resplit(0); resplit(1);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < T; ++t) {
        const uint4* ps = s_P + (t & (SLOTS - 1)) * P_F4;
        const f32x4* gs = s_G + ((t & (SLOTS - 1)) * NCH + w) * 4 * 64;
        f32x4 acc0[4], acc1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { acc0[m] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#define PA(M, TERM, KS) __builtin_bit_cast(bf16x8, ps[(((M) * 3 + (TERM)) * 2 + (KS)) * 64 + lane])
#define PS(ACC, M, PT, ST, KS) ACC[M] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA(M, PT, KS), sb[ST][KS], ACC[M], 0, 0, 0)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                PS(acc0, m, 2, 0, ks); PS(acc1, m, 0, 2, ks); PS(acc0, m, 1, 1, ks);
                PS(acc1, m, 1, 0, ks); PS(acc0, m, 0, 1, ks); PS(acc1, m, 0, 0, ks);
            }
        const float alpha = 0.9f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 gt = gs[m * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) S[m][r] = alpha * (acc0[m][r] + acc1[m][r]) + gt[r];
        }
        resplit(0); resplit(1);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * NCH + w] = t1 - t0;
#pragma unroll
    for (int m = 0; m < 4; ++m) reinterpret_cast<f32x4*>(out)[((blockIdx.x * NCH + w) * 4 + m) * 64 + lane] = S[m];
}

// fp16-pair chain: P = Ph + Pl, S = Sh + Sl (fp16), three products per k step; software-pipelined by hand: the k-step-0 MFMAs of
// frame t+1 (they need only tiles 0,1 of S_t) are issued before the re-split of tiles 2,3.
template <int NCH>
__global__ __launch_bounds__(64 * NCH) void chain16_kernel(const uint4* pimg, const f32x4* gimg, float* out, unsigned long long* cyc, int T)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4* s_P = smem;                                  // [SLOTS][4 m][2 terms][2 s][64]  (the probe reuses the 3-term buffer)
    f32x4* s_G = reinterpret_cast<f32x4*>(smem + SLOTS * P_F4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < SLOTS * P_F4; i += 64 * NCH) s_P[i] = pimg[i];
    for (int i = tid; i < SLOTS * NCH * 4 * 64; i += 64 * NCH) s_G[i] = gimg[i % (SLOTS * 4 * 64)];
    __syncthreads();
    f32x4 S[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = f32x4{0.01f * lane, 0.02f, -0.01f * m, 0.03f};
    f16x8 sb[2][2];                                     // [term][kstep]
    auto resplit = [&](int s) __attribute__((always_inline)) {
        uint2 a0, a1, b0, b1;
        split2x4(S[2 * s], a0, a1);
        split2x4(S[2 * s + 1], b0, b1);
        sb[0][s] = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
        sb[1][s] = __builtin_bit_cast(f16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
    };
    resplit(0); resplit(1);
    f32x4 acc[4];
#define PA16(PS_, M, TERM, KS) __builtin_bit_cast(f16x8, PS_[(((M) * 2 + (TERM)) * 2 + (KS)) * 64 + lane])
#define MF(M, PT, ST, KS, PS_) acc[M] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PA16(PS_, M, PT, KS), sb[ST][KS], acc[M], 0, 0, 0)
    const unsigned long long t0 = __builtin_readcyclecounter();
    {   // prologue: k step 0 of frame 0
        const uint4* ps = s_P;
#pragma unroll
        for (int m = 0; m < 4; ++m) { acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; MF(m, 1, 0, 0, ps); MF(m, 0, 1, 0, ps); MF(m, 0, 0, 0, ps); }
    }
    for (int t = 0; t < T; ++t) {
        const uint4* ps = s_P + (t & (SLOTS - 1)) * P_F4;
        const uint4* pn = s_P + ((t + 1) & (SLOTS - 1)) * P_F4;
        const f32x4* gs = s_G + ((t & (SLOTS - 1)) * NCH + w) * 4 * 64;
        const float alpha = 0.9f;
        f32x4 gt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) gt[m] = gs[m * 64 + lane];
        // k step 1 of frame t, tile by tile; tile m is final after its three MFMAs
#pragma unroll
        for (int m = 0; m < 2; ++m) { MF(m, 1, 0, 1, ps); MF(m, 0, 1, 1, ps); MF(m, 0, 0, 1, ps); }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[m][r] = alpha * acc[m][r] + gt[m][r];
        resplit(0);
#pragma unroll
        for (int m = 2; m < 4; ++m) { MF(m, 1, 0, 1, ps); MF(m, 0, 1, 1, ps); MF(m, 0, 0, 1, ps); }
        // k step 0 of frame t+1 needs only sb[.][0]
        f32x4 nacc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            nacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            nacc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PA16(pn, m, 1, 0), sb[0][0], nacc[m], 0, 0, 0);
            nacc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PA16(pn, m, 0, 0), sb[1][0], nacc[m], 0, 0, 0);
            nacc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PA16(pn, m, 0, 0), sb[0][0], nacc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 2; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[m][r] = alpha * acc[m][r] + gt[m][r];
        resplit(1);
#ifdef SGB
        // 6 MFMA (tiles 0,1) | 6 x (1 MFMA of tiles 2,3 + VALU of tiles 0,1) | 12 x (1 MFMA of the next frame + VALU of tiles 2,3)
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x002, SGB_A, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
#pragma unroll
        for (int i = 0; i < 12; ++i) { __builtin_amdgcn_sched_group_barrier(0x002, SGB_B, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
#endif
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = nacc[m];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * NCH + w] = t1 - t0;
#pragma unroll
    for (int m = 0; m < 4; ++m) reinterpret_cast<f32x4*>(out)[((blockIdx.x * NCH + w) * 4 + m) * 64 + lane] = S[m] + acc[m];
}

template <int NCH>
void run16(int grid, int T, const uint4* dp, const f32x4* dg, float* dout, unsigned long long* dcyc)
{
    const size_t lds = (size_t)SLOTS * P_F4 * 16 + (size_t)SLOTS * NCH * 4 * 64 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain16_kernel<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain16_kernel<NCH>, dim3(grid), dim3(64 * NCH), lds, 0, dp, dg, dout, dcyc, T);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid * NCH);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto x : c) mx = x > mx ? x : mx;
    printf("fp16-pair NCH=%d grid=%d T=%d: %.3f us total, %.1f ns/frame (wall), %.1f counter ticks/frame (max wave)\n", NCH, grid, T, ms * 1e3,
           ms * 1e6 / T, (double)mx / T);
}

// fp16-pair chain, phase-structured by hand (sched_barrier between phases), next frame's operands prefetched into registers
template <int NCH>
__global__ __launch_bounds__(64 * NCH) void chain16b_kernel(const uint4* pimg, const f32x4* gimg, float* out, unsigned long long* cyc, int T)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4* s_P = smem;
    f32x4* s_G = reinterpret_cast<f32x4*>(smem + SLOTS * P_F4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < SLOTS * P_F4; i += 64 * NCH) s_P[i] = pimg[i];
    for (int i = tid; i < SLOTS * NCH * 4 * 64; i += 64 * NCH) s_G[i] = gimg[i % (SLOTS * 4 * 64)];
    __syncthreads();
    f32x4 S[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = f32x4{0.01f * lane, 0.02f, -0.01f * m, 0.03f};
    f16x8 sb[2][2];
    auto split_tile = [&](int m, uint2& h, uint2& l) __attribute__((always_inline)) { split2x4(S[m], h, l); };
    struct Ops { f16x8 p[4][2][2]; f32x4 g[4]; };
    auto load_ops = [&](int t, Ops& o) __attribute__((always_inline)) {
        const uint4* ps = s_P + (t & (SLOTS - 1)) * P_F4;
        const f32x4* gs = s_G + ((t & (SLOTS - 1)) * NCH + w) * 4 * 64;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) o.p[m][tm][ks] = __builtin_bit_cast(f16x8, ps[((m * 2 + tm) * 2 + ks) * 64 + lane]);
            o.g[m] = gs[m * 64 + lane];
        }
    };
    f32x4 acc[4];
    uint2 h0, l0, h1, l1;
    split_tile(0, h0, l0); split_tile(1, h1, l1);
    sb[0][0] = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y)); sb[1][0] = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
    split_tile(2, h0, l0); split_tile(3, h1, l1);
    sb[0][1] = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y)); sb[1][1] = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
    Ops oa, ob;
    load_ops(0, oa);
#define M3(ACC, O, M, KS) \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(O.p[M][1][KS], sb[0][KS], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(O.p[M][0][KS], sb[1][KS], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(O.p[M][0][KS], sb[0][KS], ACC, 0, 0, 0)
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; M3(acc[m], oa, m, 0); }
    const float alpha = 0.9f;
    auto frame = [&](int t, Ops& cur, Ops& nxt) __attribute__((always_inline)) {
        load_ops(t + 1, nxt);                                  // P1: prefetch
        __builtin_amdgcn_sched_barrier(0);
        M3(acc[0], cur, 0, 1); M3(acc[1], cur, 1, 1);          // P2
        __builtin_amdgcn_sched_barrier(0);
        uint2 ha, la, hb, lb;
        M3(acc[2], cur, 2, 1);                                 // P3: + VALU tile 0
#pragma unroll
        for (int r = 0; r < 4; ++r) S[0][r] = alpha * acc[0][r] + cur.g[0][r];
        split_tile(0, ha, la);
        __builtin_amdgcn_sched_barrier(0);
        M3(acc[3], cur, 3, 1);                                 // P4: + VALU tile 1
#pragma unroll
        for (int r = 0; r < 4; ++r) S[1][r] = alpha * acc[1][r] + cur.g[1][r];
        split_tile(1, hb, lb);
        sb[0][0] = __builtin_bit_cast(f16x8, make_uint4(ha.x, ha.y, hb.x, hb.y)); sb[1][0] = __builtin_bit_cast(f16x8, make_uint4(la.x, la.y, lb.x, lb.y));
        __builtin_amdgcn_sched_barrier(0);
        f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f}, n3 = {0.f, 0.f, 0.f, 0.f};
        M3(n0, nxt, 0, 0); M3(n1, nxt, 1, 0);                  // P5: + VALU tile 2
#pragma unroll
        for (int r = 0; r < 4; ++r) S[2][r] = alpha * acc[2][r] + cur.g[2][r];
        split_tile(2, ha, la);
        __builtin_amdgcn_sched_barrier(0);
        M3(n2, nxt, 2, 0); M3(n3, nxt, 3, 0);                  // P6: + VALU tile 3
#pragma unroll
        for (int r = 0; r < 4; ++r) S[3][r] = alpha * acc[3][r] + cur.g[3][r];
        split_tile(3, hb, lb);
        sb[0][1] = __builtin_bit_cast(f16x8, make_uint4(ha.x, ha.y, hb.x, hb.y)); sb[1][1] = __builtin_bit_cast(f16x8, make_uint4(la.x, la.y, lb.x, lb.y));
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = n0; acc[1] = n1; acc[2] = n2; acc[3] = n3;
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < T; t += 2) { frame(t, oa, ob); frame(t + 1, ob, oa); }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * NCH + w] = t1 - t0;
#pragma unroll
    for (int m = 0; m < 4; ++m) reinterpret_cast<f32x4*>(out)[((blockIdx.x * NCH + w) * 4 + m) * 64 + lane] = S[m] + acc[m];
}

template <int NCH>
void run16b(int grid, int T, const uint4* dp, const f32x4* dg, float* dout, unsigned long long* dcyc)
{
    const size_t lds = (size_t)SLOTS * P_F4 * 16 + (size_t)SLOTS * NCH * 4 * 64 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain16b_kernel<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain16b_kernel<NCH>, dim3(grid), dim3(64 * NCH), lds, 0, dp, dg, dout, dcyc, T);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid * NCH);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto x : c) mx = x > mx ? x : mx;
    printf("fp16-pair-B NCH=%d grid=%d T=%d: %.3f us total, %.1f ns/frame (wall), %.1f counter ticks/frame (max wave)\n", NCH, grid, T, ms * 1e3,
           ms * 1e6 / T, (double)mx / T);
}

template <int NCH>
void run(int grid, int T, const uint4* dp, const f32x4* dg, float* dout, unsigned long long* dcyc)
{
    const size_t lds = (size_t)SLOTS * P_F4 * 16 + (size_t)SLOTS * NCH * 4 * 64 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain_kernel<NCH>, dim3(grid), dim3(64 * NCH), lds, 0, dp, dg, dout, dcyc, T);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid * NCH);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto x : c) mx = x > mx ? x : mx;
    printf("NCH=%d grid=%d T=%d: %.3f us total, %.1f ns/frame (wall), %.1f counter ticks/frame (max wave)\n", NCH, grid, T, ms * 1e3,
           ms * 1e6 / T, (double)mx / T);
}

int main(int argc, char** argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 512;
    std::vector<unsigned short> hp((size_t)SLOTS * P_F4 * 8);
    srand(1);
    for (auto& x : hp) { float f = ((rand() % 2001) - 1000) * 1e-4f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
    std::vector<float> hg((size_t)SLOTS * 4 * 64 * 4);
    for (auto& x : hg) x = ((rand() % 2001) - 1000) * 1e-3f;
    uint4* dp; f32x4* dg; float* dout; unsigned long long* dcyc;
    hipMalloc(&dp, hp.size() * 2); hipMalloc(&dg, hg.size() * 4); hipMalloc(&dout, 256 * 4 * 4 * 64 * 16); hipMalloc(&dcyc, 256 * 4 * 8);
    hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    for (int grid : {1, 128, 256}) {
        run<1>(grid, T, dp, dg, dout, dcyc);
        run<2>(grid, T, dp, dg, dout, dcyc);
        run<4>(grid, T, dp, dg, dout, dcyc);
        run16<1>(grid, T, dp, dg, dout, dcyc);
        run16<2>(grid, T, dp, dg, dout, dcyc);
        run16<4>(grid, T, dp, dg, dout, dcyc);
        run16b<1>(grid, T, dp, dg, dout, dcyc);
        run16b<2>(grid, T, dp, dg, dout, dcyc);
        run16b<4>(grid, T, dp, dg, dout, dcyc);
    }
    return 0;
}
