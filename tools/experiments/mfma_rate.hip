// mfma_rate.hip -- issue rate of v_mfma_f32_16x16x32_bf16 on gfx950: W waves per CU, each looping over NACC independent
// accumulators.  Prints cycles per MFMA per SIMD (s_memtime) and the wall-clock rate.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void rate(float* out, unsigned long long* cyc, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int threads, int iters)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 64 / 4.0, n = (double)iters * NACC;
    printf("waves/CU %2d  NACC %2d: %6.1f cycles per MFMA per wave, %5.1f per SIMD;  %7.1f us -> %.2f PFLOP/s\n", threads / 64, NACC,
           c / n, c / n / (waves_per_simd < 1 ? 1 : waves_per_simd), ms * 1e3, 256.0 * (threads / 64) * n * 16384 / (ms * 1e-3) / 1e15);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<14>(256, 4000); run<14>(512, 4000); run<4>(256, 4000); run<4>(512, 4000); run<2>(512, 4000); run<14>(1024, 4000);
    return 0;
}
