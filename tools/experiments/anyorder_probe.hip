// anyorder_probe.hip -- does hipExtAnyOrderLaunch let a kernel start while the PREVIOUS kernel of the same stream is still running (no
// barrier bit on its dispatch packet) on gfx950?  (hip_ext.h notes the flag as unsupported on GFX9xx boards for one of the three entry
// points.)  Kernel A: 32 workgroups that spin ~300 us; kernel B: 512 short workgroups.  Both stamp s_memrealtime (100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o anyorder_probe anyorder_probe.hip && ./anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long rt() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

__global__ void spin(unsigned long long* stamps, unsigned long long ticks)
{
    const unsigned long long t0 = rt();
    while (rt() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = rt(); }
}
__global__ void quick(unsigned long long* stamps)
{
    const unsigned long long t0 = rt();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = rt(); }
}

int main()
{
    const int NA = 32, NB = 512;
    unsigned long long *sa, *sb, ha[2 * NA], hb[2 * NB];
    CHECK(hipMalloc(&sa, sizeof(ha))); CHECK(hipMalloc(&sb, sizeof(hb)));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int flags : {0, 1}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin, dim3(NA), dim3(256), 0, st, sa, 30000ull);                 // 300 us
            hipExtLaunchKernelGGL(quick, dim3(NB), dim3(256), 0, st, nullptr, nullptr, flags, sb);
            hipLaunchKernelGGL(quick, dim3(1), dim3(64), 0, st, sb);                            // an ordinary launch behind both
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(ha, sa, sizeof(ha), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hb, sb, sizeof(hb), hipMemcpyDeviceToHost));
            unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull;
            for (int i = 0; i < NA; ++i) { if (ha[2 * i] < a0) a0 = ha[2 * i]; if (ha[2 * i + 1] > a1) a1 = ha[2 * i + 1]; }
            for (int i = 1; i < NB; ++i) if (hb[2 * i] < b0) b0 = hb[2 * i];
            printf("flags %d rep %d: kernel A runs %.1f us; kernel B's first workgroup starts %+.1f us after A's start; the ordinary launch behind them starts at %+.1f us\n",
                   flags, rep, (double)(a1 - a0) / 100.0, ((double)b0 - (double)a0) / 100.0, ((double)hb[0] - (double)a0) / 100.0);
        }
    }
    return 0;
}
