// fence_probe.hip -- what do agent-scope release / acquire cost inside one kernel on gfx950?  Producer blocks (low block ids) write
// a frame's worth of operands (80 KB), fence and raise a flag; consumer blocks (high block ids) wait for the flag of frame t and
// read 20 KB of it, frame after frame -- the traffic pattern of a fused prep + scan launch.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -o fence_probe fence_probe.hip && ./fence_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int FRAME_F4 = 80 * 1024 / 16;     // float4 per frame-head
constexpr int READ_F4 = 20 * 1024 / 16;      // what one consumer reads of a frame

// mode bit 2: producer stores are device-scope write-through (sc1), no fence;  bit 0: producer does __threadfence() before the flag; bit 1: consumer flag load is an acquire (else relaxed + no fence)
__global__ __launch_bounds__(256) void probe(float4* data, unsigned* flags, float* sink, int nprod, int BH, int T, int nsl, int mode, int work)
{
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nprod) {
        // producer: frames in time-major order, idx = t * BH + bh
        for (int idx = blockIdx.x; idx < T * BH; idx += nprod) {
            float acc = (float)idx;
            for (int i = 0; i < work; ++i) acc = acc * 1.0001f + 0.5f;         // stand-in for the solve
            float4* dst = data + (size_t)idx * FRAME_F4;
            if (mode & 4) {                        // device-scope (write-through) stores instead of a release fence
                for (int i = tid; i < FRAME_F4; i += 256) {
                    typedef float vf4 __attribute__((ext_vector_type(4)));
                    const vf4 v = {acc, (float)i, 0.f, 1.f};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + i), "v"(v) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else
            for (int i = tid; i < FRAME_F4; i += 256) dst[i] = make_float4(acc, (float)i, 0.f, 1.f);
            if (mode & 1) __threadfence();
            __syncthreads();
            if (tid == 0) {
                if (mode & 1) __hip_atomic_store(flags + idx, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(flags + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    const int c = blockIdx.x - nprod, bh = c / nsl, sl = c % nsl;
    float s = 0.f;
    for (int t = 0; t < T; ++t) {
        const int idx = t * BH + bh;
        if (mode & 2) { while (__hip_atomic_load(flags + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1); }
        else { while (__hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1); }
        const float4* src = data + (size_t)idx * FRAME_F4 + (sl % 4) * READ_F4;
        for (int i = tid; i < READ_F4; i += 256) { const float4 v = src[i]; s += v.x + v.w; }
        __syncthreads();
    }
    sink[(size_t)c * 256 + tid] = s;
}

int main()
{
    const int BH = 16, T = 32, nsl = 16, nprod = 256, ncons = BH * nsl;
    float4* data; unsigned* flags; float* sink;
    CHECK(hipMalloc(&data, (size_t)BH * T * FRAME_F4 * sizeof(float4)));
    CHECK(hipMalloc(&flags, BH * T * sizeof(unsigned)));
    CHECK(hipMalloc(&sink, (size_t)ncons * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int work : {0, 20000}) {
        for (int mode : {0, 1, 2, 3, 4}) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipMemsetAsync(flags, 0, BH * T * sizeof(unsigned), 0));
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(probe, dim3(nprod + ncons), dim3(256), 0, 0, data, flags, sink, nprod, BH, T, nsl, mode, work);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("work %6d  producer fence %d  sc1 stores %d  consumer acquire %d : %8.1f us\n", work, mode & 1, (mode >> 2) & 1, (mode >> 1) & 1, best * 1e3f);
        }
    }
    // producers alone, and consumers alone on ready flags: the two serial pieces
    return 0;
}
