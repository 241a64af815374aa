// xcd_flag_probe.hip -- can a kernel on one stream consume, frame by frame, what a kernel running AT THE SAME TIME on another stream
// produces, across XCDs (each XCD has its own L2), without fences?  Diagnostic for csrc/gdr_pipeline.hip.
//   producer (stream A): NP workgroups, each writes chunks of CHUNK bytes (values derived from epoch and index), waits for its stores,
//                        and adds 1 to the chunk's flag (agent-scope relaxed atomic).
//   consumer (stream B): NC workgroups, each walks ALL chunks in order: polls the flag (agent-scope atomic load), reads the chunk, checks it.
// The same buffers are re-used over several epochs with different values, so a stale line in a consumer XCD's L2 shows as a mismatch.
// Every spin loop gives up after a bounded number of polls (the kernels always drain).
//   hipcc --offload-arch=gfx950 -O3 -o xcd_flag_probe xcd_flag_probe.hip && ./xcd_flag_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK_Q = 16 * 1024 / 16;        // 16-byte units per chunk

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ unsigned long long rt() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

// mode bit 0: producer stores carry sc1 (agent scope: write-through); bit 1: consumer data loads carry sc1
__global__ __launch_bounds__(256) void producer(u32x4* data, unsigned* flags, int nchunk, unsigned epoch, int mode, int work, unsigned long long* stamps)
{
    const int tid = threadIdx.x;
    if (tid == 0) stamps[2 * blockIdx.x] = rt();
    for (int idx = blockIdx.x; idx < nchunk; idx += gridDim.x) {
        float acc = (float)idx;
        for (int i = 0; i < work; ++i) acc = acc * 1.0001f + 0.5f;             // stand-in for the fold's arithmetic
        const unsigned salt = acc > 1e30f ? 1u : 0u;
        u32x4* dst = data + (size_t)idx * CHUNK_Q;
        for (int i = tid; i < CHUNK_Q; i += 256) {
            const u32x4 v = {epoch * 0x9e3779b9u + (unsigned)idx, (unsigned)i + salt, epoch, (unsigned)idx ^ (unsigned)i};
            if (mode & 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + i), "v"(v) : "memory");
            else dst[i] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(flags + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) stamps[2 * blockIdx.x + 1] = rt();
}

__global__ __launch_bounds__(256) void consumer(const u32x4* data, unsigned* flags, int nchunk, unsigned epoch, int mode, unsigned* result, unsigned long long* stamps)
{
    const int tid = threadIdx.x;
    unsigned bad = 0, timeouts = 0;
    if (tid == 0) stamps[2 * blockIdx.x] = rt();
    for (int idx = 0; idx < nchunk; ++idx) {
        int spins = 0;
        while (__hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && spins < 2000000) { __builtin_amdgcn_s_sleep(2); ++spins; }
        if (spins >= 2000000) ++timeouts;
        const u32x4* src = data + (size_t)idx * CHUNK_Q;
        for (int i = tid; i < CHUNK_Q; i += 256) {
            u32x4 v;
            if (mode & 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src + i) : "memory");
            else v = src[i];
            if (v[0] != epoch * 0x9e3779b9u + (unsigned)idx || v[1] != (unsigned)i || v[2] != epoch || v[3] != ((unsigned)idx ^ (unsigned)i)) ++bad;
        }
    }
    if (tid == 0) stamps[2 * blockIdx.x + 1] = rt();
    atomicAdd(result, bad);
    if (tid == 0) { atomicAdd(result + 1, timeouts); atomicOr(result + 2, 1u << xcc_id()); }
}

int main()
{
    const int nchunk = 2048, NP = 512, NC = 32;
    u32x4* data; unsigned* flags; unsigned* result; unsigned long long *sp, *sc;
    CHECK(hipMalloc(&data, (size_t)nchunk * CHUNK_Q * sizeof(u32x4)));
    CHECK(hipMalloc(&flags, nchunk * sizeof(unsigned)));
    CHECK(hipMalloc(&result, 4 * sizeof(unsigned)));
    CHECK(hipMalloc(&sp, 2 * NP * sizeof(unsigned long long)));
    CHECK(hipMalloc(&sc, 2 * NC * sizeof(unsigned long long)));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    unsigned long long hp[2 * NP], hc[2 * NC];
    for (int work : {0, 4000}) {
        for (int mode : {0, 1, 2, 3}) {
            CHECK(hipMemset(flags, 0, nchunk * sizeof(unsigned)));
            CHECK(hipMemset(result, 0, 4 * sizeof(unsigned)));
            CHECK(hipDeviceSynchronize());
            float best = 1e9f;
            double c_first = 0, p_last = 0, c_last = 0;
            for (unsigned epoch = 1; epoch <= 6; ++epoch) {     // flags count up: epoch e waits for the value e
                CHECK(hipEventRecord(e0, sa));
                CHECK(hipStreamWaitEvent(sb, e0, 0));
                hipLaunchKernelGGL(consumer, dim3(NC), dim3(256), 0, sb, data, flags, nchunk, epoch, mode, result, sc);
                hipLaunchKernelGGL(producer, dim3(NP), dim3(256), 0, sa, data, flags, nchunk, epoch, mode, work, sp);
                CHECK(hipEventRecord(e2, sb));
                CHECK(hipStreamWaitEvent(sa, e2, 0));
                CHECK(hipEventRecord(e1, sa));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (epoch > 1 && ms < best) best = ms;
                CHECK(hipMemcpy(hp, sp, sizeof(hp), hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(hc, sc, sizeof(hc), hipMemcpyDeviceToHost));
                unsigned long long p0 = ~0ull, p1 = 0, c0 = ~0ull, c1 = 0;
                for (int i = 0; i < NP; ++i) { if (hp[2 * i] < p0) p0 = hp[2 * i]; if (hp[2 * i + 1] > p1) p1 = hp[2 * i + 1]; }
                for (int i = 0; i < NC; ++i) { if (hc[2 * i] < c0) c0 = hc[2 * i]; if (hc[2 * i + 1] > c1) c1 = hc[2 * i + 1]; }
                c_first = ((double)c0 - (double)p0) / 100.0; p_last = (double)(p1 - p0) / 100.0; c_last = ((double)c1 - (double)p0) / 100.0;   // 100 MHz counter -> us
            }
            unsigned hr[4];
            CHECK(hipMemcpy(hr, result, sizeof(hr), hipMemcpyDeviceToHost));
            printf("work %5d  producer sc1 %d  consumer sc1 %d : best %7.1f us   mismatching 16-byte units %u   timeouts %u   consumer XCD mask 0x%x   "
                   "(last epoch: consumer starts %+.1f us after the producer, producer ends %.1f, consumer ends %.1f)\n",
                   work, mode & 1, (mode >> 1) & 1, best * 1e3f, hr[0], hr[1], hr[2], c_first, p_last, c_last);
        }
    }
    // the producer alone and the consumer alone on ready flags: the two serial pieces
    return 0;
}
