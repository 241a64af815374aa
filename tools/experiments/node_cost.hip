// node_cost.hip -- stand-alone probe (round 6): what ONE dependent kernel costs on this stack when the kernel itself is tiny -- the regime of the per-frame
// step mode (13 dependent launches per frame on 16 frames' worth of data) and of a one-clip request (24 launches for 0.34 ms).
//   a chain of N kernels (each: one workgroup adds 1 to 256 floats, or 256 workgroups doing the same on 64 K floats), timed as
//     (a) plain launches on one stream (the host loop is C++: ~2 us per launch, far below the device-side cost),
//     (b) one hipGraph of the N kernel nodes in a chain (stream capture),
//     (c) one hipGraph of TWO independent chains of N / 2 nodes (fork / join through a second stream).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/_build/node_cost tools/experiments/node_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void bump(float* p) { p[blockIdx.x * 256 + threadIdx.x] += 1.0f; }

static double now_us(hipEvent_t a, hipEvent_t b) { float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1e3; }

int main()
{
    const int N = 416;                                      // (the step mode's node count for 32 frames)
    float *x, *y;
    CK(hipMalloc(&x, 256 * 256 * sizeof(float))); CK(hipMalloc(&y, 256 * 256 * sizeof(float)));
    CK(hipMemset(x, 0, 256 * 256 * sizeof(float))); CK(hipMemset(y, 0, 256 * 256 * sizeof(float)));
    hipStream_t s, s2;
    CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, fork, join;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    for (int grid : {1, 256}) {
        // (a) plain launches
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, x);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        }
        const double t_plain = now_us(e0, e1) / N;
        // (b) one chain in a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, x);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); }
        const double t_graph = now_us(e0, e1) / N;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        // (c) two independent chains in one graph
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s2, fork, 0));
        for (int i = 0; i < N / 2; ++i) { hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, x); hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s2, y); }
        CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s, join, 0));
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); }
        const double t_two = now_us(e0, e1) / N;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        printf("%3d workgroup(s) per kernel, %d dependent kernels: plain launches %.2f us per kernel | one chain in a hipGraph %.2f | two chains of %d in one hipGraph %.2f us per kernel (of %d)\n",
               grid, N, t_plain, t_graph, N / 2, t_two, N);
    }
    return 0;
}
