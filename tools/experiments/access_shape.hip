// access_shape.hip -- stand-alone probe (round 6): what a global store / load costs as a function of the SHAPE of a wave's access, at equal bytes.
// A workgroup of 4 waves owns a tile of ROWS rows x 256 bytes inside rows of `pitch` bytes (the other part of each row belongs to another
// workgroup, as with a 16-column slice of a wider tensor); every byte of the buffer is written (read) exactly once.  Variants differ only in
// which lane touches which 8 / 16 bytes:
//   seg32   b64 : a wave instruction = 16 rows x 32 bytes   (an MFMA accumulator tile of 16 columns stored as it comes: gdr_readout_kernel, round 5)
//   seg64   b128: 16 rows x 64 bytes                         (conv3x3_tile / conv3x3_c64 epilogues: 8 channels per lane, 4 lanes per pixel)
//   seg128  b128: 8 rows x 128 bytes
//   seg256  b128: 4 rows x 256 bytes                         (gdr_readout_rows_kernel)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/_build/access_shape tools/experiments/access_shape.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 64;      // rows per workgroup tile

// mode: 0 seg32, 1 seg64, 2 seg128, 3 seg256.   LOAD: read instead of write (sum kept alive through a never-true store)
template <int MODE, bool LOAD>
__global__ __launch_bounds__(256) void shape_kernel(char* buf, int pitch, int tiles_per_row, unsigned long long* sink)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
    const size_t tile = blockIdx.x;
    const size_t trow = tile / tiles_per_row, tcol = tile % tiles_per_row;
    char* base = buf + trow * ROWS * (size_t)pitch + tcol * 256;
    unsigned long long acc = 0;
    if constexpr (MODE == 0) {
#pragma unroll
        for (int t = 0; t < ROWS / 16; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                uint2* p = reinterpret_cast<uint2*>(base + (size_t)(16 * t + li) * pitch + (2 * w + c) * 32 + 8 * g);
                if constexpr (LOAD) { const uint2 v = *p; acc += v.x + v.y; } else *p = make_uint2(tid, t);
            }
    } else {
        constexpr int SEG = MODE == 1 ? 64 : MODE == 2 ? 128 : 256;
        constexpr int LPR = SEG / 16;                      // lanes per row piece
        constexpr int RPI = 64 / LPR;                      // rows per instruction
        constexpr int PPR = 256 / SEG;                     // pieces per row (covered by different waves / instructions)
        // instruction index i covers (row group, piece): the workgroup's 4 waves x NI instructions cover ROWS x PPR pieces
        constexpr int NI = ROWS * PPR / RPI / 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int unit = i * 4 + w;                    // 0 .. ROWS*PPR/RPI - 1
            const int piece = unit % PPR, rg = unit / PPR;
            const int row = rg * RPI + lane / LPR, off = piece * SEG + (lane % LPR) * 16;
            uint4* p = reinterpret_cast<uint4*>(base + (size_t)row * pitch + off);
            if constexpr (LOAD) { const uint4 v = *p; acc += v.x + v.y + v.z + v.w; } else *p = make_uint4(tid, i, li, g);
        }
    }
    if (LOAD && acc == 0x123456789abcdefull) sink[0] = acc;
}

template <int MODE, bool LOAD>
float run(char* buf, size_t bytes, int pitch, unsigned long long* sink, int iters)
{
    const int tiles_per_row = pitch / 256;
    const size_t tiles = bytes / (256 * ROWS);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((shape_kernel<MODE, LOAD>), dim3((unsigned)tiles), dim3(256), 0, 0, buf, pitch, tiles_per_row, sink);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((shape_kernel<MODE, LOAD>), dim3((unsigned)tiles), dim3(256), 0, 0, buf, pitch, tiles_per_row, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main()
{
    const size_t bytes = (size_t)512 << 20;                // 512 MB: beyond the 256 MB Infinity Cache
    char* buf; unsigned long long* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 1, bytes));
    for (int pitch : {256, 512, 1024}) {
        printf("row pitch %4d bytes (a workgroup owns 256 of them), %zu MB, us per pass / TB/s:\n", pitch, bytes >> 20);
        float t;
        t = run<0, false>(buf, bytes, pitch, sink, 10); printf("  store seg32  b64   %8.1f us  %5.2f TB/s\n", t, bytes / t / 1e6);
        t = run<1, false>(buf, bytes, pitch, sink, 10); printf("  store seg64  b128  %8.1f us  %5.2f TB/s\n", t, bytes / t / 1e6);
        t = run<2, false>(buf, bytes, pitch, sink, 10); printf("  store seg128 b128  %8.1f us  %5.2f TB/s\n", t, bytes / t / 1e6);
        t = run<3, false>(buf, bytes, pitch, sink, 10); printf("  store seg256 b128  %8.1f us  %5.2f TB/s\n", t, bytes / t / 1e6);
    }
    return 0;
}
