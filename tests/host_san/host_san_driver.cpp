// host_san_driver.cpp -- the HOST halves of the HIP library under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not
// available on the pool).  Built by tests/host_san/build.sh (hipcc -fsanitize=address,undefined -fno-gpu-sanitize: host code instrumented,
// device code compiled as usual and never run) from the product's own sources:
// gdkvm_api.hip, gdr_segmented.hip, gdr_normalizer.hip, gdr_step.hip + csrc/gdr_ws.hpp; the entry points those files call into the
// kernel-heavy translation units (gdkvm_scan_fwd / _prep / _apply / _transition / _stitch) are stubbed here -- what runs is workspace
// carving, size arithmetic, argument checking and error reporting, with every workspace malloc'ed at EXACTLY the size the library asks
// for and every carved region written end to end, so a carve that walks past its own size request is a heap-buffer-overflow report.
// There is no GPU in this process: calls that get as far as the device check return GDKVM_ERR_ARCH, which is a checked outcome here.
// TEST INFRASTRUCTURE (tests/test_abi_cpu.py builds and runs it); exit code 0 = clean.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gdr_ws.hpp"

// ---- stubs of the kernel-launching entry points the composed calls reach (argument recording only) -------------------------------------
static int g_calls = 0;
extern "C" size_t gdkvm_scan_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv) { return gdr_workspace_bytes(B, T, Hh, N, Dk, Dv); }
extern "C" int gdkvm_scan_fwd(const void*, const void*, const void*, const float*, const float*, const float*, void*, float*, float*, void* ws, size_t wsb,
                              int B, int T, int Hh, int N, int Dk, int Dv, int, int, int, void*)
{
    ++g_calls;
    if (wsb < gdr_workspace_bytes(B, T, Hh, N, Dk, Dv)) return GDKVM_ERR_WORKSPACE;
    if (ws && wsb) { static_cast<char*>(ws)[0] = 1; static_cast<char*>(ws)[wsb - 1] = 1; }      // the inner workspace really is wsb bytes long
    return GDKVM_OK;
}
extern "C" int gdkvm_scan_prep(const void*, const void*, const void*, const float*, void* ws, size_t wsb, int B, int T, int Hh, int N, int Dk, int Dv, int, int, int, void*)
{
    ++g_calls;
    WsView v;
    if (int rc = carve("stub_prep", ws, wsb, B, T, Hh, N, Dk, Dv, &v)) return rc;
    memset(v.trash, 0, 1024);                                           // the LAST region of the layout: in bounds iff the whole carve is
    return GDKVM_OK;
}
extern "C" int gdkvm_scan_apply(const void*, const float*, const float*, void*, float*, float*, const void*, size_t, int, int, int, int, int, int, int, int, void*) { ++g_calls; return GDKVM_OK; }
extern "C" int gdkvm_scan_transition(const void*, const float*, float* phi, const void*, size_t, int B, int, int Hh, int, int Dk, int, int, int, void*)
{
    ++g_calls;
    memset(phi, 0, sizeof(float) * (size_t)B * Hh * Dk * Dk);           // phi_out is [B, Hh, Dk, Dk]: the segmented call must have carved that much
    return GDKVM_OK;
}
extern "C" int gdkvm_scan_stitch(const float*, const float*, const float*, float* starts, float*, int B, int S, int Hh, int Dk, int Dv, void*)
{
    ++g_calls;
    memset(starts, 0, sizeof(float) * (size_t)B * S * Hh * Dk * Dv);
    return GDKVM_OK;
}

// ---- the product's entry points under test ----------------------------------------------------------------------------------------------
extern "C" {
size_t gdkvm_scan_segmented_workspace_bytes(int, int, int, int, int, int, int);
int gdkvm_scan_segments(int, int, int, int, int);
int gdkvm_scan_fwd_segmented(const void*, const void*, const void*, const float*, const float*, const float*, void*, float*, void*, size_t,
                             int, int, int, int, int, int, int, int, int, int, void*);
size_t gdkvm_scan_normalizer_workspace_bytes(int, int, int, int, int, int, int);
int gdkvm_scan_fwd_normalizer(const void*, const void*, const void*, const float*, const float*, const float*, const float*, void*, float*, float*,
                              void*, size_t, int, int, int, int, int, int, int, int, int, float, void*);
int gdkvm_lkva_read(const void*, const float*, const float*, void*, int, int, int, int, int, int, int, void*);
int gdkvm_mask_embed_add(const uint8_t*, const float*, void*, int, int, int, int, int, int, int, void*);
}

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "host_san: line %d: %s   (last error: %s)\n", __LINE__, #cond, gdkvm_last_error()); ++fails; } } while (0)

int main()
{
    // 1. the scan workspace layout: every region of carve() lies inside gdr_workspace_bytes(), for frames below / at / above the 64-token
    //    chunk, several heads, narrow and wide values
    const int shapes[][6] = {{1, 1, 1, 1, 64, 16}, {2, 3, 1, 49, 64, 256}, {1, 2, 2, 65, 64, 48}, {2, 2, 1, 256, 64, 256}, {1, 1, 1, 1024, 64, 64}, {3, 5, 2, 64, 64, 32}};
    for (const auto& sh : shapes) {
        const size_t need = gdr_workspace_bytes(sh[0], sh[1], sh[2], sh[3], sh[4], sh[5]);
        char* buf = static_cast<char*>(malloc(need));
        WsView v;
        EXPECT(carve("carve", buf, need, sh[0], sh[1], sh[2], sh[3], sh[4], sh[5], &v) == GDKVM_OK);
        EXPECT(carve("carve", buf, need - 1, sh[0], sh[1], sh[2], sh[3], sh[4], sh[5], &v) == GDKVM_ERR_WORKSPACE);
        float* regions[] = {v.wt, v.knT, v.ut, v.kn, v.wtT, v.qnT, v.tii, v.wti, v.ppt, v.qinv, v.pp, v.gg, v.x0, v.ppc, v.ggc, v.simg, v.gmax, v.esc, v.zero};
        for (size_t i = 0; i + 1 < sizeof(regions) / sizeof(regions[0]); ++i) {
            EXPECT(regions[i] <= regions[i + 1]);                       // laid out in order, never overlapping backwards
            if (regions[i] < regions[i + 1]) memset(regions[i], 0, (size_t)(regions[i + 1] - regions[i]) * sizeof(float));
        }
        memset(v.zero, 0, 1024);
        memset(v.trash, 0, 1024);                                       // ends exactly at `need`
        EXPECT(v.trash + 1024 == buf + need);
        free(buf);
    }
    // 2. the segmented scan: its carve (inner workspace, phi, s_loc, starts) inside the size it reports; argument errors
    {
        const int B = 2, T = 64, Hh = 1, N = 70, Dk = 64, Dv = 32, S = 4;
        EXPECT(gdkvm_scan_segments(B, T, Hh, Dv, S) == S);
        EXPECT(gdkvm_scan_segments(B, T, Hh, Dv, 5) == 1);              // does not divide T
        const size_t need = gdkvm_scan_segmented_workspace_bytes(B, T, Hh, N, Dk, Dv, S);
        char* ws = static_cast<char*>(aligned_alloc(256, (need + 255) & ~(size_t)255));
        char* io = static_cast<char*>(aligned_alloc(256, 4096));
        g_calls = 0;
        EXPECT(gdkvm_scan_fwd_segmented(io, io, io, (float*)io, (float*)io, nullptr, io, (float*)io, ws, need, B, T, Hh, N, Dk, Dv, S, GDKVM_BF16, 2, 3, nullptr) == GDKVM_OK);
        EXPECT(g_calls == 5);                                           // prep, transition, apply, stitch, apply
        EXPECT(gdkvm_scan_fwd_segmented(io, io, io, (float*)io, (float*)io, nullptr, io, (float*)io, ws, need - 512, B, T, Hh, N, Dk, Dv, S, GDKVM_BF16, 2, 3, nullptr) == GDKVM_ERR_WORKSPACE);
        EXPECT(gdkvm_scan_fwd_segmented(io, io, io, (float*)io, (float*)io, nullptr, io, (float*)io, ws, need, B, T, Hh, N, Dk, Dv, 5, GDKVM_BF16, 2, 3, nullptr) == GDKVM_ERR_SHAPE);
        EXPECT(gdkvm_scan_fwd_segmented(io + 4, io, io, (float*)io, (float*)io, nullptr, io, (float*)io, ws, need, B, T, Hh, N, Dk, Dv, S, GDKVM_BF16, 2, 3, nullptr) == GDKVM_ERR_ARG);
        EXPECT(gdkvm_scan_fwd_segmented(io, io, io, (float*)io, (float*)io, nullptr, io, (float*)io, ws, need, B, T, Hh, N, 48, Dv, S, GDKVM_BF16, 2, 3, nullptr) == GDKVM_ERR_SHAPE);
        EXPECT(gdkvm_scan_fwd_segmented(io, io, io, (float*)io, (float*)io, nullptr, io, (float*)io, ws, need, B, T, Hh, N, Dk, Dv, S, 7, 2, 3, nullptr) == GDKVM_ERR_DTYPE);
        free(ws); free(io);
    }
    // 3. the normalizer call: argument checks in front of the device check; its size request covers the augmented problem
    {
        const int B = 2, T = 3, Hh = 1, N = 20, Dk = 64, Dv = 32;
        const size_t need = gdkvm_scan_normalizer_workspace_bytes(B, T, Hh, N, Dk, Dv, GDKVM_F32);
        EXPECT(need > gdr_workspace_bytes(B, T, Hh, N, Dk, Dv + 16) + 2 * (size_t)B * T * N * Hh * (Dv + 16) * 4);
        EXPECT(gdkvm_scan_normalizer_workspace_bytes(0, 0, 1, 0, 64, 16, GDKVM_BF16) >= 256);
        char* io = static_cast<char*>(aligned_alloc(256, 4096));
        EXPECT(gdkvm_scan_fwd_normalizer(io, io, io, (float*)io, (float*)io, nullptr, nullptr, io, nullptr, nullptr, io, need, B, T, Hh, N, Dk, Dv, GDKVM_F32, 2, 3, 0.f, nullptr) == GDKVM_ERR_SHAPE);   // eps
        EXPECT(gdkvm_scan_fwd_normalizer(io, io, io, (float*)io, (float*)io, nullptr, nullptr, io, nullptr, nullptr, io, need, B, T, Hh, N, Dk, Dv, GDKVM_F32, 2, 4, 1e-6f, nullptr) == GDKVM_ERR_SHAPE);  // GDKVM_FLAG_TRAIN
        EXPECT(gdkvm_scan_fwd_normalizer(io, io, io, (float*)io, (float*)io, nullptr, nullptr, io, nullptr, nullptr, io, need, B, T, Hh, N, Dk, Dv, GDKVM_F32, 9, 3, 1e-6f, nullptr) == GDKVM_ERR_SHAPE);  // rule
        EXPECT(gdkvm_scan_fwd_normalizer(nullptr, io, io, (float*)io, (float*)io, nullptr, nullptr, io, nullptr, nullptr, io, need, B, T, Hh, N, Dk, Dv, GDKVM_F32, 2, 3, 1e-6f, nullptr) == GDKVM_ERR_ARG);
        EXPECT(gdkvm_scan_fwd_normalizer(io, io, io, (float*)io, (float*)io, nullptr, (float*)(io + 4), io, nullptr, nullptr, io, need, B, T, Hh, N, Dk, Dv, GDKVM_F32, 2, 3, 1e-6f, nullptr) == GDKVM_ERR_ARG);
        EXPECT(gdkvm_scan_fwd_normalizer(io, io, io, (float*)io, (float*)io, nullptr, nullptr, io, nullptr, nullptr, io, need, B, T, Hh, N, Dk, 24, GDKVM_F32, 2, 3, 1e-6f, nullptr) == GDKVM_ERR_SHAPE);    // Dv % 16
        // valid arguments: reaches the device check -- no GPU here
        EXPECT(gdkvm_scan_fwd_normalizer(io, io, io, (float*)io, (float*)io, nullptr, nullptr, io, nullptr, nullptr, io, need, B, T, Hh, N, Dk, Dv, GDKVM_F32, 2, 3, 1e-6f, nullptr) == GDKVM_ERR_ARCH);
        EXPECT(strlen(gdkvm_last_error()) > 0);
        // 4. the step-mode kernels' argument checks
        EXPECT(gdkvm_lkva_read(io, nullptr, (float*)io, io, 2, 49, 1, 64, 256, GDKVM_BF16, 1, nullptr) == GDKVM_ERR_ARCH);
        EXPECT(gdkvm_lkva_read(io, nullptr, (float*)io, io, 0, 49, 1, 64, 256, GDKVM_BF16, 1, nullptr) == GDKVM_OK);            // nothing to do
        EXPECT(gdkvm_lkva_read(io, (float*)io, (float*)io, io, 2, 49, 1, 64, 256, GDKVM_BF16, 0, nullptr) == GDKVM_ERR_ARG);     // norms without the flag
        EXPECT(gdkvm_lkva_read(io, nullptr, nullptr, io, 2, 49, 1, 64, 256, GDKVM_BF16, 1, nullptr) == GDKVM_ERR_ARG);
        EXPECT(gdkvm_lkva_read(io, nullptr, (float*)io, io, 2, 49, 1, 32, 256, GDKVM_BF16, 1, nullptr) == GDKVM_ERR_SHAPE);
        EXPECT(gdkvm_mask_embed_add((uint8_t*)io, (float*)io, io, 2, 112, 112, 7, 7, 256, GDKVM_BF16, nullptr) == GDKVM_ERR_ARCH);
        EXPECT(gdkvm_mask_embed_add((uint8_t*)io, (float*)io, io, 2, 4, 4, 7, 7, 256, GDKVM_BF16, nullptr) == GDKVM_ERR_SHAPE);   // more tokens than pixels
        EXPECT(gdkvm_mask_embed_add(nullptr, (float*)io, io, 2, 112, 112, 7, 7, 256, GDKVM_BF16, nullptr) == GDKVM_ERR_ARG);
        EXPECT(gdkvm_mask_embed_add((uint8_t*)io + 3, (float*)io, io, 0, 112, 112, 7, 7, 256, GDKVM_BF16, nullptr) == GDKVM_OK); // masks are bytes: any alignment
        free(io);
    }
    if (fails) { fprintf(stderr, "host_san: %d check(s) failed\n", fails); return 1; }
    printf("host_san: ok\n");
    return 0;
}
