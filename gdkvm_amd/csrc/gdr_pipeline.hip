// gdr_pipeline.hip -- gdkvm_scan_fwd as overlapping time blocks ("GDR memory-state carry across chunks",
// BASELINE.json:11).
//
// One gdkvm_scan_fwd is three kinds of work: the frame-parallel fold (gdr_prepm_kernel [+ gdr_compose_kernel]: fills the device, knows
// nothing of the state), the serial recurrence (gdr_affine_scan_kernel: one workgroup per clip-head and 16-column slice -- 32 of 256
// CUs at 2 clips x 256 columns) and, for frames of more than 64 tokens, the frame-parallel read-out (gdr_readout_kernel).  Run one
// after the other the call costs their SUM while most of the device idles under the recurrence.  Here the call's frames are cut into
// time blocks and the three run as a pipeline on three streams:
//
//     helper stream P   prep(0) prep(1) prep(2) ...
//     caller's stream         scan(0) scan(1) scan(2) ...          scan(c) waits for prep(c); the state is carried block to block
//     helper stream R                 read(0) read(1) ...          read(c) waits for scan(c)
//
// Every kernel is the one the plain sequence launches, on a window of the same tensors and the same workspace regions (gdr_ws.hpp:
// gdr_ws_window; kernels stride clips by the whole clip's length), and a clip processed block by block with the state carried is
// bit-identical to one pass by gdkvm_scan_fwd's own chunking contract -- per frame the same operations in the same order.  The helper
// streams fork from and join back into the caller's stream through events, so the call stays stream-ordered for the caller (and
// capturable in a graph once the helper streams exist).
#include <stdlib.h>
#include <string.h>

#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

struct PipeRes {
    bool tried = false, ok = false;
    hipStream_t sp = nullptr, sr = nullptr, ss = nullptr;   // ss: the serial kernel of the concurrent form (highest priority)
    hipEvent_t fork = nullptr, join_r = nullptr, join_s = nullptr, evp[GDR_MAX_BLOCKS] = {}, evs[GDR_MAX_BLOCKS] = {};
};

// helper streams and events per host thread and device (calls on different host threads never share them), created on first use and
// kept for the life of the process.  Lowest priority: where a CU could take a workgroup of either, the recurrence goes first.
PipeRes* pipe_res(hipStream_t user)
{
    constexpr int MAXDEV = 16;
    thread_local PipeRes res[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    PipeRes& r = res[dev];
    if (r.tried) return r.ok ? &r : nullptr;
    // (never created inside a capture: the first captured call runs the plain sequence)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(user, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    r.tried = true;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    bool ok = hipStreamCreateWithPriority(&r.sp, hipStreamNonBlocking, least) == hipSuccess
              && hipStreamCreateWithPriority(&r.sr, hipStreamNonBlocking, least) == hipSuccess
              && hipStreamCreateWithPriority(&r.ss, hipStreamNonBlocking, greatest) == hipSuccess
              && hipEventCreateWithFlags(&r.join_s, hipEventDisableTiming) == hipSuccess
              && hipEventCreateWithFlags(&r.fork, hipEventDisableTiming) == hipSuccess
              && hipEventCreateWithFlags(&r.join_r, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < GDR_MAX_BLOCKS; ++i)
        ok = hipEventCreateWithFlags(&r.evp[i], hipEventDisableTiming) == hipSuccess
             && hipEventCreateWithFlags(&r.evs[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    r.ok = ok;
    return ok ? &r : nullptr;
}

// Block boundaries 0 = b[0] < b[1] < ... < b[C] = T.  GDKVM_SCAN_BLOCKS = n: n equal blocks (0 / 1: the plain sequence);
// GDKVM_SCAN_BLOCK_LIST = "t1,t2,...": explicit interior boundaries (experiments).  Without either: by shape (scan_blocks_auto).
int scan_blocks_auto(int B, int T, int Hh, int N, int Dv, int cus)
{
    (void)B; (void)T; (void)Hh; (void)N; (void)Dv; (void)cus;
    return 1;
}

int plan_blocks(int B, int T, int Hh, int N, int Dv, int* bnd)
{
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    }
    if (const char* e = getenv("GDKVM_SCAN_BLOCK_LIST")) {
        int C = 0;
        bnd[0] = 0;
        const char* p = e;
        while (*p && C < GDR_MAX_BLOCKS - 1) {
            char* end = nullptr;
            const long t = strtol(p, &end, 10);
            if (end == p) break;
            if (t > bnd[C] && t < T) bnd[++C] = (int)t;
            p = *end == ',' ? end + 1 : end;
        }
        bnd[++C] = T;
        return C;
    }
    int C = scan_blocks_auto(B, T, Hh, N, Dv, cus);
    if (const char* e = getenv("GDKVM_SCAN_BLOCKS")) C = atoi(e);
    C = C < 1 ? 1 : (C > GDR_MAX_BLOCKS ? GDR_MAX_BLOCKS : C);
    if (C > T) C = T;
    for (int c = 0; c <= C; ++c) bnd[c] = (int)((long)T * c / C);
    return C;
}

#define PIPE_HIP(call)                                                                                       \
    do {                                                                                                     \
        hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess) return gdkvm_fail(GDKVM_ERR_LAUNCH, "scan_fwd blocks: %s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

}  // namespace

int gdr_scan_fwd_blocks(const void* q, const void* k, const void* v, const float* alpha, const float* beta, const float* norms,
                        const float* s_in, void* r_out, float* s_out, void* workspace, size_t workspace_bytes,
                        int B, int T, int Hh, int N, int Dv, int io_dtype, int rule, int flags, hipStream_t st)
{
    if (B <= 0 || T < 2 || N <= 0 || Hh <= 0 || Dv <= 0 || (flags & GDKVM_FLAG_TRAIN)) return 1;
    int bnd[GDR_MAX_BLOCKS + 1];
    const int C = plan_blocks(B, T, Hh, N, Dv, bnd);
    if (C < 2) return 1;
    // from here on the call is ours: the plain sequence's own checks first (it reports them under its own names)
    if (int rc = check_common("scan_fwd", B, T, Hh, N, GDKVM_DK, Dv, io_dtype, flags)) return rc;
    if (rule < 0 || rule > 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd: rule=%d", rule);
    if (norms && (!(flags & GDKVM_FLAG_NORMALIZE_QK) || !gdkvm_aligned16(norms)))
        return gdkvm_fail(GDKVM_ERR_ARG, "scan_fwd_normed: norms go with GDKVM_FLAG_NORMALIZE_QK, 16-byte aligned");
    if (int rc = check_ptrs("scan_fwd", {q, k, v, alpha, beta, workspace}, {s_in, r_out, s_out})) return rc;
    WsView ws;
    if (int rc = carve("scan_fwd", workspace, workspace_bytes, B, T, Hh, N, GDKVM_DK, Dv, &ws)) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    PipeRes* pr = pipe_res(st);
    if (!pr) return 1;

    const size_t es = io_dtype == GDKVM_F32 ? 4 : 2;
    const size_t rowk = (size_t)N * Hh * GDKVM_DK * es, rowv = (size_t)N * Hh * Dv * es;       // bytes of one frame of q / k and of v / r
    const bool defer = ws.nb > 4 && r_out != nullptr;     // (as gdr_apply_window decides)
    int fuse = GDR_FUSE_AUTO;                              // frames of > 64 tokens: the variant the WHOLE call would take -- other blocks fill the device
    {
        int cus = 256, dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        fuse = (long)B * T * Hh >= cus;
        if (const char* e = getenv("GDKVM_SCAN_BLOCK_FUSE")) fuse = atoi(e);
    }
    float* carry = s_out ? s_out : ws.carry;

    PIPE_HIP(hipEventRecord(pr->fork, st));
    PIPE_HIP(hipStreamWaitEvent(pr->sp, pr->fork, 0));
    if (defer) PIPE_HIP(hipStreamWaitEvent(pr->sr, pr->fork, 0));
    auto at = [](const void* p, size_t bytes) { return p ? static_cast<const char*>(p) + bytes : nullptr; };
    auto prep = [&](int c) -> int {
        const int t0 = bnd[c], Tb = bnd[c + 1] - t0;
        const WsView w = gdr_ws_window(ws, B, Hh, N, Dv, t0, c);
        if (int rc = gdr_prep_window(at(q, t0 * rowk), at(k, t0 * rowk), at(v, t0 * rowv), beta + (size_t)t0 * N * Hh,
                                     norms ? norms + (size_t)t0 * N * Hh * 2 : nullptr, w, B, Tb, T, Hh, N, Dv, io_dtype, rule, flags,
                                     (c == 0 && fuse == 1 && getenv("GDKVM_SCAN_BLOCK_FIRST_UNFUSED")) ? 0 : fuse, pr->sp)) return rc;
        PIPE_HIP(hipEventRecord(pr->evp[c], pr->sp));
        return GDKVM_OK;
    };
    if (int rc = prep(0)) return rc;
    for (int c = 0; c < C; ++c) {
        if (c + 1 < C) if (int rc = prep(c + 1)) return rc;
        const int t0 = bnd[c], Tb = bnd[c + 1] - t0;
        const WsView w = gdr_ws_window(ws, B, Hh, N, Dv, t0, c);
        const void* qc = at(q, t0 * rowk);
        void* rc_out = r_out ? static_cast<char*>(r_out) + t0 * rowv : nullptr;
        const float* al = alpha + (size_t)t0 * Hh;
        PIPE_HIP(hipStreamWaitEvent(st, pr->evp[c], 0));
        if (int rc = gdr_apply_window(qc, al, c == 0 ? s_in : carry, rc_out, c + 1 == C ? s_out : carry, nullptr, w,
                                      B, Tb, T, Hh, N, Dv, io_dtype, flags, 1, st)) return rc;
        if (defer) {
            PIPE_HIP(hipEventRecord(pr->evs[c], st));
            PIPE_HIP(hipStreamWaitEvent(pr->sr, pr->evs[c], 0));
            if (int rc = gdr_apply_window(qc, al, nullptr, rc_out, nullptr, nullptr, w, B, Tb, T, Hh, N, Dv, io_dtype, flags, 2, pr->sr)) return rc;
        }
    }
    if (defer) {
        PIPE_HIP(hipEventRecord(pr->join_r, pr->sr));
        PIPE_HIP(hipStreamWaitEvent(st, pr->join_r, 0));
    }
    return GDKVM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The concurrent form (frames of more than 64 tokens, fp16-pair operands).  Event dependencies between the stages cost ~8-10 us each on
// this system, more than any block of frames hides (profiles/r04_b_time_blocks_on_streams.txt), so here the stages are three kernels that
// RUN AT THE SAME TIME and hand frames over through counters in the workspace (gdr_device.hpp, "flags"):
//
//     caller's stream   [clear counters] fold, time-major ........................ read-out, time-major (waits per frame group on `prog`)
//     helper stream            serial recurrence (waits per frame group on `prep`, raises `prog`) ........|
//
// The fold never waits, the recurrence waits only for the fold, the read-out is launched behind the fold (stream order) and waits only for
// the recurrence, which is resident by then: no cycle, and every wait is bounded anyway.  One event forks the helper stream at the start
// (its latency hides behind the fold's first frames) and one joins it at the end (recorded behind the recurrence, which finishes before
// the read-out does).  Per frame every kernel does what the plain sequence's kernel does, in the same order: the results are
// bit-identical (tests/test_scan_gpu.py, tests/test_configs_gpu.py at cfg5's full size).
int gdr_scan_fwd_pipe(const void* q, const void* k, const void* v, const float* alpha, const float* beta, const float* norms,
                      const float* s_in, void* r_out, float* s_out, void* workspace, size_t workspace_bytes,
                      int B, int T, int Hh, int N, int Dv, int io_dtype, int rule, int flags, hipStream_t st)
{
    if (B <= 0 || T < 2 || N <= 64 || Hh <= 0 || Dv <= 0 || !r_out || (flags & (GDKVM_FLAG_TRAIN | GDKVM_FLAG_WIDE_RANGE))) return 1;
    // Opt-in (GDKVM_SCAN_PIPE=1), NOT taken by shape: measured at 2 x 512 frames of 256 tokens the form is bit-identical and no faster than
    // the plain sequence (358 against 346 us) -- the serial kernel is bound by what its CU can fetch per frame, and beside kernels that load
    // the memory system it slows from 390 to 460-750 ns per frame, which eats what the overlap saves (DESIGN.md §2.6,
    // profiles/r04_i_pipe_timeline_cfg5.txt, r04_j_scan_alone.txt).
    const char* take_e = getenv("GDKVM_SCAN_PIPE");
    if (!take_e || take_e[0] != '1') return 1;
    if (int rc = check_common("scan_fwd", B, T, Hh, N, GDKVM_DK, Dv, io_dtype, flags)) return rc;
    if (rule < 0 || rule > 2) return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd: rule=%d", rule);
    if (norms && (!(flags & GDKVM_FLAG_NORMALIZE_QK) || !gdkvm_aligned16(norms)))
        return gdkvm_fail(GDKVM_ERR_ARG, "scan_fwd_normed: norms go with GDKVM_FLAG_NORMALIZE_QK, 16-byte aligned");
    if (int rc = check_ptrs("scan_fwd", {q, k, v, alpha, beta, r_out, workspace}, {s_in, s_out})) return rc;
    WsView ws;
    if (int rc = carve("scan_fwd", workspace, workspace_bytes, B, T, Hh, N, GDKVM_DK, Dv, &ws)) return rc;
    if (int rc = gdkvm_check_device()) return rc;
    PipeRes* pr = pipe_res(st);
    if (!pr) return 1;
    ws.pipe.prep_per_frame = (unsigned)gdr_prep_producers(ws, B, T, Hh, Dv, io_dtype, rule, flags, GDR_FUSE_AUTO);
    if (const char* e = getenv("GDKVM_PIPE_DBG")) ws.pipe.dbg = atoi(e);
    // GDKVM_PIPE_STAGES (timing experiments; results are meaningless unless all three run): bit 0 the fold, bit 1 the recurrence, bit 2 the
    // read-out; the counters of a stage that is left out are preset as "done"
    int stages = 7;
    if (const char* e = getenv("GDKVM_PIPE_STAGES")) stages = atoi(e) & 7;
    const size_t fbytes = (size_t)B * Hh * ws.pipe.ngrp * sizeof(unsigned);
    if (stages == 7) PIPE_HIP(hipMemsetAsync(ws.pipe.prep, 0, 2 * fbytes, st));      // prep and prog (adjacent)
    else {
        PIPE_HIP(hipMemsetAsync(ws.pipe.prep, (stages & 1) ? 0 : 0x7f, fbytes, st));
        PIPE_HIP(hipMemsetAsync(ws.pipe.prog, (stages & 2) ? 0 : 0x7f, fbytes, st));
    }
    // the recurrence goes FIRST, on the caller's stream: its workgroups take their CUs while the device is empty (8 waves x 192 registers:
    // beside two workgroups of the fold -- 2 x 232 registers per SIMD -- one never fits, and with the fold launched first the recurrence
    // got its CUs only when the fold's last round drained: profiles/r04_e_pipe_probe.txt, S3 = the sum); the fold does not fit beside it
    // either, so the serial chain has its CUs to itself.  The frame-parallel kernels follow on the helper stream.
    PIPE_HIP(hipEventRecord(pr->fork, st));
    PIPE_HIP(hipStreamWaitEvent(pr->sp, pr->fork, 0));
    if (stages & 2)
        if (int rc = gdr_apply_window(q, alpha, s_in, r_out, s_out, nullptr, ws, B, T, T, Hh, N, Dv, io_dtype, flags, GDR_PHASE_SCAN | GDR_PHASE_PIPE, st)) return rc;
    if (stages & 1)
        if (int rc = gdr_prep_window(q, k, v, beta, norms, ws, B, T, T, Hh, N, Dv, io_dtype, rule, flags, GDR_FUSE_AUTO, pr->sp, true)) return rc;
    if (stages & 4)
        if (int rc = gdr_apply_window(q, alpha, nullptr, r_out, nullptr, nullptr, ws, B, T, T, Hh, N, Dv, io_dtype, flags, GDR_PHASE_READOUT | GDR_PHASE_PIPE, pr->sp)) return rc;
    PIPE_HIP(hipEventRecord(pr->join_r, pr->sp));
    PIPE_HIP(hipStreamWaitEvent(st, pr->join_r, 0));
    return GDKVM_OK;
}
