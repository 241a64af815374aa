// gdr_segmented.hip -- SURVEY.md §8f row n3 inside one GPU, at the C ABI: gdkvm_scan_fwd_segmented.
//
// A long clip on few clips x heads x column slices leaves most CUs idle in the serial recurrence (cfg5: 2 clips x 16 slices = 32
// workgroups on 256 CUs).  The time axis is cut into S equal segments that run concurrently: the clip viewed as B*S clips of T/S
// frames is prepared once (gdkvm_scan_prep), every segment's transition matrix Phi_c (gdkvm_scan_transition) and zero-start end
// state S_loc_c (gdkvm_scan_apply without a read-out) are computed in parallel, one small kernel stitches the true start states
// (gdkvm_scan_stitch: start_{c+1} = Phi_c start_c + S_loc_c, exact fp32), and every segment is scanned again from its start state
// with the read-out.  2.25x the recurrence work on S times the workgroups: cfg5 269 us against 346 us serial (round 3).
// NOT bit-identical to gdkvm_scan_fwd (fp32 re-association through Phi), which is why gdkvm_scan_fwd -- whose contract is "a clip
// processed as consecutive calls is bit-identical to one call" -- never switches to this path by itself.
#include "gdkvm_common.hpp"
#include "gdr_ws.hpp"

namespace {

inline size_t seg_up256(size_t x) { return (x + 255) & ~(size_t)255; }

// S = 0: the smallest power of two that divides T, leaves segments of at least 8 frames and fills the device twice over with
// (clip, head, 16-column slice, segment) workgroups; 1 (= serial) when the serial grid keeps half the CUs busy or fewer than 4
// segments come out.
int seg_resolve(int B, int T, int Hh, int Dv, int segments)
{
    if (segments > 0) return segments;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    }
    const long serial = (long)B * Hh * (Dv / 16);
    if (2 * serial >= cus) return 1;                  // the serial grid keeps at least half the CUs busy
    int s = 1;
    while (serial * s < 2L * cus && T % (2 * s) == 0 && T / (2 * s) >= 8) s *= 2;
    return s >= 4 ? s : 1;                            // 2.25x the recurrence work: fewer than 4 segments cannot pay
}

struct SegView { char* ws; size_t ws_bytes; float* phi; float* s_loc; float* starts; size_t total; };

SegView seg_carve(void* base, int B, int T, int Hh, int N, int Dk, int Dv, int S)
{
    SegView v{};
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = base ? static_cast<char*>(base) + off : nullptr; off += seg_up256(bytes); return r; };
    v.ws_bytes = gdkvm_scan_workspace_bytes(B * S, T / S, Hh, N, Dk, Dv);
    v.ws = take(v.ws_bytes);
    const size_t BH = (size_t)B * S * Hh;
    v.phi = reinterpret_cast<float*>(take(BH * Dk * Dk * sizeof(float)));
    v.s_loc = reinterpret_cast<float*>(take(BH * Dk * Dv * sizeof(float)));
    v.starts = reinterpret_cast<float*>(take(BH * Dk * Dv * sizeof(float)));
    v.total = off;
    return v;
}

}  // namespace

extern "C" int gdkvm_scan_segments(int B, int T, int Hh, int Dv, int segments)
{
    if (B <= 0 || T <= 0 || Hh <= 0 || Dv <= 0 || segments < 0) return 1;
    const int s = seg_resolve(B, T, Hh, Dv, segments);
    return (s < 1 || T % s) ? 1 : s;
}

extern "C" size_t gdkvm_scan_segmented_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv, int segments)
{
    if (B <= 0 || T <= 0 || Hh <= 0 || N <= 0 || Dk <= 0 || Dv <= 0) return 256;
    const int S = gdkvm_scan_segments(B, T, Hh, Dv, segments);
    if (S == 1) return gdkvm_scan_workspace_bytes(B, T, Hh, N, Dk, Dv);
    return seg_carve(nullptr, B, T, Hh, N, Dk, Dv, S).total + 256;
}

extern "C" int gdkvm_scan_fwd_segmented(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                                        const float* s_in, void* r_out, float* s_out, void* workspace, size_t workspace_bytes,
                                        int B, int T, int Hh, int N, int Dk, int Dv, int segments,
                                        int io_dtype, int rule, int flags, void* stream)
{
    if (int rc = check_common("scan_fwd_segmented", B, T, Hh, N, Dk, Dv, io_dtype, flags)) return rc;
    if (segments < 0 || (segments > 0 && T % segments))
        return gdkvm_fail(GDKVM_ERR_SHAPE, "scan_fwd_segmented: segments=%d must divide T=%d (0 = chosen by shape)", segments, T);
    const int S = (B > 0 && T > 0) ? gdkvm_scan_segments(B, T, Hh, Dv, segments) : 1;
    if (S == 1 || N == 0)
        return gdkvm_scan_fwd(q, k, v, alpha, beta, s_in, r_out, s_out, nullptr, workspace, workspace_bytes, B, T, Hh, N, Dk, Dv, io_dtype, rule, flags, stream);
    if (int rc = check_ptrs("scan_fwd_segmented", {q, k, v, alpha, beta, r_out, workspace}, {s_in, s_out})) return rc;
    const SegView sv = seg_carve(workspace, B, T, Hh, N, Dk, Dv, S);
    if (workspace_bytes < sv.total) return gdkvm_fail(GDKVM_ERR_WORKSPACE, "scan_fwd_segmented: workspace %zu < %zu bytes", workspace_bytes, sv.total);
    if (rule == GDKVM_RULE_DELTA_PARALLEL) flags |= GDKVM_FLAG_WIDE_RANGE;      // as gdkvm_scan_fwd
    const int BS = B * S, Ts = T / S;                                              // [B, T, ...] viewed as [B*S, T/S, ...]
    if (int rc = gdkvm_scan_prep(q, k, v, beta, sv.ws, sv.ws_bytes, BS, Ts, Hh, N, Dk, Dv, io_dtype, rule, flags, stream)) return rc;
    if (int rc = gdkvm_scan_transition(q, alpha, sv.phi, sv.ws, sv.ws_bytes, BS, Ts, Hh, N, Dk, Dv, io_dtype, flags, stream)) return rc;
    if (int rc = gdkvm_scan_apply(q, alpha, nullptr, nullptr, sv.s_loc, nullptr, sv.ws, sv.ws_bytes, BS, Ts, Hh, N, Dk, Dv, io_dtype, flags, stream)) return rc;
    if (int rc = gdkvm_scan_stitch(sv.phi, sv.s_loc, s_in, sv.starts, s_out, B, S, Hh, Dk, Dv, stream)) return rc;
    return gdkvm_scan_apply(q, alpha, sv.starts, r_out, nullptr, nullptr, sv.ws, sv.ws_bytes, BS, Ts, Hh, N, Dk, Dv, io_dtype, flags, stream);
}
