/* oracle_selftest.c -- a small driver of the scalar C oracle for sanitizer builds (make -C oracle SAN=1: AddressSanitizer + UBSan on the
 * CPU; GPU sanitizers are not available on the pool).  TEST INFRASTRUCTURE ONLY.  It walks every entry point of gdkvm_oracle.c over
 * ragged shapes (token counts off any tile edge, several heads, zero frames, carried state, every rule, the normalizer variant) with
 * exactly sized heap buffers -- an out-of-bounds index or a read of uninitialised scratch aborts under ASan -- and checks two analytic
 * known answers (SURVEY.md A.6 K1 recall, K12 normalised read-out) so that a silently wrong build fails too.  Exit code 0 = clean. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int gdkvm_oracle_scan_f64(const float*, const float*, const float*, const float*, const float*, const float*, float*, float*,
                          int, int, int, int, int, int, int, int);
int gdkvm_oracle_scan_f32(const float*, const float*, const float*, const float*, const float*, const float*, float*, float*,
                          int, int, int, int, int, int, int, int);
int gdkvm_oracle_scan_normalizer_f64(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                                     float*, float*, float*, int, int, int, int, int, int, int, int, double);
int gdkvm_oracle_scan_normalizer_f32(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                                     float*, float*, float*, int, int, int, int, int, int, int, int, double);
int gdkvm_oracle_kpff_f64(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                          int, int, int, int, int, int);
int gdkvm_oracle_kpff_f32(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                          int, int, int, int, int, int);
int gdkvm_oracle_argmax_dice(const float*, const uint8_t*, uint8_t*, int32_t*, int, int, int, int);
int gdkvm_oracle_upsample_argmax_dice(const float*, const uint8_t*, uint8_t*, int32_t*, int, int, int, int, int, int);

static uint32_t rs = 12345u;
static float rnd(void) { rs = rs * 1664525u + 1013904223u; return (float)((rs >> 8) & 0xffff) / 32768.0f - 1.0f; }
static float* buf(size_t n) { float* p = (float*)malloc(sizeof(float) * (n ? n : 1)); for (size_t i = 0; i < n; ++i) p[i] = rnd(); return p; }

static int fail(const char* what) { fprintf(stderr, "oracle_selftest: %s\n", what); return 1; }

int main(void)
{
    /* ragged sweeps, exact-size buffers */
    const int shapes[][6] = {{1, 2, 1, 1, 8, 4}, {2, 3, 5, 2, 8, 12}, {1, 1, 17, 1, 16, 20}, {3, 0, 4, 1, 8, 4}, {2, 2, 0, 1, 8, 4}};
    for (unsigned si = 0; si < sizeof(shapes) / sizeof(shapes[0]); ++si) {
        const int B = shapes[si][0], T = shapes[si][1], N = shapes[si][2], Hh = shapes[si][3], Dk = shapes[si][4], Dv = shapes[si][5];
        const size_t rows = (size_t)B * T * N * Hh;
        float *q = buf(rows * Dk), *k = buf(rows * Dk), *v = buf(rows * Dv), *a = buf((size_t)B * T * Hh), *b = buf(rows);
        float *s0 = buf((size_t)B * Hh * Dk * Dv), *z0 = buf((size_t)B * Hh * Dk);
        float *r = buf(rows * Dv), *s = buf((size_t)B * Hh * Dk * Dv), *z = buf((size_t)B * Hh * Dk);
        for (int rule = 0; rule < 3; ++rule)
            for (int flags = 0; flags < 4; ++flags) {
                if (gdkvm_oracle_scan_f64(q, k, v, a, b, flags & 1 ? s0 : NULL, r, s, B, T, N, Hh, Dk, Dv, rule, flags)) return fail("scan_f64 rc");
                if (gdkvm_oracle_scan_f32(q, k, v, a, b, s0, r, NULL, B, T, N, Hh, Dk, Dv, rule, flags)) return fail("scan_f32 rc");
                if (gdkvm_oracle_scan_normalizer_f64(q, k, v, a, b, s0, flags & 2 ? z0 : NULL, r, s, z, B, T, N, Hh, Dk, Dv, rule, flags, 1e-6))
                    return fail("scan_normalizer_f64 rc");
                if (gdkvm_oracle_scan_normalizer_f32(q, k, v, a, b, NULL, NULL, r, NULL, NULL, B, T, N, Hh, Dk, Dv, rule, flags, 1e-6))
                    return fail("scan_normalizer_f32 rc");
            }
        free(q); free(k); free(v); free(a); free(b); free(s0); free(z0); free(r); free(s); free(z);
    }
    {   /* KPFF: 5 x 3 tokens (cells of scale 2 and 4 cut by both edges) */
        const int BT = 2, Ck = 8, Cv = 12, Cp = 4, h = 5, w = 3, N = h * w, Cin = Cp + Ck + Cv;
        float *L = buf((size_t)BT * N * Ck), *G = buf((size_t)BT * N * Cv), *P = buf((size_t)BT * N * Cp), *Wa = buf((size_t)2 * Cp * Cin),
              *ba = buf(2 * Cp), *Wl = buf((size_t)Cp * Ck), *Wg = buf((size_t)Cp * Cv), *F = buf((size_t)BT * N * Cp);
        if (gdkvm_oracle_kpff_f64(L, G, P, Wa, ba, Wl, Wg, F, BT, Ck, Cv, Cp, h, w)) return fail("kpff_f64 rc");
        if (gdkvm_oracle_kpff_f32(L, G, P, Wa, ba, Wl, Wg, F, BT, Ck, Cv, Cp, h, w)) return fail("kpff_f32 rc");
        free(L); free(G); free(P); free(Wa); free(ba); free(Wl); free(Wg); free(F);
    }
    {   /* argmax / Dice, odd sizes, labels out of range */
        const int BT = 3, ncls = 3, hl = 3, wl = 5, H = 11, W = 19;
        float* lg = buf((size_t)BT * ncls * H * W);
        float* lo = buf((size_t)BT * ncls * hl * wl);
        uint8_t* tg = (uint8_t*)malloc((size_t)BT * H * W);
        uint8_t* mk = (uint8_t*)malloc((size_t)BT * H * W);
        int32_t* ct = (int32_t*)malloc(sizeof(int32_t) * BT * ncls * 3);
        for (int i = 0; i < BT * H * W; ++i) tg[i] = (uint8_t)(i % 5 == 4 ? 255 : i % 3);
        if (gdkvm_oracle_argmax_dice(lg, tg, mk, ct, BT, ncls, H, W)) return fail("argmax rc");
        if (gdkvm_oracle_argmax_dice(lg, NULL, mk, NULL, BT, ncls, H, W)) return fail("argmax (no target) rc");
        if (gdkvm_oracle_upsample_argmax_dice(lo, tg, mk, ct, BT, ncls, hl, wl, H, W)) return fail("upsample_argmax rc");
        free(lg); free(lo); free(tg); free(mk); free(ct);
    }
    {   /* K1 recall and K12 normalised read-out: two orthonormal keys, beta = alpha = 1; frame 1 reads with q = (k1 + k2) / sqrt 2 */
        enum { Dk = 8, Dv = 4, N = 2, T = 2 };
        float q[T * N * Dk] = {0}, k[T * N * Dk] = {0}, v[T * N * Dv] = {0}, a[T] = {1, 1}, b[T * N] = {1, 1, 1, 1};
        float r[T * N * Dv], s[Dk * Dv], z[Dk];
        k[0 * Dk + 1] = 1.f; k[1 * Dk + 4] = 1.f;                       /* frame 0: keys e1, e4 */
        for (int c = 0; c < Dv; ++c) { v[0 * Dv + c] = (float)(c + 1); v[1 * Dv + c] = (float)(10 * (c + 1)); }
        const float h2 = (float)sqrt(0.5);
        q[(N + 0) * Dk + 1] = 1.f;                                       /* frame 1, token 0: q = e1            */
        q[(N + 1) * Dk + 1] = h2; q[(N + 1) * Dk + 4] = h2;              /* frame 1, token 1: q = (e1 + e4)/sqrt2 */
        k[(N + 0) * Dk + 6] = 1.f; k[(N + 1) * Dk + 7] = 1.f;
        if (gdkvm_oracle_scan_f64(q, k, v, a, b, NULL, r, s, 1, T, N, 1, Dk, Dv, 2, 0)) return fail("K1 rc");
        for (int c = 0; c < Dv; ++c)
            if (r[(N + 0) * Dv + c] != v[c]) return fail("K1: recall of an orthonormal key is not exact");
        if (gdkvm_oracle_scan_normalizer_f64(q, k, v, a, b, NULL, NULL, r, s, z, 1, T, N, 1, Dk, Dv, 2, 0, 1e-6)) return fail("K12 rc");
        for (int c = 0; c < Dv; ++c) {
            const double want0 = v[c] / (1.0 + 1e-6), want1 = 0.5 * ((double)v[c] + v[Dv + c]);   /* q.z = 1 resp. sqrt 2: the MEAN of v1, v2 */
            if (fabs(r[(N + 0) * Dv + c] - want0) > 1e-5 || fabs(r[(N + 1) * Dv + c] - want1) > 1e-4 * want1) return fail("K12: normalised read-out");
        }
        if (z[1] != 1.f || z[4] != 1.f || z[6] != 1.f || z[7] != 1.f || z[0] != 0.f) return fail("K12: z is not the sum of the written keys");
    }
    printf("oracle_selftest: ok\n");
    return 0;
}
