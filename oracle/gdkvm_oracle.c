/* gdkvm_oracle.c -- scalar C restatement of the GDKVM memory path (CPU oracle + CPU baseline "port").
 *
 * TEST INFRASTRUCTURE ONLY.  Linked/loaded only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg (through oracle/c_oracle.py).  Never by gdkvm_amd/.
 *
 * PARITY UNPINNED: /root/reference holds no implementation, test or golden vector for this path
 * (the model code lives in the repo named at /root/reference/README.md:1 and is
 * git-ignored at /root/reference/.gitignore:73-76).  The functions below restate the builder's SPEC-v0
 * (SURVEY.md Appendix A), whose only reference-side sources are /root/reference/README.md:20 and
 * /root/reference/website/src/content/homepage/en.json:20 (names and roles of LKVA / GDR / KPFF) and
 * BASELINE.json north_star (gating formula).  Independent of oracle/gdkvm_oracle.py (numpy) so that the
 * two restatements check each other; both are pinned by the analytic KATs of SURVEY.md A.6.
 *
 * Build: see oracle/Makefile  ->  oracle/_build/libgdkvm_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SUF _f64
#include "gdkvm_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL float
#define SUF _f32
#include "gdkvm_oracle_impl.h"
#undef REAL
#undef SUF

/* Row a6: argmax over classes (ties -> lowest index) + integer Dice counts.
 * logits [BT, ncls, H, W] fp32; target may be NULL (then no counts are produced).
 * counts [BT, ncls, 3] = { |A n B|, |A|, |B| }. */
int gdkvm_oracle_argmax_dice(const float* logits, const uint8_t* target, uint8_t* mask, int32_t* counts,
                             int BT, int ncls, int H, int W)
{
    if (BT < 0 || ncls <= 0 || ncls > 255 || H <= 0 || W <= 0) return -1;
    const size_t HW = (size_t)H * W;
    if (counts && target) memset(counts, 0, sizeof(int32_t) * (size_t)BT * ncls * 3);
    for (int f = 0; f < BT; ++f)
        for (size_t p = 0; p < HW; ++p) {
            const float* lp = logits + (size_t)f * ncls * HW + p;
            int best = 0;
            float bv = lp[0];
            for (int c = 1; c < ncls; ++c) {
                const float x = lp[(size_t)c * HW];
                if (x > bv) { bv = x; best = c; }
            }
            mask[(size_t)f * HW + p] = (uint8_t)best;
            if (counts && target) {
                const int tc = target[(size_t)f * HW + p];
                counts[((size_t)f * ncls + best) * 3 + 1] += 1;
                if (tc < ncls) {
                    counts[((size_t)f * ncls + tc) * 3 + 2] += 1;
                    if (tc == best) counts[((size_t)f * ncls + best) * 3 + 0] += 1;
                }
            }
        }
    return 0;
}

/* Row a6, fused form: bilinear upsampling (align_corners = false, the PyTorch formula, all in fp32 and WITHOUT fused
 * multiply-add -- this file is built with -ffp-contract=off) of low-resolution logits [BT, ncls, hl, wl] to H x W, then
 * argmax over classes (ties -> lowest index) and the integer Dice counts.  The full-resolution logits never exist. */
static float up_src(float scale, int dst) { const float s = scale * ((float)dst + 0.5f) - 0.5f; return s < 0.f ? 0.f : s; }

int gdkvm_oracle_upsample_argmax_dice(const float* logits, const uint8_t* target, uint8_t* mask, int32_t* counts,
                                      int BT, int ncls, int hl, int wl, int H, int W)
{
    if (BT < 0 || ncls <= 0 || ncls > 255 || hl <= 0 || wl <= 0 || H <= 0 || W <= 0) return -1;
    const float sy = (float)hl / (float)H, sx = (float)wl / (float)W;
    if (counts && target) memset(counts, 0, sizeof(int32_t) * (size_t)BT * ncls * 3);
    for (int f = 0; f < BT; ++f)
        for (int y = 0; y < H; ++y) {
            const float fy = up_src(sy, y);
            const int y0 = (int)fy, y1 = y0 + 1 < hl ? y0 + 1 : hl - 1;
            const float ly = fy - (float)y0, hy = 1.0f - ly;
            for (int x = 0; x < W; ++x) {
                const float fx = up_src(sx, x);
                const int x0 = (int)fx, x1 = x0 + 1 < wl ? x0 + 1 : wl - 1;
                const float lx = fx - (float)x0, hx = 1.0f - lx;
                int best = 0;
                float bv = 0.f;
                for (int c = 0; c < ncls; ++c) {
                    const float* L = logits + ((size_t)f * ncls + c) * hl * wl;
                    const float top = hx * L[y0 * wl + x0] + lx * L[y0 * wl + x1];
                    const float bot = hx * L[y1 * wl + x0] + lx * L[y1 * wl + x1];
                    const float v = hy * top + ly * bot;
                    if (c == 0 || v > bv) { bv = v; best = c; }
                }
                const size_t p = ((size_t)f * H + y) * W + x;
                mask[p] = (uint8_t)best;
                if (counts && target) {
                    const int tc = target[p];
                    counts[((size_t)f * ncls + best) * 3 + 1] += 1;
                    if (tc < ncls) {
                        counts[((size_t)f * ncls + tc) * 3 + 2] += 1;
                        if (tc == best) counts[((size_t)f * ncls + best) * 3 + 0] += 1;
                    }
                }
            }
        }
    return 0;
}

int gdkvm_oracle_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
