/* Scalar C restatement of the GDKVM memory path -- body, included once per arithmetic type.
 * TEST INFRASTRUCTURE ONLY (see gdkvm_oracle.c).  REAL = arithmetic type, SUF = symbol suffix.
 * I/O is always float (fp32); bf16 callers round on the host first. */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

static REAL FN(sigm)(REAL x) { return (REAL)1 / ((REAL)1 + (REAL)exp(-(double)x)); }

/* Row a3 = a1 + a2 (+ a5 prologue), definitional token-sequential form.
 * q,k [B,T,N,Hh,Dk]  v,r [B,T,N,Hh,Dv]  alpha [B,T,Hh]  beta [B,T,N,Hh]  s [B,Hh,Dk,Dv]. */
/* normalizer (SURVEY.md A.1 flag): z [B,Hh,Dk] is the state column of a value channel that is 1 for every token -- it follows the
 * SAME write rule as every column of S -- and the read-out of token n becomes R_t[n,:] / (|q_n . z_{t-1}| + eps).  z_in NULL = zeros. */
static int FN(scan_core)(const float* q, const float* k, const float* v, const float* alpha,
                         const float* beta, const float* s_in, const float* z_in, float* r_out, float* s_out, float* z_out,
                         int B, int T, int N, int Hh, int Dk, int Dv, int rule, int flags, int normalizer, double eps)
{
    if (B < 0 || T < 0 || N < 0 || Hh <= 0 || Dk <= 0 || Dv <= 0 || rule < 0 || rule > 2) return -1;
    int status = 0;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < Hh; ++h) {
            REAL* S = (REAL*)malloc(sizeof(REAL) * (size_t)Dk * Dv);
            REAL* kn = (REAL*)malloc(sizeof(REAL) * (size_t)(N > 0 ? N : 1) * Dk);
            REAL* qn = (REAL*)malloc(sizeof(REAL) * (size_t)Dk);
            REAL* e = (REAL*)malloc(sizeof(REAL) * (size_t)(N > 0 ? N : 1) * Dv);
            REAL* z = (REAL*)malloc(sizeof(REAL) * (size_t)Dk);
            REAL* ez = (REAL*)malloc(sizeof(REAL) * (size_t)(N > 0 ? N : 1));
            if (!S || !kn || !qn || !e || !z || !ez) { status = -5; free(S); free(kn); free(qn); free(e); free(z); free(ez); continue; }
            for (int i = 0; i < Dk * Dv; ++i)
                S[i] = s_in ? (REAL)s_in[((size_t)b * Hh + h) * Dk * Dv + i] : (REAL)0;
            for (int d = 0; d < Dk; ++d) z[d] = (normalizer && z_in) ? (REAL)z_in[((size_t)b * Hh + h) * Dk + d] : (REAL)0;
            for (int t = 0; t < T; ++t) {
                const size_t bt = (size_t)b * T + t;
                REAL a = (REAL)alpha[bt * Hh + h];
                if (flags & 2) a = FN(sigm)(a);
                /* a1: LKVA read against the state BEFORE this frame's write */
                for (int n = 0; n < N; ++n) {
                    const float* qp = q + ((bt * N + n) * Hh + h) * Dk;
                    REAL inv = 1;
                    if (flags & 1) {
                        REAL ss = 0;
                        for (int d = 0; d < Dk; ++d) ss += (REAL)qp[d] * (REAL)qp[d];
                        inv = (REAL)1 / (REAL)sqrt((double)(ss + (REAL)1e-12));
                    }
                    for (int d = 0; d < Dk; ++d) qn[d] = (REAL)qp[d] * inv;
                    float* rp = r_out + ((bt * N + n) * Hh + h) * Dv;
                    REAL den = 1;
                    if (normalizer) {
                        REAL qz = 0;
                        for (int d = 0; d < Dk; ++d) qz += qn[d] * z[d];
                        den = (REAL)fabs((double)qz) + (REAL)eps;
                    }
                    for (int c = 0; c < Dv; ++c) {
                        REAL acc = 0;
                        for (int d = 0; d < Dk; ++d) acc += qn[d] * S[(size_t)d * Dv + c];
                        rp[c] = (float)(normalizer ? acc / den : acc);
                    }
                }
                /* a5: key normalisation */
                for (int n = 0; n < N; ++n) {
                    const float* kp = k + ((bt * N + n) * Hh + h) * Dk;
                    REAL inv = 1;
                    if (flags & 1) {
                        REAL ss = 0;
                        for (int d = 0; d < Dk; ++d) ss += (REAL)kp[d] * (REAL)kp[d];
                        inv = (REAL)1 / (REAL)sqrt((double)(ss + (REAL)1e-12));
                    }
                    for (int d = 0; d < Dk; ++d) kn[(size_t)n * Dk + d] = (REAL)kp[d] * inv;
                }
                /* a2: GDR write */
                for (int i = 0; i < Dk * Dv; ++i) S[i] *= a;
                for (int d = 0; d < Dk; ++d) z[d] *= a;
                if (rule == 1) /* parallel: every token's error against the decayed state */
                    for (int n = 0; n < N; ++n) {
                        const float* vp = v + ((bt * N + n) * Hh + h) * Dv;
                        for (int c = 0; c < Dv; ++c) {
                            REAL acc = 0;
                            for (int d = 0; d < Dk; ++d) acc += kn[(size_t)n * Dk + d] * S[(size_t)d * Dv + c];
                            e[(size_t)n * Dv + c] = (REAL)vp[c] - acc;
                        }
                        REAL accz = 0;
                        for (int d = 0; d < Dk; ++d) accz += kn[(size_t)n * Dk + d] * z[d];
                        ez[n] = (REAL)1 - accz;
                    }
                for (int n = 0; n < N; ++n) {
                    const float* vp = v + ((bt * N + n) * Hh + h) * Dv;
                    REAL bt_n = (REAL)beta[(bt * N + n) * Hh + h];
                    if (flags & 2) bt_n = FN(sigm)(bt_n);
                    const REAL* kr = kn + (size_t)n * Dk;
                    if (rule == 2) {
                        for (int c = 0; c < Dv; ++c) {
                            REAL acc = 0;
                            for (int d = 0; d < Dk; ++d) acc += kr[d] * S[(size_t)d * Dv + c];
                            e[(size_t)n * Dv + c] = (REAL)vp[c] - acc;
                        }
                        REAL accz = 0;
                        for (int d = 0; d < Dk; ++d) accz += kr[d] * z[d];
                        ez[n] = (REAL)1 - accz;
                    } else if (rule == 0) {
                        for (int c = 0; c < Dv; ++c) e[(size_t)n * Dv + c] = (REAL)vp[c];
                        ez[n] = (REAL)1;
                    }
                    for (int d = 0; d < Dk; ++d) {
                        const REAL bk = bt_n * kr[d];
                        for (int c = 0; c < Dv; ++c) S[(size_t)d * Dv + c] += bk * e[(size_t)n * Dv + c];
                        z[d] += bk * ez[n];
                    }
                }
            }
            if (s_out)
                for (int i = 0; i < Dk * Dv; ++i) s_out[((size_t)b * Hh + h) * Dk * Dv + i] = (float)S[i];
            if (normalizer && z_out)
                for (int d = 0; d < Dk; ++d) z_out[((size_t)b * Hh + h) * Dk + d] = (float)z[d];
            free(S); free(kn); free(qn); free(e); free(z); free(ez);
        }
    return status;
}

int FN(gdkvm_oracle_scan)(const float* q, const float* k, const float* v, const float* alpha,
                          const float* beta, const float* s_in, float* r_out, float* s_out,
                          int B, int T, int N, int Hh, int Dk, int Dv, int rule, int flags)
{
    return FN(scan_core)(q, k, v, alpha, beta, s_in, NULL, r_out, s_out, NULL, B, T, N, Hh, Dk, Dv, rule, flags, 0, 0.0);
}

/* Row a1 with the normalizer flag of SURVEY.md A.1: as gdkvm_oracle_scan, with z [B,Hh,Dk] carried in / out (either may be NULL). */
int FN(gdkvm_oracle_scan_normalizer)(const float* q, const float* k, const float* v, const float* alpha,
                                     const float* beta, const float* s_in, const float* z_in, float* r_out, float* s_out, float* z_out,
                                     int B, int T, int N, int Hh, int Dk, int Dv, int rule, int flags, double eps)
{
    return FN(scan_core)(q, k, v, alpha, beta, s_in, z_in, r_out, s_out, z_out, B, T, N, Hh, Dk, Dv, rule, flags, 1, eps);
}

/* Row a4, KPFF.  L [BT,N,Ck] G [BT,N,Cv] P [BT,N,Cp], N = h*w; Wa [2Cp, Cp+Ck+Cv],
 * ba [2Cp], Wl [Cp,Ck], Wg [Cp,Cv]; out F [BT,N,Cp]. */
int FN(gdkvm_oracle_kpff)(const float* L, const float* G, const float* P, const float* Wa,
                          const float* ba, const float* Wl, const float* Wg, float* F,
                          int BT, int Ck, int Cv, int Cp, int h, int w)
{
    if (BT < 0 || Ck <= 0 || Cv <= 0 || Cp <= 0 || h <= 0 || w <= 0) return -1;
    const int N = h * w, Cin = Cp + Ck + Cv;
    static const int scales[3] = {1, 2, 4};
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int f = 0; f < BT; ++f) {
        REAL* gms = (REAL*)calloc((size_t)N * Cv, sizeof(REAL));
        REAL* x = (REAL*)malloc(sizeof(REAL) * (size_t)Cin);
        REAL* cell = (REAL*)malloc(sizeof(REAL) * (size_t)Cv);
        if (!gms || !x || !cell) { status = -5; free(gms); free(x); free(cell); continue; }
        const float* Gf = G + (size_t)f * N * Cv;
        for (int si = 0; si < 3; ++si) {
            const int s = scales[si];
            for (int y0 = 0; y0 < h; y0 += s)
                for (int x0 = 0; x0 < w; x0 += s) {
                    const int y1 = y0 + s < h ? y0 + s : h, x1 = x0 + s < w ? x0 + s : w;
                    for (int c = 0; c < Cv; ++c) cell[c] = 0;
                    for (int yy = y0; yy < y1; ++yy)
                        for (int xx = x0; xx < x1; ++xx)
                            for (int c = 0; c < Cv; ++c) cell[c] += (REAL)Gf[((size_t)yy * w + xx) * Cv + c];
                    const REAL inv = (REAL)1 / (REAL)((y1 - y0) * (x1 - x0));
                    for (int yy = y0; yy < y1; ++yy)
                        for (int xx = x0; xx < x1; ++xx)
                            for (int c = 0; c < Cv; ++c) gms[((size_t)yy * w + xx) * Cv + c] += cell[c] * inv;
                }
        }
        for (size_t i = 0; i < (size_t)N * Cv; ++i) gms[i] /= (REAL)3;
        for (int n = 0; n < N; ++n) {
            const float* Pn = P + ((size_t)f * N + n) * Cp;
            const float* Ln = L + ((size_t)f * N + n) * Ck;
            for (int c = 0; c < Cp; ++c) x[c] = (REAL)Pn[c];
            for (int c = 0; c < Ck; ++c) x[Cp + c] = (REAL)Ln[c];
            for (int c = 0; c < Cv; ++c) x[Cp + Ck + c] = gms[(size_t)n * Cv + c];
            for (int o = 0; o < Cp; ++o) {
                REAL gl = (REAL)ba[o], gg = (REAL)ba[Cp + o], lp = 0, gp = 0;
                const float* wal = Wa + (size_t)o * Cin;
                const float* wag = Wa + (size_t)(Cp + o) * Cin;
                for (int c = 0; c < Cin; ++c) { gl += x[c] * (REAL)wal[c]; gg += x[c] * (REAL)wag[c]; }
                for (int c = 0; c < Ck; ++c) lp += x[Cp + c] * (REAL)Wl[(size_t)o * Ck + c];
                for (int c = 0; c < Cv; ++c) gp += x[Cp + Ck + c] * (REAL)Wg[(size_t)o * Cv + c];
                F[((size_t)f * N + n) * Cp + o] = (float)(x[o] + FN(sigm)(gl) * lp + FN(sigm)(gg) * gp);
            }
        }
        free(gms); free(x); free(cell);
    }
    return status;
}

#undef FN
#undef CAT
#undef CAT_
