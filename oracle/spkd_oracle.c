/*
 * spkd_oracle.c — CPU restatement (plain C, fp64) of the reference's hot path on
 * additive sufficient statistics.  TEST INFRASTRUCTURE ONLY: linked/loaded by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
 * product package.  It exists so parity can be checked at sizes where the
 * raw-frame numpy oracle (oracle/numpy_engine.py) would take hours.
 *
 * Pinned: tests/test_oracle_golden.py runs every golden command-line case of
 * tests/golden/cli_cases.json (produced by the reference scripts themselves)
 * through this library and requires byte-identical recipes.
 *
 * Independent of the HIP implementation where that is cheap: every determinant
 * here is a partial-pivoting LU (LAPACK dgetf2 order, the algorithm behind
 * scipy.linalg.det), pinv is a Jacobi eigen-decomposition with scipy's cut-off,
 * the KL2 means are accumulated in float32 exactly like np.mean(float32, axis=0).
 *
 * Reference lines restated (paths relative to the reference tree):
 *   np.cov / det / log composition ... spk-clustering.py:88-100,103-121,124-133
 *   dist_gw ........................... spk-change-detection.py:180-288
 *   dist_sw (distances) ............... spk-change-detection.py:304-312
 *   spk_cluster_hi v1 / v2 ............ spk-clustering.py:178-240 / spk-clustering2.py:173-222
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define D 39
#define DA 40
#define REC 820
#define G 8
#define MAXINT_F 9223372036854775808.0
#define PEN_UNIT (39.0 + 0.5 * 39.0 * 40.0)

enum { ORC_BIC = 0, ORC_GLR = 1, ORC_KL2 = 2 };

static int pk(int r, int c) { return r * DA - (r * (r - 1)) / 2 + (c - r); }

/* ------------------------------------------------------------------ statistics */
void orc_accumulate(const float *frames, int64_t begin, int64_t end, double *rec) {
    for (int64_t t = begin; t < end; ++t) {
        const float *x = frames + t * D;
        int e = 0;
        for (int r = 0; r < DA; ++r) {
            const double xr = r < D ? (double)x[r] : 1.0;
            for (int c = r; c < DA; ++c, ++e) rec[e] += xr * (c < D ? (double)x[c] : 1.0);
        }
    }
}

/* float32 running sum of frames, the way np.mean(float32 array, axis=0) adds rows */
void orc_accumulate_f32(const float *frames, int64_t begin, int64_t end, float *sum) {
    for (int64_t t = begin; t < end; ++t)
        for (int j = 0; j < D; ++j) sum[j] += frames[t * D + j];
}

static void cov_from_rec(const double *rec, double *S /* D*D */) {
    const double n = rec[REC - 1];
    const double inv_n = 1.0 / n, f = 1.0 / (n - 1.0);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            const int r = i < j ? i : j, c = i < j ? j : i;
            S[i * D + j] = (rec[pk(r, c)] - rec[pk(i, D)] * rec[pk(j, D)] * inv_n) * f;
        }
}

/* log(det(S)) with numpy/scipy semantics; *nonfinite set when S has NaN/inf
 * (scipy.linalg.det raises ValueError there). */
static double logdet(double *A, int *nonfinite) {
    for (int i = 0; i < D * D; ++i)
        if (!(fabs(A[i]) < INFINITY)) { *nonfinite = 1; return NAN; }
    double det = 1.0;
    int neg = 0;
    for (int k = 0; k < D; ++k) {
        int p = k;
        double best = fabs(A[k * D + k]);
        for (int i = k + 1; i < D; ++i)
            if (fabs(A[i * D + k]) > best) { best = fabs(A[i * D + k]); p = i; }
        if (p != k) {
            for (int j = 0; j < D; ++j) { double t = A[k * D + j]; A[k * D + j] = A[p * D + j]; A[p * D + j] = t; }
            neg = !neg;
        }
        const double piv = A[k * D + k];
        det *= piv;
        if (piv != 0.0) {
            const double r = 1.0 / piv;
            for (int i = k + 1; i < D; ++i) {
                const double l = A[i * D + k] * r;
                if (l != 0.0)
                    for (int j = k + 1; j < D; ++j) A[i * D + j] -= l * A[k * D + j];
            }
        }
    }
    if (neg) det = -det;
    return log(det);
}

double orc_logdet_cov(const double *rec, int *nonfinite) {
    double S[D * D];
    cov_from_rec(rec, S);
    return logdet(S, nonfinite);
}

/* symmetric pinv diagonal via cyclic Jacobi; cut-off = 39 * eps * max|lambda|
 * (scipy.linalg.pinv: rtol = max(M, N) * eps on the singular values) */
static void pinv_diag(const double *S, double *diag) {
    double A[D * D], V[D * D];
    memcpy(A, S, sizeof A);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) V[i * D + j] = i == j;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < D; ++i)
            for (int j = i + 1; j < D; ++j) off += A[i * D + j] * A[i * D + j];
        if (off < 1e-300) break;
        for (int p = 0; p < D; ++p)
            for (int q = p + 1; q < D; ++q) {
                const double apq = A[p * D + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * D + q] - A[p * D + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < D; ++k) {
                    const double akp = A[k * D + p], akq = A[k * D + q];
                    A[k * D + p] = c * akp - s * akq;
                    A[k * D + q] = s * akp + c * akq;
                }
                for (int k = 0; k < D; ++k) {
                    const double apk = A[p * D + k], aqk = A[q * D + k];
                    A[p * D + k] = c * apk - s * aqk;
                    A[q * D + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < D; ++k) {
                    const double vkp = V[k * D + p], vkq = V[k * D + q];
                    V[k * D + p] = c * vkp - s * vkq;
                    V[k * D + q] = s * vkp + c * vkq;
                }
            }
    }
    double lmax = 0.0;
    for (int i = 0; i < D; ++i) if (fabs(A[i * D + i]) > lmax) lmax = fabs(A[i * D + i]);
    const double cut = 39.0 * 2.220446049250313e-16 * lmax;
    for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        for (int k = 0; k < D; ++k) {
            const double lam = A[k * D + k];
            if (fabs(lam) > cut) acc += V[i * D + k] * V[i * D + k] / lam;
        }
        diag[i] = acc;
    }
}

/* KL2 as coded (elementwise products; float32 means): mean1/mean2 are the float32
 * means np.mean would give, or NULL to derive them from the records (rounded). */
double orc_kl2(const double *ra, const double *rb, const float *mean1, const float *mean2) {
    double S1[D * D], S2[D * D], p1[D], p2[D];
    cov_from_rec(ra, S1);
    cov_from_rec(rb, S2);
    pinv_diag(S1, p1);
    pinv_diag(S2, p2);
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < D; ++i) {
        const float m1 = mean1 ? mean1[i] : (float)(ra[pk(i, D)] / ra[REC - 1]);
        const float m2 = mean2 ? mean2[i] : (float)(rb[pk(i, D)] / rb[REC - 1]);
        const double delta = (double)(float)(m1 - m2);
        t1 += (S1[i * D + i] - S2[i * D + i]) * (p2[i] - p1[i]);
        t2 += ((p1[i] + p2[i]) * delta) * delta;
    }
    return 0.5 * t1 + 0.5 * t2;
}

/* out: n1 n2 ld1 ld2 ldU ldW kl2 0 ; returns 1 when a covariance was non-finite */
int orc_pair_terms(const double *ra, const double *rb, int want_glr, int want_kl2,
                   const float *mean1, const float *mean2, double *out) {
    int nf = 0;
    double u[REC];
    for (int e = 0; e < REC; ++e) u[e] = ra[e] + rb[e];
    out[0] = ra[REC - 1];
    out[1] = rb[REC - 1];
    out[2] = orc_logdet_cov(ra, &nf);
    out[3] = orc_logdet_cov(rb, &nf);
    out[4] = orc_logdet_cov(u, &nf);
    out[5] = NAN;
    out[6] = NAN;
    out[7] = 0.0;
    if (want_glr) {
        double S1[D * D], S2[D * D];
        const double n1 = out[0], n2 = out[1], n = n1 + n2;
        cov_from_rec(ra, S1);
        cov_from_rec(rb, S2);
        for (int i = 0; i < D * D; ++i) S1[i] = (n1 / n) * S1[i] + (n2 / n) * S2[i];
        out[5] = logdet(S1, &nf);
    }
    if (want_kl2) out[6] = orc_kl2(ra, rb, mean1, mean2);
    return nf;
}

static double bic_terms(double n1, double ld1, double n2, double ld2, double ldu, double lambdac) {
    const double n = n1 + n2;
    double d = 0.5 * n * ldu - 0.5 * n1 * ld1 - 0.5 * n2 * ld2;
    d -= lambdac * 0.5 * PEN_UNIT * log(n);
    return d;
}

static double glr_terms(double n1, double ld1, double n2, double ld2, double ldw) {
    const double n = n1 + n2;
    return -(n / 2.0) * ((n1 / n) * ld1 + (n2 / n) * ld2 - ldw);
}

/* ------------------------------------------------------------------ prefix stats of a turn */
typedef struct {
    const float *fr;
    int64_t n;
    double *snap; /* (n / G + 1) records */
} prefix_t;

static int prefix_build(prefix_t *P, const float *fr, int64_t n) {
    P->fr = fr;
    P->n = n;
    P->snap = (double *)calloc((size_t)(n / G + 1) * REC, sizeof(double));
    if (!P->snap) return -1;
    double acc[REC];
    memset(acc, 0, sizeof acc);
    for (int64_t t = 0; t < n; ++t) {
        orc_accumulate(fr, t, t + 1, acc);
        if ((t + 1) % G == 0) memcpy(P->snap + ((t + 1) / G) * REC, acc, sizeof acc);
    }
    return 0;
}

static void prefix_at(const prefix_t *P, int64_t t, double *rec) {
    const int64_t j = t / G;
    memcpy(rec, P->snap + j * REC, REC * sizeof(double));
    orc_accumulate(P->fr, j * G, t, rec);
}

static double range_logdet(const prefix_t *P, int64_t a, int64_t b, int *nf) {
    double ra[REC], rb[REC];
    prefix_at(P, a, ra);
    prefix_at(P, b, rb);
    for (int e = 0; e < REC; ++e) rb[e] -= ra[e];
    return orc_logdet_cov(rb, nf);
}

static void range_rec(const prefix_t *P, int64_t a, int64_t b, double *rec) {
    double ra[REC];
    prefix_at(P, a, ra);
    prefix_at(P, b, rec);
    for (int e = 0; e < REC; ++e) rec[e] -= ra[e];
}

static void range_mean_f32(const prefix_t *P, int64_t a, int64_t b, float *mean) {
    float s[D];
    memset(s, 0, sizeof s);
    orc_accumulate_f32(P->fr, a, b, s);
    for (int j = 0; j < D; ++j) mean[j] = s[j] / (float)(b - a);
}

/* distance of the two-part window [a,b) | [b,c) */
static double split_distance(const prefix_t *P, int kind, double lambdac, int64_t a, int64_t b,
                             int64_t c, double *memo_left, double *memo_union, int *nf) {
    const double n1 = (double)(b - a), n2 = (double)(c - b), n = n1 + n2;
    if (kind == ORC_KL2) {
        double r1[REC], r2[REC];
        float m1[D], m2[D];
        range_rec(P, a, b, r1);
        range_rec(P, b, c, r2);
        range_mean_f32(P, a, b, m1);
        range_mean_f32(P, b, c, m2);
        return orc_kl2(r1, r2, m1, m2);
    }
    const double ld1 = memo_left && !isnan(*memo_left) ? *memo_left : range_logdet(P, a, b, nf);
    if (memo_left) *memo_left = ld1;
    const double ld2 = range_logdet(P, b, c, nf);
    if (kind == ORC_BIC) {
        const double ldu = memo_union && !isnan(*memo_union) ? *memo_union : range_logdet(P, a, c, nf);
        if (memo_union) *memo_union = ldu;
        double d = 0.5 * n * ldu - 0.5 * n1 * ld1 - 0.5 * n2 * ld2;
        d -= lambdac * 0.5 * PEN_UNIT * log(n);
        return d;
    }
    double r1[REC], r2[REC], S1[D * D], S2[D * D];
    range_rec(P, a, b, r1);
    range_rec(P, b, c, r2);
    cov_from_rec(r1, S1);
    cov_from_rec(r2, S2);
    for (int i = 0; i < D * D; ++i) S1[i] = (n1 / n) * S1[i] + (n2 / n) * S2[i];
    return glr_terms(n1, ld1, n2, ld2, logdet(S1, nf));
}

/* ------------------------------------------------------------------ growing window */
typedef struct {
    int32_t kind, trace;
    double lambdac, threshold, winsize, winstep, deltaws, rate;
} orc_cd_params;

typedef struct {
    int32_t coarse;
    int32_t pad;
    int64_t win;
    double start, i, d;
    int64_t n1, n2;
} orc_cand_log;

/* one turn; outputs like spkd_gw for a single turn.  Returns 0, or 1 when a
 * covariance was non-finite, or -1 on capacity / allocation failure. */
int orc_gw_turn(const float *fr, int64_t n, const orc_cd_params *P, int64_t ev_cap,
                int32_t *n_win, double *win_maxd, int32_t *win_det, double *det_start,
                double *det_maxi, double *det_d, double *final_start,
                orc_cand_log *log, int64_t log_cap, int64_t *log_count) {
    prefix_t pf;
    if (prefix_build(&pf, fr, n)) return -1;
    int nf = 0;
    const double winsize = P->winsize, winstep = P->winstep, rate = P->rate;
    double start = 0.0, end = start + winsize * 2;
    const double minfeas = rate / 2, istep = rate / 10;
    double ws = minfeas, dws = P->deltaws;
    const int64_t memo_cap = (int64_t)((double)n / istep) + 16;
    double *memo = (double *)malloc((size_t)memo_cap * sizeof(double));
    if (!memo) { free(pf.snap); return -1; }
    for (int64_t k = 0; k < memo_cap; ++k) memo[k] = NAN;
    int nw = 0, nd = 0;
    int64_t nlog = 0;
    int rc = 0;
    while (end <= (double)n) {
        if (nw >= ev_cap) { rc = -1; break; }
        const int64_t a = (int64_t)start, c = (int64_t)end;
        double maxd = -MAXINT_F, maxi = 0.0;
        double ldu = NAN;            /* log det of the pooled window: same for every candidate */
        int found = 0;
        int64_t k = 0;
        for (double i = minfeas; i < end - start - minfeas; i += istep, ++k) {
            const int64_t b = (int64_t)(start + i);
            const double d = split_distance(&pf, P->kind, P->lambdac, a, b, c,
                                            k < memo_cap ? &memo[k] : NULL, &ldu, &nf);
            if (P->trace || d == INFINITY || d == -INFINITY) {
                if (log && nlog < log_cap) {
                    orc_cand_log r = {1, 0, nw, start, i, d, b - a, c - b};
                    log[nlog] = r;
                }
                ++nlog;
            }
            if (d > maxd && d != INFINITY) { maxd = d; maxi = i; found = 1; }
        }
        win_maxd[nw] = found ? maxd : NAN;
        win_det[nw] = 0;
        ++nw;
        if (found && maxd > P->threshold) {
            for (double i = maxi - istep, endtune = maxi + istep; i < endtune; i += 1) {
                const int64_t b = (int64_t)(start + i);
                const double d = split_distance(&pf, P->kind, P->lambdac, a, b, c, NULL, &ldu, &nf);
                if (d == INFINITY || d == -INFINITY) {
                    if (log && nlog < log_cap) {
                        orc_cand_log r = {0, 0, nw - 1, start, i, d, b - a, c - b};
                        log[nlog] = r;
                    }
                    ++nlog;
                }
                if (d > maxd && d != INFINITY) { maxd = d; maxi = i; }
            }
            det_start[nd] = start;
            det_maxi[nd] = maxi;
            det_d[nd] = maxd;
            win_det[nw - 1] = 1;
            ++nd;
            for (int64_t q = 0; q < memo_cap; ++q) memo[q] = NAN;
            start += maxi;
            if (start + winsize * 2 <= (double)n) {
                end = start + winsize * 2;
                ws = minfeas;
                dws = P->deltaws;
            } else {
                break;
            }
        } else {
            if (end + ws <= (double)n) {
                end += ws;
                if (ws < winstep) { ws += dws; dws *= 2; }
                if (ws > winstep) ws = winstep;
            } else if (end != (double)n) {
                end = (double)n;
            } else {
                break;
            }
        }
    }
    *n_win = nw;
    *final_start = start;
    if (log_count) *log_count = nlog;
    free(memo);
    free(pf.snap);
    if (rc) return rc;
    return nf ? 1 : 0;
}

/* ------------------------------------------------------------------ sliding window */
int orc_sw_turn(const float *fr, int64_t n, const orc_cd_params *P, double *d_out) {
    prefix_t pf;
    if (prefix_build(&pf, fr, n)) return -1;
    int nf = 0;
    const int64_t w = (int64_t)P->winsize;
    int64_t k = 0;
    for (double s = 0; s + 2 * P->winsize <= (double)n; s += P->winstep, ++k) {
        const int64_t a = (int64_t)s;
        d_out[k] = split_distance(&pf, P->kind, P->lambdac, a, a + w, a + 2 * w, NULL, NULL, &nf);
    }
    free(pf.snap);
    return nf ? 1 : 0;
}

/* ------------------------------------------------------------------ agglomerative clustering */
typedef struct {
    int32_t variant, kind, max_spk, reserved;
    double lambdac, threshold;
} orc_ahc_params;

static double cluster_distance(int kind, double lambdac, const double *ra, double lda,
                               const double *rb, double ldb, int *nf) {
    if (kind == ORC_KL2) return orc_kl2(ra, rb, NULL, NULL);
    const double n1 = ra[REC - 1], n2 = rb[REC - 1], n = n1 + n2;
    if (kind == ORC_BIC) {
        double u[REC];
        for (int e = 0; e < REC; ++e) u[e] = ra[e] + rb[e];
        return bic_terms(n1, lda, n2, ldb, orc_logdet_cov(u, nf), lambdac);
    }
    double S1[D * D], S2[D * D];
    cov_from_rec(ra, S1);
    cov_from_rec(rb, S2);
    for (int i = 0; i < D * D; ++i) S1[i] = (n1 / n) * S1[i] + (n2 / n) * S2[i];
    return glr_terms(n1, lda, n2, ldb, logdet(S1, nf));
}

/* Threads: the pair distances of a clustering problem are independent, so the loops over
 * them (and the scans of the matrix) are OpenMP loops.  Every distance is still computed
 * by the same scalar code on one thread and every reduction is combined in index order,
 * so results do not depend on the thread count.  orc_set_threads(1) gives the scalar
 * port, orc_set_threads(0) all cores; orc_get_threads() what a parallel region gets. */
int orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

int orc_get_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* stats: n records (copied; merged in place on the copy).  merge_a/b/d: capacity n.
 * stat_max / stat_min as spkd_ahc. */
int orc_ahc(const double *stats_in, int64_t n, const orc_ahc_params *P, int32_t *n_merges,
            int32_t *merge_a, int32_t *merge_b, double *merge_d, double *stat_max, double *stat_min) {
    int nf = 0;
    double *st = (double *)malloc((size_t)n * REC * sizeof(double));
    double *ld = (double *)malloc((size_t)n * sizeof(double));
    double *dm = (double *)malloc((size_t)n * n * sizeof(double));
    double *dm2 = (double *)malloc((size_t)n * n * sizeof(double));  /* np.delete target */
    int64_t *ids = (int64_t *)malloc((size_t)n * sizeof(int64_t));   /* compacted position -> slot */
    double *rs = (double *)malloc((size_t)(4 * n + 8) * sizeof(double));   /* per-row scan results */
    if (!st || !ld || !dm || !dm2 || !ids || !rs) { free(st); free(ld); free(dm); free(dm2); free(ids); free(rs); return -1; }
    memcpy(st, stats_in, (size_t)n * REC * sizeof(double));
    double smax = NAN, smin = NAN;
#pragma omp parallel for schedule(static) reduction(|:nf)
    for (int64_t i = 0; i < n; ++i) {
        ids[i] = i;
        ld[i] = P->kind == ORC_KL2 ? 0.0 : orc_logdet_cov(st + i * REC, &nf);
    }
    /* the matrix is kept compacted exactly like np.delete does */
    int64_t m = n;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) dm[i * n + j] = P->variant == 1 ? (i == j ? MAXINT_F : 0.0) : INFINITY;
#define NOTE(d) do { if ((d) == (d) && fabs(d) < INFINITY) { \
        if (smax != smax || (d) > smax) smax = (d); if (smin != smin || (d) < smin) smin = (d); } } while (0)
#pragma omp parallel for schedule(dynamic, 4) reduction(|:nf)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j) {
            const double d = cluster_distance(P->kind, P->lambdac, st + i * REC, ld[i], st + j * REC, ld[j], &nf);
            dm[i * n + j] = d;
            if (P->variant == 1) dm[j * n + i] = d;
        }
    if (P->variant == 1)
        for (int64_t i = 0; i < n; ++i)
            for (int64_t j = i + 1; j < n; ++j) NOTE(dm[i * n + j]);
    int nm = 0;
    double fmax = NAN, fmin = NAN;
    for (;;) {
        /* numpy min / argmin over the m x m compacted matrix (row stride m): per-row
         * partial results, combined in row order (first occurrence wins) */
        double mind = INFINITY, mx = -INFINITY;
        int64_t idx = -1, nan_idx = -1;
        {
            double *rmin = rs, *rmax = rs + m;
            int64_t *ridx = (int64_t *)(rs + 2 * m), *rnan = ridx + m;
            {
#pragma omp parallel for schedule(static)
                for (int64_t r = 0; r < m; ++r) {
                    double mv = INFINITY, xv = -INFINITY;
                    int64_t mi = -1, ni = -1;
                    for (int64_t c = 0; c < m; ++c) {
                        const double v = dm[r * m + c];
                        if (v != v) { if (ni < 0) ni = r * m + c; continue; }
                        if (mi < 0 || v < mv) { mv = v; mi = r * m + c; }
                        if (v > xv) xv = v;
                    }
                    rmin[r] = mv; rmax[r] = xv; ridx[r] = mi; rnan[r] = ni;
                }
                for (int64_t r = 0; r < m; ++r) {
                    if (rnan[r] >= 0 && nan_idx < 0) nan_idx = rnan[r];
                    if (ridx[r] >= 0 && (idx < 0 || rmin[r] < mind)) { mind = rmin[r]; idx = ridx[r]; }
                    if (rmax[r] > mx) mx = rmax[r];
                }
            }
        }
        if (nan_idx >= 0) { mind = NAN; idx = nan_idx; mx = NAN; }
        fmax = mx;
        fmin = mind;
        if (!(mind <= P->threshold || (P->max_spk > 0 && m > P->max_spk))) break;
        int64_t a = idx / m, b = idx % m;
        if (a > b) { int64_t t = a; a = b; b = t; }
        if (a == b) break;
        merge_a[nm] = (int32_t)a;
        merge_b[nm] = (int32_t)b;
        merge_d[nm] = mind;
        ++nm;
        double *ra = st + ids[a] * REC;
        const double *rb = st + ids[b] * REC;
        for (int e = 0; e < REC; ++e) ra[e] += rb[e];
        /* np.delete(row b), np.delete(col b) */
#pragma omp parallel for schedule(static)
        for (int64_t i2 = 0; i2 < m - 1; ++i2) {
            const double *src = dm + (i2 < b ? i2 : i2 + 1) * m;
            double *dst = dm2 + i2 * (m - 1);
            for (int64_t j = 0; j < b; ++j) dst[j] = src[j];
            for (int64_t j = b + 1; j < m; ++j) dst[j - 1] = src[j];
        }
        { double *t = dm; dm = dm2; dm2 = t; }
        for (int64_t i = b; i + 1 < m; ++i) ids[i] = ids[i + 1];
        --m;
        if (P->kind != ORC_KL2) ld[ids[a]] = orc_logdet_cov(ra, &nf);
#pragma omp parallel for schedule(dynamic, 8) reduction(|:nf)
        for (int64_t s2 = 0; s2 < m; ++s2) {
            if (s2 == a) continue;
            const double d = cluster_distance(P->kind, P->lambdac, ra, ld[ids[a]], st + ids[s2] * REC,
                                              ld[ids[s2]], &nf);
            dm[a * m + s2] = d;
            if (P->variant == 1) dm[s2 * m + a] = d;
        }
        if (P->variant == 1)
            for (int64_t s2 = 0; s2 < m; ++s2)
                if (s2 != a) NOTE(dm[a * m + s2]);
    }
    *n_merges = nm;
    if (P->variant == 1) { *stat_max = smax; *stat_min = smin; }
    else { *stat_max = fmax; *stat_min = fmin; }
    free(st); free(ld); free(dm); free(dm2); free(ids); free(rs);
    return nf ? 1 : 0;
}
