// Full-square quad elimination (superseded in the library by spkd_tri.hpp); kept for
// tools/lu_bench.hip and tools/blocked_bench.hip, which measure it against the
// column-blocked and the symmetric forms.
#pragma once
#include "spkd_quad.hpp"

namespace spkd {

template <int K>
struct QuadStep {
    static __device__ __forceinline__ void run(QuadRows& q, double& det, bool& ok) {
        constexpr int S = K / QL, T = K % QL;
        const double piv = bcast16<T>(q.r[S][K]);
        ok = ok && (piv > 0.0) && (piv < __builtin_huge_val());
        det *= piv;
        const double inv = fast_recip(piv);
        double l[QS];
#pragma unroll
        for (int s = S; s < QS; ++s) l[s] = -(q.r[s][K] * inv);
#pragma unroll
        for (int j = K + 1; j < D; ++j) {
            const double u = bcast16<T>(q.r[S][j]);
#pragma unroll
            for (int s = S; s < QS; ++s) q.r[s][j] = fma(l[s], u, q.r[s][j]);
        }
        QuadStep<K + 1>::run(q, det, ok);
    }
};

template <>
struct QuadStep<D> {
    static __device__ __forceinline__ void run(QuadRows&, double&, bool&) {}
};

// det (per DPP row, i.e. per matrix) of four symmetric positive definite matrices.
// Returns, per lane, whether its matrix met only positive finite pivots.
__device__ __forceinline__ bool quad_det_nopivot(QuadRows& q, double& det_out) {
    double det = 1.0;
    bool ok = true;
    QuadStep<0>::run(q, det, ok);
    det_out = det;
    return ok;
}


// q += w * record rows; sv2[s] = that record's sums column
__device__ __forceinline__ void quad_fma_record(const double* __restrict__ qr, int t, double w,
                                                QuadRows& q, double (&sv2)[QS]) {
#pragma unroll
    for (int s = 0; s < QS; ++s) {
#pragma unroll
        for (int j = 0; j < D; ++j) q.r[s][j] = fma(w, qr[(s * DA + j) * 16 + t], q.r[s][j]);
        sv2[s] = qr[(s * DA + D) * 16 + t];
    }
}

// rank-one update q[s][j] += c[s] * bcast(v)[j] for all j, where v[s] holds the
// per-row values of a vector (same distribution as the rows)
template <int J>
struct QuadRank1 {
    static __device__ __forceinline__ void run(QuadRows& q, const double (&c)[QS], const double (&v)[QS]) {
        constexpr int S = J / QL, T = J % QL;
        const double vj = bcast16<T>(v[S]);
#pragma unroll
        for (int s = 0; s < QS; ++s) q.r[s][J] = fma(c[s], vj, q.r[s][J]);
        QuadRank1<J + 1>::run(q, c, v);
    }
};
template <>
struct QuadRank1<D> {
    static __device__ __forceinline__ void run(QuadRows&, const double (&)[QS], const double (&)[QS]) {}
};

// np.cov semantics on raw moments, in place: S_ij = (Q_ij - s_i s_j / n) / (n - 1)
__device__ __forceinline__ void quad_cov(QuadRows& q, const double (&sv)[QS], double n) {
    const double inv_n = 1.0 / n, f = 1.0 / (n - 1.0);
    double c[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) c[s] = -(sv[s] * inv_n);
    QuadRank1<0>::run(q, c, sv);
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
        for (int j = 0; j < D; ++j) q.r[s][j] *= f;
}

// per-lane flag: every entry of the lane's matrix is finite (DPP-row wide AND)
__device__ __forceinline__ bool quad_finite(const QuadRows& q, int m) {
    bool ok = true;
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
        for (int j = 0; j < D; ++j) ok = ok && (fabs(q.r[s][j]) < __builtin_huge_val());
    ok = ok || ((lane_id() & 15) >= QL);        // padding lanes carry no rows
    const unsigned long long bad = __ballot(!ok);
    return ((bad >> (16 * m)) & 0xffffull) == 0ull;
}

// log(det) of the four matrices held in q (per lane: its own matrix).  Matrices
// that meet a non-positive pivot are redone one at a time in the row-per-lane
// layout with partial pivoting: form_single(mi, a) must fill matrix mi there.
// Non-finite matrices raise ERR_NONFINITE and give NaN.
template <class FormSingle>
__device__ __forceinline__ double quad_logdet(QuadRows& q, int m, int* err, FormSingle form_single) {
    // a NaN / inf entry always surfaces as a pivot that is not a positive finite
    // number, so the finite check of the reference (scipy raises ValueError) is left
    // to the fallback, which re-forms the matrix and tests it
    double det;
    const bool ok = quad_det_nopivot(q, det);
    double ld = log(det);
    const unsigned long long badmask = __ballot(!ok);
#ifndef SPKD_NO_FALLBACK
    if (badmask) {
#pragma unroll 1
        for (int mi = 0; mi < 4; ++mi) {
            if (((badmask >> (16 * mi)) & 0xffffull) == 0ull) continue;   // wave-uniform
            double a[DA];
            form_single(mi, a);
            const double v = logdet_pivoted_fn(a, err);
            if (m == mi) ld = v;
        }
    }
#endif
    return ld;
}

}  // namespace spkd
