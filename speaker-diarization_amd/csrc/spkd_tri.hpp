// Symmetric ("triangular") quad elimination: the quad layout of spkd_quad.hpp
// (four 39x39 matrices per wave64, lane t of a DPP row holds rows t, 13 + t, 26 + t)
// keeping only the LOWER triangle: slot s needs columns 0 .. 13 (s + 1) - 1, i.e.
// 13 + 26 + 39 = 78 doubles per lane instead of 117.  Step k updates
//     a[i][j] -= (a[i][k] / a[k][k]) * a[j][k]        for k < j <= i
// where a[j][k] (row j, column k: lower triangle, not yet touched by step k) is
// broadcast from its owner lane j % 13 / slot j / 13 with one row_newbcast DPP
// move -- the symmetric counterpart of broadcasting the pivot row.  Lanes whose
// rows lie above column j compute into registers nobody reads.  1 144 fp64 FMAs
// + 780 DPP moves per four determinants (full square form: 1 820 + 741), and
// 156 VGPRs of matrix instead of 234: two waves per SIMD without column blocking.
#pragma once
#include "spkd_quad.hpp"

namespace spkd {

// columns held by slot s: 0 .. tri_cols(s) - 1
__host__ __device__ constexpr int tri_cols(int s) { return QL * (s + 1); }

template <int K, int J>
struct TriCol {
    static __device__ __forceinline__ void run(QuadRows& q, const double (&l)[QS]) {
        if constexpr (J < D) {
            constexpr int SJ = J / QL, TJ = J % QL;
            const double u = bcast16<TJ>(q.r[SJ][K]);            // a[J][K]
#pragma unroll
            for (int s = SJ; s < QS; ++s) q.r[s][J] = fma(l[s], u, q.r[s][J]);
            TriCol<K, J + 1>::run(q, l);
        }
    }
};

template <int K>
struct TriStep {
    static __device__ __forceinline__ void run(QuadRows& q, double& det, bool& ok) {
        if constexpr (K < D) {
            constexpr int S = K / QL, T = K % QL;
            const double piv = bcast16<T>(q.r[S][K]);
            ok = ok && (piv > 0.0) && (piv < __builtin_huge_val());
            det *= piv;
            const double inv = fast_recip(piv);
            double l[QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) l[s] = (s >= S) ? -(q.r[s][K] * inv) : 0.0;
            TriCol<K, K + 1>::run(q, l);
            TriStep<K + 1>::run(q, det, ok);
        }
    }
};

// det (per DPP row) of four symmetric matrices given by their lower triangles.
__device__ __forceinline__ bool tri_det_nopivot(QuadRows& q, double& det_out) {
    double det = 1.0;
    bool ok = true;
    TriStep<0>::run(q, det, ok);
    det_out = det;
    return ok;
}

// q[s][J] += c[s] * v_J for the lower-triangle columns (v distributed like the rows)
template <int J>
struct TriRank1 {
    static __device__ __forceinline__ void run(QuadRows& q, const double (&c)[QS], const double (&v)[QS]) {
        if constexpr (J < D) {
            constexpr int SJ = J / QL, TJ = J % QL;
            const double vj = bcast16<TJ>(v[SJ]);
#pragma unroll
            for (int s = SJ; s < QS; ++s) q.r[s][J] = fma(c[s], vj, q.r[s][J]);
            TriRank1<J + 1>::run(q, c, v);
        }
    }
};

// log(det) of the four lower triangles held in q (per lane: its own matrix).
// Matrices that meet a pivot that is not a positive finite number (this is also how
// NaN / inf entries surface) are redone one at a time in the row-per-lane layout
// with partial pivoting: form_single(mi, a) must fill matrix mi there.
template <class FormSingle>
__device__ __forceinline__ double tri_logdet(QuadRows& q, int m, int* err, FormSingle form_single) {
    double det;
    const bool ok = tri_det_nopivot(q, det);
    double ld = log(det);
    const unsigned long long badmask = __ballot(!ok);
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
    return ld;
}

}  // namespace spkd
