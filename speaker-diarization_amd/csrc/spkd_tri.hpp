// Symmetric ("triangular") quad elimination: the quad layout of spkd_quad.hpp
// (four 39x39 matrices per wave64, lane t of a DPP row holds rows t, 13 + t, 26 + t)
// keeping only the LOWER triangle: slot s needs columns 0 .. 13 (s + 1) - 1, i.e.
// 13 + 26 + 39 = 78 doubles per lane instead of 117.  Step k updates
//     a[i][j] -= (a[i][k] / a[k][k]) * a[j][k]        for k < j <= i
// where a[j][k] (row j, column k: lower triangle, not yet touched by step k) is
// broadcast from its owner lane j % 13 / slot j / 13 with one row_newbcast DPP
// move -- the symmetric counterpart of broadcasting the pivot row.  Lanes whose
// rows lie above column j compute into registers nobody reads.  The broadcast is the
// DPP operand of the FMA itself (v_fmac_f64_dpp): 1 144 instructions per four
// determinants (full square form: 1 820 FMAs + 741 DPP moves), and 156 VGPRs of
// matrix instead of 234: two waves per SIMD without column blocking.
#pragma once
#include "spkd_quad.hpp"

namespace spkd {

// columns held by slot s: 0 .. tri_cols(s) - 1
__host__ __device__ constexpr int tri_cols(int s) { return QL * (s + 1); }

// acc += bcast(src, lane T of the DPP row) * m as ONE instruction: gfx950 has the DPP
// form of v_fmac_f64 for row_newbcast, which the compiler never selects (it emits a
// v_mov_b64_dpp and a plain FMA: 780 extra instructions per elimination).
// Hazard: a VALU write of a VGPR needs two wait states before a DPP read of it, and
// the compiler's hazard recognizer does not look inside inline asm, so the first
// read of a source in a run carries its own s_nop.  The statements are volatile:
// their program order is the order the hazard analysis below relies on.
template <int T, bool NOP>
__device__ __forceinline__ void fmac_bcast16(double& acc, double src, double m) {
    if constexpr (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(src), "v"(m), "n"(T));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(src), "v"(m), "n"(T));
}

// broadcast with its own wait states (the source may have been written by the asm
// FMA just before, which the compiler does not see as a VALU write)
template <int T, bool NOP = true>
__device__ __forceinline__ double bcast16_nop(double v) {
    double r;
    if constexpr (NOP)
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                     : "=v"(r) : "v"(v), "n"(T));
    else
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                     : "=v"(r) : "v"(v), "n"(T));
    return r;
}

// column J of step K: a[i][J] += l_i * a[J][K].  The DPP sources of step K are the
// column-K registers, last written by the FIRST column of step K - 1; every FMA of step
// K comes after the step's pivot broadcast statement (an s_nop 1 + a DPP move, in program
// order because both are volatile), i.e. more than two wait states after that write, so
// the column FMAs carry no s_nop of their own.  tools/check_dpp_hazard.py verifies the
// rule on the disassembly of the built library (the compiler could in principle copy a
// source register right in front of a statement).  -DSPKD_TRI_COLNOP=1 restores a wait
// in front of every column (the round-1 form).
#ifndef SPKD_TRI_COLNOP
#define SPKD_TRI_COLNOP 0
#endif
template <int K, int J>
struct TriCol {
    static __device__ __forceinline__ void run(QuadRows& q, const double (&l)[QS]) {
        if constexpr (J < D) {
            constexpr int SJ = J / QL, TJ = J % QL;
            fmac_bcast16<TJ, SPKD_TRI_COLNOP != 0>(q.r[SJ][J], q.r[SJ][K], l[SJ]);
#pragma unroll
            for (int s = SJ + 1; s < QS; ++s) fmac_bcast16<TJ, false>(q.r[s][J], q.r[SJ][K], l[s]);
            TriCol<K, J + 1>::run(q, l);
        }
    }
};

// Step K.  The pivot a[K][K] was last written by the first column of step K - 1; the
// remaining columns of that step (two instructions or more while K <= 36) separate that
// write from the broadcast below, so only the first step (its source comes straight out
// of the formation code) and the last two carry wait states of their own.
// The smallest pivot is tracked instead of a per-step test: a pivot that is not a
// positive finite number shows as pmin <= 0 or as a determinant that is NaN or infinite
// (v_min_f64 drops a NaN operand, the product keeps it).
template <int K>
struct TriStep {
    static __device__ __forceinline__ void run(QuadRows& q, double& det, double& pmin) {
        if constexpr (K < D) {
            constexpr int S = K / QL, T = K % QL;
            const double piv = bcast16_nop<T, (K == 0 || K > 36)>(q.r[S][K]);
            asm("v_min_f64 %0, %1, %2" : "=v"(pmin) : "v"(pmin), "v"(piv));    // (fmin() adds a canonicalisation)
            det *= piv;
            const double inv = fast_recip(piv);
            double l[QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) l[s] = (s >= S) ? -(q.r[s][K] * inv) : 0.0;
            TriCol<K, K + 1>::run(q, l);
            TriStep<K + 1>::run(q, det, pmin);
        }
    }
};

// det (per DPP row) of four symmetric matrices given by their lower triangles.
// false: some pivot was not a positive finite number (or the product left the range).
__device__ __forceinline__ bool tri_det_nopivot(QuadRows& q, double& det_out) {
    double det = 1.0;
    double pmin = __builtin_huge_val();
    TriStep<0>::run(q, det, pmin);
    det_out = det;
    return (pmin > 0.0) && (det == det) && (det < __builtin_huge_val());
}

// q[s][J] += c[s] * v_J for the lower-triangle columns (v distributed like the rows)
template <int J>
struct TriRank1 {
    static __device__ __forceinline__ void run(QuadRows& q, const double (&c)[QS], const double (&v)[QS]) {
        if constexpr (J < D) {
            constexpr int SJ = J / QL, TJ = J % QL;
            fmac_bcast16<TJ, true>(q.r[SJ][J], v[SJ], c[SJ]);
#pragma unroll
            for (int s = SJ + 1; s < QS; ++s) fmac_bcast16<TJ, false>(q.r[s][J], v[SJ], c[s]);
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
