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

// ---------------------------------------------------------------------------
// The elimination with the pivot chain of step K + 1 interleaved into step K.
//
// Per step the multipliers need 1 / a[K][K]: a DPP broadcast, v_rcp_f64, Newton steps,
// then l_i = -a[i][K] / a[K][K] -- about nine dependent instructions, ~100 cycles of
// latency in an in-order wave, 39 times per elimination.  The FMAs are opaque volatile
// statements (their DPP form does not exist for the compiler), so the scheduler cannot
// move that chain under them; left alone it sits between two steps and stalls the wave
// (measured: a pass ran at 55 - 67 % of its issue bound).  Here the chain for pivot K + 1
// starts as soon as column K + 1 of step K is final (it is updated first) and its
// instructions are dealt out between the remaining column updates of step K, which do
// not depend on it.  Steps with too few columns left run the rest of the chain back to back.
//   op 0      piv = bcast a[P][P]                       (inline asm: a DPP move)
//   op 1      det *= pivot of the CURRENT step, sign |= its sign bit
//   op 2      r = rcp(piv)
//   op 3..5   one cubic refinement step (below)
//   op 6..8   l_next[s] = -a[row of slot s][P] * r   (column P is final too)
// Ops 1..8 are ordinary C++ pinned in place by scheduling barriers, NOT inline asm: the
// compiler's hazard recogniser counts an inline-asm statement as zero wait states and
// assumes the worst of it, so every asm statement that reads a register written by an
// earlier asm statement with nothing but asm in between gets an s_nop in front (gfx950's
// dst-forwarding rule) -- as asm the chain cost ~400 such s_nop per elimination, 2.7
// cycles of issue each.  A compiler-visible instruction between two asm statements ends
// that look-back, which is also why op 1 sits between the broadcast and the rcp.
// A pivot that is not a positive finite number shows in the OR of the pivots' sign bits, or
// as a determinant that is NaN or infinite (a zero pivot gives an infinite reciprocal).
// ---------------------------------------------------------------------------
struct PivotChain {
    double piv, r, e;
};

// 1 / pivot: v_rcp_f64 is good to 4.6e-8 (tools/rcp_precision.hip); one cubic step
//     e = 1 - piv r ;  t = e + e^2 ;  r += r t         (r (1 + e + e^2), error e^3 ~ 1e-22)
// brings it to the last bit in three FMAs (two quadratic Newton steps take four).
constexpr int NEWTON_OPS = 3;
constexpr int CHAIN_OPS = 3 + NEWTON_OPS + QS;

struct DetAcc {
    double det;
    int sign;          // OR of the high words of the pivots
};

template <int P, int OP>
__device__ __forceinline__ void chain_op(QuadRows& q, PivotChain& ch, double (&ln)[QS], DetAcc& da, double piv_cur) {
    constexpr int S = P / QL, T = P % QL;
    if constexpr (OP == 0) {
        // DPP read of a[P][P], written by the first FMA of column P of the running step: two
        // wait states are needed in between.  QS - 1 - S further FMAs of that column follow
        // the write, and the compiler puts one s_nop 0 in front of this statement (see
        // above); the rest is supplied here.  tools/check_dpp_hazard.py checks the result.
        constexpr int have = (QS - 1 - S) + 1;
        if constexpr (P == 0 || have < 2)
            asm volatile("s_nop %3\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                         : "=v"(ch.piv) : "v"(q.r[S][P]), "n"(T), "n"(P == 0 ? 1 : 1 - have));
        else
            asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                         : "=v"(ch.piv) : "v"(q.r[S][P]), "n"(T));
    } else if constexpr (OP == 1) {
        if constexpr (P > 0) {
            da.det *= piv_cur;
            da.sign |= __double2hiint(piv_cur);
        }
    } else if constexpr (OP == 2) {
        ch.r = __builtin_amdgcn_rcp(ch.piv);
    } else if constexpr (OP < 3 + NEWTON_OPS) {
        if constexpr (OP == 3) ch.e = fma(-ch.piv, ch.r, 1.0);
        else if constexpr (OP == 4) ch.e = fma(ch.e, ch.e, ch.e);
        else ch.r = fma(ch.r, ch.e, ch.r);
    } else {
        constexpr int s = OP - 3 - NEWTON_OPS;
        if constexpr (s >= S) ln[s] = -q.r[s][P] * ch.r;
        else ln[s] = 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int P, int OP>
__device__ __forceinline__ void chain_rest(QuadRows& q, PivotChain& ch, double (&ln)[QS], DetAcc& da, double piv_cur) {
    if constexpr (OP < CHAIN_OPS) {
        chain_op<P, OP>(q, ch, ln, da, piv_cur);
        chain_rest<P, OP + 1>(q, ch, ln, da, piv_cur);
    }
}

// column J of step K (see TriCol), followed by the chain operations that are due
template <int K, int J, int OP>
struct TriColAhead {
    static __device__ __forceinline__ void run(QuadRows& q, const double (&l)[QS], PivotChain& ch, double (&ln)[QS],
                                               DetAcc& da, double piv_k) {
        if constexpr (J < D) {
            constexpr int SJ = J / QL, TJ = J % QL;
            fmac_bcast16<TJ, false>(q.r[SJ][J], q.r[SJ][K], l[SJ]);
#pragma unroll
            for (int s = SJ + 1; s < QS; ++s) fmac_bcast16<TJ, false>(q.r[s][J], q.r[SJ][K], l[s]);
            constexpr int first = K + 1;                       // the column that holds the next pivot
            constexpr int avail = D - 1 - first;               // columns behind it
            constexpr int gap = avail >= 2 * CHAIN_OPS ? 2 : 1;
            constexpr bool emit = OP < CHAIN_OPS && ((J - first) % gap == 0);
            if constexpr (emit) {
                __builtin_amdgcn_sched_barrier(0);
                chain_op<K + 1, OP>(q, ch, ln, da, piv_k);
            }
            TriColAhead<K, J + 1, emit ? OP + 1 : OP>::run(q, l, ch, ln, da, piv_k);
        } else {
            __builtin_amdgcn_sched_barrier(0);
            chain_rest<K + 1, OP>(q, ch, ln, da, piv_k);       // what did not fit between the columns
        }
    }
};

// A hook runs between the steps of an elimination: after_step<K>() is called when step K is
// complete, i.e. when column K of the matrix (13, 26 or 39 rows of it, by slot) is dead and
// its registers are free.  The kernels use it to load column K of the NEXT pass's record
// into exactly those registers (PackedColumns below): a wave then never waits for a pass's
// record loads, they are in flight under the previous pass's arithmetic.
struct NoHook {
    template <int K> __device__ __forceinline__ void after_step() {}
    __device__ __forceinline__ void redo() {}
};

template <int K>
struct TriStepAhead {
    // l: multipliers of step K (ready); piv_k: its pivot (for the determinant)
    template <class Hook>
    static __device__ __forceinline__ void run(QuadRows& q, DetAcc& da, const double (&l)[QS], double piv_k, Hook& hook) {
        if constexpr (K + 1 < D) {
            PivotChain ch;
            double ln[QS];
            TriColAhead<K, K + 1, 0>::run(q, l, ch, ln, da, piv_k);
            hook.template after_step<K>();
            TriStepAhead<K + 1>::run(q, da, ln, ch.piv, hook);
        } else {
            da.det *= piv_k;
            da.sign |= __double2hiint(piv_k);
            hook.template after_step<K>();
        }
    }
};

// det (per DPP row) of four symmetric matrices given by their lower triangles.
// false: some pivot was not a positive finite number (or the product left the range).
template <class Hook>
__device__ __forceinline__ bool tri_det_nopivot(QuadRows& q, double& det_out, Hook& hook) {
    DetAcc da;
    da.det = 1.0;
    da.sign = 0;
    PivotChain ch;
    double l0[QS];
    __builtin_amdgcn_sched_barrier(0);
    chain_rest<0, 0>(q, ch, l0, da, 1.0);                      // the first pivot's chain: nothing to hide it under
    TriStepAhead<0>::run(q, da, l0, ch.piv, hook);
    det_out = da.det;
    return (da.sign >= 0) && (da.det == da.det) && (da.det < __builtin_huge_val());
}

__device__ __forceinline__ bool tri_det_nopivot(QuadRows& q, double& det_out) {
    NoHook h;
    return tri_det_nopivot(q, det_out, h);
}

// The hook the kernels use: the lower-triangle columns of a PACKED record (SPKD_REC doubles:
// column j holds rows j .. 39 contiguously from pk_off(j), so the quad load of (slot s,
// column j) is 13 consecutive doubles at pk_off(j) + 13 s - j + t; lanes of a diagonal block
// that sit above the diagonal read the previous column's tail, inside the record, into
// registers nobody reads) -> `dst`, column K after step K; the sums column (three more
// doubles per lane) comes with column SUMS_AT, late enough to cost no register while the
// matrix is still large and early enough to have landed when the elimination ends.
// rt0 = record + t (t = min(lane in the DPP row, 12)), rt1 = rt0 + 512: the immediate offset
// of a global load spans 4 KB, so two bases reach the whole record.
struct PackedColumns {
    static constexpr int SUMS_AT = 2 * QL;
    const SPKD_GLOBAL double* rt0;
    const SPKD_GLOBAL double* rt1;
    QuadRows* dst;
    double* sums;         // [QS]

    __device__ __forceinline__ void set_record(const double* rec, int t12) {
        long long o1 = 512;                 // opaque, so that the bases stay separate registers
        asm volatile("" : "+v"(o1));
        rt0 = (const SPKD_GLOBAL double*)rec + t12;
        rt1 = rt0 + o1;
    }
    template <int J>
    __device__ __forceinline__ void column() {
#pragma unroll
        for (int s = J / QL; s < QS; ++s) {
            const int e = pk_off(J) + QL * s - J;       // + t (in the base)
            dst->r[s][J] = e < 512 ? rt0[e] : rt1[e - 512];
        }
    }
    // (39, c) for this lane's row c = 13 s + t of slot s: record[pk_off(c) + 39 - c]
    __device__ __forceinline__ void sums_column() {
        int t = lane_id() & 15;
        t = t < QL ? t : QL - 1;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const int c = QL * s + t;
            sums[s] = rt0[pk_off(c) + D - c - t];
        }
    }
    template <int K>
    __device__ __forceinline__ void after_step() {
        column<K>();
        if constexpr (K == SUMS_AT) sums_column();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void redo() { all(); }
    // the whole record at once (the first pass of a loop: nothing to hide the loads under)
    template <int J = 0>
    __device__ __forceinline__ void all() {
        if constexpr (J < D) {
            column<J>();
            all<J + 1>();
        } else {
            sums_column();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

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

// det of the four lower triangles held in q (per lane: its own matrix).
// Matrices that meet a pivot that is not a positive finite number (this is also how
// NaN / inf entries surface) are redone one at a time in the row-per-lane layout
// with partial pivoting: form_single(mi, a) must fill matrix mi there.
// The LOG is left to the caller: a wave holds four determinants, so a log taken here costs
// a full instruction sequence per four values; the kernels take it later, in their dense
// one-thread-per-candidate phases (same value, same rounding).
template <class FormSingle, class Hook>
__device__ __forceinline__ double tri_det(QuadRows& q, int m, int* err, FormSingle form_single, Hook& hook) {
    double det;
    const bool ok = tri_det_nopivot(q, det, hook);
    const unsigned long long badmask = __ballot(!ok);
    if (badmask) {
#pragma unroll 1
        for (int mi = 0; mi < 4; ++mi) {
            if (((badmask >> (16 * mi)) & 0xffffull) == 0ull) continue;   // wave-uniform
            double a[DA];
            form_single(mi, a);
            const double v = det_pivoted_fn(a, err);
            if (m == mi) det = v;
        }
        // what the hook has loaded is loaded again: nothing of it has to survive the calls
        // above (156 registers would, in scratch, with the stores on the fast path)
        hook.redo();
    }
    return det;
}

template <class FormSingle>
__device__ __forceinline__ double tri_det(QuadRows& q, int m, int* err, FormSingle form_single) {
    NoHook h;
    return tri_det(q, m, err, form_single, h);
}

template <class FormSingle>
__device__ __forceinline__ double tri_logdet(QuadRows& q, int m, int* err, FormSingle form_single) {
    return log(tri_det(q, m, err, form_single));
}

}  // namespace spkd
