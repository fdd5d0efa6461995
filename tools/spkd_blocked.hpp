// Column-blocked quad elimination: same arithmetic as spkd_quad.hpp (four 39x39
// matrices per wave, DPP row broadcast), restructured so that a wave never holds a
// whole matrix in registers:
//   phase A  panel: columns 0..22 formed and eliminated in registers; the
//            multipliers overwrite the eliminated entries (69 doubles);
//   phase B  the 16 trailing columns stream through in chunks of 4: form, apply
//            the 23 panel steps, write the Schur-complement rows 23..38 to a
//            per-wave LDS tile (the loads of the next chunk are in flight
//            meanwhile);
//   phase C  the 16x16 Schur complement, one row per lane, eliminated from LDS.
// Peak ~190 VGPRs instead of ~300, i.e. two waves per SIMD instead of one, which is
// what hides the memory / LDS latency the one-wave form leaves exposed.
//
#pragma once
#include "quad_square.hpp"
#ifdef SPKD_NO_SB
#define SPKD_SB() ((void)0)
#else
#define SPKD_SB() __builtin_amdgcn_sched_barrier(0)
#endif

namespace spkd {

constexpr int BP = 23;              // panel width
constexpr int BT = D - BP;          // trailing columns = Schur dimension (16)
constexpr int BW = 4;               // chunk width

// ---- phase A step K restricted to the panel columns
template <int K>
struct PanelStep {
    static __device__ __forceinline__ void run(double (&p)[QS][BP], double& det, bool& ok) {
        constexpr int S = K / QL, T = K % QL;
        const double piv = bcast16<T>(p[S][K]);
        ok = ok && (piv > 0.0) && (piv < __builtin_huge_val());
        det *= piv;
        const double inv = fast_recip(piv);
        double l[QS];
#pragma unroll
        for (int s = S; s < QS; ++s) l[s] = -(p[s][K] * inv);
#pragma unroll
        for (int j = K + 1; j < BP; ++j) {
            const double u = bcast16<T>(p[S][j]);
#pragma unroll
            for (int s = S; s < QS; ++s) p[s][j] = fma(l[s], u, p[s][j]);
        }
#pragma unroll
        for (int s = S; s < QS; ++s) p[s][K] = l[s];      // keep the multipliers
        PanelStep<K + 1>::run(p, det, ok);
    }
};
template <>
struct PanelStep<BP> {
    static __device__ __forceinline__ void run(double (&)[QS][BP], double&, bool&) {}
};

// ---- phase B: apply panel step K to a chunk of W trailing columns
template <int K, int W>
struct ApplyStep {
    static __device__ __forceinline__ void run(const double (&p)[QS][BP], double (&t)[QS][W]) {
        constexpr int S = K / QL, T = K % QL;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const double u = bcast16<T>(t[S][w]);
#pragma unroll
            for (int s = S; s < QS; ++s) t[s][w] = fma(p[s][K], u, t[s][w]);
        }
        ApplyStep<K + 1, W>::run(p, t);
    }
};
template <int W>
struct ApplyStep<BP, W> {
    static __device__ __forceinline__ void run(const double (&)[QS][BP], double (&)[QS][W]) {}
};

// ---- phase C: 16x16, lane r of each DPP row holds row r
template <int K>
struct SchurStep {
    static __device__ __forceinline__ void run(double (&c)[BT], double& det, bool& ok) {
        const double piv = bcast16<K>(c[K]);
        ok = ok && (piv > 0.0) && (piv < __builtin_huge_val());
        det *= piv;
        const double l = -(c[K] * fast_recip(piv));
#pragma unroll
        for (int j = K + 1; j < BT; ++j) c[j] = fma(l, bcast16<K>(c[j]), c[j]);
        SchurStep<K + 1>::run(c, det, ok);
    }
};
template <>
struct SchurStep<BT> {
    static __device__ __forceinline__ void run(double (&)[BT], double&, bool&) {}
};

// The matrix is described by a "source" with two compile-time-indexed methods:
//   src.template load<J>(col)     col[3] = the slow (global memory) part of column J
//   src.template finish<J>(col)   col[3] = the finished column J, in place
// so that many loads can be in flight before the first finish.
constexpr int BTP = BT + 1;         // padded row stride of the LDS Schur tile
constexpr int SCHUR_TILE = 4 * (BT + 1) * BTP;   // + one dump row per matrix for idle lanes

template <int J, int JEND, class Src, int NC>
struct LoadCols {
    static __device__ __forceinline__ void run(Src& src, double (&a)[QS][NC], int j0) {
        if constexpr (J < JEND) {
            double col[QS];
            src.template load<J>(col);
#pragma unroll
            for (int s = 0; s < QS; ++s) a[s][J - (JEND - NC)] = col[s];
            LoadCols<J + 1, JEND, Src, NC>::run(src, a, j0);
        }
    }
};

template <int J, int JEND, class Src, int NC>
struct FinishCols {
    static __device__ __forceinline__ void run(Src& src, double (&a)[QS][NC]) {
        if constexpr (J < JEND) {
            double col[QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) col[s] = a[s][J - (JEND - NC)];
            src.template finish<J>(col);
#pragma unroll
            for (int s = 0; s < QS; ++s) a[s][J - (JEND - NC)] = col[s];
            if constexpr ((J & 3) == 3) SPKD_SB();   // <= 4 columns of LDS reads in flight
            FinishCols<J + 1, JEND, Src, NC>::run(src, a);
        }
    }
};

template <int C, class Src>
struct TrailChunks {
    // chunk C covers columns BP + 4 C .. BP + 4 C + 3; tt arrives with its loads issued
    static __device__ __forceinline__ void run(Src& src, const double (&p)[QS][BP],
                                               double (&tt)[QS][BW], double* schur, int m, int t) {
        if constexpr (C < BT / BW) {
            constexpr int J0 = BP + BW * C;
            double tn[QS][BW];
            if constexpr (C + 1 < BT / BW)           // next chunk's slow loads go out now
                LoadCols<J0 + BW, J0 + 2 * BW, Src, BW>::run(src, tn, 0);
            SPKD_SB();       // (barriers: keep the phases from being
            FinishCols<J0, J0 + BW, Src, BW>::run(src, tt);   //  merged into one big load cluster)
            SPKD_SB();
            ApplyStep<0, BW>::run(p, tt);
            SPKD_SB();
            // rows 23..38 of these columns -> LDS tile [matrix][row - 23][col - 23]
            // slot 1 holds rows 13..25 (t >= 10 -> rows 23..25), slot 2 rows 26..38
            // branch-free: lanes that hold no Schur row write to the dump row
            const int r1 = (t >= BP - QL && t < QL) ? (QL + t - BP) : BT;
            const int r2 = (t < QL) ? (2 * QL + t - BP) : BT;
            double* d1 = schur + (m * (BT + 1) + r1) * BTP + (J0 - BP);
            double* d2 = schur + (m * (BT + 1) + r2) * BTP + (J0 - BP);
#pragma unroll
            for (int w = 0; w < BW; ++w) {
                d1[w] = tt[1][w];
                d2[w] = tt[2][w];
            }
            SPKD_SB();
            if constexpr (C + 1 < BT / BW) TrailChunks<C + 1, Src>::run(src, p, tn, schur, m, t);
        }
    }
};

// log det (per DPP row) of the four matrices described by src.  schur: per-wave LDS
// tile of SCHUR_TILE doubles.  ok_out: all pivots positive finite.
template <class Src>
__device__ __forceinline__ double quad_logdet_blocked(Src& src, double* schur, const QuadLane& L,
                                                      bool& ok_out) {
    double det = 1.0;
    bool ok = true;
    {
        double p[QS][BP];
        LoadCols<0, BP, Src, BP>::run(src, p, 0);        // 69 loads in flight, one latency
        double tt[QS][BW];
        LoadCols<BP, BP + BW, Src, BW>::run(src, tt, 0);
        SPKD_SB();
        FinishCols<0, BP, Src, BP>::run(src, p);
        SPKD_SB();
        PanelStep<0>::run(p, det, ok);
        SPKD_SB();
        TrailChunks<0, Src>::run(src, p, tt, schur, L.m, L.t);
    }
    SPKD_SB();
    // phase C (the same wave wrote the tile: LDS operations of a wave are in order)
    double c[BT];
#pragma unroll
    for (int j = 0; j < BT; ++j) c[j] = schur[(L.m * (BT + 1) + L.t) * BTP + j];
    SchurStep<0>::run(c, det, ok);
    ok_out = ok;
    return log(det);
}

}  // namespace spkd
