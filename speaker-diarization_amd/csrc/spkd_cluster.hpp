// Pair distances and agglomerative clustering on packed statistics records.
//
//   k_pair_terms   : one wave per (set A, set B) job -> n, log det terms, KL2
//                    (bodies of bic/glr/kl2: spk-clustering.py:81-133,
//                    spk-change-detection.py:72-133)
//   k_cluster_prep : per record, log det S and the KL2 vectors
//   k_matrix       : initial N x N distance matrix of spk_cluster_hi
//                    (spk-clustering.py:188-200, spk-clustering2.py:178-184)
//   k_ahc          : the merge loop (spk-clustering.py:201-240,
//                    spk-clustering2.py:185-222), one workgroup per problem,
//                    device resident: arg-min with numpy semantics, statistics
//                    merge, row recompute, v1 / v2 matrix update rules.
//   k_ahc_update / k_ahc_select / k_ahc_pairs / k_ahc_final : the same loop as a chain
//                    of chip-wide launches, for calls with few (long) problems.
//
// Per evaluated pair the algorithmic traffic is two records in, one double out
// = 13 128 B; one factorisation per pair (the union), because
// the per-cluster log det terms are cached.
#pragma once
#include "spkd_device.hpp"
#include "spkd_quad.hpp"
#include "spkd_tri.hpp"
#include "../../include/spkd.h"

namespace spkd {

constexpr double MAXINT_F = 9223372036854775808.0;     // float(sys.maxint) = 2^63
constexpr double PEN_UNIT = 39.0 + 0.5 * 39.0 * 40.0;  // p + 0.5 p (p + 1), p = 39
constexpr int AUX = 3 * DA;                            // diag S | diag pinv S | mean (as f32 value)
constexpr int ERR_DEGENERATE_MERGE = 2;

__device__ __forceinline__ unsigned long long dkey(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dkey_inv(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
// Wave-wide integer reductions without LDS: four DPP steps inside each row of 16 lanes (min and
// max are idempotent: quad_perm xor 1, xor 2, row_half_mirror, row_mirror leave the row's result
// in every lane), then the four rows by v_readlane and scalar arithmetic.  The result is uniform.
// (The selection of k_ahc_step used 64-bit __shfl_xor butterflies with fp64 compares: six
// dependent steps of ~14 ds_bpermute each, 3.8 k cycles per wave; these take ~150.)
template <bool MAXOP>
__device__ __forceinline__ unsigned wave_red_u32(unsigned x) {
    auto op = [](unsigned a, unsigned b) { return MAXOP ? (a > b ? a : b) : (a < b ? a : b); };
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));   // row_half_mirror
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));   // row_mirror
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)x, 0), b = (unsigned)__builtin_amdgcn_readlane((int)x, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)x, 32), d = (unsigned)__builtin_amdgcn_readlane((int)x, 48);
    return op(op(a, b), op(c, d));
}
// inclusive prefix sum over the wave: four shifts inside a row of 16 (zeros shifted in), then
// lane 15 of rows 0 and 2 into rows 1 and 3, lane 31 into rows 2 and 3
__device__ __forceinline__ int wave_scan_i32(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return x;
}
template <bool MAXOP>
__device__ __forceinline__ unsigned long long wave_red_u64(unsigned long long k) {
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned h = wave_red_u32<MAXOP>(hi);
    const unsigned l = wave_red_u32<MAXOP>(hi == h ? lo : (MAXOP ? 0u : 0xffffffffu));
    return ((unsigned long long)h << 32) | l;
}
// the smallest (key, idx) pair of the wave, lexicographically
__device__ __forceinline__ void wave_argmin_key(unsigned long long key, unsigned idx,
                                                unsigned long long& okey, unsigned& oidx) {
    okey = wave_red_u64<false>(key);
    oidx = wave_red_u32<false>(key == okey ? idx : 0xffffffffu);
}

__device__ __forceinline__ bool stat_valid(double d) {       // "d != inf and d != -inf", NaN never updates
    return d == d && fabs(d) < __builtin_huge_val();
}

// KL2 "as coded" (elementwise products, only diagonals contribute; means rounded
// to float32 like np.mean of float32 frames): spk-clustering.py:124-133.
__device__ __forceinline__ double kl2_combine(double s1, double p1, double m1,
                                              double s2, double p2, double m2);
__device__ __forceinline__ double kl2_from_aux(const double* __restrict__ a1,
                                               const double* __restrict__ a2) {
    int i = lane_id();
    i = i >= D ? D - 1 : i;
    return kl2_combine(a1[i], a1[DA + i], a1[2 * DA + i], a2[i], a2[DA + i], a2[2 * DA + i]);
}

// per-lane KL2 ingredients of one record held as covariance rows in a (consumed):
// ds = S_ii, dp = pinv(S)_ii (inverse; NaN when S is not positive definite),
// mu = mean_i rounded to float32.
__device__ __forceinline__ void kl2_lane_terms(double (&a)[DA], double mean_i,
                                               double& ds, double& dp, double& mu) {
    double out[2];
    spd_diag_terms_fn(a, out);
    ds = out[0];
    dp = out[1];
    mu = (double)(float)mean_i;
}

__device__ __forceinline__ double kl2_combine(double s1, double p1, double m1,
                                              double s2, double p2, double m2) {
    const float dm = (float)m1 - (float)m2;
    const double delta = (double)dm;
    const double t1 = wave_sum39((s1 - s2) * (p2 - p1));
    const double t2 = wave_sum39(((p1 + p2) * delta) * delta);
    return 0.5 * t1 + 0.5 * t2;
}

__device__ __forceinline__ void kl2_aux_from_cov(double (&a)[DA], double mean_i,
                                                 double* __restrict__ aux) {
    double ds, dp, mu;
    kl2_lane_terms(a, mean_i, ds, dp, mu);
    const int lane = lane_id();
    if (lane < D) {
        aux[lane] = ds;
        aux[DA + lane] = dp;
        aux[2 * DA + lane] = mu;
    }
}

// ---------------------------------------------------------------------------
constexpr int PT_WAVES = 4;

// Packed -> quad records (see spkd_quad.hpp).  The packed form (6 560 B) stays the
// ABI format; the clustering kernels keep their working set as quad records.
__global__ __launch_bounds__(256) void k_to_quadrec(const double* __restrict__ packed, int64_t n_rec,
                                                    double* __restrict__ qr) {
    const int64_t c = blockIdx.x;
    if (c >= n_rec) return;
    const double* g = packed + c * REC;
    double* o = qr + c * QREC;
    for (int e = threadIdx.x; e < QREC; e += 256) {
        const int t = e & 15, sj = e >> 4;
        const int s = sj / DA, j = sj - s * DA;
        double v = 0.0;
        if (t < QL) {
            const int i = QL * s + t;
            const int r = i < j ? i : j, cc = i < j ? j : i;
            v = g[pk(r, cc)];
        } else if (e == QREC_COUNT_AT) {
            v = g[REC - 1];
        }
        o[e] = v;
    }
}

// row-per-lane (single matrix) rows of a packed record, by symmetry (k_gw's cache records)
__device__ __forceinline__ void single_rows_from_packed(const double* __restrict__ rec, double (&q)[DA]) {
    int i = lane_id();
    i = i >= D ? D - 1 : i;           // lanes >= 39 mirror row 38 (ignored)
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] = rec[j <= i ? pk_low(i, j) : pk_low(j, i)];
    q[D] = rec[pk_low(D, i)];
}

// row-per-lane (single matrix) forms used by the pivoting fallback and by KL2
__device__ __forceinline__ void single_rows_from_qr(const double* __restrict__ qr, double (&q)[DA]) {
    int i = lane_id();
    i = i >= D ? D - 1 : i;           // lanes >= 39 mirror row 38 (ignored)
    const int base = (i / QL) * DA * 16 + (i % QL);
#pragma unroll
    for (int j = 0; j < DA; ++j) q[j] = qr[base + j * 16];
}

// matrix whose log det a pair distance needs, row-per-lane layout
__device__ __forceinline__ void single_pair_matrix(int kind, const double* __restrict__ qrA,
                                                   const double* __restrict__ qrC, bool self,
                                                   double (&q)[DA]) {
    const double nA = qr_count(qrA);
    single_rows_from_qr(qrA, q);
    if (self) { cov_rows(q, nA); return; }
    const double nC = qr_count(qrC), n = nA + nC;
    double q2[DA];
    single_rows_from_qr(qrC, q2);
    if (kind == SPKD_BIC) {
#pragma unroll
        for (int j = 0; j < DA; ++j) q[j] += q2[j];
        cov_rows(q, n);
    } else {
        const double al1 = (nA / n) / (nA - 1.0), al2 = (nC / n) / (nC - 1.0);
        const double be1 = al1 / nA, be2 = al2 / nC;
        const double s1i = q[D], s2i = q2[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double s1j = readlane_d(s1i, j), s2j = readlane_d(s2i, j);
            double v = fma(al2, q2[j], al1 * q[j]);
            v = fma(-(be1 * s1i), s1j, v);
            q[j] = fma(-(be2 * s2i), s2j, v);
        }
    }
}

// log det of the matrix a pair distance needs, for FOUR partner records at once
// (DPP row m: partner qrC, or the cluster itself when self).  BIC: covariance of
// the union; GLR: (nA S_A + nC S_C) / N.  Lower triangle only (spkd_tri.hpp):
//   entry (i, j) = wa * A_ij + wc * C_ij + k1 v1_i v1_j + k2 v2_i v2_j
// with the covariance scale folded into wa, wc, k1 (BIC: v1 = sums of the union,
// k1 = -f / n, k2 = 0;  GLR: v1 = sums of A, v2 = sums of C).  The second rank-one
// term is a compile-time switch (a run-time branch per column wrecks the register
// allocation).
// ldsA: quad record of cluster A staged in LDS (all four matrices share it);
// gA: the same record in global memory (fallback path only).
// pkC: the same partner record in the packed ABI layout (6 560 B) -- the hot loads read
// that copy: by symmetry a packed record is the lower triangle column by column, so the
// load of (slot s, column j) is 13 consecutive doubles at pk_off(j) + 13 s - j + t (lanes of
// a diagonal block above the diagonal read the previous column's tail: inside the record,
// into registers nobody reads).  A pass of the merge loop streams its partners from HBM
// (a problem's records exceed its share of L2): tools/pair_bench.hip, partners streaming,
// 559 M pairs/s from quad records, 686 M pairs/s from packed ones.  The quad copy (qrC)
// serves the rare pivoting fallback and the in-place merges' LDS staging.
#ifdef SPKD_PROFILE
// profiling builds: cycles a wave waits for a pass's record loads, cycles of whole passes, passes
__device__ unsigned long long g_pass_prof[4];
#define PASS_T0() const unsigned long long pass_t0_ = clock64()
#define PASS_LOADED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long pass_t1_ = clock64()
#define PASS_DONE() do { if (lane_id() == 0 && ((pass_t0_ >> 4) & 63) == 0) { const unsigned long long pass_t2_ = clock64(); /* 1 pass in 64 */ \
        atomicAdd(&g_pass_prof[0], pass_t1_ - pass_t0_); atomicAdd(&g_pass_prof[1], pass_t2_ - pass_t0_); \
        atomicAdd(&g_pass_prof[2], 1ull); } } while (0)
#else
#define PASS_T0() ((void)0)
#define PASS_LOADED() ((void)0)
#define PASS_DONE() ((void)0)
#endif

template <bool TWO>
__device__ __forceinline__ double quad_pair_det(int kind, const double* ldsA, double nA,
                                                   const double* __restrict__ gA,
                                                   const double* __restrict__ qrC,
                                                   const double* __restrict__ pkC, bool self,
                                                   const QuadLane& L, int* err, double nC_known = -1.0) {
    // (nC_known >= 0: the caller has the partner's frame count already -- the step chain keeps
    // the counts in an array of their own, so that a pass does not wait for the count before it
    // can issue its 81 record loads: two memory round trips in a row on the chain's critical path)
    const double nC = self ? 0.0 : (nC_known >= 0.0 ? nC_known : pkC[REC - 1]);
    const double n = nA + nC;
    const bool glr = TWO && (kind == SPKD_GLR && !self);
    // (an IEEE fp64 division is ~25 instructions; the covariance scale uses the Newton
    // reciprocal, the GLR weights exist only in the GLR instantiation)
    const double f = fast_recip(n - 1.0);
    double wa = f, wc = self ? 0.0 : f;
    double k1 = -(f * fast_recip(n)), k2 = 0.0;
    if constexpr (TWO) {
        if (glr) {
            wa = (nA / n) / (nA - 1.0);
            wc = (nC / n) / (nC - 1.0);
            k1 = -(wa / nA);
            k2 = -(wc / nC);
        }
    }
    int ta = L.t;
    asm volatile("" : "+v"(ta));          // keeps A's LDS reads inside the caller's loop
    QuadRows q;
    double sc[QS];
    PASS_T0();
    {
        // 81 loads in flight, one latency; one base pointer per 4 KB of the record (the
        // immediate offset of a global load spans 4 KB)
        const int t12 = ta < QL ? ta : QL - 1;          // idle lanes 13..15 ride with lane 12
        const double* rt[2];
        long long o1 = 512;                 // opaque, so that the bases stay separate registers
        asm volatile("" : "+v"(o1));
        rt[0] = pkC + t12;
        rt[1] = rt[0] + o1;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
#pragma unroll
            for (int j = 0; j < tri_cols(s); ++j) {
                const int e = pk_off(j) + QL * s - j;   // + t12 (in the base)
                q.r[s][j] = rt[e / 512][e % 512];
            }
            const int c = QL * s + t12;                 // this lane's row of slot s
            sc[s] = pkC[pk_off(c) + D - c];             // (39, c): the sums entry of column c
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    PASS_LOADED();
    double v1[QS], v2[QS], c1[QS], c2[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
#pragma unroll
        for (int j = 0; j < tri_cols(s); ++j)
            q.r[s][j] = fma(wa, ldsA[(s * DA + j) * 16 + ta], wc * q.r[s][j]);
        const double sa = ldsA[(s * DA + D) * 16 + ta];
        const double scs = self ? 0.0 : sc[s];
        v1[s] = glr ? sa : sa + scs;
        v2[s] = glr ? scs : 0.0;
        c1[s] = k1 * v1[s];
        c2[s] = k2 * v2[s];
        __builtin_amdgcn_sched_barrier(0);
    }
    TriRank1<0>::run(q, c1, v1);
    if constexpr (TWO) TriRank1<0>::run(q, c2, v2);
    auto form_single = [&](int mi, double (&a)[DA]) {
        // (the record and the flag of DPP row mi, from its first lane: no per-pass arrays,
        // which live in scratch and cost 2.3 KB of stores per wave pass)
        const double* rm = (const double*)__shfl((unsigned long long)qrC, 16 * mi);
        single_pair_matrix(kind, gA, rm, __shfl((int)self, 16 * mi) != 0, a);
    };
    const double det = tri_det(q, L.m, err, form_single);
    PASS_DONE();
    return det;
}

__device__ __forceinline__ void stage_record(double* lds, const double* __restrict__ g, int tid, int nthreads) {
    for (int e = tid; e < QREC; e += nthreads) lds[e] = g[e];
}

// ---------------------------------------------------------------------------
// Pair terms for the host-driven modes (merge_rec, spk_cluster_in with KL2): per pair the four
// log determinants a BIC or GLR distance is made of -- cov(A), cov(B), cov(A u B), and the
// count-weighted mean covariance -- as ONE wave pass: the four matrices of a pair are the four
// DPP rows of the quad elimination, all formed from A's record (staged in LDS in the quad
// layout) and B's (loaded like any partner record) with a row's own weights:
//     M = wa A + wc B + w1 v1 v1' + w2 v2 v2',   v1 = k1a s_A + k1c s_B,  v2 = k2a s_A + k2c s_B
//   row 0  cov(A):      wa = 1/(n1-1), wc = 0,        v1 = s_A,       w1 = -wa/n1
//   row 1  cov(B):      wa = 0,        wc = 1/(n2-1), v1 = s_B,       w1 = -wc/n2
//   row 2  cov(A u B):  wa = wc = 1/(n-1),            v1 = s_A + s_B, w1 = -wa/n
//   row 3  GLR:         wa = (n1/n)/(n1-1), wc = (n2/n)/(n2-1), v1 = s_A, w1 = -wa/n1,
//                       v2 = s_B, w2 = -wc/n2     (without SPKD_WANT_GLR: row 2 again)
// A matrix that meets a pivot that is not a positive finite number is redone with partial
// pivoting in the row-per-lane layout (tri_det), the definition the clustering kernels use.
// Until round 3 this kernel did the determinants one after the other in the row-per-lane
// layout (342 VGPRs, scratch): the last of its kind.  KL2 (pseudo-inverse diagonals) is
// row-per-lane work by nature and stays so.
// Two waves per workgroup, a pair each: A's and B's quad copies in LDS (2 x 10 KB a wave).
// ---------------------------------------------------------------------------
constexpr int PT2_WAVES = 2;

// the 81 loads of a partner record (packed layout) into the elimination's registers, as
// quad_pair_det issues them (one base pointer per 4 KB of the record)
__device__ __forceinline__ void quad_load_packed(const double* __restrict__ pkC, int ta, QuadRows& q, double (&sc)[QS]) {
    const int t12 = ta < QL ? ta : QL - 1;          // idle lanes 13..15 ride with lane 12
    const double* rt[2];
    long long o1 = 512;                 // opaque, so that the bases stay separate registers
    asm volatile("" : "+v"(o1));
    rt[0] = pkC + t12;
    rt[1] = rt[0] + o1;
#pragma unroll
    for (int s = 0; s < QS; ++s) {
#pragma unroll
        for (int j = 0; j < tri_cols(s); ++j) {
            const int e = pk_off(j) + QL * s - j;   // + t12 (in the base)
            q.r[s][j] = rt[e / 512][e % 512];
        }
        const int c = QL * s + t12;                 // this lane's row of slot s
        sc[s] = pkC[pk_off(c) + D - c];             // (39, c): the sums entry of column c
    }
}

__device__ __forceinline__ void quad_from_packed_lds(double* lds, const double* __restrict__ g, int lane) {
    for (int e = lane; e < QREC; e += WAVE) {             // (k_to_quadrec's map, by one wave)
        const int t = e & 15, sj = e >> 4;
        const int s = sj / DA, j = sj - s * DA;
        double v = 0.0;
        if (t < QL) {
            const int i = QL * s + t;
            const int r = i < j ? i : j, cc = i < j ? j : i;
            v = g[pk(r, cc)];
        } else if (e == QREC_COUNT_AT) {
            v = g[REC - 1];
        }
        lds[e] = v;
    }
}

__global__ __launch_bounds__(PT2_WAVES * WAVE) void k_pair_terms(
        const double* __restrict__ stats, const int32_t* __restrict__ ia,
        const int32_t* __restrict__ ib, int64_t n_pairs, int flags,
        double* __restrict__ out, int* err) {
    __shared__ double slabs[PT2_WAVES][2][QREC];
    const int wave = threadIdx.x >> 6;
    const int lane = lane_id();
    const int64_t pair = (int64_t)blockIdx.x * PT2_WAVES + wave;
    if (pair >= n_pairs) return;                           // (whole waves: no workgroup barrier below)
    double* ldsA = slabs[wave][0];
    double* ldsB = slabs[wave][1];
    const double* A = stats + (int64_t)ia[pair] * REC;
    const double* B = stats + (int64_t)ib[pair] * REC;
    const double n1 = A[REC - 1], n2 = B[REC - 1];
    const double n = n1 + n2;
    const QuadLane L = quad_lane();
    int ta = L.t;
    asm volatile("" : "+v"(ta));
    QuadRows q;
    double sc[QS];
    quad_load_packed(B, ta, q, sc);                        // (in flight while the quad copies are made)
    quad_from_packed_lds(ldsA, A, lane);
    quad_from_packed_lds(ldsB, B, lane);
    // this row's weights
    const bool want_glr = (flags & SPKD_WANT_GLR) != 0;
    const int mode = (L.m == 3 && !want_glr) ? 2 : L.m;
    double wa, wc, k1a, k1c, w1, k2c, w2;
    {
        const double fA = 1.0 / (n1 - 1.0), fB = 1.0 / (n2 - 1.0), fU = 1.0 / (n - 1.0);
        const double al1 = (n1 / n) / (n1 - 1.0), al2 = (n2 / n) / (n2 - 1.0);
        wa = mode == 0 ? fA : (mode == 1 ? 0.0 : (mode == 2 ? fU : al1));
        wc = mode == 0 ? 0.0 : (mode == 1 ? fB : (mode == 2 ? fU : al2));
        k1a = mode == 1 ? 0.0 : 1.0;
        k1c = (mode == 1 || mode == 2) ? 1.0 : 0.0;
        w1 = mode == 0 ? -(fA / n1) : (mode == 1 ? -(fB / n2) : (mode == 2 ? -(fU / n) : -(al1 / n1)));
        k2c = mode == 3 ? 1.0 : 0.0;
        w2 = mode == 3 ? -(al2 / n2) : 0.0;
    }
    double v1[QS], v2[QS], c1[QS], c2[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
#pragma unroll
        for (int j = 0; j < tri_cols(s); ++j)
            q.r[s][j] = fma(wa, ldsA[(s * DA + j) * 16 + ta], wc * q.r[s][j]);
        const double sa = ldsA[(s * DA + D) * 16 + ta];
        v1[s] = fma(k1c, sc[s], k1a * sa);
        v2[s] = k2c * sc[s];
        c1[s] = w1 * v1[s];
        c2[s] = w2 * v2[s];
        __builtin_amdgcn_sched_barrier(0);
    }
    TriRank1<0>::run(q, c1, v1);
    TriRank1<0>::run(q, c2, v2);
    auto form_single = [&](int mi, double (&a)[DA]) {
        const int md = (mi == 3 && !want_glr) ? 2 : mi;
        if (md == 0) single_pair_matrix(SPKD_BIC, ldsA, ldsA, true, a);
        else if (md == 1) single_pair_matrix(SPKD_BIC, ldsB, ldsB, true, a);
        else single_pair_matrix(md == 2 ? SPKD_BIC : SPKD_GLR, ldsA, ldsB, false, a);
    };
    const double ld = log(tri_det(q, L.m, err, form_single));
    double res[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) res[mi] = __shfl(ld, 16 * mi);
    double kl = __builtin_nan("");
    if (flags & SPKD_WANT_KL2) {
        double* slab = ldsA;                               // (REC <= QREC; the quad copies are done with)
        double a[DA];
        double ds[2], dp[2], mu[2];
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            stage1(slab, t ? B : A);
            row_from_slab(slab, a);
            const double nn = t ? n2 : n1;
            const double mean_i = a[D] / nn;
            cov_rows(a, nn);
            kl2_lane_terms(a, mean_i, ds[t], dp[t], mu[t]);
        }
        kl = kl2_combine(ds[0], dp[0], mu[0], ds[1], dp[1], mu[1]);
    }
    if (lane == 0) {
        double* o = out + pair * 8;
        o[0] = n1; o[1] = n2; o[2] = res[0]; o[3] = res[1]; o[4] = res[2];
        o[5] = want_glr ? res[3] : __builtin_nan("");
        o[6] = kl; o[7] = 0.0;
    }
}

// ---------------------------------------------------------------------------
// per-record cached terms: log det S (BIC / GLR) or the KL2 vectors
__global__ __launch_bounds__(PT_WAVES * WAVE) void k_cluster_prep(
        const double* __restrict__ qr, int64_t n_rec, int kind,
        double* __restrict__ ld, double* __restrict__ aux, int* err) {
    const int wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * PT_WAVES + wave;
    if (kind == SPKD_KL2) {
        if (w >= n_rec) return;
        const double* R = qr + w * QREC;
        double a[DA];
        single_rows_from_qr(R, a);
        const double n = qr_count(R);
        const double mean_i = a[D] / n;
        cov_rows(a, n);
        kl2_aux_from_cov(a, mean_i, aux + w * AUX);
        if (lane_id() == 0) ld[w] = 0.0;
        return;
    }
    if (w * 4 >= n_rec) return;
    const QuadLane L = quad_lane();
    const double* recs[4];
    bool selfs[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        int64_t c = w * 4 + mi;
        c = c < n_rec ? c : n_rec - 1;
        recs[mi] = qr + c * QREC;
        selfs[mi] = true;
    }
    int64_t c = w * 4 + L.m;
    const bool valid = c < n_rec;
    c = valid ? c : n_rec - 1;
    const double* R = qr + c * QREC;
    // every DPP row takes its own record as "A": formed in the single-matrix path
    // of quad_pair_logdet with self = true through a per-lane pointer
    QuadRows q;
    double sv[QS];
    const double nR = qr_count(R), fR = fast_recip(nR - 1.0);
    quad_load_scaled(R, L.t, fR, q, sv);
    double c1[QS];
    const double gR = -(fR * fast_recip(nR));
#pragma unroll
    for (int s = 0; s < QS; ++s) c1[s] = gR * sv[s];
    TriRank1<0>::run(q, c1, sv);
    auto form_single = [&](int mi, double (&a)[DA]) { single_pair_matrix(kind, recs[mi], recs[mi], true, a); };
    const double v = tri_logdet(q, L.m, err, form_single);
    if (valid && L.t == 0) ld[c] = v;
}

__device__ __forceinline__ double finish_distance(int kind, double lambdac, double nA, double ldA,
                                                  double nC, double ldC, double ldx) {
    // no contraction into FMAs here: the matrix kernel and the merge loops (four launch shapes)
    // must round this expression alike whatever code surrounds the call
#pragma clang fp contract(off)
    const double n = nA + nC;
    if (kind == SPKD_BIC) {
        double d = 0.5 * n * ldx - 0.5 * nA * ldA - 0.5 * nC * ldC;
        d -= lambdac * 0.5 * PEN_UNIT * log(n);
        return d;
    }
    return -(n / 2.0) * ((nA / n) * ldA + (nC / n) * ldC - ldx);
}

__device__ __forceinline__ int find_problem(const int64_t* __restrict__ seg_off, int64_t n_prob, int64_t g) {
    int64_t lo = 0, hi = n_prob;            // seg_off[lo] <= g < seg_off[hi]
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (seg_off[mid] <= g) lo = mid; else hi = mid;
    }
    return (int)lo;
}

#ifndef SPKD_MX_WAVES
#define SPKD_MX_WAVES 4
#endif
constexpr int MX_WAVES = SPKD_MX_WAVES;

// One block per record, in the launch order `sched` (XCD-aware, built by the host: the
// rows of a problem run on one XCD); block b computes row a = g - seg_off[p] of problem
// p = rec_prob[g], g = sched[b]: D[a][c] for c > a (and D[c][a] for variant 1), plus the
// diagonal / lower-triangle initial values.  Each wave evaluates four partners per pass.
// Four waves per block, two blocks per CU: a block's start-up (staging its row's record,
// a dependent chain of global loads) and its ragged end overlap with the other block's
// eliminations -- with one eight-wave block per CU the SIMDs idled through both.
template <bool TWO>
__global__ __launch_bounds__(MX_WAVES * WAVE, 2) void k_matrix(
        const double* __restrict__ ex, const double* __restrict__ pk, const int64_t* __restrict__ seg_off,
        const int32_t* __restrict__ rec_prob, const int32_t* __restrict__ sched, int variant, int kind, double lambdac,
        const double* __restrict__ ld, const double* __restrict__ aux,
        double* __restrict__ mat, const int64_t* __restrict__ mat_off,
        unsigned long long* stat_max, unsigned long long* stat_min, int* err) {
    __shared__ double ldsA[QREC];
    const int64_t g = sched[blockIdx.x];
    if (g < 0) return;                       // (padding of the launch order: block-uniform)
    const int p = rec_prob[g];
    const int64_t off = seg_off[p];
    const int64_t N = seg_off[p + 1] - off;
    const int64_t ra = g - off;
    double* Dm = mat + mat_off[p];
    const double* A = ex + g * QREC;
    stage_record(ldsA, A, threadIdx.x, MX_WAVES * WAVE);
    if (variant == 1) {
        if (threadIdx.x == 0) Dm[ra * N + ra] = MAXINT_F;
    } else {
        for (int64_t c = threadIdx.x; c <= ra; c += MX_WAVES * WAVE) Dm[ra * N + c] = __builtin_huge_val();
    }
    __syncthreads();
    const double nA = ldsA[QREC_COUNT_AT];
    const double ldA = ld[g];
    const int wave = threadIdx.x >> 6;
    const QuadLane L = quad_lane();
    double wmax = __builtin_nan(""), wmin = __builtin_nan("");
    if (kind == SPKD_KL2) {
        for (int64_t rc = ra + 1 + wave; rc < N; rc += MX_WAVES) {
            const double d = kl2_from_aux(aux + g * AUX, aux + (off + rc) * AUX);
            if (lane_id() == 0) {
                Dm[ra * N + rc] = d;
                if (variant == 1) Dm[rc * N + ra] = d;
            }
            if (stat_valid(d)) {
                wmax = (wmax != wmax || d > wmax) ? d : wmax;
                wmin = (wmin != wmin || d < wmin) ? d : wmin;
            }
        }
    } else {
        for (int64_t base = ra + 1 + 4 * wave; base < N; base += 4 * MX_WAVES) {
            int64_t rc = base + L.m;
            const bool valid = rc < N;
            rc = valid ? rc : N - 1;
            const double* C = ex + (off + rc) * QREC;
            const double* Cp = pk + (off + rc) * REC;
            const double ldx = log(quad_pair_det<TWO>(kind, ldsA, nA, A, C, Cp, false, L, err));
            const double d = finish_distance(kind, lambdac, nA, ldA, Cp[REC - 1], ld[off + rc], ldx);
            if (valid && L.t == 0) {
                Dm[ra * N + rc] = d;
                if (variant == 1) Dm[rc * N + ra] = d;
            }
            if (valid && stat_valid(d)) {
                wmax = (wmax != wmax || d > wmax) ? d : wmax;
                wmin = (wmin != wmin || d < wmin) ? d : wmin;
            }
        }
    }
    if (variant == 1 && (kind == SPKD_KL2 ? lane_id() == 0 : L.t == 0)) {
        if (wmax == wmax) atomicMax(stat_max + p, dkey(wmax));
        if (wmin == wmin) atomicMin(stat_min + p, dkey(wmin));
    }
}

// ---------------------------------------------------------------------------
constexpr int AHC_WAVES = 8;
constexpr int AHC_TPB = AHC_WAVES * WAVE;

struct ArgMin {
    double v;
    long long idx;
    long long nan_idx;
};

__device__ __forceinline__ void argmin_merge(ArgMin& x, const ArgMin& y) {
    if (y.v < x.v || (y.v == x.v && y.idx < x.idx)) { x.v = y.v; x.idx = y.idx; }
    if (y.nan_idx < x.nan_idx) x.nan_idx = y.nan_idx;
}

// One workgroup per problem.  ex is a private working copy of the quad records
// (summed in place as clusters merge).  Dynamic LDS: int32 ids[N + 1].
//
// The arg-min keeps a per-row cache (row minimum, its first column, first NaN
// column); a merge only invalidates the rows whose cached column was one of the
// two merged clusters, so an iteration costs O(N) plus a few row rescans instead
// of an O(N^2) scan, with numpy's first-occurrence / NaN semantics intact.
#ifdef SPKD_PROFILE
__device__ unsigned long long g_step_prof[8];     // profiling builds: clocks inside the selection of k_ahc_step
__device__ unsigned long long g_ahc_prof[8];      // profiling builds: rows rescanned, merges, 5 phase clocks of k_ahc
#endif
constexpr int NO_COL = 0x7fffffff;
constexpr int AHC_MAX_N = 65536;       // records per problem (row-flag masks live in LDS)

// Row-cache refresh: every alive row flagged dirty gets its minimum, the first
// column holding it and its first NaN column recomputed.  The flags are read 64
// rows per load and the dirty ones taken from the ballot, so a round with few
// dirty rows costs a few memory latencies, not one per row.
__device__ __forceinline__ void refresh_rows(const double* __restrict__ Dm, long long N,
                                             const int32_t* __restrict__ al, int32_t* __restrict__ dirty,
                                             double* __restrict__ rmin, int32_t* __restrict__ rarg,
                                             int32_t* __restrict__ rnan, unsigned long long* masks,
                                             int tid, int wave, int lane) {
    // phase A: snapshot of the flags, one 64-row mask per LDS word
    const long long n_chunks = (N + WAVE - 1) / WAVE;
    for (long long ch = wave; ch < n_chunks; ch += AHC_WAVES) {
        const long long rl = ch * WAVE + lane;
        const long long rc = rl < N ? rl : N - 1;
        const int fa = al[rc], fd = dirty[rc];
        const unsigned long long mk = __ballot(rl < N && fa && fd);
        if (lane == 0) masks[ch] = mk;
    }
    __syncthreads();
    // phase B: wave w takes every 8th dirty row (the snapshot makes the count the
    // same in every wave although rows are being marked clean meanwhile)
    int ord = 0;
    for (long long ch = 0; ch < n_chunks; ++ch) {
        const long long r0 = ch * WAVE;
        unsigned long long todo = masks[ch];
        while (todo) {
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1;
            if ((ord++ & (AHC_WAVES - 1)) != wave) continue;
            const long long r = r0 + b;
            const double* row = Dm + r * N;
            double mv = __builtin_huge_val();
            int mc = NO_COL, nc = NO_COL;
            for (long long c0 = 0; c0 < N; c0 += 4 * WAVE) {
                double v[4];
                int a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long c = c0 + u * WAVE + lane;
                    const long long cc = c < N ? c : N - 1;
                    a[u] = (c < N) ? al[cc] : 0;
                    v[u] = row[cc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = (int)(c0 + u * WAVE + lane);
                    if (!a[u]) continue;
                    if (v[u] != v[u]) { if (c < nc) nc = c; continue; }
                    if (v[u] < mv || (v[u] == mv && c < mc)) { mv = v[u]; mc = c; }
                }
            }
            {
                // (the wave's minimum and its first column in integers: DPP row steps + v_readlane)
                unsigned long long okey;
                unsigned oc;
                wave_argmin_key(mc != NO_COL ? dkey(mv) : ~0ull, (unsigned)mc, okey, oc);
                mv = okey != ~0ull ? dkey_inv(okey) : __builtin_huge_val();
                mc = okey != ~0ull ? (int)oc : NO_COL;
                nc = __any(nc != NO_COL) ? (int)wave_red_u32<false>((unsigned)nc) : NO_COL;
            }
            if (lane == 0) { rmin[r] = mv; rarg[r] = mc; rnan[r] = nc; dirty[r] = 0; }
#ifdef SPKD_PROFILE
            if (lane == 0) atomicAdd(&g_ahc_prof[0], 1ull);      // rows rescanned
#endif
        }
    }
}

template <bool TWO>
__global__ __launch_bounds__(AHC_TPB) void k_ahc(
        double* __restrict__ ex, double* __restrict__ pk, const int64_t* __restrict__ seg_off,
        int variant, int kind, int max_spk, double lambdac, double threshold,
        double* __restrict__ ld, double* __restrict__ aux,
        double* __restrict__ mat, const int64_t* __restrict__ mat_off,
        int32_t* __restrict__ alive, double* __restrict__ tmp,
        double* __restrict__ rmin_all, int32_t* __restrict__ rcache_all,
        int32_t* __restrict__ out_n, int32_t* __restrict__ out_a, int32_t* __restrict__ out_b,
        double* __restrict__ out_d, unsigned long long* stat_max, unsigned long long* stat_min,
        double* __restrict__ final_max, double* __restrict__ final_min, int* err) {
    extern __shared__ int32_t ids[];                    // alive partner slots of the merged cluster
    __shared__ double ldsA[QREC];
    __shared__ unsigned long long red3[AHC_WAVES][3];
    __shared__ unsigned long long s_masks[AHC_MAX_N / WAVE];
    __shared__ int s_cnt[2];
    __shared__ int s_nids;
    __shared__ int s_next;                              // next quad of partners of the running merge
    __shared__ double s_tmax[AHC_WAVES];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const QuadLane L = quad_lane();
    const int p = blockIdx.x;
    const int64_t off = seg_off[p];
    const long long N = seg_off[p + 1] - off;
    double* Dm = mat + mat_off[p];
    int32_t* al = alive + off;
    double* ldp = ld + off;
    double* tp = tmp + off;
    double* rmin = rmin_all + off;
    int32_t* rarg = rcache_all + 3 * off;
    int32_t* rnan = rarg + N;
    int32_t* dirty = rnan + N;
    for (long long c = tid; c < N; c += AHC_TPB) { al[c] = 1; dirty[c] = 1; }
    __syncthreads();
    long long m = N;
    int n_merges = 0;
    const long long INF_IDX = 0x7fffffffffffffffLL;
    double fmin = 0.0;
#ifdef SPKD_PROFILE
    unsigned long long ahc_acc[5] = {0ull, 0ull, 0ull, 0ull, 0ull}, ahc_t = clock64();
#define AHC_TICK(i) do { const unsigned long long now_ = clock64(); ahc_acc[i] += now_ - ahc_t; ahc_t = now_; } while (0)
#else
#define AHC_TICK(i) ((void)0)
#endif
    while (true) {
        // ---- 0. refresh the cache of the rows a merge invalidated (wave per row)
        refresh_rows(Dm, N, al, dirty, rmin, rarg, rnan, s_masks, tid, wave, lane);
        __syncthreads();
        AHC_TICK(0);
        // ---- 1. min / argmin over the alive sub-matrix, numpy semantics: first
        // occurrence in row-major order; any NaN -> min is NaN and argmin the first
        // NaN (distances.min(), distances.argmin(): CL1:203-204)
        // in integers, like the selection of the step chain below: a distance by its order-
        // preserving key, a thread meets its rows in ascending order (a strict "<" keeps the first
        // of equal keys), wave reductions by DPP row steps + v_readlane (wave_red_u64), every
        // thread folds the eight waves' results for itself -- no second barrier
        unsigned long long kb = ~0ull, ib = ~0ull, inan = ~0ull;
        for (long long r = tid; r < N; r += AHC_TPB) {
            if (!al[r]) continue;
            const double v = rmin[r];
            const int c = rarg[r], nc = rnan[r];
            if (nc != NO_COL) { const unsigned long long l = (unsigned long long)(r * N + nc); inan = l < inan ? l : inan; }
            if (c != NO_COL) {
                const unsigned long long kv = dkey(v);
                if (kv < kb) { kb = kv; ib = (unsigned long long)(r * N + c); }
            }
        }
        {
            const unsigned long long wk = wave_red_u64<false>(kb);
            const unsigned long long wi = wave_red_u64<false>(kb == wk ? ib : ~0ull);
            const unsigned long long wn = __any(inan != ~0ull) ? wave_red_u64<false>(inan) : ~0ull;
            if (lane == 0) { red3[wave][0] = wk; red3[wave][1] = wi; red3[wave][2] = wn; }
        }
        __syncthreads();
        unsigned long long bk = red3[0][0], bi = red3[0][1], bn = red3[0][2];
#pragma unroll
        for (int w = 1; w < AHC_WAVES; ++w) {
            const unsigned long long ok = red3[w][0], oi = red3[w][1], on = red3[w][2];
            const bool t = ok < bk || (ok == bk && oi < bi);
            bk = t ? ok : bk; bi = t ? oi : bi;
            bn = on < bn ? on : bn;
        }
        const bool has_nan = bn != ~0ull;
        const double mind = has_nan ? __builtin_nan("") : (bi != ~0ull ? dkey_inv(bk) : __builtin_huge_val());
        const long long index = has_nan ? (long long)bn : (bi != ~0ull ? (long long)bi : INF_IDX);
        fmin = mind;
        const bool go = (mind <= threshold) || (max_spk > 0 && m > max_spk);
        if (!go) break;
        AHC_TICK(1);
        const long long r0 = index / N, c0 = index - r0 * N;
        if (r0 == c0) {                       // a diagonal cell won: degenerate
            if (tid == 0) atomicOr(err, ERR_DEGENERATE_MERGE);
            break;
        }
        const long long sa = r0 < c0 ? r0 : c0, sb = r0 < c0 ? c0 : r0;
        // compacted indices = number of alive slots in front
        if (tid < 2) s_cnt[tid] = 0;
        if (tid == 0) { s_nids = 1; s_next = 0; }
        __syncthreads();
        {
            int ca = 0, cb = 0;
            for (long long c = tid; c < sb; c += AHC_TPB) {
                if (al[c]) { cb++; if (c < sa) ca++; }
            }
            // (wave sums by a DPP prefix scan: the total sits in lane 63)
            ca = __builtin_amdgcn_readlane(wave_scan_i32(ca), WAVE - 1);
            cb = __builtin_amdgcn_readlane(wave_scan_i32(cb), WAVE - 1);
            if (lane == 0 && ca) atomicAdd(&s_cnt[0], ca);
            if (lane == 0 && cb) atomicAdd(&s_cnt[1], cb);
        }
        __syncthreads();
        if (tid == 0) {
            const int64_t o = off + n_merges;
            out_a[o] = s_cnt[0]; out_b[o] = s_cnt[1]; out_d[o] = mind;
            al[sb] = 0;
            ids[0] = (int32_t)sa;             // job 0 = the merged cluster itself
        }
        n_merges++;
        m--;
#ifdef SPKD_PROFILE
        if (tid == 0) atomicAdd(&g_ahc_prof[1], 1ull);
#endif
        // ---- 2. merge the statistics (speakers[a].extend(speakers[b]))
        double* A = ex + (off + sa) * QREC;
        const double* B = ex + (off + sb) * QREC;
        {
            // both forms of both records asked for before the first sum is stored (A and B alias as
            // far as the compiler knows: written as two loops they are two trips to memory in a row)
            double* Ap = pk + (off + sa) * REC;       // the packed copies the pair passes load from
            const double* Bp = pk + (off + sb) * REC;
            constexpr int NQ = (QREC + AHC_TPB - 1) / AHC_TPB, NP = (REC + AHC_TPB - 1) / AHC_TPB;
            double qa[NQ], qb[NQ], pa[NP], pb[NP];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int e = tid + u * AHC_TPB;
                qa[u] = A[e < QREC ? e : 0]; qb[u] = B[e < QREC ? e : 0];
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int e = tid + u * AHC_TPB;
                pa[u] = Ap[e < REC ? e : 0]; pb[u] = Bp[e < REC ? e : 0];
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int e = tid + u * AHC_TPB;
                const double v = qa[u] + qb[u];
                if (e < QREC) { A[e] = v; ldsA[e] = v; }
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int e = tid + u * AHC_TPB;
                if (e < REC) Ap[e] = pa[u] + pb[u];
            }
        }
        __syncthreads();
        // partner list (order is irrelevant: results are scattered by slot); one LDS atomic
        // per wave and 64 slots, the lanes take their places from the ballot
        for (long long c0 = 0; c0 < N; c0 += AHC_TPB) {
            const long long c = c0 + tid;
            const bool take = c < N && c != sa && al[c < N ? c : N - 1];
            const unsigned long long mk = __ballot(take);
            int base = 0;
            if (lane == 0 && mk) base = atomicAdd(&s_nids, __popcll(mk));
            base = __shfl(base, 0);
            if (take) ids[base + __popcll(mk & ((1ull << lane) - 1ull))] = (int32_t)c;
        }
        __syncthreads();
        AHC_TICK(2);
        const int nids = s_nids;
        const double nA = ldsA[QREC_COUNT_AT];
        // ---- 3. the merged cluster's own term and the log dets of its unions
        if (kind == SPKD_KL2) {
            if (wave == 0) {
                double a[DA];
                single_rows_from_qr(A, a);
                const double mean_i = a[D] / nA;
                cov_rows(a, nA);
                kl2_aux_from_cov(a, mean_i, aux + (off + sa) * AUX);
            }
        } else {
            // (the waves take the quads of partners from a counter, not by stride: a wave whose
            // records come late takes fewer, nobody idles at the barrier for the slowest's sixth pass)
            for (;;) {
                int q4 = 0;
                if (lane == 0) q4 = atomicAdd(&s_next, 1);
                const int base = 4 * __builtin_amdgcn_readfirstlane(q4);
                if (base >= nids) break;
                int k = base + L.m;
                const bool valid = k < nids;
                k = valid ? k : nids - 1;
                const int32_t slot = ids[k];
                // determinants: the partners' logs are taken in step 4, one thread per partner
                const double v = quad_pair_det<TWO>(kind, ldsA, nA, A, ex + (off + slot) * QREC, pk + (off + slot) * REC, k == 0, L, err);
                if (valid && L.t == 0) {
                    if (k == 0) ldp[sa] = log(v); else tp[slot] = v;
                }
            }
        }
        __syncthreads();
        AHC_TICK(3);
        // ---- 4. finish the distances, update row (and column) sa and the row caches
        double wmax = __builtin_nan(""), wmin = __builtin_nan("");
        const double ldA = ldp[sa];
        for (long long c = tid; c < N; c += AHC_TPB) {
            if (c == sa || !al[c]) continue;
            double d;
            if (kind == SPKD_KL2) {
                // (one lane per pair here: 39-term sums done serially)
                const double* a1 = aux + (off + sa) * AUX;
                const double* a2 = aux + (off + c) * AUX;
                double t1 = 0.0, t2 = 0.0;
                for (int i = 0; i < D; ++i) {
                    const float dm = (float)a1[2 * DA + i] - (float)a2[2 * DA + i];
                    const double delta = (double)dm;
                    t1 += (a1[i] - a2[i]) * (a2[DA + i] - a1[DA + i]);
                    t2 += ((a1[DA + i] + a2[DA + i]) * delta) * delta;
                }
                d = 0.5 * t1 + 0.5 * t2;
            } else {
                const double nC = qr_count(ex + (off + c) * QREC);
                d = finish_distance(kind, lambdac, nA, ldA, nC, ldp[c], log(tp[c]));
            }
            Dm[sa * N + c] = d;
            const int ra = rarg[c], rn = rnan[c];
            if (variant == 1) {
                Dm[c * N + sa] = d;
                if (ra == sa || ra == sb || rn == sa || rn == sb) {
                    // a strictly smaller value at sa is the new row minimum whatever the rest holds
                    if (rn != sa && rn != sb && d < rmin[c]) { rmin[c] = d; rarg[c] = (int)sa; }
                    else dirty[c] = 1;
                }
                else if (d != d) { if ((int)sa < rn) rnan[c] = (int)sa; }
                else if (d < rmin[c] || (d == rmin[c] && (int)sa < ra)) { rmin[c] = d; rarg[c] = (int)sa; }
                if (stat_valid(d)) {
                    wmax = (wmax != wmax || d > wmax) ? d : wmax;
                    wmin = (wmin != wmin || d < wmin) ? d : wmin;
                }
            } else {
                if (ra == sb || rn == sb) dirty[c] = 1;   // column sa keeps its stale value (A-9)
            }
        }
        if (tid == 0) dirty[sa] = 1;
        if (variant == 1) {
            const unsigned long long kmax = wave_red_u64<true>(wmax == wmax ? dkey(wmax) : 0ull);
            const unsigned long long kmin = wave_red_u64<false>(wmin == wmin ? dkey(wmin) : ~0ull);
            if (lane == 0) {
                if (kmax != 0ull) atomicMax(stat_max + p, kmax);
                if (kmin != ~0ull) atomicMin(stat_min + p, kmin);
            }
        }
        __syncthreads();
        AHC_TICK(4);
    }
#ifdef SPKD_PROFILE
    if (tid == 0) for (int i = 0; i < 5; ++i) atomicAdd(&g_ahc_prof[2 + i], ahc_acc[i]);
#endif
    // max over the final alive sub-matrix (variant 2 reports distances.max()); NaN propagates
    double tmax = -__builtin_huge_val();
    bool anynan = false;
    for (long long r = wave; r < N; r += AHC_WAVES) {
        if (!al[r]) continue;
        const double* row = Dm + r * N;
        for (long long c = lane; c < N; c += WAVE) {
            if (!al[c]) continue;
            const double v = row[c];
            if (v != v) anynan = true; else tmax = v > tmax ? v : tmax;
        }
    }
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        const double t2 = __shfl_xor(tmax, s);
        tmax = t2 > tmax ? t2 : tmax;
    }
    anynan = __any(anynan);
    if (lane == 0) s_tmax[wave] = anynan ? __builtin_nan("") : tmax;
    __syncthreads();
    if (tid == 0) {
        double mx = s_tmax[0];
        for (int w = 1; w < AHC_WAVES; ++w) {
            const double x = s_tmax[w];
            if (mx == mx) mx = (x != x) ? x : (x > mx ? x : mx);
        }
        out_n[p] = n_merges;
        final_max[p] = mx;
        final_min[p] = fmin;
    }
}


// ---------------------------------------------------------------------------
// The same merge loop split over kernel launches ("wide" form): when there are
// few problems (one long file, BASELINE.json configs 2 / 5) one workgroup per
// problem leaves the chip idle and the loop is a serial chain of N - 1 rounds.
// Here a round is ONE launch, k_ahc_round, of ceil(partners / 31) workgroups per
// problem:
//   every workgroup  the log dets of the merged cluster with its 31 partners (and the
//                merged cluster's own, which each of them needs), the finished distances,
//                row / column sa of the matrix and the row caches of its partner rows
//                (rows whose cached minimum was invalidated are rescanned on the spot);
//   the workgroup that finishes last (an arrival ticket behind an agent-scope release;
//                it takes an agent-scope acquire)  the next round's selection: row sa's
//                own cache, arg-min over the row caches, the stop decision, the merge
//                of the two records, the partner list.
// Two small launches start the chain (row caches of the full matrix, first selection).
// The host enqueues N - 1 rounds back to back without reading anything: a problem
// that stopped sets state.done and its later launches return at once.  Nothing ever
// waits for another workgroup, so there is no co-residency requirement.  Round 1's
// three launches per merge (update / select / pairs) took 30 - 45 us per merge.
// Arithmetic, tie-breaks and NaN rules are those of k_ahc.
struct AhcState {
    int32_t done, n_merges, nids, ticket;
    long long sa, sb;
    double nA, fmin;
};

// distance of the merged cluster sa to cluster c from the finished log dets
// (k_ahc step 4; KL2 from the auxiliary records, 39 terms in index order)
__device__ __forceinline__ double ahc_finish(int kind, double lambdac, const double* __restrict__ ex,
                                             const double* __restrict__ aux, const double* __restrict__ ldp,
                                             double ldx, int64_t off, long long sa,
                                             long long c, double nA, double ldA) {
    if (kind == SPKD_KL2) {
        const double* a1 = aux + (off + sa) * AUX;
        const double* a2 = aux + (off + c) * AUX;
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < D; ++i) {
            const float dm = (float)a1[2 * DA + i] - (float)a2[2 * DA + i];
            const double delta = (double)dm;
            t1 += (a1[i] - a2[i]) * (a2[DA + i] - a1[DA + i]);
            t2 += ((a1[DA + i] + a2[DA + i]) * delta) * delta;
        }
        return 0.5 * t1 + 0.5 * t2;
    }
    const double nC = qr_count(ex + (off + c) * QREC);
    return finish_distance(kind, lambdac, nA, ldA, nC, ldp[c], ldx);
}

// wave-wide (min, first column, first NaN column) of one row; sub_col >= 0
// replaces that column's stored value by sub_val (the caller's own fresh write)
__device__ __forceinline__ void ahc_scan_row(const double* __restrict__ row, long long N,
                                             const int32_t* __restrict__ al, bool all_alive,
                                             long long sub_col, double sub_val, int lane,
                                             double& mv, int& mc, int& nc) {
    mv = __builtin_huge_val();
    mc = NO_COL; nc = NO_COL;
    for (long long c0 = 0; c0 < N; c0 += 4 * WAVE) {
        double v[4];
        int a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long c = c0 + u * WAVE + lane;
            const long long cc = c < N ? c : N - 1;
            a[u] = (c < N) ? (all_alive ? 1 : al[cc]) : 0;
            v[u] = row[cc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long cl = c0 + u * WAVE + lane;
            const int c = (int)cl;
            if (!a[u]) continue;
            const double x = (cl == sub_col) ? sub_val : v[u];
            if (x != x) { if (c < nc) nc = c; continue; }
            if (x < mv || (x == mv && c < mc)) { mv = x; mc = c; }
        }
    }
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        const double v2 = __shfl_xor(mv, s);
        const int c2 = __shfl_xor(mc, s), n2 = __shfl_xor(nc, s);
        if (v2 < mv || (v2 == mv && c2 < mc)) { mv = v2; mc = c2; }
        nc = n2 < nc ? n2 : nc;
    }
}

// grid (ceil(n_max / 8), n_prob); wave w of block x owns row 8 x + w: the row caches of
// the full initial matrix, every record alive
__global__ __launch_bounds__(AHC_TPB) void k_ahc_init_rows(
        const int64_t* __restrict__ seg_off, const double* __restrict__ mat,
        const int64_t* __restrict__ mat_off, int32_t* __restrict__ alive,
        double* __restrict__ rmin_all, int32_t* __restrict__ rcache_all) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int p = blockIdx.y;
    const int64_t off = seg_off[p];
    const long long N = seg_off[p + 1] - off;
    const long long r = (long long)blockIdx.x * AHC_WAVES + wave;
    if (r >= N) return;
    const double* Dm = mat + mat_off[p];
    int32_t* al = alive + off;
    double* rmin = rmin_all + off;
    int32_t* rarg = rcache_all + 3 * off;
    int32_t* rnan = rarg + N;
    double mv;
    int mc, nc;
    ahc_scan_row(Dm + r * N, N, al, true, -1, 0.0, lane, mv, mc, nc);
    if (lane == 0) { rmin[r] = mv; rarg[r] = mc; rnan[r] = nc; al[r] = 1; }
}

// The selection step of one problem, by one whole workgroup (AHC_TPB threads): with
// it > 0 first the cache of the merged cluster's own row (this round's workgroups wrote
// it) and the running statistics of variant 1 over the distances just evaluated; then
// arg-min over the row caches (k_ahc step 1), the stop decision, the merge of the two
// records and the partner list of the next round.
__device__ __forceinline__ void ahc_select_body(
        int it, int p, double* __restrict__ ex, const int64_t* __restrict__ seg_off,
        int variant, int kind, int max_spk, double threshold, double* __restrict__ aux,
        const double* __restrict__ mat, const int64_t* __restrict__ mat_off,
        int32_t* __restrict__ alive, double* __restrict__ rmin_all,
        int32_t* __restrict__ rcache_all, int32_t* __restrict__ ids_all,
        AhcState* __restrict__ state, int32_t* __restrict__ out_a, int32_t* __restrict__ out_b,
        double* __restrict__ out_d, unsigned long long* stat_max, unsigned long long* stat_min,
        int* err, double* __restrict__ pk) {
    struct RowRed { double mv, wmax, wmin; int mc, nc; };
    __shared__ RowRed rred[AHC_WAVES];
    __shared__ ArgMin red[AHC_WAVES];
    __shared__ ArgMin best;
    __shared__ int s_cnt[2];
    __shared__ int s_nids;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int64_t off = seg_off[p];
    const long long N = seg_off[p + 1] - off;
    int32_t* al = alive + off;
    double* rmin = rmin_all + off;
    int32_t* rarg = rcache_all + 3 * off;
    int32_t* rnan = rarg + N;
    int32_t* ids = ids_all + off;
    AhcState* S = state + p;
    const long long INF_IDX = 0x7fffffffffffffffLL;
    const int n_merges = it > 0 ? S->n_merges : 0;
    const long long psa = it > 0 ? S->sa : -1;       // the row the round just rewrote (none before the first merge)
    const long long m = N - n_merges;
    // ---- 0 + 1 in ONE pass over x = 0 .. N - 1 (one memory round trip): x as a COLUMN of
    // row psa -- that row's fresh cache and, variant 1, the running statistics over the
    // distances just evaluated -- and x as a ROW of the arg-min over the row caches (k_ahc
    // step 1; row psa enters with its fresh values below)
    const double* rowp = mat + mat_off[p] + (psa >= 0 ? psa : 0) * N;
    double mv = __builtin_huge_val(), wmax = __builtin_nan(""), wmin = __builtin_nan("");
    int mc = NO_COL, nc = NO_COL;
    ArgMin mine;
    mine.v = __builtin_huge_val(); mine.idx = INF_IDX; mine.nan_idx = INF_IDX;
#pragma unroll 2
    for (long long x = tid; x < N; x += AHC_TPB) {
        const int a = al[x];
        const double d = rowp[x];
        const double v = rmin[x];
        const int c = rarg[x], rn = rnan[x];
        if (!a) continue;
        if (psa >= 0) {
            if (variant == 1 && x != psa && stat_valid(d)) {
                wmax = (wmax != wmax || d > wmax) ? d : wmax;
                wmin = (wmin != wmin || d < wmin) ? d : wmin;
            }
            if (d != d) { if ((int)x < nc) nc = (int)x; }
            else if (d < mv || (d == mv && (int)x < mc)) { mv = d; mc = (int)x; }
        }
        if (x != psa) {
            if (rn != NO_COL) { const long long l = x * N + rn; if (l < mine.nan_idx) mine.nan_idx = l; }
            if (c != NO_COL) {
                const long long l = x * N + c;
                if (v < mine.v || (v == mine.v && l < mine.idx)) { mine.v = v; mine.idx = l; }
            }
        }
    }
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        const double v2 = __shfl_xor(mv, s);
        const int c2 = __shfl_xor(mc, s), n2 = __shfl_xor(nc, s);
        if (v2 < mv || (v2 == mv && c2 < mc)) { mv = v2; mc = c2; }
        nc = n2 < nc ? n2 : nc;
        const double x = __shfl_xor(wmax, s), y = __shfl_xor(wmin, s);
        if (x == x && (wmax != wmax || x > wmax)) wmax = x;
        if (y == y && (wmin != wmin || y < wmin)) wmin = y;
        ArgMin o;
        o.v = __shfl_xor(mine.v, s); o.idx = __shfl_xor(mine.idx, s); o.nan_idx = __shfl_xor(mine.nan_idx, s);
        argmin_merge(mine, o);
    }
    if (lane == 0) {
        rred[wave].mv = mv; rred[wave].mc = mc; rred[wave].nc = nc; rred[wave].wmax = wmax; rred[wave].wmin = wmin;
        red[wave] = mine;
    }
    if (tid < 2) s_cnt[tid] = 0;
    if (tid == 0) s_nids = 1;
    __syncthreads();
    if (tid == 0) {
        ArgMin b = red[0];
        for (int w = 1; w < AHC_WAVES; ++w) argmin_merge(b, red[w]);
        if (psa >= 0) {
            for (int w = 1; w < AHC_WAVES; ++w) {
                const RowRed o = rred[w];
                if (o.mv < mv || (o.mv == mv && o.mc < mc)) { mv = o.mv; mc = o.mc; }
                nc = o.nc < nc ? o.nc : nc;
                if (o.wmax == o.wmax && (wmax != wmax || o.wmax > wmax)) wmax = o.wmax;
                if (o.wmin == o.wmin && (wmin != wmin || o.wmin < wmin)) wmin = o.wmin;
            }
            rmin[psa] = mv; rarg[psa] = mc; rnan[psa] = nc;
            if (variant == 1) {
                if (wmax == wmax) atomicMax(stat_max + p, dkey(wmax));
                if (wmin == wmin) atomicMin(stat_min + p, dkey(wmin));
            }
            ArgMin own;                              // row psa as a candidate of the arg-min
            own.v = mc != NO_COL ? mv : __builtin_huge_val();
            own.idx = mc != NO_COL ? psa * N + mc : INF_IDX;
            own.nan_idx = nc != NO_COL ? psa * N + nc : INF_IDX;
            argmin_merge(b, own);
        }
        best = b;
    }
    __syncthreads();
    const bool has_nan = best.nan_idx != INF_IDX;
    const double mind = has_nan ? __builtin_nan("") : best.v;
    const long long index = has_nan ? best.nan_idx : best.idx;
    const bool go = (mind <= threshold) || (max_spk > 0 && m > max_spk);
    const long long r0 = index / N, c0 = index - r0 * N;
    if (!go || r0 == c0) {
        if (tid == 0) {
            if (go) atomicOr(err, ERR_DEGENERATE_MERGE);
            S->done = 1;
            S->n_merges = n_merges;
            S->fmin = mind;
        }
        return;
    }
    const long long sa = r0 < c0 ? r0 : c0, sb = r0 < c0 ? c0 : r0;
    // compacted indices = alive slots in front; partner list (any order)
    // (one LDS atomic per wave and 64 slots; the lanes take their places from the ballot --
    // an atomic per alive cluster serialised ~4 000 LDS operations per merge on a 10 h file)
    {
        int ca = 0, cb = 0;
        for (long long c0 = 0; c0 < N; c0 += AHC_TPB) {
            const long long c = c0 + tid;
            const bool alive_c = c < N && al[c < N ? c : N - 1];
            if (alive_c && c < sb) { cb++; if (c < sa) ca++; }
            const bool take = alive_c && c != sa && c != sb;
            const unsigned long long mk = __ballot(take);
            int base = 0;
            if (lane == 0 && mk) base = atomicAdd(&s_nids, __popcll(mk));
            base = __shfl(base, 0);
            if (take) ids[base + __popcll(mk & ((1ull << lane) - 1ull))] = (int32_t)c;
        }
        // (summed inside the wave first: one LDS atomic per wave, not per thread)
#pragma unroll
        for (int s = 1; s < WAVE; s <<= 1) { ca += __shfl_xor(ca, s); cb += __shfl_xor(cb, s); }
        if (lane == 0 && ca) atomicAdd(&s_cnt[0], ca);
        if (lane == 0 && cb) atomicAdd(&s_cnt[1], cb);
    }
    // ---- 2. merge the statistics
    double* A = ex + (off + sa) * QREC;
    const double* B = ex + (off + sb) * QREC;
    for (int e = tid; e < QREC; e += AHC_TPB) A[e] = A[e] + B[e];
    {                                                 // the packed copies the pair passes load from
        double* Ap = pk + (off + sa) * REC;
        const double* Bp = pk + (off + sb) * REC;
        for (int e = tid; e < REC; e += AHC_TPB) Ap[e] = Ap[e] + Bp[e];
    }
    __syncthreads();
    const double nA = A[QREC_COUNT_AT];
    if (kind == SPKD_KL2 && wave == 0) {
        double a[DA];
        single_rows_from_qr(A, a);
        const double mean_i = a[D] / nA;
        cov_rows(a, nA);
        kl2_aux_from_cov(a, mean_i, aux + (off + sa) * AUX);
    }
    if (tid == 0) {
        const int64_t o = off + n_merges;
        out_a[o] = s_cnt[0]; out_b[o] = s_cnt[1]; out_d[o] = mind;
        al[sb] = 0;
        ids[0] = (int32_t)sa;
        S->done = 0;
        S->sa = sa; S->sb = sb; S->nA = nA; S->nids = s_nids;
        S->n_merges = n_merges + 1;
        S->fmin = mind;
    }
}

// the first selection of every problem (grid: n_prob)
__global__ __launch_bounds__(AHC_TPB) void k_ahc_select0(
        double* __restrict__ ex, double* __restrict__ pk, const int64_t* __restrict__ seg_off,
        int variant, int kind, int max_spk, double threshold, double* __restrict__ aux,
        const double* __restrict__ mat, const int64_t* __restrict__ mat_off,
        int32_t* __restrict__ alive, double* __restrict__ rmin_all,
        int32_t* __restrict__ rcache_all, int32_t* __restrict__ ids_all,
        AhcState* __restrict__ state, int32_t* __restrict__ out_a, int32_t* __restrict__ out_b,
        double* __restrict__ out_d, unsigned long long* stat_max, unsigned long long* stat_min,
        int* err) {
    if (threadIdx.x == 0) state[blockIdx.x].ticket = 0;
    ahc_select_body(0, blockIdx.x, ex, seg_off, variant, kind, max_spk, threshold, aux, mat, mat_off, alive,
                    rmin_all, rcache_all, ids_all, state, out_a, out_b, out_d, stat_max, stat_min, err, pk);
}

// partners per workgroup: every wave pass holds four matrices; the first slot of the
// workgroup's first pass is the merged cluster itself
constexpr int RND_PARTNERS = 4 * AHC_WAVES - 1;

// one merge round; grid (ceil((n_max - 1) / RND_PARTNERS), n_prob)
template <bool TWO>
__global__ __launch_bounds__(AHC_TPB) void k_ahc_round(
        int it, double* __restrict__ ex, double* __restrict__ pk, const int64_t* __restrict__ seg_off,
        int variant, int kind, int max_spk, double lambdac, double threshold,
        double* __restrict__ ld, double* __restrict__ aux,
        double* __restrict__ mat, const int64_t* __restrict__ mat_off,
        int32_t* __restrict__ alive, double* __restrict__ rmin_all,
        int32_t* __restrict__ rcache_all, int32_t* __restrict__ ids_all,
        AhcState* __restrict__ state, int32_t* __restrict__ out_a, int32_t* __restrict__ out_b,
        double* __restrict__ out_d, unsigned long long* stat_max, unsigned long long* stat_min,
        int* err) {
    __shared__ double ldsA[QREC];
    __shared__ double s_ldx[4 * AHC_WAVES];
    __shared__ double s_dfin[4 * AHC_WAVES];
    __shared__ int s_rescan[4 * AHC_WAVES];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const QuadLane L = quad_lane();
    const int p = blockIdx.y;
    AhcState* S = state + p;
    if (S->done) return;
    const int nids = S->nids;                       // 1 + partners
    const int nb = nids > 1 ? (nids - 1 + RND_PARTNERS - 1) / RND_PARTNERS : 1;
    if ((int)blockIdx.x >= nb) return;
    const int64_t off = seg_off[p];
    const long long N = seg_off[p + 1] - off;
    const int32_t* ids = ids_all + off;
    const long long sa = S->sa, sb = S->sb;
    const double nA = S->nA;
    const double* A = ex + (off + sa) * QREC;
    double* Dm = mat + mat_off[p];
    int32_t* al = alive + off;
    double* ldp = ld + off;
    double* rmin = rmin_all + off;
    int32_t* rarg = rcache_all + 3 * off;
    int32_t* rnan = rarg + N;
    // this workgroup's items: item 0 = the merged cluster itself, item j >= 1 = partner
    // ids[first + j - 1]
    const int first = 1 + (int)blockIdx.x * RND_PARTNERS;
    const int mine = nids - first < RND_PARTNERS ? nids - first : RND_PARTNERS;     // partners here (>= 0)
    if (kind != SPKD_KL2) {
        for (int e = tid; e < QREC; e += AHC_TPB) ldsA[e] = A[e];
        __syncthreads();
        const int base = 4 * wave;                  // items base .. base + 3 of this wave
        if (base <= mine) {
            int j = base + L.m;
            const bool valid = j <= mine;
            j = valid ? j : mine;
            const long long cs = j == 0 ? sa : (long long)ids[first + j - 1];
            const double* C = ex + (off + cs) * QREC;
            const double v = quad_pair_det<TWO>(kind, ldsA, nA, A, C, pk + (off + cs) * REC, j == 0, L, err);
            if (valid && L.t == 0) s_ldx[j] = v;          // determinants
        }
        __syncthreads();
    }
    const double ldA = kind == SPKD_KL2 ? 0.0 : log(s_ldx[0]);
    if (kind != SPKD_KL2 && blockIdx.x == 0 && tid == 0) ldp[sa] = ldA;      // the merged cluster's cached term
    // ---- finish this workgroup's distances, row / column sa, the partner rows' caches: one
    // THREAD per partner (their loads travel together: one memory round trip for the
    // chunk, where a wave per row made one per row); the rows whose cached minimum was
    // invalidated are then rescanned, a wave per row
    {
        const int j = tid;                           // partner j = 1 .. mine (<= 31: the first wave)
        bool rescan = false;
        double d = 0.0;
        if (j >= 1 && j <= mine) {
            const long long r = ids[first + j - 1];
            d = ahc_finish(kind, lambdac, ex, aux, ldp, kind == SPKD_KL2 ? 0.0 : log(s_ldx[j]), off, sa, r, nA, ldA);
            const int ra = rarg[r], rn = rnan[r];
            const double rm = rmin[r];
            Dm[sa * N + r] = d;
            if (variant == 1) {
                Dm[r * N + sa] = d;
                const bool nan_hit = (rn == sa || rn == sb);
                if (ra == sa || ra == sb || nan_hit) {
                    if (!nan_hit && d < rm) { rmin[r] = d; rarg[r] = (int)sa; }    // still (or now) the strict row minimum
                    else rescan = true;
                } else if (d != d) {
                    if ((int)sa < rn) rnan[r] = (int)sa;
                } else if (d < rm || (d == rm && (int)sa < ra)) {
                    rmin[r] = d; rarg[r] = (int)sa;
                }
            } else if (ra == sb || rn == sb) {       // column sa keeps its stale value (A-9)
                rescan = true;
            }
        }
        if (tid < 4 * AHC_WAVES) { s_rescan[tid] = rescan ? 1 : 0; s_dfin[tid] = d; }
        __syncthreads();
        for (int jj = 1 + wave; jj <= mine; jj += AHC_WAVES) {
            if (!s_rescan[jj]) continue;             // (wave-uniform)
            const long long r = ids[first + jj - 1];
            double mv;
            int mc, nc;
            // variant 1: the row's own fresh cell (column sa) is taken from the value just computed
            ahc_scan_row(Dm + r * N, N, al, false, variant == 1 ? sa : -1, s_dfin[jj], lane, mv, mc, nc);
            if (lane == 0) { rmin[r] = mv; rarg[r] = mc; rnan[r] = nc; }
        }
    }
    // ---- arrival: everything this workgroup wrote is released at agent scope before its
    // ticket; the last arriver acquires and runs the next selection
    __syncthreads();                                 // (waits for every wave's stores)
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = atomicAdd(&S->ticket, 1);
        const int last = (t == nb - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            S->ticket = 0;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    ahc_select_body(it, p, ex, seg_off, variant, kind, max_spk, threshold, aux, mat, mat_off, alive,
                    rmin_all, rcache_all, ids_all, state, out_a, out_b, out_d, stat_max, stat_min, err, pk);
}

__global__ __launch_bounds__(AHC_TPB) void k_ahc_final(
        const int64_t* __restrict__ seg_off, const double* __restrict__ mat,
        const int64_t* __restrict__ mat_off, const int32_t* __restrict__ alive,
        const AhcState* __restrict__ state, int32_t* __restrict__ out_n,
        double* __restrict__ final_max, double* __restrict__ final_min) {
    __shared__ double s_tmax[AHC_WAVES];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int p = blockIdx.x;
    const int64_t off = seg_off[p];
    const long long N = seg_off[p + 1] - off;
    const double* Dm = mat + mat_off[p];
    const int32_t* al = alive + off;
    double tmax = -__builtin_huge_val();
    bool anynan = false;
    for (long long r = wave; r < N; r += AHC_WAVES) {
        if (!al[r]) continue;
        const double* row = Dm + r * N;
        for (long long c = lane; c < N; c += WAVE) {
            if (!al[c]) continue;
            const double v = row[c];
            if (v != v) anynan = true; else tmax = v > tmax ? v : tmax;
        }
    }
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        const double t2 = __shfl_xor(tmax, s);
        tmax = t2 > tmax ? t2 : tmax;
    }
    anynan = __any(anynan);
    if (lane == 0) s_tmax[wave] = anynan ? __builtin_nan("") : tmax;
    __syncthreads();
    if (tid == 0) {
        double mx = s_tmax[0];
        for (int w = 1; w < AHC_WAVES; ++w) {
            const double x = s_tmax[w];
            if (mx == mx) mx = (x != x) ? x : (x > mx ? x : mx);
        }
        out_n[p] = state[p].n_merges;
        final_max[p] = mx;
        final_min[p] = state[p].fmin;
    }
}


// ---------------------------------------------------------------------------
// The wide form without hand-offs ("step" chain): one launch per merge, NO workgroup waits
// for or signals another one inside a launch -- the only synchronisation is the kernel
// boundary.  The selection of merge k (the O(N) pass over the row caches, the stop decision,
// the partner list, the merged record) is done REDUNDANTLY by every workgroup of round k from
// what round k - 1 left; the code and the reduction order are the same everywhere, so every
// workgroup arrives at the same merge.  What the round-k workgroups write is kept apart from
// what round k reads:
//   * row caches and the scalar state are double-buffered by the parity of k;
//   * the new row of the matrix also goes to a staging row (parity of k), which is what the
//     next selection reads -- row sa of the matrix itself gets its symmetric entries written
//     while slow workgroups may still be selecting;
//   * a dead cluster is marked with the ROUND it died in (alive at the start of round k:
//     death >= k), so the mark may land at any time during the round;
//   * a merged record goes to a fresh slot (N + k - 1); a cluster's current slot is one 64-bit
//     word (round | new | old) that reads correctly before and after workgroup 0 replaces it.
// Against the ticket form (k_ahc_round: release fence, ticket, acquire fence, then one
// workgroup selects while the others have left) a merge loses ~4 us of fences and atomics and
// three dependent passes of the last arriver: 25.5 -> 14 us per merge at N = 380 (with the
// selection in integers, the bookkeeper workgroup and three partners a workgroup, below).
// Arithmetic, tie-breaks and NaN rules are those of k_ahc; results are bit-identical.
// ---------------------------------------------------------------------------
constexpr int STEP_WAVES = 4;
constexpr int STEP_WIDE_FROM = 1024;                       // problems larger than this: eight waves per workgroup
constexpr int STEP_PARTNERS = 4 * STEP_WAVES - 1;          // + the merged cluster itself: 16 items, one wave pass each
constexpr int ALIVE_ROUND = 0x7fffffff;
constexpr int STEP_MAX_N = 16384;                          // (two int arrays of N in LDS; positions packed as row << 14 | column)
constexpr int STEP_IDX_SHIFT = 14;

struct StepState {
    int32_t done, n_merges;
    long long psa;           // the row the previous round rewrote (-1: none)
    double fmin;
    double diag;             // D[psa][psa] (never rewritten; carried here so that no load depends on psa)
};

__host__ __device__ inline unsigned long long step_slot_word(int round, int new_slot, int old_slot) {
    return ((unsigned long long)(unsigned)round << 40) | ((unsigned long long)(unsigned)new_slot << 20) |
           (unsigned long long)(unsigned)old_slot;
}
__device__ __forceinline__ int step_slot_of(unsigned long long w, int k) {
    return ((int)(w >> 40) < k) ? (int)((w >> 20) & 0xfffffull) : (int)(w & 0xfffffull);
}

// what the selection reads per cluster, one 32-byte line piece per round parity: the row cache
// (minimum, its first column, first NaN column), the cluster's entry of the newest matrix row
// (written in round k for the selection of round k + 1: the same parity as the caches), and
// the round the cluster died in (kept in BOTH parities; ALIVE_ROUND while alive)
struct StepSel {
    double rmin, newrow;
    int32_t rarg, rnan, death, pad;
};

struct StepArrays {
    double* ex;              // initial records, quad layout (slots 0 .. N - 1 of a problem)
    double* pk;              //                  packed
    double* exm;             // merged records (slot N + m of a problem = record off + m here)
    double* pkm;
    int32_t* death;          // [n_total] (the row rescans and the final statistics read this copy)
    StepSel* sel2;           // [2][n_total]
    unsigned long long* sw;  // [n_total]
    double* cnt;             // [n_total] frame count of every cluster (a merged one: written by its round)
    StepState* state2;       // [2][n_prob]
    int64_t n_total;
    int32_t n_prob;
};

// row caches of the full initial matrix -> buffer 1 (what round 1 reads), every cluster alive,
// every cluster in its own slot.  grid (ceil(n_max / 8), n_prob), AHC_TPB threads.
__global__ __launch_bounds__(AHC_TPB) void k_step_init(
        const int64_t* __restrict__ seg_off, const double* __restrict__ mat,
        const int64_t* __restrict__ mat_off, StepArrays Q) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int p = blockIdx.y;
    const int64_t off = seg_off[p];
    const long long N = seg_off[p + 1] - off;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        StepState S;
        S.done = 0; S.n_merges = 0; S.psa = -1; S.fmin = 0.0; S.diag = 0.0;
        Q.state2[(size_t)1 * Q.n_prob + p] = S;
        Q.state2[(size_t)0 * Q.n_prob + p] = S;
    }
    const long long r = (long long)blockIdx.x * AHC_WAVES + wave;
    if (r >= N) return;
    double mv;
    int mc, nc;
    ahc_scan_row(mat + mat_off[p] + r * N, N, nullptr, true, -1, 0.0, lane, mv, mc, nc);
    if (lane == 0) {
        StepSel e;
        e.rmin = mv; e.newrow = 0.0; e.rarg = mc; e.rnan = nc; e.death = ALIVE_ROUND; e.pad = 0;
        Q.sel2[Q.n_total + off + r] = e;
        Q.sel2[off + r] = e;
        Q.death[off + r] = ALIVE_ROUND;
        Q.sw[off + r] = step_slot_word(0, (int)r, (int)r);
        Q.cnt[off + r] = Q.pk[(off + r) * REC + REC - 1];
    }
}

// wave-wide (min, first column, first NaN column) of one row over the clusters alive AFTER
// merge k (death > k, and not sb: its mark may not have landed yet); sub_col as in ahc_scan_row
__device__ __forceinline__ void step_scan_row(const double* __restrict__ row, long long N,
                                              const int32_t* __restrict__ death, int k, long long sb,
                                              long long sub_col, double sub_val, int lane,
                                              double& mv, int& mc, int& nc) {
    mv = __builtin_huge_val();
    mc = NO_COL; nc = NO_COL;
    for (long long c0 = 0; c0 < N; c0 += 4 * WAVE) {
        double v[4];
        int a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long c = c0 + u * WAVE + lane;
            const long long cc = c < N ? c : N - 1;
            a[u] = (c < N && c != sb) ? (death[cc] > k ? 1 : 0) : 0;
            v[u] = row[cc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long cl = c0 + u * WAVE + lane;
            const int c = (int)cl;
            if (!a[u]) continue;
            const double x = (cl == sub_col) ? sub_val : v[u];
            if (x != x) { if (c < nc) nc = c; continue; }
            if (x < mv || (x == mv && c < mc)) { mv = x; mc = c; }
        }
    }
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        const double v2 = __shfl_xor(mv, s);
        const int c2 = __shfl_xor(mc, s), n2 = __shfl_xor(nc, s);
        if (v2 < mv || (v2 == mv && c2 < mc)) { mv = v2; mc = c2; }
        nc = n2 < nc ? n2 : nc;
    }
}

// round k (k = 1 .. n_max - 1): merge k of every problem that has not stopped.
// grid (max(1, ceil((n_max - k - 1) / SP)) + 1, n_prob); dynamic LDS: 2 N_max ints + the chunk masks.
// SP partners of the merge per workgroup (3, 7 or 15: the host picks per round).
// SW waves per workgroup: the first STEP_WAVES of them eliminate (one wave per SIMD, the pass needs
// 156 VGPRs); with SW = 8 four more share the selection, the partner list and the rescans, which
// are issue-bound per wave (15 clusters per thread at N = 3 860 with four waves) and idle at the
// barriers through the pass.
template <bool TWO, int SW, int SP>
__global__ __launch_bounds__(SW * WAVE) void k_ahc_step(
        int k, const int64_t* __restrict__ seg_off, int variant, int kind, int max_spk, double lambdac,
        double threshold, double* __restrict__ ld, double* __restrict__ aux,
        double* __restrict__ mat, const int64_t* __restrict__ mat_off, StepArrays Q,
        int32_t* __restrict__ out_a, int32_t* __restrict__ out_b, double* __restrict__ out_d,
        unsigned long long* stat_max, unsigned long long* stat_min, int* err) {
    constexpr int TPB = SW * WAVE;
    extern __shared__ int32_t s_dyn[];               // [N] partner list | [N] current slot of a cluster (-1: dead)
    __shared__ double ldsA[QREC];
    __shared__ double s_auxA[AUX];
    __shared__ double s_ldx[4 * STEP_WAVES];
    __shared__ double s_dfin[4 * STEP_WAVES];
    __shared__ int s_rescan[4 * STEP_WAVES];
    struct RowRed { double mv; int mc, nc; };
    struct WaveRed { unsigned long long kb, kr, kmax, kmin; unsigned ib, inan; int mc, nc; };
    __shared__ WaveRed wred[SW];
    __shared__ int s_cnt[2];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const QuadLane L = quad_lane();
    const int p = blockIdx.y;
#ifdef SPKD_PROFILE
    // profiling builds: phase clocks of thread 0 of workgroup 0 (tools/step_phase_profile.py)
    unsigned long long st_t = clock64();
#define STEP_TICK(i) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_ahc_prof[i], now_ - st_t); st_t = now_; } while (0)
    unsigned long long st_f = st_t;
#define STEP_FINE(i) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_step_prof[i], now_ - st_f); st_f = now_; } while (0)
#else
#define STEP_TICK(i) ((void)0)
#define STEP_FINE(i) ((void)0)
#endif
    // The prologue's scalar loads in two batches -- the kernel arguments, then what they point to
    // (the state, the problem's offsets) -- pinned here: left to itself the compiler loads each
    // argument in the block that first uses it, six dependent trips to memory one after the
    // other before the selection's first vector load is issued.
    asm volatile("" :: "s"(Q.state2), "s"(Q.sel2), "s"(Q.sw), "s"(Q.ex), "s"(Q.exm), "s"(Q.n_total), "s"(Q.n_prob),
                 "s"(seg_off), "s"(mat_off), "s"(mat), "s"(k));
    StepState S = Q.state2[(size_t)(k & 1) * Q.n_prob + p];
    StepState* Snext = Q.state2 + (size_t)((k + 1) & 1) * Q.n_prob + p;
    int64_t off = seg_off[p], off_end = seg_off[p + 1], moff = mat_off[p];
    asm volatile("" :: "s"(S.done), "s"(S.n_merges), "s"(off), "s"(off_end), "s"(moff));
    const long long N = off_end - off;
    const bool lead = blockIdx.x == 0;
    if (S.done) {                                    // a stopped problem: carry its state forward
        if (lead && tid == 0) *Snext = S;
        return;
    }
    STEP_FINE(0);                                    // the state and the offsets are here
    const int n_merges = S.n_merges;
    const long long psa = S.psa;
    const long long m = N - n_merges;                // clusters alive now
    const int n_part = (int)(m - 2);                 // partners of this merge, if it happens
    constexpr int sp = SP;
    // SP partners per workgroup (+ the merged cluster itself): 15 = all four waves pass; 7 = two
    // of them, when the chip has CUs to spare -- a workgroup's record loads go through one CU's
    // address unit (41 KB a wave), so half the passes per CU are back in half the time
    const int nb = n_part > 0 ? (n_part + sp - 1) / sp : 1;
    // workgroups 0 .. nb - 1 take the partners; workgroup nb is the BOOKKEEPER: it selects like
    // everybody, forms the merged record, and writes it out (both layouts), the slot word, the
    // count, the death mark and the merge log while the others eliminate -- 5 k cycles that sat
    // on the critical path of workgroup 0 (tools/step_phase_profile.py)
    if ((int)blockIdx.x > nb) return;
    const bool keeper = (int)blockIdx.x == nb;
    int32_t* s_ids = s_dyn;
    int32_t* s_slot = s_dyn + N;
    const int nch = (int)((N + WAVE - 1) / WAVE);    // 64-cluster chunks: alive masks and their running counts
    int32_t* s_base = s_dyn + 2 * N;
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_dyn + ((2 * N + nch + 1) & ~1LL));
    const int rb = k & 1, wb = (k + 1) & 1;
    const int32_t* death = Q.death + off;
    const StepSel* sel_r = Q.sel2 + (size_t)rb * Q.n_total + off;
    StepSel* sel_w = Q.sel2 + (size_t)wb * Q.n_total + off;
    double* Dm = mat + moff;
    double* ldp = ld + off;
    const long long INF_IDX = 0x7fffffffffffffffLL;
    auto rec_quad = [&](int slot) -> const double* {
        return slot < N ? Q.ex + (off + slot) * QREC : Q.exm + (off + (slot - N)) * QREC;
    };
    auto rec_packed = [&](int slot) -> const double* {
        return slot < N ? Q.pk + (off + slot) * REC : Q.pkm + (off + (slot - N)) * REC;
    };
    // ---- S. the selection, by every workgroup: ONE pass over x = 0 .. N - 1 -- x as a column
    // of the row psa (its fresh cache; variant 1: the running statistics over the distances
    // round k - 1 evaluated) and x as a row of the arg-min over the row caches
    // Everything is compared as integers: a distance by its order-preserving 64-bit key (dkey),
    // a matrix position (row, column) by row << 14 | column (N <= 16 384: the order of the linear index).  A thread meets its clusters in
    // ascending order, so on equal keys the one met first has the smaller index: a strict "<"
    // against a running best that starts at the all-ones key (above every number's key) is the
    // reference's (value, then index) order.  No branches per cluster, only selects.
    const double diag_psa = S.diag;
    const bool hp = psa >= 0, v1 = variant == 1;             // (uniform)
    const unsigned NONE = 0xffffffffu;
    unsigned long long kb = ~0ull, kr = ~0ull, kmax = 0ull, kmin = ~0ull;
    unsigned ib = NONE, inan = NONE;
    int mc = NO_COL, nc = NO_COL;
    // (eight clusters per thread at a time, all their loads in flight before the first is looked
    // at: a wave's lanes hold 64 consecutive clusters, chunk x / 64)
    constexpr int SEL_U = 8;
    for (long long xb = 0; xb < N; xb += (long long)SEL_U * TPB) {       // (xb: uniform)
        const long long x0 = xb + tid;
        int dth[SEL_U], c_[SEL_U], rn_[SEL_U];
        unsigned long long w_[SEL_U];
        double dn_[SEL_U], v_[SEL_U];
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) {
            if (xb + (long long)u * TPB < N) {         // (uniform: nobody has a cluster beyond)
                const long long x = x0 + (long long)u * TPB;
                const long long xc = x < N ? x : N - 1;
                const StepSel e = sel_r[xc];              // (two 16-byte loads)
                w_[u] = Q.sw[off + xc];
                dth[u] = e.death; dn_[u] = e.newrow; v_[u] = e.rmin; c_[u] = e.rarg; rn_[u] = e.rnan;
            }
        }
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) {
            if (xb + (long long)u * TPB >= N) break;
            const long long x = x0 + (long long)u * TPB;
            const unsigned xu = (unsigned)x;
            const bool in = x < N;                     // (a wave covers one chunk)
            const bool a = in && dth[u] >= k;
            const unsigned long long mk = __ballot(a);
            if (in) {
                s_slot[x] = a ? step_slot_of(w_[u], k) : -1;
                if (lane == 0) s_mask[x >> 6] = mk;
            }
            const bool np = a && x != psa;             // a row of the arg-min; a column of row psa other than psa
            const unsigned long long kv = dkey(v_[u]);
            const bool bet = np && c_[u] != NO_COL && kv < kb;
            kb = bet ? kv : kb;
            ib = bet ? ((xu << STEP_IDX_SHIFT) | (unsigned)c_[u]) : ib;
            const unsigned ni = (np && rn_[u] != NO_COL) ? ((xu << STEP_IDX_SHIFT) | (unsigned)rn_[u]) : NONE;
            inan = ni < inan ? ni : inan;
            if (hp) {
                const double d = x == psa ? diag_psa : dn_[u];
                const bool dnan = d != d;
                const unsigned long long kd = dkey(d);
                const bool bt = a && !dnan && kd < kr;
                kr = bt ? kd : kr;
                mc = bt ? (int)x : mc;
                nc = (a && dnan && (int)x < nc) ? (int)x : nc;
                if (v1) {
                    const bool sv = np && stat_valid(d);
                    kmax = (sv && kd > kmax) ? kd : kmax;
                    kmin = (sv && kd < kmin) ? kd : kmin;
                }
            }
        }
    }
    STEP_FINE(1);                                    // the pass over the clusters
    {
        WaveRed w;
        wave_argmin_key(kb, ib, w.kb, w.ib);
        w.inan = __any(inan != NONE) ? wave_red_u32<false>(inan) : NONE;
        w.kr = ~0ull; w.mc = NO_COL; w.nc = NO_COL; w.kmax = 0ull; w.kmin = ~0ull;
        if (hp) {
            unsigned mcu;
            wave_argmin_key(kr, (unsigned)mc, w.kr, mcu);
            w.mc = (int)mcu;
            if (__any(nc != NO_COL)) w.nc = (int)wave_red_u32<false>((unsigned)nc);
            if (v1) { w.kmax = wave_red_u64<true>(kmax); w.kmin = wave_red_u64<false>(kmin); }
        }
        STEP_FINE(2);                                // the wave's reduction
        if (lane == 0) wred[wave] = w;
    }
    __syncthreads();
    // every thread folds the waves' results for itself (SW entries, broadcast reads): no second
    // barrier, no trip of the winner through LDS
    // (as a tree: the merges of one level do not depend on each other)
    WaveRed fw[SW];
#pragma unroll
    for (int w = 0; w < SW; ++w) fw[w] = wred[w];
#pragma unroll
    for (int st = 1; st < SW; st <<= 1) {
#pragma unroll
        for (int w = 0; w + st < SW; w += 2 * st) {
            WaveRed& x = fw[w];
            const WaveRed& o = fw[w + st];
            const bool b1 = o.kb < x.kb || (o.kb == x.kb && o.ib < x.ib);
            x.kb = b1 ? o.kb : x.kb; x.ib = b1 ? o.ib : x.ib;
            x.inan = o.inan < x.inan ? o.inan : x.inan;
            const bool b2 = o.kr < x.kr || (o.kr == x.kr && o.mc < x.mc);
            x.kr = b2 ? o.kr : x.kr; x.mc = b2 ? o.mc : x.mc;
            x.nc = o.nc < x.nc ? o.nc : x.nc;
            x.kmax = o.kmax > x.kmax ? o.kmax : x.kmax;
            x.kmin = o.kmin < x.kmin ? o.kmin : x.kmin;
        }
    }
    WaveRed f = fw[0];
    // row psa: its fresh cache (what the finish reads for r == psa), and its place in the arg-min
    RowRed s_psa;
    s_psa.mv = f.mc != NO_COL ? dkey_inv(f.kr) : __builtin_huge_val();
    s_psa.mc = f.mc; s_psa.nc = f.nc;
    if (hp) {
        if (lead && tid == 0 && v1) {
            if (f.kmax != 0ull) atomicMax(stat_max + p, f.kmax);
            if (f.kmin != ~0ull) atomicMin(stat_min + p, f.kmin);
        }
        if (f.mc != NO_COL) {
            const unsigned io = ((unsigned)psa << STEP_IDX_SHIFT) | (unsigned)f.mc;
            if (f.kr < f.kb || (f.kr == f.kb && io < f.ib)) { f.kb = f.kr; f.ib = io; }
        }
        if (f.nc != NO_COL) {
            const unsigned io = ((unsigned)psa << STEP_IDX_SHIFT) | (unsigned)f.nc;
            f.inan = io < f.inan ? io : f.inan;
        }
    }
    STEP_FINE(3);                                    // barrier + the fold over the waves
    STEP_TICK(2);                                    // selection: one pass + reductions
    const bool has_nan = f.inan != NONE;
    const double mind = has_nan ? __builtin_nan("") : (f.ib != NONE ? dkey_inv(f.kb) : __builtin_huge_val());
    const unsigned pos = has_nan ? f.inan : f.ib;    // (NONE: nothing to merge; `go` is false then unless max_spk forces)
    const bool go = (mind <= threshold) || (max_spk > 0 && m > max_spk);
    const long long r0 = pos >> STEP_IDX_SHIFT, c0 = pos & ((1u << STEP_IDX_SHIFT) - 1u);
    if (!go || r0 == c0 || pos == NONE) {            // (pos == NONE with go: no pair left to force)
        if (lead && tid == 0) {
            if (go) atomicOr(err, ERR_DEGENERATE_MERGE);
            StepState T;
            T.done = 1; T.n_merges = n_merges; T.psa = psa; T.fmin = mind; T.diag = S.diag;
            *Snext = T;
        }
        return;
    }
    const long long sa = r0 < c0 ? r0 : c0, sb = r0 < c0 ? c0 : r0;
    double diag_sa = 0.0;
    if (lead && tid == 0) diag_sa = Dm[sa * N + sa];   // for the next round's state
    // ---- M (first half). the two old records of the merged pair on their way to registers
    // while the partner list is being built
    constexpr int M_PER = (QREC + TPB - 1) / TPB;
    double m_a[M_PER], m_b[M_PER];
    {
        const double* A0 = rec_quad(s_slot[sa]);
        const double* B0 = rec_quad(s_slot[sb]);
#pragma unroll
        for (int u = 0; u < M_PER; ++u) {
            const int e = tid + u * TPB;
            m_a[u] = A0[e < QREC ? e : 0];
            m_b[u] = B0[e < QREC ? e : 0];
        }
    }
    // ---- partner list, in ascending slot order (the same in every workgroup): the alive rank of
    // a cluster = the running count of its chunk + the alive lanes in front of it; the merge's
    // compacted indices are the alive ranks of sa and sb; a partner's place in the list is its
    // alive rank less the merged pair in front of it.  (One wave scans the chunk counts, every
    // thread places its clusters: no loop over the clusters by a single wave.)
    if (wave == 0) {
        int run = 0;
        for (int c1 = 0; c1 < nch; c1 += WAVE) {
            const int ch = c1 + lane;
            const int cnt = ch < nch ? __popcll(s_mask[ch]) : 0;
            const int inc = wave_scan_i32(cnt);
            if (ch < nch) s_base[ch] = run + inc - cnt;
            run += __builtin_amdgcn_readlane(inc, WAVE - 1);
        }
    }
    __syncthreads();
    const int ca = s_base[sa >> 6] + __popcll(s_mask[sa >> 6] & ((1ull << (sa & 63)) - 1ull));
    const int cb = s_base[sb >> 6] + __popcll(s_mask[sb >> 6] & ((1ull << (sb & 63)) - 1ull));
    for (long long x = tid; x < N; x += TPB) {
        if (s_slot[x] < 0 || x == sa || x == sb) continue;
        const int ar = s_base[x >> 6] + __popcll(s_mask[x >> 6] & ((1ull << (x & 63)) - 1ull));
        s_ids[ar - (ar > ca ? 1 : 0) - (ar > cb ? 1 : 0)] = (int32_t)x;
    }
    if (tid == 0) { s_cnt[0] = ca; s_cnt[1] = cb; }
    // ---- M (second half). the merged record, formed by every workgroup from the two old ones
#pragma unroll
    for (int u = 0; u < M_PER; ++u) {
        const int e = tid + u * TPB;
        if (e < QREC) ldsA[e] = m_a[u] + m_b[u];
    }
    __syncthreads();
    STEP_TICK(3);                                    // partner list + merged record
    const double nA = ldsA[QREC_COUNT_AT];
    const int new_slot = (int)N + n_merges;
    if (keeper) {
        // the books: the merged record in its fresh slot (both layouts), the slot word, the
        // count, the death mark, the merge log
        double* Am = Q.exm + (off + n_merges) * QREC;
        for (int e = tid; e < QREC; e += TPB) Am[e] = ldsA[e];
        const double* Ap = rec_packed(s_slot[sa]);
        const double* Bp = rec_packed(s_slot[sb]);
        double* Pm = Q.pkm + (off + n_merges) * REC;
        for (int e = tid; e < REC; e += TPB) Pm[e] = Ap[e] + Bp[e];
        if (tid == 0) {
            Q.sw[off + sa] = step_slot_word(k, new_slot, s_slot[sa]);
            Q.cnt[off + sa] = nA;                    // (read by later rounds only: sa is nobody's partner now)
            Q.death[off + sb] = k;
            Q.sel2[off + sb].death = k;
            Q.sel2[Q.n_total + off + sb].death = k;
            const int64_t o = off + n_merges;
            out_a[o] = s_cnt[0]; out_b[o] = s_cnt[1]; out_d[o] = mind;
        }
    }
    if (kind == SPKD_KL2 && wave == 0) {
        double a[DA];
        single_rows_from_qr(ldsA, a);
        const double mean_i = a[D] / nA;
        cov_rows(a, nA);
        kl2_aux_from_cov(a, mean_i, s_auxA);
        if (keeper && lane < D) {
            double* ga = aux + (off + sa) * AUX;
            ga[lane] = s_auxA[lane]; ga[DA + lane] = s_auxA[DA + lane]; ga[2 * DA + lane] = s_auxA[2 * DA + lane];
        }
    }
    if (keeper) return;
    STEP_TICK(4);
    // ---- P. this workgroup's items: item 0 = the merged cluster itself, item j >= 1 = partner
    // s_ids[first + j - 1]
    const int first = (int)blockIdx.x * sp;
    const int mine_n = n_part - first < sp ? n_part - first : sp;   // partners here (>= 0)
    // what the finish needs of a partner (its count, its log term, its row cache), asked for now:
    // the answers arrive under the pass
    const bool fin = tid >= 1 && tid <= mine_n;
    const long long fin_r = fin ? (long long)s_ids[first + tid - 1] : 0;
    double fin_nC = 0.0, fin_ld = 0.0;
    StepSel fin_sel;
    fin_sel.rmin = 0.0; fin_sel.newrow = 0.0; fin_sel.rarg = 0; fin_sel.rnan = 0; fin_sel.death = 0; fin_sel.pad = 0;
    if (fin) {
        fin_sel = sel_r[fin_r];
        if (kind != SPKD_KL2) { fin_nC = Q.cnt[off + fin_r]; fin_ld = ldp[fin_r]; }
    }
    if (kind != SPKD_KL2) {
        const int base = 4 * wave;
        if (base <= mine_n) {
            int j = base + L.m;
            const bool valid = j <= mine_n;
            j = valid ? j : mine_n;
            const long long cx = j == 0 ? sa : (long long)s_ids[first + j - 1];
            const int cslot = s_slot[cx];
            const double nCx = j == 0 ? 0.0 : Q.cnt[off + cx];
            // (the fallback's "global A" is the LDS copy: the merged record is not in global
            // memory yet for anybody but workgroup 0)
            const double v = quad_pair_det<TWO>(kind, ldsA, nA, ldsA, rec_quad(cslot), rec_packed(cslot), j == 0, L, err, nCx);
            if (valid && L.t == 0) s_ldx[j] = v;
        }
    }
    __syncthreads();
    STEP_TICK(5);                                    // the pass
    // the logs of the 16 determinants side by side (the merged cluster's own term among them)
    if (kind != SPKD_KL2 && tid <= mine_n) s_ldx[tid] = log(s_ldx[tid]);
    __syncthreads();
    const double ldA = kind == SPKD_KL2 ? 0.0 : s_ldx[0];
    if (kind != SPKD_KL2 && lead && tid == 0) ldp[sa] = ldA;
    // ---- finish the distances: a thread per partner; row sa (+ its staging copy), column sa
    // (variant 1), the partner rows' caches into the other buffer
    {
        const int j = tid;
        bool rescan = false;
        double d = 0.0;
        if (fin) {
            const long long r = fin_r;
            if (kind == SPKD_KL2) {
                const double* a2 = aux + (off + r) * AUX;
                double t1 = 0.0, t2 = 0.0;
                for (int i = 0; i < D; ++i) {
                    const float dm = (float)s_auxA[2 * DA + i] - (float)a2[2 * DA + i];
                    const double delta = (double)dm;
                    t1 += (s_auxA[i] - a2[i]) * (a2[DA + i] - s_auxA[DA + i]);
                    t2 += ((s_auxA[DA + i] + a2[DA + i]) * delta) * delta;
                }
                d = 0.5 * t1 + 0.5 * t2;
            } else {
                d = finish_distance(kind, lambdac, nA, ldA, fin_nC, fin_ld, s_ldx[j]);
            }
            double rm;
            int ra, rn;
            if (r == psa) { rm = s_psa.mv; ra = s_psa.mc; rn = s_psa.nc; }
            else { rm = fin_sel.rmin; ra = fin_sel.rarg; rn = fin_sel.rnan; }
            Dm[sa * N + r] = d;
            sel_w[r].newrow = d;
            if (variant == 1) {
                Dm[r * N + sa] = d;
                const bool nan_hit = (rn == sa || rn == sb);
                if (ra == sa || ra == sb || nan_hit) {
                    if (!nan_hit && d < rm) { rm = d; ra = (int)sa; }    // still (or now) the strict row minimum
                    else rescan = true;
                } else if (d != d) {
                    if ((int)sa < rn) rn = (int)sa;
                } else if (d < rm || (d == rm && (int)sa < ra)) {
                    rm = d; ra = (int)sa;
                }
            } else if (ra == sb || rn == sb) {       // column sa keeps its stale value (A-9)
                rescan = true;
            }
            if (!rescan) { sel_w[r].rmin = rm; sel_w[r].rarg = ra; sel_w[r].rnan = rn; }
        }
        if (tid < 4 * STEP_WAVES) { s_rescan[tid] = rescan ? 1 : 0; s_dfin[tid] = d; }
        __syncthreads();
        for (int jj = 1 + wave; jj <= mine_n; jj += SW) {
            if (!s_rescan[jj]) continue;             // (wave-uniform)
            const long long r = s_ids[first + jj - 1];
            double mv2;
            int mc2, nc2;
            step_scan_row(Dm + r * N, N, death, k, sb, variant == 1 ? sa : -1, s_dfin[jj], lane, mv2, mc2, nc2);
            if (lane == 0) { sel_w[r].rmin = mv2; sel_w[r].rarg = mc2; sel_w[r].rnan = nc2; }
        }
    }
    STEP_TICK(6);                                    // logs, distances, row caches, rescans
#ifdef SPKD_PROFILE
    if (lead && tid == 0) atomicAdd(&g_ahc_prof[1], 1ull);
#endif
    if (lead && tid == 0) {
        StepState T;
        T.done = 0; T.n_merges = n_merges + 1; T.psa = sa; T.fmin = mind; T.diag = diag_sa;
        *Snext = T;
    }
}

// after the last round R: merge counts and the statistics of the final matrix
__global__ __launch_bounds__(AHC_TPB) void k_step_final(
        int last_round, const int64_t* __restrict__ seg_off, const double* __restrict__ mat,
        const int64_t* __restrict__ mat_off, StepArrays Q, int32_t* __restrict__ out_n,
        double* __restrict__ final_max, double* __restrict__ final_min) {
    __shared__ double s_tmax[AHC_WAVES];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int p = blockIdx.x;
    const int64_t off = seg_off[p];
    const long long N = seg_off[p + 1] - off;
    const double* Dm = mat + mat_off[p];
    const int32_t* death = Q.death + off;
    const StepState S = Q.state2[(size_t)((last_round + 1) & 1) * Q.n_prob + p];
    double tmax = -__builtin_huge_val();
    bool anynan = false;
    for (long long r = wave; r < N; r += AHC_WAVES) {
        if (death[r] != ALIVE_ROUND) continue;
        const double* row = Dm + r * N;
        for (long long c = lane; c < N; c += WAVE) {
            if (death[c] != ALIVE_ROUND) continue;
            const double v = row[c];
            if (v != v) anynan = true; else tmax = v > tmax ? v : tmax;
        }
    }
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        const double t2 = __shfl_xor(tmax, s);
        tmax = t2 > tmax ? t2 : tmax;
    }
    anynan = __any(anynan);
    if (lane == 0) s_tmax[wave] = anynan ? __builtin_nan("") : tmax;
    __syncthreads();
    if (tid == 0) {
        double mx = s_tmax[0];
        for (int w = 1; w < AHC_WAVES; ++w) {
            const double x = s_tmax[w];
            if (mx == mx) mx = (x != x) ? x : (x > mx ? x : mx);
        }
        out_n[p] = S.n_merges;
        final_max[p] = mx;
        final_min[p] = S.fmin;
    }
}

// ---------------------------------------------------------------------------
// spk_cluster_in (spk-clustering.py:136-175 / spk-clustering2.py:135-170) as ONE device-resident
// chain over the statistics records of the recipe's lines, in recipe order: record 0 founds
// cluster 0; every later record is compared with every cluster so far -- the cluster is the
// first argument of the distance, the new segment the second -- and joins the first arg-min
// over the finite distances if that is <= threshold, else founds a new cluster.  A cluster's
// record is the sum of its members' (the reference concatenates their frames); BIC's own term
// of a grown cluster is the union determinant of the pair that grew it, GLR's takes one more
// pass.  One workgroup: the chain is serial, a step is one wave pass for up to 16 clusters.
// Host-driven (a library call and a synchronisation per segment and per stage) the same took
// 345 us per segment; this takes ~10.
// dist: all distances in evaluation order, dist_off[s] .. dist_off[s + 1] those of record s
// (the host replays the reference's prints and statistics from them).  done[0]: records
// processed (< n: a non-finite covariance or a full dist buffer stopped the chain at that
// record), done[1]: clusters.
// ---------------------------------------------------------------------------
constexpr int CIN_WAVES = 4;
constexpr int CIN_TPB = CIN_WAVES * WAVE;

template <bool TWO>
__global__ __launch_bounds__(CIN_TPB) void k_cluster_in(
        const double* __restrict__ seg_ex, const double* __restrict__ seg_pk, const double* __restrict__ seg_ld,
        const double* __restrict__ seg_aux, long long n, int kind, double lambdac, double threshold,
        double* __restrict__ clu_ex, double* __restrict__ clu_pk, double* __restrict__ clu_ld, double* __restrict__ clu_aux,
        double* __restrict__ tmp, int32_t* __restrict__ label, double* __restrict__ dist, long long dist_cap,
        long long* __restrict__ dist_off, long long* __restrict__ done, int* err) {
    __shared__ double ldsA[QREC];
    __shared__ int s_best, s_stop;
    __shared__ double s_mind;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const QuadLane L = quad_lane();
    for (int e = tid; e < QREC; e += CIN_TPB) clu_ex[e] = seg_ex[e];
    for (int e = tid; e < REC; e += CIN_TPB) clu_pk[e] = seg_pk[e];
    if (tid == 0) { clu_ld[0] = seg_ld[0]; label[0] = 0; dist_off[0] = 0; dist_off[1] = 0; }
    const bool kl2 = kind == SPKD_KL2;               // distances from the records' KL2 vectors (k_cluster_prep), no eliminations
    if (kl2) for (int e = tid; e < AUX; e += CIN_TPB) clu_aux[e] = seg_aux[e];
    long long K = 1, off = 0, s = 1;
    __syncthreads();
    for (; s < n; ++s) {
        const double* A = seg_ex + s * QREC;
        stage_record(ldsA, A, tid, CIN_TPB);
        __syncthreads();
        const double nA = ldsA[QREC_COUNT_AT], ldS = seg_ld[s];
        for (long long base = 4 * wave; base < K && !kl2; base += 4 * CIN_WAVES) {
            const long long k = base + L.m;
            const bool valid = k < K;
            const long long kk = valid ? k : K - 1;
            const double v = quad_pair_det<TWO>(kind, ldsA, nA, A, clu_ex + kk * QREC, clu_pk + kk * REC, false, L, err);
            if (valid && L.t == 0) tmp[k] = v;
        }
        __syncthreads();
        const bool room = off + K <= dist_cap;
        for (long long k = tid; k < K && room; k += CIN_TPB) {
            if (kl2) {
                // (one lane per pair, 39-term sums done serially: the arithmetic of k_ahc's finish)
                const double* a1 = clu_aux + k * AUX;
                const double* a2 = seg_aux + s * AUX;
                double t1 = 0.0, t2 = 0.0;
                for (int i = 0; i < D; ++i) {
                    const float dm = (float)a1[2 * DA + i] - (float)a2[2 * DA + i];
                    const double delta = (double)dm;
                    t1 += (a1[i] - a2[i]) * (a2[DA + i] - a1[DA + i]);
                    t2 += ((a1[DA + i] + a2[DA + i]) * delta) * delta;
                }
                dist[off + k] = 0.5 * t1 + 0.5 * t2;
                continue;
            }
            const double ldx = log(tmp[k]);
            tmp[k] = ldx;
            dist[off + k] = finish_distance(kind, lambdac, clu_pk[k * REC + REC - 1], clu_ld[k], nA, ldS, ldx);
        }
        __syncthreads();
        if (tid == 0) {
            double mind = MAXINT_F;                      // sys.maxint
            int best = -1;
            for (long long k = 0; k < K && room; ++k) {
                const double d = dist[off + k];
                // "if d != inf and d != -inf" (a NaN passes and then fails every comparison)
                if (d != __builtin_huge_val() && d != -__builtin_huge_val() && d < mind) { mind = d; best = (int)k; }
            }
            s_mind = mind; s_best = best;
            int stop = room ? 0 : 1;
            if (!room) atomicOr(err, 4);                 // the host gave too small a buffer
            if (*reinterpret_cast<volatile int*>(err) & ERR_NONFINITE) stop = 1;
            s_stop = stop;
        }
        __syncthreads();
        if (s_stop) break;
        const int best = s_best;
        if (s_mind <= threshold && best >= 0) {
            double* Cx = clu_ex + (long long)best * QREC;
            double* Cp = clu_pk + (long long)best * REC;
            const double* Sp = seg_pk + s * REC;
            for (int e = tid; e < QREC; e += CIN_TPB) {
                const double m = Cx[e] + ldsA[e];
                Cx[e] = m;
                ldsA[e] = m;
            }
            for (int e = tid; e < REC; e += CIN_TPB) Cp[e] = Cp[e] + Sp[e];
            __syncthreads();
            if (kl2) {
                if (wave == 0) {                         // the grown cluster's KL2 vectors
                    double a[DA];
                    single_rows_from_qr(ldsA, a);
                    const double nM = ldsA[QREC_COUNT_AT];
                    const double mean_i = a[D] / nM;
                    cov_rows(a, nM);
                    kl2_aux_from_cov(a, mean_i, clu_aux + (long long)best * AUX);
                }
            } else if (TWO && kind == SPKD_GLR) {
                if (wave == 0) {
                    const double v = quad_pair_det<TWO>(kind, ldsA, ldsA[QREC_COUNT_AT], Cx, Cx, Cp, true, L, err);
                    if (lane == 0) clu_ld[best] = log(v);
                }
            } else if (tid == 0) {
                clu_ld[best] = tmp[best];                // the union of the pair IS the grown cluster
            }
            if (tid == 0) label[s] = best;
        } else {
            double* Cx = clu_ex + K * QREC;
            double* Cp = clu_pk + K * REC;
            const double* Sp = seg_pk + s * REC;
            for (int e = tid; e < QREC; e += CIN_TPB) Cx[e] = ldsA[e];
            for (int e = tid; e < REC; e += CIN_TPB) Cp[e] = Sp[e];
            if (kl2) for (int e = tid; e < AUX; e += CIN_TPB) clu_aux[K * AUX + e] = seg_aux[s * AUX + e];
            if (tid == 0) { clu_ld[K] = ldS; label[s] = (int32_t)K; }
        }
        off += K;
        if (!(s_mind <= threshold && best >= 0)) ++K;
        if (tid == 0) dist_off[s + 1] = off;
        __syncthreads();
    }
    if (tid == 0) { done[0] = s; done[1] = K; }
}

}  // namespace spkd
