// Wave-level (64-lane) building blocks shared by every kernel in this library.
//
// Layout convention ("row layout"): lane i (0..38) owns row i of a 39x39 matrix
// in 39 doubles held in registers; lane 39 owns the augmented row (sums, count);
// lanes 40..63 ride along with row 39's values and are ignored.  All loops over
// columns are fully unrolled so register indices are static; cross-lane data
// moves with v_readlane (static or wave-uniform source lane) -- no LDS.
//
// gfx950 only: wave64, no MFMA (the statistics are O(d^2) per frame and the
// factorisation is a 39-step dependent chain; throughput comes from many
// independent matrices in flight).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spkd {

constexpr int D = 39;          // feature dimension
constexpr int DA = 40;         // augmented dimension ([x; 1])
constexpr int REC = 820;       // packed upper triangle of DA x DA
constexpr int WAVE = 64;
// pointer to global memory in a non-kernel function: a plain pointer there is generic,
// its loads and stores are flat_* and count in lgkmcnt too, so every LDS wait would also
// wait for the global loads and stores in flight
#define SPKD_GLOBAL __attribute__((address_space(1)))
constexpr int REC_PER_LANE = 13;   // ceil(820 / 64)

__host__ __device__ constexpr int pk_off(int r) { return r * DA - (r * (r - 1)) / 2; }
__host__ __device__ constexpr int pk(int r, int c) { return pk_off(r) + (c - r); }   // r <= c

// entry (row r, column j <= r) of the lower triangle in a packed record: by symmetry
// pk(j, r), so column j holds its rows j .. 39 contiguously from pk_off(j)
__host__ __device__ constexpr int pk_low(int r, int j) { return pk_off(j) + (r - j); }

__device__ __forceinline__ void decode_entry(int e, int& r, int& c) {
    int rr = 0;
#pragma unroll 1
    while (rr + 1 < DA && pk_off(rr + 1) <= e) ++rr;
    r = rr;
    c = rr + (e - pk_off(rr));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ double readlane_d(double v, int src_lane) {
    // src_lane must be wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ long long uniform_ll(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

template <class T>
__device__ __forceinline__ T* uniform_p(T* p) { return (T*)uniform_ll((long long)p); }

__device__ __forceinline__ double uniform_d(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------
// Packed record -> row layout through a per-wave LDS slab of REC doubles.
//
// stage(): coalesced copy (13 doubles per lane) of  g1 [+ sgn * g2]  into the slab.
// row_from_slab(): q[j] = M(lane, j) for j = 0..39 (q[39] = sum of x_lane;
// lane 39 gets the sums row, its q[39] is the frame count).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void stage1(double* slab, const double* __restrict__ g) {
    const int lane = lane_id();
#pragma unroll
    for (int m = 0; m < REC_PER_LANE; ++m) {
        int e = lane + WAVE * m;
        if (e < REC) slab[e] = g[e];
    }
}

__device__ __forceinline__ void stage2(double* slab, const double* __restrict__ g1,
                                       const double* __restrict__ g2, double sgn) {
    const int lane = lane_id();
#pragma unroll
    for (int m = 0; m < REC_PER_LANE; ++m) {
        int e = lane + WAVE * m;
        if (e < REC) slab[e] = g1[e] + sgn * g2[e];
    }
}

__device__ __forceinline__ void row_from_slab(const double* slab, double (&q)[DA]) {
    int i = lane_id();
    i = i > D ? D : i;
    const int base_i = pk_off(i) - i;
#pragma unroll
    for (int j = 0; j < DA; ++j) {
        const int cj = pk_off(j) - j;
        int idx = (i <= j) ? (base_i + j) : (cj + i);
        q[j] = slab[idx];
    }
}

// q[j] += sgn * M(lane, j) from a slab
__device__ __forceinline__ void row_acc_slab(const double* slab, double (&q)[DA], double sgn) {
    int i = lane_id();
    i = i > D ? D : i;
    const int base_i = pk_off(i) - i;
#pragma unroll
    for (int j = 0; j < DA; ++j) {
        const int cj = pk_off(j) - j;
        int idx = (i <= j) ? (base_i + j) : (cj + i);
        q[j] = fma(sgn, slab[idx], q[j]);
    }
}

// Expanded LDS image of one record: ex[j * DA + i] = M(i, j), i, j in 0..39.
// Conflict-free for row-layout reads (consecutive lanes -> consecutive doubles).
__device__ __forceinline__ void expand_to_lds(double* ex, const double* __restrict__ g,
                                              int tid, int nthreads) {
    for (int t = tid; t < DA * DA; t += nthreads) {
        int j = t / DA, i = t - j * DA;
        int r = i < j ? i : j, c = i < j ? j : i;
        ex[t] = g[pk(r, c)];
    }
}

__device__ __forceinline__ void row_from_expanded(const double* ex, double (&q)[DA]) {
    int i = lane_id();
    i = i > D ? D : i;
#pragma unroll
    for (int j = 0; j < DA; ++j) q[j] = ex[j * DA + i];
}

// ---------------------------------------------------------------------------
// Covariance rows from second-moment rows:  np.cov(x, rowvar=0) semantics
// (numpy: subtract the mean, X^T X, multiply by 1/(n-1)) restated on raw
// moments: S_ij = (Q_ij - s_i s_j / n) * (1/(n-1)).
// q: second-moment row (q[39] = s_lane), overwritten by the covariance row;
// n = frame count (uniform).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void cov_rows(double (&q)[DA], double n) {   // in place
    const double inv_n = 1.0 / n;
    const double f = 1.0 / (n - 1.0);
    const double mi = -(q[D] * inv_n);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double sj = readlane_d(q[D], j);
        q[j] = fma(mi, sj, q[j]) * f;
    }
}

// true when every entry of every real row (lanes 0..38) is finite
__device__ __forceinline__ bool rows_finite(const double (&a)[DA]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < D; ++j) ok = ok && (fabs(a[j]) < __builtin_huge_val());
    ok = ok || (lane_id() >= D);
    return __all(ok);
}

// ---------------------------------------------------------------------------
// Determinant by Gaussian elimination, row layout.
//
// det_nopivot: for the symmetric positive definite case (every covariance of
// >= 40 real frames).  Straight-line code: 741 fp64 FMAs, pivot rows broadcast
// with static-lane v_readlane.  Returns false if some pivot is not a positive
// finite number; the caller then re-forms the matrix and calls det_pivoted.
//
// det_pivoted: partial pivoting with LAPACK dgetf2 semantics (first maximum in
// current row order, zero pivot => that elimination step is skipped), product
// of pivots in order k = 0..38 times the permutation sign -- the quantity
// scipy.linalg.det returns (scipy/linalg/_basic.py det -> getrf + diagonal
// product).  Destroys a.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool det_nopivot(double (&a)[DA], double& det_out) {
    double det = 1.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double piv = readlane_d(a[k], k);
        ok = ok && (piv > 0.0) && (piv < __builtin_huge_val());
        det *= piv;
        const double l = -(a[k] * (1.0 / piv));
#pragma unroll
        for (int j = k + 1; j < D; ++j) {
            const double u = readlane_d(a[j], k);
            a[j] = fma(l, u, a[j]);
        }
    }
    det_out = det;
    return ok;
}

__device__ __forceinline__ double det_pivoted(double (&a)[DA]) {
    const int lane = lane_id();
    int pos = lane;                       // current position of this row
    double det = 1.0;
    bool neg = false;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        // arg max |a[k]| over rows at positions >= k; ties -> smallest position
        double v = (lane < D && pos >= k) ? fabs(a[k]) : -1.0;
        if (v != v) v = -0.5;             // NaNs never win (idamax compares with >)
        int p = pos;
        int src = lane;
#pragma unroll
        for (int m = 1; m < WAVE; m <<= 1) {
            double v2 = __shfl_xor(v, m);
            int p2 = __shfl_xor(p, m);
            int s2 = __shfl_xor(src, m);
            bool take = (v2 > v) || (v2 == v && p2 < p);
            v = take ? v2 : v;
            p = take ? p2 : p;
            src = take ? s2 : src;
        }
        const int ps = uniform_i(src);    // lane holding the pivot row
        const int pp = uniform_i(p);      // its position
        const double piv = readlane_d(a[k], ps);
        if (pp != k) {                    // row interchange
            neg = !neg;
            int newpos = pos;
            if (pos == k) newpos = pp;
            if (lane == ps) newpos = k;
            pos = newpos;
        }
        det *= piv;
        double l = 0.0;
        if (piv != 0.0 && pos > k && lane < D) l = -(a[k] * (1.0 / piv));
#pragma unroll
        for (int j = k + 1; j < D; ++j) {
            const double u = readlane_d(a[j], ps);
            a[j] = fma(l, u, a[j]);
        }
    }
    return neg ? -det : det;
}

// In-place inverse of a symmetric positive definite matrix (Gauss-Jordan without
// pivoting, row layout); returns false when a pivot is not positive finite.
// Used for the diagonal of pinv(S) in KL2 (full rank: pinv == inverse).
__device__ __forceinline__ bool invert_spd(double (&a)[DA]) {
    const int lane = lane_id();
    bool ok = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double piv = readlane_d(a[k], k);
        ok = ok && (piv > 0.0) && (piv < __builtin_huge_val());
        const double r = 1.0 / piv;
        const double aik = a[k];
        const bool isk = (lane == k);
        const double l = isk ? 0.0 : -(aik * r);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (j == k) continue;
            const double u = readlane_d(a[j], k);
            a[j] = isk ? u * r : fma(l, u, a[j]);
        }
        a[k] = isk ? r : l;               // column k of the inverse-in-progress
    }
    return ok;
}

// a[lane] without dynamic register indexing
__device__ __forceinline__ double diag_of(const double (&a)[DA]) {
    const int lane = lane_id();
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) v = (lane == j) ? a[j] : v;
    return v;
}

__device__ __forceinline__ double wave_sum39(double v) {
    v = (lane_id() < D) ? v : 0.0;
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// ---------------------------------------------------------------------------
// log(det(A)) of a matrix produced by `form(a)` (fills a[0..38] per lane).
// Non-finite entries: the reference's scipy.linalg.det raises ValueError
// ("array must not contain infs or NaNs"); here the error bit is raised and
// NaN returned.  Fast path = det_nopivot; if that meets a non-positive pivot the
// matrix is formed again and eliminated with partial pivoting.
// numpy.log semantics: log(0) = -inf, log(negative) = NaN.
// ---------------------------------------------------------------------------
constexpr int ERR_NONFINITE = 1;

// Out-of-line slow paths.  The straight-line eliminations are several thousand
// instructions each; inlining the rarely taken ones at every call site blows the
// kernels far past the instruction cache (k_gw was > 500 KB of code), so they are
// real functions, emitted once per code object, taking the row array by pointer.
// determinant by partial pivoting (NaN + the error bit for non-finite input); its log,
// taken by the caller, is what scipy.linalg.det + numpy.log give: 0 -> -inf, negative -> NaN
__device__ __noinline__ double det_pivoted_fn(const double* rows, int* err) {
    double a[DA];
#pragma unroll
    for (int j = 0; j < DA; ++j) a[j] = rows[j];
    if (!rows_finite(a)) {
        if (lane_id() == 0) atomicOr(err, ERR_NONFINITE);
        return __builtin_nan("");
    }
    return det_pivoted(a);
}

__device__ __forceinline__ double logdet_pivoted_fn(const double* rows, int* err) {
    return log(det_pivoted_fn(rows, err));
}

template <class Form>
__device__ __forceinline__ double logdet_formed(double (&a)[DA], int* err, Form form) {
    form(a);
    if (!rows_finite(a)) {
        if (lane_id() == 0) atomicOr(err, ERR_NONFINITE);
        return __builtin_nan("");
    }
    double det;
    if (!det_nopivot(a, det)) {
        form(a);
        return logdet_pivoted_fn(a, err);
    }
    return log(det);
}

// out[0] = S_ii, out[1] = (S^-1)_ii (NaN if S is not positive definite), for the
// lane's row of the covariance held in rows[] (row-per-lane layout)
__device__ __noinline__ void spd_diag_terms_fn(const double* rows, double* out) {
    double a[DA];
#pragma unroll
    for (int j = 0; j < DA; ++j) a[j] = rows[j];
    out[0] = diag_of(a);
    const bool ok = invert_spd(a);
    double dp = diag_of(a);
    if (!ok) dp = __builtin_nan("");
    out[1] = dp;
}

}  // namespace spkd
