// Quad layout: FOUR 39x39 matrices per wave64, one per 16-lane DPP row.
//
// Inside a DPP row, lanes t = 0..12 each hold three rows of their matrix
// (row 13*s + t in slot s = 0, 1, 2); lanes 13..15 idle.  Row / column k of the
// matrices lives in lane (k % 13) slot (k / 13) of every DPP row, so one
// row_newbcast DPP operand reaches the element of all four matrices at once -- no
// v_readlane, no SGPR round trip, no LDS.  The elimination the kernels use works on
// the lower triangle only (spkd_tri.hpp); the full-square and column-blocked forms
// it superseded live in tools/ with the micro-benchmarks that compare them.
#pragma once
#include "spkd_device.hpp"

namespace spkd {

constexpr int QL = 13;        // row-carrying lanes per DPP row
constexpr int QS = 3;         // row slots per lane

struct QuadRows {
    double r[QS][D];
};

// broadcast lane K (0..15) of every 16-lane row to the whole row
template <int K>
__device__ __forceinline__ double bcast16(double v) {
    const long long x = __double_as_longlong(v);
    return __longlong_as_double(__builtin_amdgcn_update_dpp(0ll, x, 0x150 + K, 0xf, 0xf, true));
}

// 1 / x to ~1 ulp: hardware estimate + two Newton steps (5 instructions instead of the
// ~15 of an IEEE division; the multipliers are not part of any parity contract)
__device__ __forceinline__ double fast_recip(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

}  // namespace spkd

// ===========================================================================
// Quad records: the storage format the clustering kernels keep their working set
// in.  QREC = 3 slots x 40 columns x 16 lanes doubles (15 360 B):
//     qr[(s * 40 + j) * 16 + t] = M(13 s + t, j)   for t < 13 (M = augmented moments)
// so that a wave's load of (slot s, column j) is one fully used 128-B line per
// DPP row -- perfectly coalesced for the quad row layout.  Column 39 holds the
// sums; the frame count sits in the padding lane 15 of (slot 0, column 39).
// Records add component-wise (padding stays zero).
// ===========================================================================
namespace spkd {

constexpr int QREC = QS * DA * 16;
constexpr int QREC_COUNT_AT = (0 * DA + D) * 16 + 15;

__host__ __device__ constexpr int qr_index(int i, int j) { return ((i / QL) * DA + j) * 16 + (i % QL); }

struct QuadLane {
    int m;       // matrix index inside the wave (DPP row), 0..3
    int t;       // lane inside the DPP row, 0..15
};

__device__ __forceinline__ QuadLane quad_lane() {
    QuadLane L;
    const int lane = lane_id();
    L.m = lane >> 4;
    L.t = lane & 15;
    return L;
}

__device__ __forceinline__ double qr_count(const double* __restrict__ qr) { return qr[QREC_COUNT_AT]; }

// q = w * record rows (global or LDS pointer); sv[s] = sums column
__device__ __forceinline__ void quad_load_scaled(const double* __restrict__ qr, int t, double w,
                                                 QuadRows& q, double (&sv)[QS]) {
#pragma unroll
    for (int s = 0; s < QS; ++s) {
#pragma unroll
        for (int j = 0; j < D; ++j) q.r[s][j] = w * qr[(s * DA + j) * 16 + t];
        sv[s] = qr[(s * DA + D) * 16 + t];
    }
}


// ===========================================================================
// Tri records: the quad lane layout restricted to what the symmetric elimination reads,
// in 128-byte lines: line (tri_off(s) + j) holds M(13 s + t, j) for t = 0..12 in its first
// 13 doubles, j < 13 (s + 1); lines 78..80 hold the sums column of slot s; the frame count
// sits in lane 15 of line 78.  81 lines = 10 368 B; every load of (slot s, column j) is one
// aligned line per DPP row.
// ===========================================================================
constexpr int TLINES = 3 * QL + 3 * QL + 3;                  // 13 + 26 + 39 + 3 = 81
constexpr int TREC = TLINES * 16;                            // 1 296 doubles = 10 368 B
constexpr int TREC_SUMS = 6 * QL;                            // first sums line (78)
constexpr int TREC_COUNT_AT = TREC_SUMS * 16 + 15;
__host__ __device__ constexpr int tri_off(int s) { return s == 0 ? 0 : (s == 1 ? QL : 3 * QL); }

// entry (row r, column j <= r) of the 40x40 augmented matrix -> index in a tri record
__device__ __forceinline__ int tri_slot(int r, int j) {
    if (r < D) { const int s = r / QL; return (tri_off(s) + j) * 16 + (r - QL * s); }
    if (j < D) { const int s = j / QL; return (TREC_SUMS + s) * 16 + (j - QL * s); }
    return TREC_COUNT_AT;
}


// tri-record index idx -> packed index holding the same entry (-1: padding)
__device__ __forceinline__ int tri_image_source(int idx) {
    const int t = idx & 15, line = idx >> 4;
    if (line >= TREC_SUMS) {                     // sums lines: (39, 13 s + t), the count in lane 15 of the first
        const int s = line - TREC_SUMS;
        if (t < QL) return pk_low(D, QL * s + t);
        return idx == TREC_COUNT_AT ? REC - 1 : -1;
    }
    if (t >= QL) return -1;
    const int s = line < QL ? 0 : (line < 3 * QL ? 1 : 2);
    const int j = line - tri_off(s);
    const int r = QL * s + t;
    return r >= j ? pk_low(r, j) : pk_low(j, r);
}


}  // namespace spkd
