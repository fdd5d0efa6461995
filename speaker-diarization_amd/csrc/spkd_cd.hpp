// Change-detection kernels: growing window (dist_gw, spk-change-detection.py:180-288)
// and sliding window (dist_sw, spk-change-detection.py:291-357, distances only).
//
// One workgroup per VAD turn.  Phase 1 turns the turn's frames into prefix
// statistics snapshots every SNAP_G frames, stored as quad records (each frame is
// read once, coalesced, staged through LDS, fp64 accumulation in frame order).
// Phase 2 evaluates candidate split points: the statistics of any frame range
// [x, y) are R(y) - R(x) with R(t) = snapshot[t / G] + the <= G-1 edge frames, so a
// candidate costs one record read and one or two 39x39 eliminations instead of two
// np.cov passes over raw frames.  Four candidates share a wave (quad layout,
// spkd_quad.hpp).  The growing-window decision chain (a float state machine with
// int() truncation) runs on the device with the exact operation
// order of the reference.
#pragma once
#include "spkd_device.hpp"
#include "spkd_quad.hpp"
#include "spkd_cluster.hpp"
#include "../../include/spkd.h"

namespace spkd {

constexpr int SNAP_G = 16;
constexpr int CD_TILE = 64;
constexpr double NEG_MAXINT_M1 = -9223372036854775808.0;   // -sys.maxint - 1
constexpr int QUSE = QS * DA * QL;                         // row-carrying entries of a quad record

struct TurnDesc {
    int64_t begin;       // first frame of the turn in the frame array
    int64_t len;         // frames
    int64_t snap_off;    // first snapshot record of this turn
    int64_t cand_off;    // first candidate slot
    int64_t cand_cap;    // candidate slots owned by this turn
    int64_t ev_off;      // first event slot
    int64_t ev_cap;      // event slots owned by this turn
    int64_t id;          // caller's turn index (blocks are launched longest turn first)
};

// e in [0, QUSE) -> quad-record index and the (row, column) it holds
__device__ __forceinline__ void quse_decode(int e, int& idx, int& i, int& j) {
    const int s = e / (DA * QL), rem = e - s * (DA * QL);
    j = rem / QL;
    const int t = rem - j * QL;
    i = QL * s + t;
    idx = (s * DA + j) * 16 + t;
}

// Phase 1: snapshots snap[k] = moments of turn frames [0, G*k), k = 0 .. len / G.
// Thread (row i, column group g) owns the entries (i, g + 6 m), m = 0..6: per frame
// it reads x_i once and one x_j per entry from the LDS tile (already converted to
// double when staged), i.e. 8 LDS reads + 7 FMAs per frame.  Sums are in frame order.
constexpr int P1_GROUPS = 6;
constexpr int P1_COLS = (DA + P1_GROUPS - 1) / P1_GROUPS;     // 7

template <int TPB>
__device__ __forceinline__ void build_prefix(const float* __restrict__ fr, long long n,
                                             double* __restrict__ snap, double* xs) {
    static_assert(TPB >= D * P1_GROUPS, "phase 1 needs 234 threads");
    const int tid = threadIdx.x;
    const bool on = tid < D * P1_GROUPS;
    const int i = on ? tid % D : 0, g = on ? tid / D : 0;
    const int rbase = (i / QL) * DA * 16 + (i % QL);          // + j * 16
    double acc[P1_COLS];
#pragma unroll
    for (int m = 0; m < P1_COLS; ++m) {
        acc[m] = 0.0;
        const int j = g + P1_GROUPS * m;
        if (on && j < DA) snap[rbase + j * 16] = 0.0;
    }
    if (tid == 0) snap[QREC_COUNT_AT] = 0.0;
    for (long long t0 = 0; t0 < n; t0 += CD_TILE) {
        const int tl = (int)((n - t0) < CD_TILE ? (n - t0) : CD_TILE);
        const float* src = fr + t0 * D;
        for (int idx = tid; idx < tl * D; idx += TPB) {
            int f = idx / D, c = idx - f * D;
            xs[f * DA + c] = (double)src[idx];
        }
        if (tid < tl) xs[tid * DA + D] = 1.0;
        __syncthreads();
        for (int f = 0; f < tl; ++f) {
            const double xi = xs[f * DA + i];
#pragma unroll
            for (int m = 0; m < P1_COLS; ++m) {
                const int j = g + P1_GROUPS * m;
                acc[m] = fma(xi, xs[f * DA + (j < DA ? j : 0)], acc[m]);
            }
            const long long t = t0 + f + 1;
            if ((t % SNAP_G) == 0) {
                double* dst = snap + (t / SNAP_G) * QREC;
#pragma unroll
                for (int m = 0; m < P1_COLS; ++m) {
                    const int j = g + P1_GROUPS * m;
                    if (on && j < DA) dst[rbase + j * 16] = acc[m];
                }
                if (tid == 0) dst[QREC_COUNT_AT] = (double)t;
            }
        }
        __syncthreads();
    }
}

// LDS quad record of R(t), built by the whole workgroup (padding lanes untouched:
// the caller zeroes them once).  The <= G-1 edge frames are staged in LDS first
// (one coalesced read), then every entry adds its products in frame order.
// Contains block-wide barriers: call from uniform control flow.
__device__ __forceinline__ void build_record_lds(double* lds, const double* __restrict__ snap,
                                                 const float* __restrict__ fr, long long t,
                                                 double* xs, int tid, int nthreads) {
    const long long k = t / SNAP_G;
    const int ne = (int)(t - k * SNAP_G);
    const float* src = fr + k * SNAP_G * D;
    for (int idx = tid; idx < ne * D; idx += nthreads) {
        const int f = idx / D, c = idx - f * D;
        xs[f * DA + c] = (double)src[idx];
    }
    if (tid < ne) xs[tid * DA + D] = 1.0;
    __syncthreads();
    const double* s = snap + k * QREC;
    for (int e = tid; e < QUSE; e += nthreads) {
        int idx, i, j;
        quse_decode(e, idx, i, j);
        double v = s[idx];
        for (int f = 0; f < ne; ++f) v = fma(xs[f * DA + i], xs[f * DA + j], v);
        lds[idx] = v;
    }
    __syncthreads();
}

// Quad-layout prefix statistics: DPP row m gets R(tpos) for its own tpos.
// q = rows (columns 0..38), sv = sums column.  The <= G-1 edge frames of the four
// positions are first copied to a per-wave LDS buffer (coalesced, few registers),
// the snapshot rows are loaded straight into the row registers, then the edge
// frames are added in frame order (positions with fewer edge frames add zeros).
constexpr int EDGE_FLOATS = (SNAP_G - 1) * D;          // per position (<= G/2 used since snapshots are two-sided)
constexpr int EDGE_WAVE_FLOATS = 4 * EDGE_FLOATS;      // per wave

__device__ __forceinline__ void quad_prefix_rows(QuadRows& q, double (&sv)[QS],
                                                 const double* __restrict__ snap,
                                                 const float* __restrict__ fr, long long tpos,
                                                 long long nturn, const QuadLane& L,
                                                 float* edge /* per-wave LDS */) {
    // nearest snapshot: below (add the frames [G k, t)) or above (subtract [t, G (k+1)))
    long long k = tpos / SNAP_G;
    int ne = (int)(tpos - k * SNAP_G);
    double sgn = 1.0;
    long long first = k * SNAP_G;                 // first edge frame
    if (ne > SNAP_G / 2 && (k + 1) * SNAP_G <= nturn) {
        k += 1;
        first = tpos;
        ne = SNAP_G - ne;
        sgn = -1.0;
    }
    const double* s = snap + k * QREC;
    const int lane = lane_id();
    // stage the edge frames of the four positions (contiguous floats each)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const long long fm = __shfl(first, 16 * mi);
        const int nem = __shfl(ne, 16 * mi);
        const float* src = fr + fm * D;
        for (int idx = lane; idx < nem * D; idx += WAVE) edge[mi * EDGE_FLOATS + idx] = src[idx];
    }
#pragma unroll
    for (int ss = 0; ss < QS; ++ss) {
#pragma unroll
        for (int j = 0; j < tri_cols(ss); ++j) q.r[ss][j] = s[(ss * DA + j) * 16 + L.t];
        sv[ss] = s[(ss * DA + D) * 16 + L.t];
    }
    int mx = ne;
    mx = max(mx, __shfl_xor(mx, 16));
    mx = max(mx, __shfl_xor(mx, 32));
    mx = uniform_i(mx);
    const float* mine = edge + L.m * EDGE_FLOATS;
    const bool carrier = L.t < QL;
#pragma unroll 1
    for (int e = 0; e < mx; ++e) {
        double x[QS], cx[QS];
#pragma unroll
        for (int ss = 0; ss < QS; ++ss) {
            const float xf = (carrier && e < ne) ? mine[e * D + QL * ss + (carrier ? L.t : 0)] : 0.0f;
            x[ss] = (double)xf;
            cx[ss] = sgn * x[ss];
            sv[ss] += cx[ss];
        }
        TriRank1<0>::run(q, cx, x);
    }
}

// Row-per-lane (single matrix) prefix rows, used by KL2 and by the pivoting fallback.
__device__ __forceinline__ void single_prefix_rows(double (&q)[DA], const double* __restrict__ snap,
                                                   const float* __restrict__ fr, long long t) {
    const long long k = t / SNAP_G;
    single_rows_from_qr(snap + k * QREC, q);
    const int lane = lane_id();
    for (long long f = k * SNAP_G; f < t; ++f) {
        const float x = (lane < D) ? fr[f * D + lane] : 0.0f;
        const double xi = (double)x;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const double xc = readlane_d(xi, c);
            q[c] = fma(xi, xc, q[c]);
        }
        q[D] += xi;
    }
}

__device__ __forceinline__ void single_rows_from_lds(const double* lds, double (&q)[DA]) {
    int i = lane_id();
    i = i >= D ? D - 1 : i;
    const int base = (i / QL) * DA * 16 + (i % QL);
#pragma unroll
    for (int j = 0; j < DA; ++j) q[j] = lds[base + j * 16];
}

struct BestD {
    double d;
    long long k;
};

// first-index arg max over candidates accepted by "d > maxd and d != inf"
// starting from `floor` (exclusive); k = -1 when none.
template <int NWAVES>
__device__ __forceinline__ BestD block_argmax(const double* __restrict__ vals, long long count,
                                              double floor_excl, BestD* red) {
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    BestD b;
    b.d = floor_excl; b.k = -1;
    for (long long k = tid; k < count; k += NWAVES * WAVE) {
        const double d = vals[k];
        if (d > b.d && d != __builtin_huge_val()) { b.d = d; b.k = k; }   // NaN fails d > b.d
    }
#pragma unroll
    for (int s = 1; s < WAVE; s <<= 1) {
        const double d2 = __shfl_xor(b.d, s);
        const long long k2 = __shfl_xor(b.k, s);
        const bool take = (k2 >= 0) && (b.k < 0 || d2 > b.d || (d2 == b.d && k2 < b.k));
        if (take) { b.d = d2; b.k = k2; }
    }
    if (lane == 0) red[wave] = b;
    __syncthreads();
    BestD r = red[0];
    for (int w = 1; w < NWAVES; ++w) {
        const BestD o = red[w];
        const bool take = (o.k >= 0) && (r.k < 0 || o.d > r.d || (o.d == r.d && o.k < r.k));
        if (take) r = o;
    }
    __syncthreads();
    return r;
}

__device__ __forceinline__ void log_cand(spkd_cand_log* log, long long cap, unsigned long long* count,
                                         int turn, int coarse, long long seq, double start, double i,
                                         double d, long long n1, long long n2) {
    const unsigned long long slot = atomicAdd(count, 1ull);
    if ((long long)slot < cap) {
        spkd_cand_log r;
        r.turn = turn; r.coarse = coarse; r.seq = seq; r.start = start; r.i = i; r.d = d;
        r.n1 = n1; r.n2 = n2;
        log[slot] = r;
    }
}

// which matrix of a split [a, b) | [b, c) a pass forms
enum { PASS_RIGHT = 0, PASS_LEFT = 1, PASS_GLR = 2, PASS_POOLED = 3 };

// single-matrix form of a pass (fallback / KL2): rows of the moment difference,
// then covariance (or GLR's weighted covariance)
__device__ __forceinline__ void single_split_matrix(int pass, const double* ldsRa, const double* ldsRc,
                                                    const double* __restrict__ snap,
                                                    const float* __restrict__ fr, long long a,
                                                    long long b, long long c, double (&q)[DA]) {
    double ra[DA], rc[DA];
    if (pass == PASS_POOLED) {
        single_rows_from_lds(ldsRc, q);
        single_rows_from_lds(ldsRa, ra);
#pragma unroll
        for (int j = 0; j < DA; ++j) q[j] -= ra[j];
        cov_rows(q, (double)(c - a));
        return;
    }
    single_prefix_rows(q, snap, fr, b);
    const double n1 = (double)(b - a), n2 = (double)(c - b), n = n1 + n2;
    if (pass == PASS_RIGHT) {
        single_rows_from_lds(ldsRc, rc);
#pragma unroll
        for (int j = 0; j < DA; ++j) q[j] = rc[j] - q[j];
        cov_rows(q, n2);
    } else if (pass == PASS_LEFT) {
        single_rows_from_lds(ldsRa, ra);
#pragma unroll
        for (int j = 0; j < DA; ++j) q[j] -= ra[j];
        cov_rows(q, n1);
    } else {
        single_rows_from_lds(ldsRa, ra);
        single_rows_from_lds(ldsRc, rc);
        const double al1 = (n1 / n) / (n1 - 1.0), al2 = (n2 / n) / (n2 - 1.0);
        const double be1 = al1 / n1, be2 = al2 / n2;
        const double s1i = q[D] - ra[D], s2i = rc[D] - q[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double q1 = q[j] - ra[j], q2 = rc[j] - q[j];
            const double s1j = readlane_d(s1i, j), s2j = readlane_d(s2i, j);
            double v = fma(al2, q2, al1 * q1);
            v = fma(-(be1 * s1i), s1j, v);
            q[j] = fma(-(be2 * s2i), s2j, v);
        }
    }
}

// log det of one pass for the four split points held by the wave (DPP row m: split b).
// Every pass is the same straight-line code with per-pass coefficients (a switch
// over differently shaped paths makes the compiler hoist the LDS reads common to
// several branches above the switch and spill hundreds of registers):
//     q  = al * R(b) + be * R(a) + ga * R(c)           (moments, rows + sums column)
//     q += c1 v1^T + c2 v2^T   (al, be, ga, c1 carry the covariance scale f)
//   right  [b, c): al = -1, be =  0, ga = 1; c1 = -s / n2, v1 = s;  f = 1 / (n2 - 1)
//   left   [a, b): al =  1, be = -1, ga = 0; c1 = -s / n1, v1 = s;  f = 1 / (n1 - 1)
//   pooled [a, c): al =  0, be = -1, ga = 1; c1 = -s / N,  v1 = s;  f = 1 / (N - 1)
//   GLR: al1 (Rb - Ra) + al2 (Rc - Rb) - be1 s1 s1^T - be2 s2 s2^T;  f = 1
__device__ __forceinline__ double quad_split_logdet(int pass, const double* ldsRa, const double* ldsRc,
                                                    const double* __restrict__ snap,
                                                    const float* __restrict__ fr, long long a,
                                                    long long b, long long c, long long nturn,
                                                    const QuadLane& L, float* edge, int* err) {
    QuadRows q;
    double svb[QS];
    int ta = L.t;
    asm volatile("" : "+v"(ta));          // keep the LDS reads out of the callers' loops
    const double n1 = (double)(b - a), n2 = (double)(c - b), n = n1 + n2;
    double al, be, ga, f, k1a, k1b, k1c, k2a, k2b, k2c, w1, w2;
    // v1 = k1a * sb + k1b * sa + k1c * sc ;  c1 = w1 * v1  (same for v2, c2)
    if (pass == PASS_GLR) {
        const double al1 = (n1 / n) / (n1 - 1.0), al2 = (n2 / n) / (n2 - 1.0);
        al = al1 - al2; be = -al1; ga = al2; f = 1.0;
        k1a = 1.0; k1b = -1.0; k1c = 0.0; w1 = -(al1 / n1);
        k2a = -1.0; k2b = 0.0; k2c = 1.0; w2 = -(al2 / n2);
    } else {
        const double np = pass == PASS_RIGHT ? n2 : (pass == PASS_LEFT ? n1 : n);
        al = pass == PASS_RIGHT ? -1.0 : (pass == PASS_LEFT ? 1.0 : 0.0);
        be = pass == PASS_RIGHT ? 0.0 : -1.0;
        ga = pass == PASS_LEFT ? 0.0 : 1.0;
        f = 1.0 / (np - 1.0);
        k1a = al; k1b = be; k1c = ga; w1 = -(f / np);
        k2a = 0.0; k2b = 0.0; k2c = 0.0; w2 = 0.0;
        al *= f; be *= f; ga *= f;          // covariance scale folded into the combine
    }
    if (pass != PASS_POOLED) {
        quad_prefix_rows(q, svb, snap, fr, b, nturn, L, edge);
    } else {                              // al = 0: R(b) is not part of the pooled window
#pragma unroll
        for (int s = 0; s < QS; ++s) {
#pragma unroll
            for (int j = 0; j < tri_cols(s); ++j) q.r[s][j] = 0.0;
            svb[s] = 0.0;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    double v1[QS], v2[QS], c1[QS], c2[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
#pragma unroll
        for (int j = 0; j < tri_cols(s); ++j) {
            const double ra = ldsRa[(s * DA + j) * 16 + ta], rc = ldsRc[(s * DA + j) * 16 + ta];
            q.r[s][j] = fma(ga, rc, fma(be, ra, al * q.r[s][j]));
        }
        const double sa = ldsRa[(s * DA + D) * 16 + ta], sc = ldsRc[(s * DA + D) * 16 + ta];
        v1[s] = fma(k1c, sc, fma(k1b, sa, k1a * svb[s]));
        v2[s] = fma(k2c, sc, fma(k2b, sa, k2a * svb[s]));
        c1[s] = w1 * v1[s];
        c2[s] = w2 * v2[s];
        __builtin_amdgcn_sched_barrier(0);
    }
    TriRank1<0>::run(q, c1, v1);
    if (pass == PASS_GLR) TriRank1<0>::run(q, c2, v2);
    auto form_single = [&](int mi, double (&arr)[DA]) {
        const long long bm = __shfl(b, 16 * mi);
        single_split_matrix(pass, ldsRa, ldsRc, snap, fr, a, bm, c, arr);
    };
    return tri_logdet(q, L.m, err, form_single);
}

constexpr int GW_WAVES = 4;
constexpr int GW_TPB = GW_WAVES * WAVE;
constexpr int GW_EDGE_BYTES = GW_WAVES * 4 * (SNAP_G - 1) * D * 4;
constexpr int GW_TILE_BYTES = CD_TILE * DA * 8;
constexpr int GW_LDS_BYTES = 2 * QREC * 8 + (GW_EDGE_BYTES > GW_TILE_BYTES ? GW_EDGE_BYTES : GW_TILE_BYTES);

// Candidate scratch per slot k: c_i = i value, c_left = memoised left term (BIC:
// 0.5 N1 log det S1; GLR: log det S1), c_x = right log det / finished distance.
//
// The kernel is one loop over "scans": a coarse scan over the candidates
// i = minfeas, minfeas + istep, ... (CD:204-221) and, after a positive one, a
// fine scan over single-frame steps around the maximum (CD:235-251).  Both kinds
// share one body, so the elimination code exists once.
__global__ __launch_bounds__(GW_TPB, 2) void k_gw(
        const float* __restrict__ frames, const TurnDesc* __restrict__ turns, spkd_cd_params P,
        double* __restrict__ snap_all, double* __restrict__ cand_all,
        int32_t* __restrict__ n_win, double* __restrict__ win_maxd, int32_t* __restrict__ win_det,
        double* __restrict__ det_start, double* __restrict__ det_maxi, double* __restrict__ det_d,
        double* __restrict__ final_start, spkd_cand_log* clog, long long log_cap,
        unsigned long long* log_count, int* err) {
    // dynamic LDS (GW_LDS_BYTES): Ra | Rc | per-wave edge buffers; the frame tile of
    // phase 1 / build_record_lds aliases the edge buffers (never live together)
    extern __shared__ double gw_lds[];
    double* ldsRa = gw_lds;
    double* ldsRc = gw_lds + QREC;
    float* edges = (float*)(gw_lds + 2 * QREC);
    double* xs = gw_lds + 2 * QREC;
    __shared__ BestD red[GW_WAVES];
    __shared__ double s_ldS;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const QuadLane L = quad_lane();
    const TurnDesc T = turns[blockIdx.x];
    const int turn = (int)T.id;
    const long long n = T.len;
    const float* fr = frames + T.begin * D;
    double* snap = snap_all + T.snap_off * QREC;
    const long long cap = T.cand_cap;
    double* c_i = cand_all + 3 * T.cand_off;
    double* c_left = c_i + cap;
    double* c_x = c_left + cap;

    for (int e = tid; e < QREC; e += GW_TPB) { ldsRa[e] = 0.0; ldsRc[e] = 0.0; }
    build_prefix<GW_TPB>(fr, n, snap, xs);
    __syncthreads();

    const int kind = P.kind;
    const double winsize = P.winsize, winstep = P.winstep, rate = P.rate;
    double start = 0.0;
    double end = start + winsize * 2;
    const double minfeas = rate / 2;
    const double istep = rate / 10;
    double ws = minfeas;
    double dws = P.deltaws;
    long long n_memo = 0;            // coarse candidates [0, n_memo) have their left term memoised
    long long n_written = 0;         // c_i[0 .. n_written) hold the coarse i sequence
    long long C = 0;                 // coarse candidates of the current scan
    double cur_i = minfeas;          // next i of the coarse sequence for this epoch
    long long a_cur = -1, c_cur = -1;
    int nw = 0, nd = 0;
    const double fn = (double)n;
    const double pen_w = P.lambdac * 0.5 * PEN_UNIT;
    bool fine = false;               // kind of the scan about to run
    double maxd = 0.0, maxi = 0.0;   // running maximum (carried from the coarse into the fine scan)
    double fine_i0 = 0.0;
    long long F = 0;

    while (fine || end <= fn) {
        long long base, count;
        if (!fine) {
            if (nw >= T.ev_cap) { if (tid == 0) atomicOr(err, 4); break; }
            const double lim = end - start - minfeas;
            while (cur_i < lim) {
                if (C >= cap) break;
                if (C >= n_written) { if (tid == 0) c_i[C] = cur_i; n_written = C + 1; }
                ++C;
                cur_i += istep;
            }
            base = 0;
            count = C;
        } else {
            fine_i0 = maxi - istep;
            const double endtune = maxi + istep;
            F = 0;
            for (double x = fine_i0; x < endtune; x += 1) ++F;
            if (C + F > cap) { if (tid == 0) atomicOr(err, 4); break; }
            // the fine i sequence, by repeated +1 like the reference
            if (tid == 0) {
                double x = fine_i0;
                for (long long k = 0; k < F; ++k) { c_i[C + k] = x; x += 1; }
            }
            base = C;                // fine-scan scratch sits behind the coarse slots
            count = F;
        }
        const long long a = (long long)start, c = (long long)end;
        if (a != a_cur) { build_record_lds(ldsRa, snap, fr, a, xs, tid, GW_TPB); a_cur = a; }
        if (c != c_cur) { build_record_lds(ldsRc, snap, fr, c, xs, tid, GW_TPB); c_cur = c; }
        __syncthreads();
        const double N = (double)(c - a);
        const bool pooled = (kind == SPKD_BIC && !fine);
        if (kind == SPKD_KL2) {
            // one wave per split point, single-matrix layout (Gauss-Jordan inverses)
            for (long long job = wave; job < count; job += GW_WAVES) {
                const double ik = c_i[base + job];
                const long long b = (long long)(start + ik);
                const double n1 = (double)(b - a), n2 = (double)(c - b);
                double ds[2], dp[2], mu[2];
#pragma unroll 1
                for (int t = 0; t < 2; ++t) {
                    double a_[DA], r_[DA];
                    single_prefix_rows(a_, snap, fr, b);
                    single_rows_from_lds(t ? ldsRc : ldsRa, r_);
#pragma unroll
                    for (int j = 0; j < DA; ++j) a_[j] = t ? (r_[j] - a_[j]) : (a_[j] - r_[j]);
                    const double nn = t ? n2 : n1;
                    const double mean_i = a_[D] / nn;
                    cov_rows(a_, nn);
                    kl2_lane_terms(a_, mean_i, ds[t], dp[t], mu[t]);
                }
                const double dist = kl2_combine(ds[0], dp[0], mu[0], ds[1], dp[1], mu[1]);
                if (lane == 0) c_x[base + job] = dist;
            }
        } else {
            // quad jobs: job -1 = pooled window (BIC coarse scans), then groups of 4 split
            // points; ONE call site of the elimination for all passes (code size)
            const long long nquads = (count + 3) / 4;
            for (long long job = (long long)wave - (pooled ? 1 : 0); job < nquads; job += GW_WAVES) {
                const bool is_pooled = job < 0;
                long long k = is_pooled ? 0 : 4 * job + L.m;
                const bool valid = is_pooled ? false : (k < count);
                k = (k < count) ? k : count - 1;
                const long long slot = base + k;
                const double ik = c_i[slot];
                const long long b = is_pooled ? a : (long long)(start + ik);
                const double n1 = (double)(b - a), n2 = (double)(c - b);
                const bool need_left = fine || k >= n_memo;
                const bool any_left = __any(need_left && valid);
                double lds_[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
                for (int p = 0; p < 4; ++p) {
                    if (is_pooled != (p == PASS_POOLED)) continue;
                    if (p == PASS_LEFT && !any_left) continue;
                    if (p == PASS_GLR && kind != SPKD_GLR) continue;
                    const double v = quad_split_logdet(p, ldsRa, ldsRc, snap, fr, a, b, c, n, L, edges + wave * EDGE_WAVE_FLOATS, err);
                    if (p == PASS_RIGHT) lds_[0] = v; else if (p == PASS_LEFT) lds_[1] = v;
                    else if (p == PASS_GLR) lds_[2] = v; else lds_[3] = v;
                }
                if (is_pooled) {
                    if (lane == 0) s_ldS = lds_[3];
                    continue;
                }
                const double ld_right = lds_[0], ld_left = lds_[1], ld_w = lds_[2];
                if (valid && L.t == 0) {
                    if (kind == SPKD_BIC) {
                        // left term: 0.5 N1 log det S1, memoised per i inside an epoch (CD:84-90)
                        if (need_left) c_left[slot] = 0.5 * n1 * ld_left;
                        c_x[slot] = ld_right;
                    } else {
                        const double l1 = need_left ? ld_left : c_left[slot];
                        if (need_left && !fine) c_left[slot] = l1;
                        c_x[slot] = -(N / 2.0) * ((n1 / N) * l1 + (n2 / N) * ld_right - ld_w);
                    }
                }
            }
        }
        __syncthreads();
        // ---- finish the distances (BIC): d = 0.5 N log|S| - c1 - 0.5 N2 log|S2| - penalty
        if (kind == SPKD_BIC) {
            if (fine) { /* s_ldS still holds the pooled term of this window */ }
            const double ldS = s_ldS;
            const double corr = pen_w * log(N);
            for (long long k = tid; k < count; k += GW_TPB) {
                const long long slot = base + k;
                const long long b = (long long)(start + c_i[slot]);
                const double n2 = (double)(c - b);
                double d = 0.5 * N * ldS - c_left[slot] - 0.5 * n2 * c_x[slot];
                d -= corr;
                c_x[slot] = d;
            }
            __syncthreads();
        }
        for (long long k = tid; k < count; k += GW_TPB) {
            const long long slot = base + k;
            const double d = c_x[slot];
            if ((P.trace && !fine) || fabs(d) == __builtin_huge_val()) {
                const long long b = (long long)(start + c_i[slot]);
                const long long w = fine ? (long long)(nw - 1) : (long long)nw;
                log_cand(clog, log_cap, log_count, turn, fine ? 0 : 1,
                         (w << 32) | (fine ? 0x80000000LL : 0LL) | k, start, c_i[slot], d, b - a, c - b);
            }
        }
        if (!fine) {
            if (C > n_memo) n_memo = C;
            BestD best = block_argmax<GW_WAVES>(c_x, C, NEG_MAXINT_M1, red);
            const bool found = best.k >= 0;
            maxd = best.d;
            maxi = found ? c_i[best.k] : 0.0;
            if (tid == 0) {
                win_maxd[T.ev_off + nw] = found ? maxd : __builtin_nan("");
                win_det[T.ev_off + nw] = 0;
            }
            ++nw;
            if (found && maxd > P.threshold) {
                fine = true;             // next scan: fine tune around maxi (CD:231-251)
                continue;
            }
            // negative: enlarge the window (CD:271-284)
            if (end + ws <= fn) {
                end += ws;
                if (ws < winstep) { ws += dws; dws *= 2; }
                if (ws > winstep) ws = winstep;
            } else if (end != fn) {
                end = fn;
            } else {
                break;
            }
        } else {
            BestD fb = block_argmax<GW_WAVES>(c_x + C, F, maxd, red);
            if (fb.k >= 0) { maxd = fb.d; maxi = c_i[C + fb.k]; }
            if (tid == 0) {
                det_start[T.ev_off + nd] = start;
                det_maxi[T.ev_off + nd] = maxi;
                det_d[T.ev_off + nd] = maxd;
                win_det[T.ev_off + nw - 1] = 1;
            }
            ++nd;
            __syncthreads();             // reads of the scratch slots are done
            fine = false;
            n_memo = 0;
            n_written = n_written < C ? n_written : C;   // fine-scan scratch overwrote slots >= C
            C = 0;
            cur_i = minfeas;
            start += maxi;
            if (start + winsize * 2 <= fn) {
                end = start + winsize * 2;
                ws = minfeas;
                dws = P.deltaws;
            } else {
                break;
            }
        }
    }
    if (tid == 0) {
        n_win[turn] = nw;
        final_start[turn] = start;
    }
}

// ---------------------------------------------------------------------------
// Sliding window: every window is independent; one wave per window, row-per-lane
// layout (this mode is not on the DIA2 path; the quad layout is used where the
// time goes).
constexpr int SW_WAVES = 4;
constexpr int SW_TPB = SW_WAVES * WAVE;

__global__ __launch_bounds__(SW_TPB) void k_sw(
        const float* __restrict__ frames, const TurnDesc* __restrict__ turns, spkd_cd_params P,
        double* __restrict__ snap_all, double* __restrict__ d_out, int* err) {
    __shared__ double xs[CD_TILE * DA];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int turn = blockIdx.x;
    const TurnDesc T = turns[turn];
    const long long n = T.len;
    const float* fr = frames + T.begin * D;
    double* snap = snap_all + T.snap_off * QREC;
    build_prefix<SW_TPB>(fr, n, snap, xs);
    __syncthreads();
    const int kind = P.kind;
    const double winsize = P.winsize, winstep = P.winstep;
    long long W = 0;
    for (double s = 0; s + 2 * winsize <= (double)n; s += winstep) ++W;
    const long long wsz = (long long)winsize;
    for (long long w = wave; w < W; w += SW_WAVES) {
        const long long a = (long long)((double)w * winstep);
        const long long m = a + wsz, e = a + 2 * wsz;
        double qa[DA], qm[DA], qe[DA], a_[DA];
        single_prefix_rows(qa, snap, fr, a);
        single_prefix_rows(qm, snap, fr, m);
        single_prefix_rows(qe, snap, fr, e);
        const double n1 = (double)wsz, n2 = (double)wsz, N = n1 + n2;
        double r[3] = {0.0, 0.0, 0.0};
        double kl = 0.0;
        if (kind == SPKD_KL2) {
            double ds[2], dp[2], mu[2];
#pragma unroll 1
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int j = 0; j < DA; ++j) a_[j] = t ? (qe[j] - qm[j]) : (qm[j] - qa[j]);
                const double mean_i = a_[D] / n1;
                cov_rows(a_, n1);
                kl2_lane_terms(a_, mean_i, ds[t], dp[t], mu[t]);
            }
            kl = kl2_combine(ds[0], dp[0], mu[0], ds[1], dp[1], mu[1]);
        } else {
#pragma unroll 1
            for (int t = 0; t < 3; ++t) {
                auto form = [&](double (&q)[DA]) {
                    if (t == 0) {
#pragma unroll
                        for (int j = 0; j < DA; ++j) q[j] = qm[j] - qa[j];
                        cov_rows(q, n1);
                    } else if (t == 1) {
#pragma unroll
                        for (int j = 0; j < DA; ++j) q[j] = qe[j] - qm[j];
                        cov_rows(q, n2);
                    } else if (kind == SPKD_BIC) {
#pragma unroll
                        for (int j = 0; j < DA; ++j) q[j] = qe[j] - qa[j];
                        cov_rows(q, N);
                    } else {
                        const double al1 = (n1 / N) / (n1 - 1.0), al2 = (n2 / N) / (n2 - 1.0);
                        const double be1 = al1 / n1, be2 = al2 / n2;
                        const double s1i = qm[D] - qa[D];
                        const double s2i = qe[D] - qm[D];
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            const double q1 = qm[j] - qa[j];
                            const double q2 = qe[j] - qm[j];
                            const double s1j = readlane_d(s1i, j), s2j = readlane_d(s2i, j);
                            double v = fma(al2, q2, al1 * q1);
                            v = fma(-(be1 * s1i), s1j, v);
                            q[j] = fma(-(be2 * s2i), s2j, v);
                        }
                    }
                };
                r[t] = logdet_formed(a_, err, form);
            }
        }
        if (lane == 0) {
            double d;
            if (kind == SPKD_BIC) {
                d = 0.5 * N * r[2] - 0.5 * n1 * r[0] - 0.5 * n2 * r[1];
                d -= P.lambdac * 0.5 * PEN_UNIT * log(N);
            } else if (kind == SPKD_GLR) {
                d = -(N / 2.0) * ((n1 / N) * r[0] + (n2 / N) * r[1] - r[2]);
            } else {
                d = kl;
            }
            d_out[T.ev_off + w] = d;
        }
    }
}

}  // namespace spkd
