// Change-detection kernels: growing window (dist_gw, spk-change-detection.py:180-288)
// and sliding window (dist_sw, spk-change-detection.py:291-357, distances only).
//
// One workgroup per VAD turn.  Phase 1 turns the turn's frames into prefix
// statistics snapshots every SNAP_G frames (each frame is read once, coalesced,
// staged through LDS, fp64 accumulation in frame order).  Phase 2 evaluates
// candidate split points: the statistics of any frame range [x, y) are
// R(y) - R(x) with R(t) = snapshot[t / G] + the <= G-1 edge frames, so a
// candidate costs one record read and one or two 39x39 factorisations instead of
// two np.cov passes over raw frames.  The growing-window decision chain (a float
// state machine with int() truncation) runs on the device with
// the exact operation order of the reference.
#pragma once
#include "spkd_device.hpp"
#include "spkd_cluster.hpp"
#include "../../include/spkd.h"

namespace spkd {

constexpr int SNAP_G = 8;
constexpr int CD_TILE = 64;
constexpr double NEG_MAXINT_M1 = -9223372036854775808.0;   // -sys.maxint - 1

struct TurnDesc {
    int64_t begin;       // first frame of the turn in the frame array
    int64_t len;         // frames
    int64_t snap_off;    // first snapshot record of this turn
    int64_t cand_off;    // first candidate slot
    int64_t cand_cap;    // candidate slots owned by this turn
    int64_t ev_off;      // first event slot
    int64_t ev_cap;      // event slots owned by this turn
};

// Phase 1: snapshots snap[j] = sum over turn frames [0, G*j), j = 0 .. len / G.
template <int TPB>
__device__ __forceinline__ void build_prefix(const float* __restrict__ fr, long long n,
                                             double* __restrict__ snap, float* xs) {
    constexpr int EPT = (REC + TPB - 1) / TPB;
    const int tid = threadIdx.x;
    int er[EPT], ec[EPT];
    double acc[EPT];
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
        int e = tid + TPB * m;
        if (e >= REC) e = REC - 1;
        decode_entry(e, er[m], ec[m]);
        acc[m] = 0.0;
        if (tid + TPB * m < REC) snap[tid + TPB * m] = 0.0;
    }
    for (long long t0 = 0; t0 < n; t0 += CD_TILE) {
        const int tl = (int)((n - t0) < CD_TILE ? (n - t0) : CD_TILE);
        const float* src = fr + t0 * D;
        for (int idx = tid; idx < tl * D; idx += TPB) {
            int f = idx / D, c = idx - f * D;
            xs[f * DA + c] = src[idx];
        }
        if (tid < tl) xs[tid * DA + D] = 1.0f;
        __syncthreads();
        for (int f = 0; f < tl; ++f) {
#pragma unroll
            for (int m = 0; m < EPT; ++m)
                acc[m] = fma((double)xs[f * DA + er[m]], (double)xs[f * DA + ec[m]], acc[m]);
            const long long t = t0 + f + 1;
            if ((t % SNAP_G) == 0) {
                double* dst = snap + (t / SNAP_G) * REC;
#pragma unroll
                for (int m = 0; m < EPT; ++m)
                    if (tid + TPB * m < REC) dst[tid + TPB * m] = acc[m];
            }
        }
        __syncthreads();
    }
}

// Row-layout prefix statistics R(t): snapshot + edge frames, in frame order.
__device__ __forceinline__ void prefix_rows(double (&q)[DA], double* slab,
                                            const double* __restrict__ snap,
                                            const float* __restrict__ fr, long long t) {
    const long long j = t / SNAP_G;
    stage1(slab, snap + j * REC);
    row_from_slab(slab, q);
    const int lane = lane_id();
    for (long long f = j * SNAP_G; f < t; ++f) {
        const float x = (lane < D) ? fr[f * D + lane] : 1.0f;
        const double xi = (double)x;
#pragma unroll
        for (int c = 0; c < DA; ++c) {
            const double xc = readlane_d(xi, c);
            q[c] = fma(xi, xc, q[c]);
        }
    }
}

// Expanded LDS image of R(t), built by the whole workgroup.
__device__ __forceinline__ void build_expanded(double* ex, const double* __restrict__ snap,
                                               const float* __restrict__ fr, long long t,
                                               int tid, int nthreads) {
    const long long j = t / SNAP_G;
    const double* s = snap + j * REC;
    for (int idx = tid; idx < DA * DA; idx += nthreads) {
        const int c = idx / DA, i = idx - c * DA;
        const int r0 = i < c ? i : c, c0 = i < c ? c : i;
        double v = s[pk(r0, c0)];
        for (long long f = j * SNAP_G; f < t; ++f) {
            const float xi = (i < D) ? fr[f * D + i] : 1.0f;
            const float xc = (c < D) ? fr[f * D + c] : 1.0f;
            v = fma((double)xi, (double)xc, v);
        }
        ex[idx] = v;
    }
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

constexpr int GW_WAVES = 8;
constexpr int GW_TPB = GW_WAVES * WAVE;
constexpr int GW_LDS_DOUBLES = 2 * DA * DA + GW_WAVES * REC;

// Candidate scratch per slot k: cand[0*cap + k] = i value, [1*cap + k] = memoised
// left term (BIC: 0.5 N1 log det S1; GLR: log det S1), [2*cap + k] = right log det
// or finished distance.
//
// The kernel is one loop over "scans": a coarse scan over the candidates
// i = minfeas, minfeas + istep, ... (CD:204-221) and, after a positive one, a
// fine scan over single-frame steps around the maximum (CD:235-251).  Both kinds
// share one body, so the two 39x39 elimination routines exist once in the code.
__global__ __launch_bounds__(GW_TPB) void k_gw(
        const float* __restrict__ frames, const TurnDesc* __restrict__ turns, spkd_cd_params P,
        double* __restrict__ snap_all, double* __restrict__ cand_all,
        int32_t* __restrict__ n_win, double* __restrict__ win_maxd, int32_t* __restrict__ win_det,
        double* __restrict__ det_start, double* __restrict__ det_maxi, double* __restrict__ det_d,
        double* __restrict__ final_start, spkd_cand_log* clog, long long log_cap,
        unsigned long long* log_count, int* err) {
    extern __shared__ double lds[];
    double* exRa = lds;
    double* exRc = lds + DA * DA;
    double* slabs = lds + 2 * DA * DA;
    __shared__ BestD red[GW_WAVES];
    __shared__ double s_ldS;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int li = lane > D ? D : lane;
    const int turn = blockIdx.x;
    const TurnDesc T = turns[turn];
    const long long n = T.len;
    const float* fr = frames + T.begin * D;
    double* snap = snap_all + T.snap_off * REC;
    const long long cap = T.cand_cap;
    double* c_i = cand_all + 3 * T.cand_off;
    double* c_left = c_i + cap;
    double* c_x = c_left + cap;
    double* slab = slabs + wave * REC;

    build_prefix<GW_TPB>(fr, n, snap, (float*)slabs);
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
            base = C;                // fine-scan scratch sits behind the coarse slots
            count = F;
        }
        const long long a = (long long)start, c = (long long)end;
        if (a != a_cur) { build_expanded(exRa, snap, fr, a, tid, GW_TPB); a_cur = a; }
        if (c != c_cur) { build_expanded(exRc, snap, fr, c, tid, GW_TPB); c_cur = c; }
        __syncthreads();
        const double N = (double)(c - a);
        // ---- jobs: -1 = pooled window (BIC, coarse scans only), 0..count-1 = split points
        for (long long job = (long long)wave - 1; job < count; job += GW_WAVES) {
            double a_[DA];
            if (job < 0) {
                if (kind == SPKD_BIC && !fine) {
                    auto form = [&](double (&q)[DA]) {
                        row_from_expanded(exRc, q);
#pragma unroll
                        for (int j = 0; j < DA; ++j) q[j] -= exRa[j * DA + li];
                        cov_rows(q, N);
                    };
                    const double v = logdet_formed(a_, err, form);
                    if (lane == 0) s_ldS = v;
                }
                continue;
            }
            double ik;
            if (!fine) {
                ik = c_i[job];
            } else {
                ik = fine_i0;
                for (long long s2 = 0; s2 < job; ++s2) ik += 1;
            }
            const long long b = (long long)(start + ik);
            const double n1 = (double)(b - a), n2 = (double)(c - b);
            double qb[DA];
            prefix_rows(qb, slab, snap, fr, b);
            const bool need_left = fine || job >= n_memo;
            double ld_right = 0.0, ld_left = 0.0, ld_w = 0.0, dist = 0.0;
            if (kind == SPKD_KL2) {
                double ds[2], dp[2], mu[2];
#pragma unroll 1
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int j = 0; j < DA; ++j)
                        a_[j] = t ? (exRc[j * DA + li] - qb[j]) : (qb[j] - exRa[j * DA + li]);
                    const double nn = t ? n2 : n1;
                    const double mean_i = a_[D] / nn;
                    cov_rows(a_, nn);
                    kl2_lane_terms(a_, mean_i, ds[t], dp[t], mu[t]);
                }
                dist = kl2_combine(ds[0], dp[0], mu[0], ds[1], dp[1], mu[1]);
            } else {
#pragma unroll 1
                for (int t = 0; t < 3; ++t) {
                    if (t == 1 && !need_left) continue;
                    if (t == 2 && kind != SPKD_GLR) continue;
                    auto form = [&](double (&q)[DA]) {
                        if (t == 0) {
#pragma unroll
                            for (int j = 0; j < DA; ++j) q[j] = exRc[j * DA + li] - qb[j];
                            cov_rows(q, n2);
                        } else if (t == 1) {
#pragma unroll
                            for (int j = 0; j < DA; ++j) q[j] = qb[j] - exRa[j * DA + li];
                            cov_rows(q, n1);
                        } else {
                            const double al1 = (n1 / N) / (n1 - 1.0), al2 = (n2 / N) / (n2 - 1.0);
                            const double be1 = al1 / n1, be2 = al2 / n2;
                            const double s1i = qb[D] - exRa[D * DA + li];
                            const double s2i = exRc[D * DA + li] - qb[D];
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                const double q1 = qb[j] - exRa[j * DA + li];
                                const double q2 = exRc[j * DA + li] - qb[j];
                                const double s1j = readlane_d(s1i, j), s2j = readlane_d(s2i, j);
                                double v = fma(al2, q2, al1 * q1);
                                v = fma(-(be1 * s1i), s1j, v);
                                q[j] = fma(-(be2 * s2i), s2j, v);
                            }
                        }
                    };
                    const double v = logdet_formed(a_, err, form);
                    if (t == 0) ld_right = v; else if (t == 1) ld_left = v; else ld_w = v;
                }
            }
            if (lane == 0) {
                const long long slot = base + job;
                if (fine) c_i[slot] = ik;
                if (kind == SPKD_BIC) {
                    // left term: 0.5 N1 log det S1, memoised per i inside an epoch (CD:84-90)
                    if (need_left) c_left[slot] = 0.5 * n1 * ld_left;   // fine slots: scratch only
                    c_x[slot] = ld_right;
                } else if (kind == SPKD_GLR) {
                    const double l1 = need_left ? ld_left : c_left[slot];
                    if (need_left && !fine) c_left[slot] = l1;
                    c_x[slot] = -(N / 2.0) * ((n1 / N) * l1 + (n2 / N) * ld_right - ld_w);
                } else {
                    c_x[slot] = dist;
                }
            }
        }
        __syncthreads();
        // ---- finish the distances (BIC): d = 0.5 N log|S| - c1 - 0.5 N2 log|S2| - penalty
        if (kind == SPKD_BIC) {
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
constexpr int SW_WAVES = 4;
constexpr int SW_TPB = SW_WAVES * WAVE;

__global__ __launch_bounds__(SW_TPB) void k_sw(
        const float* __restrict__ frames, const TurnDesc* __restrict__ turns, spkd_cd_params P,
        double* __restrict__ snap_all, double* __restrict__ d_out, int* err) {
    __shared__ double slabs[SW_WAVES][REC];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int turn = blockIdx.x;
    const TurnDesc T = turns[turn];
    const long long n = T.len;
    const float* fr = frames + T.begin * D;
    double* snap = snap_all + T.snap_off * REC;
    build_prefix<SW_TPB>(fr, n, snap, (float*)&slabs[0][0]);
    __syncthreads();
    const int kind = P.kind;
    const double winsize = P.winsize, winstep = P.winstep;
    long long W = 0;
    for (double s = 0; s + 2 * winsize <= (double)n; s += winstep) ++W;
    const long long wsz = (long long)winsize;
    double* slab = slabs[wave];
    const int li = lane > D ? D : lane;
    for (long long w = wave; w < W; w += SW_WAVES) {
        const long long a = (long long)((double)w * winstep);
        const long long m = a + wsz, e = a + 2 * wsz;
        double qa[DA], qm[DA], qe[DA], a_[DA];
        prefix_rows(qa, slab, snap, fr, a);
        prefix_rows(qm, slab, snap, fr, m);
        prefix_rows(qe, slab, snap, fr, e);
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
        (void)li;
    }
}

}  // namespace spkd
