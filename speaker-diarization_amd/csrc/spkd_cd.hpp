// Change-detection kernels: growing window (dist_gw, spk-change-detection.py:180-288).
//
// Growing window: one workgroup per VAD turn.  The statistics of every frame range
// the scan needs are differences of running moment sums P(t) = sum of [x;1][x;1]^T
// over the frames [a, t) of the current epoch (a = the window start; a detection
// starts a new epoch).  The workgroup sweeps each frame once per epoch, in frame
// order, accumulating the lower triangle in registers (2x2 blocks, fp64), and
// leaves a copy P(b) in a per-turn cache for every candidate split point b as the
// sweep passes it, plus P(c) for the window end in LDS.  A candidate then costs one
// 6.5 KB record read and one or two 39x39 symmetric eliminations -- no raw frames,
// no edge corrections -- and is re-read, not re-formed, every time the window
// grows.  Four candidates share a wave (quad layout, spkd_quad.hpp / spkd_tri.hpp).
// The growing-window decision chain (a float state machine with int() truncation,
// SURVEY.md A-1) runs on the device with the exact operation order of the reference.
//
// Sliding window (dist_sw, spk-change-detection.py:291-357): no kernel of its own -- every
// window is a pair distance between two frame sets, computed by the statistics and
// clustering kernels (spkd_sw in spkd_hip.hip).
#pragma once
#include <type_traits>
#include "spkd_device.hpp"
#include "spkd_quad.hpp"
#include "spkd_cluster.hpp"
#include "../../include/spkd.h"

namespace spkd {

constexpr double NEG_MAXINT_M1 = -9223372036854775808.0;   // -sys.maxint - 1

struct TurnDesc {
    int64_t begin;       // first frame of the turn in the frame array
    int64_t len;         // frames
    int64_t cand_off;    // first candidate slot
    int64_t cand_cap;    // candidate slots owned by this turn
    int64_t ev_off;      // first event slot
    int64_t ev_cap;      // event slots owned by this turn
    int64_t id;          // caller's turn index (blocks are launched longest turn first)
};

struct BestD {
    double d;
    long long k;
};

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

// ---------------------------------------------------------------------------
// Records of running moment sums.
//
// In LDS (P(c), the sums at the window end) the "tri" layout: line (tri_off(s) + j)
// holds M(13 s + t, j) for t = 0..12 in its first 13 doubles, j < 13 (s + 1); lines
// 78..80 hold the sums column of slot s; the frame count sits in lane 15 of line 78.
// 81 lines of 128 B, conflict-free for the formation's reads.
//
// In global memory (the per-candidate cache, and the per-segment output of the fused
// mode) the ABI's packed record itself, SPKD_REC = 820 doubles.  Read by symmetry it
// is the lower triangle column by column: column j holds rows j .. 39 contiguously
// from pk_off(j), so the quad load of (slot s, column j) is 13 consecutive doubles at
// pk_off(j) + 13 s - j + t.  For the lanes of a diagonal block that sit above the
// diagonal (13 s + t < j) that address lies in the tail of the previous column --
// inside the record, and the value lands in a register nobody reads (spkd_tri.hpp).
// 6 560 B per candidate instead of the 10 368 B of a padded tri record.
// ---------------------------------------------------------------------------
// row-per-lane (single matrix) rows of a tri record / a packed record, by symmetry
__device__ __forceinline__ void single_rows_from_tri(const double* rec, double (&q)[DA]) {
    int i = lane_id();
    i = i >= D ? D - 1 : i;
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] = rec[j <= i ? tri_slot(i, j) : tri_slot(j, i)];
    q[D] = rec[tri_slot(D, i)];
}

// ---------------------------------------------------------------------------
// The sweep, for a workgroup of NW waves (4: a workgroup per turn when there are few
// turns; 1: a WAVE per turn when there are thousands -- no wave ever waits at a barrier
// for a serial phase of its own turn, the SIMD's other wave belongs to another turn).
// Thread b of the first NBLK owns one BR x BC block of the lower triangle of the 40x40
// augmented matrix (row-major over the block grid: row block bi has Q (bi + 1) column
// blocks, Q = BR / BC).  Per frame a thread reads x[BR bi ..] and x[BC bj ..] from the float
// tile (one vector read each) and issues BR * BC fp64 FMAs.  Every entry is its own chain of
// FMAs in frame order, so the sums do not depend on the block shape: the variants give
// bit-identical records.  In a packed record a block's column is BR consecutive doubles.
//   NW = 4: 2 x 2 blocks, 210 threads    NW = 2: 4 x 2, 110    NW = 1: 4 x 4, 55
// ---------------------------------------------------------------------------
constexpr int ERR_SWEEP = 16;                                // internal: a wanted position behind the sweep

template <int NW> struct GwShape;
// (8: the sweep of 4 -- 210 threads accumulate, all 512 stage the tile -- with twice the waves
// for the determinants: a file with fewer turns than the chip has CUs, two waves per SIMD)
template <> struct GwShape<8> { static constexpr int BR = 2, BC = 2, TILE = 128; };
template <> struct GwShape<4> { static constexpr int BR = 2, BC = 2, TILE = 128; };
template <> struct GwShape<2> { static constexpr int BR = 4, BC = 2, TILE = 64; };
#ifndef SPKD_GW1_TILE
#define SPKD_GW1_TILE 48        // frames per LDS tile with a wave per turn (32: 15.5 KB of LDS a wave, 48: 17.6)
#endif
template <> struct GwShape<1> { static constexpr int BR = 4, BC = 4, TILE = SPKD_GW1_TILE; };

template <int NW>
struct Gw {
    static constexpr int WAVES = NW;
    static constexpr int TPB = NW * WAVE;
    static constexpr int BR = GwShape<NW>::BR, BC = GwShape<NW>::BC;
    static constexpr int Q = BR / BC, RB = DA / BR;
    static constexpr int NBLK = Q * RB * (RB + 1) / 2;           // blocks = active threads of the sweep
    static constexpr int NACC = BR * BC;
    static constexpr int TILE = GwShape<NW>::TILE;               // frames per LDS tile (float [TILE][40])
    static constexpr int STAGE = (TILE * D + TPB - 1) / TPB;     // floats per thread per tile
    // dynamic LDS: P(c) tri record | float frame tile | the pooled window's determinant
    static constexpr int LDS_POOLED_AT = TREC + TILE * DA / 2;   // (in doubles)
    static constexpr int LDS_BYTES = TREC * 8 + TILE * DA * 4 + 16;
    static_assert(BR % BC == 0 && DA % BR == 0 && NBLK <= TPB && TILE <= TPB, "sweep shape");
};

template <int N> struct FloatVec;
template <> struct FloatVec<2> { using type = float2; };
template <> struct FloatVec<4> { using type = float4; };
template <int N>
__device__ __forceinline__ void load_fv(const float* p, float (&o)[N]) {
    const typename FloatVec<N>::type v = *reinterpret_cast<const typename FloatVec<N>::type*>(p);
    o[0] = v.x; o[1] = v.y;
    if constexpr (N == 4) { o[2] = v.z; o[3] = v.w; }
}

struct SweepLane {
    int bi, bj;
    bool on;
    bool diag;           // the block touches the diagonal: entries with row < column are not part of the triangle
};

template <int NW>
__device__ __forceinline__ SweepLane sweep_lane(int tid) {
    using G = Gw<NW>;
    SweepLane s;
    s.on = tid < G::NBLK;
    int rem = s.on ? tid : 0, bi = 0;
    while (rem >= G::Q * (bi + 1)) { rem -= G::Q * (bi + 1); ++bi; }
    s.bi = bi;
    s.bj = rem;
    s.diag = G::BR * bi < G::BC * rem + G::BC - 1;
    return s;
}

// acc[u * BC + v] = entry (BR bi + u, BC bj + v)
template <int NW>
__device__ __forceinline__ void sweep_dump_lds(double* rec, const SweepLane& SL, const double (&acc)[Gw<NW>::NACC]) {
    using G = Gw<NW>;
    if (!SL.on) return;
    // (opaque copies: the addresses are formed HERE, per call -- hoisted out of the sweep loop
    // they are sixteen more live values, spilled, and every reload waits on vmcnt for the record
    // dumps and the tile prefetch in flight)
    int bi = SL.bi, bj = SL.bj;
    asm volatile("" : "+v"(bi), "+v"(bj));
#pragma unroll
    for (int u = 0; u < G::BR; ++u)
#pragma unroll
        for (int v = 0; v < G::BC; ++v) {
            const int r = G::BR * bi + u, j = G::BC * bj + v;
            if (r >= j) rec[tri_slot(r, j)] = acc[u * G::BC + v];
        }
}

template <int NW>
__device__ __forceinline__ void sweep_gather_lds(const double* rec, const SweepLane& SL, double (&acc)[Gw<NW>::NACC]) {
    using G = Gw<NW>;
    int bi = SL.bi, bj = SL.bj;
    asm volatile("" : "+v"(bi), "+v"(bj));
    // (branch-free: entries outside the triangle read a valid slot and are zeroed)
#pragma unroll
    for (int u = 0; u < G::BR; ++u)
#pragma unroll
        for (int v = 0; v < G::BC; ++v) {
            const int r = G::BR * bi + u, j = G::BC * bj + v;
            const bool in = SL.on && r >= j;
            const double x = rec[tri_slot(r >= j ? r : j, r >= j ? j : r)];     // (by symmetry always a valid slot)
            acc[u * G::BC + v] = in ? x : 0.0;
        }
}

template <int NW>
__device__ __forceinline__ void sweep_dump_packed(SPKD_GLOBAL double* rec, const SweepLane& SL,
                                                  const double (&acc)[Gw<NW>::NACC]) {
    using G = Gw<NW>;
    if (!SL.on) return;
    int bi = SL.bi, bj = SL.bj;
    asm volatile("" : "+v"(bi), "+v"(bj));              // (see sweep_dump_lds)
    const int r0 = G::BR * bi, j0 = G::BC * bj;
    // element (r0, j0 + v) sits at e0 + v (39 - j0) - v (v - 1) / 2: one multiply for the block,
    // the columns by compile-time steps (pk_low per column is a multiply and a 64-bit add each)
    const int c39 = (DA - 1) - j0;
    const int e0 = pk_low(r0, j0);
    if (!SL.diag) {
#pragma unroll
        for (int v = 0; v < G::BC; ++v) {
            SPKD_GLOBAL double* col = rec + (e0 + v * c39 - (v * (v - 1)) / 2);
#pragma unroll
            for (int u = 0; u < G::BR; ++u) col[u] = acc[u * G::BC + v];
        }
    } else if constexpr (G::BR == G::BC) {
        // a square block on the diagonal (r0 == j0): column v holds rows v .. BR - 1, contiguous
        // from its diagonal element -- straight-line stores, no per-entry predicate
#pragma unroll
        for (int v = 0; v < G::BC; ++v) {
            SPKD_GLOBAL double* col = rec + (e0 + v * c39 - (v * (v - 1)) / 2);
#pragma unroll
            for (int u = v; u < G::BR; ++u) col[u] = acc[u * G::BC + v];
        }
    } else {
#pragma unroll
        for (int v = 0; v < G::BC; ++v)
#pragma unroll
            for (int u = 0; u < G::BR; ++u)
                if (r0 + u >= j0 + v) rec[pk_low(r0 + u, j0 + v)] = acc[u * G::BC + v];
    }
}

template <int NW>
__device__ __forceinline__ void sweep_gather_packed(const SPKD_GLOBAL double* rec, const SweepLane& SL,
                                                    double (&acc)[Gw<NW>::NACC]) {
    using G = Gw<NW>;
    int bi = SL.bi, bj = SL.bj;
    asm volatile("" : "+v"(bi), "+v"(bj));
    const int r0 = G::BR * bi, j0 = G::BC * bj;
#pragma unroll
    for (int v = 0; v < G::BC; ++v)
#pragma unroll
        for (int u = 0; u < G::BR; ++u) {
            const int r = r0 + u, j = j0 + v;
            const double x = rec[r >= j ? pk_low(r, j) : pk_low(j, r)];         // (by symmetry always inside the record)
            acc[u * G::BC + v] = r >= j ? x : 0.0;
        }
}

// Adds the turn frames [pos, ...) to acc in frame order.  `want` is the next
// position at which the caller wants the sums: on_reach(p) is called (uniformly,
// by all threads) when the sweep stands at p == want, BEFORE frame p is added, and
// returns the next wanted position (> p), or -1 to stop.  `limit` bounds the
// frames that may be staged (the last position that will ever be wanted).
// Frames go through an LDS tile of TILE frames; the loads of the next tile are
// issued into registers before the current one is consumed, so their latency is
// covered by the accumulation.  Contains block-wide barriers.
#ifdef SPKD_PROFILE
// phase clocks of k_gw (profiling builds only: make libspkd_hip_prof.so)
__device__ unsigned long long g_gw_prof[12];
#endif

// (branch-free: out-of-range slots re-read element 0; `t` is the caller's opaque copy
// of the thread index, so that none of this address arithmetic is hoisted out of the
// scan loop and kept alive -- i.e. spilled -- across the elimination code)
template <int NW>
__device__ __forceinline__ void sweep_issue(const SPKD_GLOBAL float* fr, long long pos, long long limit,
                                            int t, float (&st)[Gw<NW>::STAGE]) {
    using G = Gw<NW>;
    const long long left = limit - pos;
    const int tl = (int)(left < G::TILE ? (left < 0 ? 0 : left) : G::TILE);
    const SPKD_GLOBAL float* src = fr + pos * D;
    const int nfl = tl * D;
    if (nfl > 0) {                       // (uniform) nothing left: element 0 would lie behind `limit`
#pragma unroll
        for (int u = 0; u < G::STAGE; ++u) {
            const int idx = t + u * G::TPB;
            st[u] = src[idx < nfl ? idx : 0];
        }
    } else {
#pragma unroll
        for (int u = 0; u < G::STAGE; ++u) st[u] = 0.0f;
    }
}

template <int NW, class OnReach>
__device__ __forceinline__ void gw_sweep(const SPKD_GLOBAL float* fr, float* xs, const SweepLane& SL,
                                         int tid, double (&acc)[Gw<NW>::NACC], long long pos, long long want,
                                         long long limit, OnReach on_reach, int* err) {
    using G = Gw<NW>;
#ifdef SPKD_PROFILE
    unsigned long long sw_acc[3] = {0ull, 0ull, 0ull};
    unsigned long long sw_t = clock64(), sw_reached = 0ull;
#define SW_TICK(i) do { const unsigned long long now_ = clock64(); sw_acc[i] += now_ - sw_t; sw_t = now_; } while (0)
#else
#define SW_TICK(i) ((void)0)
#endif
    long long tile0 = pos, tile_end = pos;       // staged frames [tile0, tile_end)
    int t = tid, bi2 = G::BR * SL.bi, bj2 = G::BC * SL.bj;
    asm volatile("" : "+v"(t), "+v"(bi2), "+v"(bj2));
    const float* xi = xs + bi2;
    const float* xj = xs + bj2;
    float st[G::STAGE];
    sweep_issue<NW>(fr, pos, limit, t, st);
    for (;;) {
        if (pos == want) {
            SW_TICK(0);
            want = on_reach(pos);
            SW_TICK(1);
#ifdef SPKD_PROFILE
            ++sw_reached;
#endif
            if (want < 0) break;
            continue;
        }
        if (want < pos || want > limit) {        // cannot happen; never loop on it
            if (tid == 0) atomicOr(err, ERR_SWEEP);
            break;
        }
        if (pos >= tile_end) {
            const int tl = (int)((limit - pos) < G::TILE ? (limit - pos) : G::TILE);
            SW_TICK(0);
            __syncthreads();                     // the previous tile is no longer read
            // (slots beyond tl * D hold element 0: they land in tile rows >= tl, never read;
            // the LDS addresses are formed here, from an opaque copy of the thread index: kept
            // across the loop they are STAGE more live values, spilled, see sweep_dump_lds)
            int ts = t;
            asm volatile("" : "+v"(ts));
#pragma unroll
            for (int u = 0; u < G::STAGE; ++u) {
                const int idx = ts + u * G::TPB;
                const int f = idx / D, c = idx - f * D;
                if (u < G::STAGE - 1 || idx < G::TILE * D) xs[f * DA + c] = st[u];
            }
            if (t < G::TILE) xs[t * DA + D] = 1.0f;
            __syncthreads();
            tile0 = pos;
            tile_end = pos + tl;
            sweep_issue<NW>(fr, tile_end, limit, t, st);     // in flight during the accumulation
            SW_TICK(2);
        }
        const long long stop = want < tile_end ? want : tile_end;
        const int f0 = (int)(pos - tile0), f1 = (int)(stop - tile0);
        auto frame = [&](const float (&a)[G::BR], const float (&b)[G::BC]) {
            double bd[G::BC];
#pragma unroll
            for (int v = 0; v < G::BC; ++v) bd[v] = (double)b[v];
#pragma unroll
            for (int u = 0; u < G::BR; ++u) {
                const double au = (double)a[u];
#pragma unroll
                for (int v = 0; v < G::BC; ++v) acc[u * G::BC + v] = fma(au, bd[v], acc[u * G::BC + v]);
            }
        };
        // the stretch between two coarse candidates is 12 or 13 frames: those run fully
        // unrolled, all LDS reads in flight before the first FMA (one exposed LDS latency
        // per stretch instead of one per group of four frames)
        auto stretch = [&](auto nf) {
            constexpr int NF = decltype(nf)::value;
            float a[NF][G::BR], b[NF][G::BC];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                load_fv<G::BR>(xi + (f0 + f) * DA, a[f]);
                load_fv<G::BC>(xj + (f0 + f) * DA, b[f]);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) frame(a[f], b[f]);
        };
        if (f1 - f0 == 13) {
            stretch(std::integral_constant<int, 13>());
        } else if (f1 - f0 == 12) {
            stretch(std::integral_constant<int, 12>());
        } else {
#pragma unroll 4
            for (int f = f0; f < f1; ++f) {
                float a[G::BR], b[G::BC];
                load_fv<G::BR>(xi + f * DA, a);
                load_fv<G::BC>(xj + f * DA, b);
                frame(a, b);
            }
        }
        pos = stop;
    }
#ifdef SPKD_PROFILE
    SW_TICK(0);
    if (tid == 0) for (int i = 0; i < 3; ++i) atomicAdd(&g_gw_prof[8 + i], sw_acc[i]);
    if (tid == 0) atomicAdd(&g_gw_prof[11], sw_reached);            // positions reached = records left
#endif
}

// The uses of the sweep, as real functions: compiled apart from the kernel body
// they keep their few live values in registers.  Inlined into k_gw they share its
// register allocation with the elimination code, the allocator spills the sweep's
// loop invariants, and every reload (a scratch load, counted with the global stores
// in vmcnt) waits for the dump stores in flight -- measured: 6 000 cycles per dump.
struct SweepOut {
    long long pos;       // where the running sums stand now
    double next_i;       // i value of the next candidate to build
    long long built_k;   // cache records [0, built_k) exist
};

// coarse scan: the sweep runs from where it stands to the window end c, leaving P(b_k) in
// cache record k for every candidate position it passes (candidate k sits at
// (long long)(start + i_k), i_k by repeated addition of istep) and P(c) in LDS, and rests at
// c.  The candidates of THIS scan end minfeas frames before c; the handful behind them are
// the first ones of the next scan and are left now, on the way -- the sweep never walks a
// frame twice inside an epoch (resting at the scan's last candidate, as it used to, it
// re-walked the last minfeas frames of the window in every scan).  What a detection makes
// of the records built ahead is nothing: the next epoch starts from slot 0.
// The running sums at the rest position ARE P(c) of the scan before, which sits in LDS:
// the sweep picks them up there (fresh: a new epoch, the sums start from zero).
template <int NW>
__device__ __noinline__ SweepOut gw_sweep_coarse(const float* __restrict__ fr, double* __restrict__ cache,
                                                 double start, double istep, double built_i,
                                                 long long built_k, long long cap, long long sweep_pos,
                                                 long long c, int fresh, int* err) {
    using G = Gw<NW>;
    extern __shared__ double gw_lds[];
    double* ldsEnd = gw_lds;
    float* xs = (float*)(gw_lds + TREC);
    const int tid = threadIdx.x;
    const SweepLane SL = sweep_lane<NW>(tid);
    const SPKD_GLOBAL float* gfr = (const SPKD_GLOBAL float*)fr;
    SPKD_GLOBAL double* gcache = (SPKD_GLOBAL double*)cache;
    double acc[G::NACC];
    if (fresh) {
#pragma unroll
        for (int e = 0; e < G::NACC; ++e) acc[e] = 0.0;
    } else {
        sweep_gather_lds<NW>(ldsEnd, SL, acc);
    }
    __syncthreads();                        // everybody holds its sums before the record is rewritten
    // position of candidate built_k, c when it lies at or behind the window end (or no slot is left)
    auto next_stop = [&]() -> long long {
        const long long b = built_k < cap ? (long long)(start + built_i) : c;
        return b < c ? b : c;
    };
    gw_sweep<NW>(gfr, xs, SL, tid, acc, sweep_pos, next_stop(), c, [&](long long pos) -> long long {
        if (pos < c) {                      // a candidate position
            sweep_dump_packed<NW>(gcache + built_k * REC, SL, acc);
            ++built_k;
            built_i += istep;
            return next_stop();
        }
        sweep_dump_lds<NW>(ldsEnd, SL, acc);    // pos == c: the window end; the sweep rests here
        return -1;
    }, err);
    SweepOut o;
    o.pos = c;
    o.next_i = built_i;
    o.built_k = built_k;
    return o;
}

// fine scan: P at the F single-frame positions (long long)(start + fine_i0 + k)
// -> cache records first_slot + k, starting from the sums `base` (a cache record, or
// nullptr for zero) that stand at base_pos.
template <int NW>
__device__ __noinline__ void gw_sweep_fine(const float* __restrict__ fr, double* __restrict__ cache,
                                           const double* __restrict__ base, long long base_pos,
                                           double start, double fine_i0, long long F,
                                           long long first_slot, int* err) {
    using G = Gw<NW>;
    extern __shared__ double gw_lds[];
    float* xs = (float*)(gw_lds + TREC);
    const int tid = threadIdx.x;
    const SweepLane SL = sweep_lane<NW>(tid);
    const SPKD_GLOBAL float* gfr = (const SPKD_GLOBAL float*)fr;
    SPKD_GLOBAL double* gcache = (SPKD_GLOBAL double*)cache;
    double acc[G::NACC];
#pragma unroll
    for (int e = 0; e < G::NACC; ++e) acc[e] = 0.0;
    if (base) sweep_gather_packed<NW>((const SPKD_GLOBAL double*)base, SL, acc);
    double fx = fine_i0;
    long long fk = 0;
    const long long last = (long long)(start + (fine_i0 + (double)(F - 1)));
    gw_sweep<NW>(gfr, xs, SL, tid, acc, base_pos, (long long)(start + fx), last, [&](long long) -> long long {
        sweep_dump_packed<NW>(gcache + (first_slot + fk) * REC, SL, acc);
        ++fk;
        fx += 1;
        return fk < F ? (long long)(start + fx) : -1;
    }, err);
}

// fused mode, tail segment: the moments of the turn frames [from, to) -> one packed record
template <int NW>
__device__ __noinline__ void gw_sweep_tail(const float* __restrict__ fr, long long from, long long to,
                                           double* __restrict__ out, int* err) {
    using G = Gw<NW>;
    extern __shared__ double gw_lds[];
    float* xs = (float*)(gw_lds + TREC);
    const int tid = threadIdx.x;
    const SweepLane SL = sweep_lane<NW>(tid);
    const SPKD_GLOBAL float* gfr = (const SPKD_GLOBAL float*)fr;
    SPKD_GLOBAL double* gout = (SPKD_GLOBAL double*)out;
    double acc[G::NACC];
#pragma unroll
    for (int e = 0; e < G::NACC; ++e) acc[e] = 0.0;
    gw_sweep<NW>(gfr, xs, SL, tid, acc, from, to, to, [&](long long) -> long long {
        sweep_dump_packed<NW>(gout, SL, acc);
        return -1;
    }, err);
}

// single-matrix form of a pass (pivoting fallback): P(b) from the candidate's cache
// record, P(c) from LDS, P(a) = 0
__device__ __forceinline__ void single_split_matrix(int pass, const double* ldsEnd,
                                                    const double* __restrict__ rec_b,
                                                    double n1, double n2, double (&q)[DA]) {
    const double n = n1 + n2;
    double rc[DA];
    if (pass == PASS_POOLED) {
        single_rows_from_tri(ldsEnd, q);
        cov_rows(q, n);
        return;
    }
    single_rows_from_packed(rec_b, q);
    if (pass == PASS_LEFT) {
        cov_rows(q, n1);
        return;
    }
    single_rows_from_tri(ldsEnd, rc);
    if (pass == PASS_RIGHT) {
#pragma unroll
        for (int j = 0; j < DA; ++j) q[j] = rc[j] - q[j];
        cov_rows(q, n2);
    } else {
        const double al1 = (n1 / n) / (n1 - 1.0), al2 = (n2 / n) / (n2 - 1.0);
        const double be1 = al1 / n1, be2 = al2 / n2;
        const double s1i = q[D], s2i = rc[D] - q[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double q1 = q[j], q2 = rc[j] - q[j];
            const double s1j = readlane_d(s1i, j), s2j = readlane_d(s2i, j);
            double v = fma(al2, q2, al1 * q1);
            v = fma(-(be1 * s1i), s1j, v);
            q[j] = fma(-(be2 * s2i), s2j, v);
        }
    }
}

// determinant (its log is taken by the caller, densely) of four (pass, split point) items held by the wave: DPP row m forms the
// matrix of ITS pass for ITS split b (packed cache record rec_b) -- the passes of a
// scan are packed four to a wave whatever their kind, so `pass` is a per-lane value
// and everything below is straight-line code with per-lane coefficients:
//     q  = al * P(b) + ga * P(c)                (lower triangle + sums column)
//     q += c1 v1^T + c2 v2^T   (al, ga, c1 carry the covariance scale f)
//   right  [b, c): al = -1, ga = 1; v1 = s_c - s_b, c1 = -v1 / n2;  f = 1 / (n2 - 1)
//   left   [a, b): al =  1, ga = 0; v1 = s_b,       c1 = -v1 / n1;  f = 1 / (n1 - 1)
//   pooled [a, c): al =  0, ga = 1; v1 = s_c,       c1 = -v1 / N;   f = 1 / (N - 1)
//                  (rec_b: any valid record, it is multiplied by zero)
//   GLR: al1 P(b) + al2 (P(c) - P(b)) - be1 s1 s1^T - be2 s2 s2^T;  f = 1
// two: the launch has GLR items (wave-uniform; the second rank-one term is skipped
// otherwise).  DPP rows that name the same record (the left and the right item of a
// new candidate sit side by side) fetch it once: their loads coalesce.
// The split point b = (long long)(start + ik) is formed HERE, behind the record loads: ik is a
// global load of the caller's (the candidate's i value), and a pass that needed it before it
// could issue its record loads waited for two memory round trips, one after the other.
__device__ __forceinline__ double quad_split_det(int pass, bool two, const double* ldsEnd,
                                                    const double* __restrict__ rec_b,
                                                    double ik, double start, long long a, long long c,
                                                    const QuadLane& L, int* err) {
    QuadRows q;
    double svb[QS];
    PASS_T0();
    int ta = L.t;
    asm volatile("" : "+v"(ta));          // keep the LDS reads out of the callers' loops
    {
        // 81 loads in flight, one latency.  One base pointer per 4 KB (the immediate
        // offset of a global load spans 4 KB; left to itself the compiler builds a
        // separate address for every load, spills them and serialises the loads)
        const int t12 = ta < QL ? ta : QL - 1;          // idle lanes 13..15 ride with lane 12
        const double* rt[2];
        long long o1 = 512;                 // opaque, so that the bases stay separate registers
        asm volatile("" : "+v"(o1));
        rt[0] = rec_b + t12;
        rt[1] = rt[0] + o1;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
#pragma unroll
            for (int j = 0; j < tri_cols(s); ++j) {
                const int e = pk_off(j) + QL * s - j;   // + t12 (in the base)
                q.r[s][j] = rt[e / 512][e % 512];
            }
            const int c = QL * s + t12;                 // this lane's row of slot s
            svb[s] = rec_b[pk_off(c) + D - c];          // (39, c): the sums entry of column c
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const long long b = pass == PASS_POOLED ? a : (long long)(start + ik);
    const double n1 = (double)(b - a), n2 = (double)(c - b);
    const double n = n1 + n2;
    double al, ga, k1a, k1c, k2a, k2c, w1, w2;
    // v1 = k1a * s_b + k1c * s_c ;  c1 = w1 * v1  (same for v2, c2)
    {
        // (an IEEE fp64 division is ~25 instructions: the GLR weights are only formed in
        // launches that have GLR items, the covariance scale uses the Newton reciprocal)
        const bool glr = two && pass == PASS_GLR;
        double al1 = 0.0, al2 = 0.0, b1 = 0.0, b2 = 0.0;
        if (two) {                          // wave-uniform
            al1 = (n1 / n) / (n1 - 1.0);
            al2 = (n2 / n) / (n2 - 1.0);
            b1 = al1 / n1;
            b2 = al2 / n2;
        }
        const double np = pass == PASS_RIGHT ? n2 : (pass == PASS_LEFT ? n1 : n);
        const double f = fast_recip(np - 1.0);
        const double sa = pass == PASS_RIGHT ? -1.0 : (pass == PASS_LEFT ? 1.0 : 0.0);
        const double sg = pass == PASS_LEFT ? 0.0 : 1.0;
        al = glr ? al1 - al2 : sa * f;      // covariance scale folded into the combine
        ga = glr ? al2 : sg * f;
        k1a = glr ? 1.0 : sa;
        k1c = glr ? 0.0 : sg;
        w1 = glr ? -b1 : -(f * fast_recip(np));
        k2a = glr ? -1.0 : 0.0;
        k2c = glr ? 1.0 : 0.0;
        w2 = glr ? -b2 : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    PASS_LOADED();
    double v1[QS], v2[QS], c1[QS], c2[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
#pragma unroll
        for (int j = 0; j < tri_cols(s); ++j)
            q.r[s][j] = fma(ga, ldsEnd[(tri_off(s) + j) * 16 + ta], al * q.r[s][j]);
        const double sc = ldsEnd[(TREC_SUMS + s) * 16 + ta];
        v1[s] = fma(k1c, sc, k1a * svb[s]);
        v2[s] = fma(k2c, sc, k2a * svb[s]);
        c1[s] = w1 * v1[s];
        c2[s] = w2 * v2[s];
        __builtin_amdgcn_sched_barrier(0);
    }
    TriRank1<0>::run(q, c1, v1);
    if (two) TriRank1<0>::run(q, c2, v2);
    auto form_single = [&](int mi, double (&arr)[DA]) {
        const double* rm = (const double*)__shfl((unsigned long long)rec_b, 16 * mi);
        single_split_matrix(__shfl(pass, 16 * mi), ldsEnd, rm, __shfl(n1, 16 * mi), __shfl(n2, 16 * mi), arr);
    };
    const double det = tri_det(q, L.m, err, form_single);
    PASS_DONE();
    return det;
}


// Growing window over one VAD turn per workgroup (spk-change-detection.py:180-288):
// the whole float state machine runs here; outputs are the per-window events.
//
// Candidate scratch per slot k: c_i = i value, c_left = determinant of the memoised left
// covariance, c_x = determinant of the right one (KL2: the finished distance);
// cache record k = P(b_k), a packed record.
//
// The kernel is one loop over "scans": a coarse scan over the candidates
// i = minfeas, minfeas + istep, ... (CD:204-221) and, after a positive one, a
// fine scan over single-frame steps around the maximum (CD:235-251).  Both kinds
// share one body, so the elimination code exists once.
//
// Fused mode (seg_stats != nullptr): every segment the detector emits -- [start,
// start + maxi) of a detection, [final start, turn end) of the tail -- also leaves its
// packed statistics record, which is what the clustering stage needs (the frames are
// then read once for both stages): a detection's record is the cache record of the
// chosen split point, the tail's is P(turn end) of the last epoch.  Record j of turn t
// lands at seg_stats[(ev_off[t] + j) * 820], j = 0 .. n_det (the tail last).
#ifdef SPKD_PROFILE
#define GW_TICK(i) do { const unsigned long long now_ = clock64(); prof_acc[i] += now_ - prof_t; prof_t = now_; } while (0)
#else
#define GW_TICK(i) ((void)0)
#endif

// The decision chain's scalar state lives in LDS, written by thread 0 and re-read by
// everybody after a barrier.  As ordinary per-thread variables these twenty values are
// live across the elimination code, get spilled, and every reload (a scratch load)
// stalls on vmcnt behind whatever global stores are in flight: 27 k cycles per scan of
// pure waiting in the phase clocks.  LDS reads are counted in lgkmcnt and cost ~100.
struct GwState {
    double start, end, ws, dws, cur_i, built_i, maxd, maxi, fine_i0;
    long long n_memo;      // coarse candidates [0, n_memo) have their left term memoised
    long long n_written;   // c_i[0 .. n_written) hold the coarse i sequence
    long long C;           // coarse candidates of the current scan
    // the sweep of this epoch: cache records [0, built_k) exist, the running sums stand
    // at frame sweep_pos (fresh: nothing added yet; otherwise they are the P(c) in LDS);
    // built_i = i value of candidate built_k (the same repeated addition as cur_i, so the
    // same doubles)
    long long built_k, sweep_pos;
    int fresh;
    long long best_k, F, base, count;
    long long seg_src, seg_dst;    // fused mode: cache slot -> output record of the detection just made
    int nw, nd;
    int next_q;            // several waves per turn: the next quad of items of the running scan
    int fine;              // kind of the scan about to run
    int go;                // the outer loop continues (written in the decision step only)
    int abort;             // a capacity ran out while the scan was being set up
    int tail_in_lds;       // the loop ended on a negative scan whose window end is the turn end
};

template <int NW>
__global__ __launch_bounds__(Gw<NW>::TPB, NW == 8 ? 1 : 2) void k_gw(
        const float* __restrict__ frames, const TurnDesc* __restrict__ turns, spkd_cd_params P,
        double* __restrict__ cache_all, double* __restrict__ cand_all,
        int32_t* __restrict__ n_win, double* __restrict__ win_maxd, int32_t* __restrict__ win_det,
        double* __restrict__ det_start, double* __restrict__ det_maxi, double* __restrict__ det_d,
        double* __restrict__ final_start, double* __restrict__ seg_stats, spkd_cand_log* clog,
        long long log_cap, unsigned long long* log_count, int* err) {
    extern __shared__ double gw_lds[];
    double* ldsEnd = gw_lds;                         // P(c), tri record
    constexpr int GW_WAVES = NW, GW_TPB = Gw<NW>::TPB;
    __shared__ BestD red[GW_WAVES];
    __shared__ double s_ldS;
    __shared__ double s_common[2];
    __shared__ GwState S;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const QuadLane L = quad_lane();
    const TurnDesc T = turns[blockIdx.x];
    const int turn = (int)T.id;
    const long long n = T.len;
    const float* fr = frames + T.begin * D;
    double* cache = cache_all + T.cand_off * REC;
    const long long cap = T.cand_cap;
    double* c_i = cand_all + 4 * T.cand_off;
    double* c_left = c_i + cap;
    double* c_x = c_left + cap;
    double* c_w = c_x + cap;             // GLR: log det of the pooled-within matrix

#ifdef SPKD_PROFILE
    unsigned long long prof_acc[4] = {0ull, 0ull, 0ull, 0ull}, prof_t = clock64(), prof_scans = 0ull;
    unsigned long long prof_passes = 0ull, prof_items = 0ull;
#endif
    for (int e = tid; e < TREC; e += GW_TPB) ldsEnd[e] = 0.0;
    // record 0 is read (times zero) by the pooled item even before it is built
    for (int e = tid; e < REC; e += GW_TPB) cache[e] = 0.0;

    const int kind = P.kind;
    const double winsize = P.winsize, winstep = P.winstep, rate = P.rate;
    const double minfeas = rate / 2;
    const double istep = rate / 10;
    const double fn = (double)n;
    const double pen_w = P.lambdac * 0.5 * PEN_UNIT;
    if (tid == 0) {
        S.start = 0.0;
        S.end = winsize * 2;
        S.ws = minfeas;
        S.dws = P.deltaws;
        S.cur_i = minfeas;
        S.built_i = minfeas;
        S.maxd = 0.0; S.maxi = 0.0; S.fine_i0 = 0.0;
        S.n_memo = 0; S.n_written = 0; S.C = 0;
        S.built_k = 0; S.sweep_pos = 0; S.fresh = 1;
        S.best_k = 0; S.F = 0; S.base = 0; S.count = 0;
        S.seg_src = -1; S.seg_dst = -1;
        S.nw = 0; S.nd = 0; S.fine = 0;
        S.go = (S.end <= fn) ? 1 : 0;
        S.abort = 0;
        S.tail_in_lds = 0;
    }
    __syncthreads();
    GW_TICK(0);

    while (S.go) {
        // ---- (A) thread 0: the scan's candidate list (coarse: the i sequence grows with
        // the window, CD:204-206; fine: single-frame steps around the maximum, CD:235-239)
        if (tid == 0) {
            if (!S.fine) {
                if (S.nw >= T.ev_cap) {
                    atomicOr(err, 4);
                    S.abort = 1;
                } else {
                    const double lim = S.end - S.start - minfeas;
                    double cur_i = S.cur_i;
                    long long C = S.C, n_written = S.n_written;
                    while (cur_i < lim) {
                        if (C >= cap) break;
                        if (C >= n_written) { c_i[C] = cur_i; n_written = C + 1; }
                        ++C;
                        cur_i += istep;
                    }
                    if (cur_i < lim) {           // candidate slots exhausted (cannot happen with spkd_gw's sizing)
                        atomicOr(err, 4);
                        S.abort = 1;
                    }
                    S.cur_i = cur_i; S.C = C; S.n_written = n_written;
                    S.base = 0;
                    S.count = C;
                }
            } else {
                const double fine_i0 = S.maxi - istep;
                const double endtune = S.maxi + istep;
                long long F = 0;
                for (double x = fine_i0; x < endtune; x += 1) ++F;
                S.fine_i0 = fine_i0;
                S.F = F;
                if (S.C + F > cap) {
                    atomicOr(err, 4);
                    S.abort = 1;
                } else {
                    // the fine i sequence, by repeated +1 like the reference
                    double x = fine_i0;
                    for (long long k = 0; k < F; ++k) { c_i[S.C + k] = x; x += 1; }
                    S.base = S.C;            // fine-scan scratch sits behind the coarse slots
                    S.count = F;
                }
            }
        }
        if (tid == 0) S.next_q = 0;
        __syncthreads();
        if (S.abort) break;                  // (written before the barrier, never reset)
        // ---- (B) the sweep: P(b_k) for the new candidates, then P(c); fine: P at the F
        // single-frame positions, from a coarse candidate two steps below the maximum
        // (always at or below the first fine position)
        {
            const double start = S.start;
            const long long a = (long long)start, c = (long long)S.end;
            SweepOut so;
            so.pos = 0; so.next_i = 0.0; so.built_k = 0;
            const long long C = S.C, built_k = S.built_k;
            const bool coarse = !S.fine;
            if (coarse) {
                so = gw_sweep_coarse<NW>(fr, cache, start, istep, S.built_i, built_k, cap, S.sweep_pos, c, S.fresh, err);
            } else {
                const long long best_k = S.best_k;
                if (best_k >= 2)
                    gw_sweep_fine<NW>(fr, cache, cache + (best_k - 2) * REC, (long long)(start + c_i[best_k - 2]),
                                  start, S.fine_i0, S.F, C, err);
                else
                    gw_sweep_fine<NW>(fr, cache, nullptr, a, start, S.fine_i0, S.F, C, err);
            }
            __syncthreads();             // everybody has read its arguments; the records are visible
            if (tid == 0 && coarse) {
                S.sweep_pos = so.pos; S.built_i = so.next_i; S.built_k = so.built_k; S.fresh = 0;
                if (so.built_k < C) { atomicOr(err, ERR_SWEEP); }        // (cannot happen: every candidate of the scan lies before c)
            }
        }
        GW_TICK(1);
        // ---- (C) the scan's matrices
        const bool fine = S.fine != 0;
        const double start = S.start;
        const long long a = (long long)start, c = (long long)S.end;
        const long long base = S.base, count = S.count;
        {
        const long long n_memo = S.n_memo;
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
                    single_rows_from_packed(cache + (base + job) * REC, a_);
                    if (t) {
                        single_rows_from_tri(ldsEnd, r_);
#pragma unroll
                        for (int j = 0; j < DA; ++j) a_[j] = r_[j] - a_[j];
                    }
                    const double nn = t ? n2 : n1;
                    const double mean_i = a_[D] / nn;
                    cov_rows(a_, nn);
                    kl2_lane_terms(a_, mean_i, ds[t], dp[t], mu[t]);
                }
                const double dist = kl2_combine(ds[0], dp[0], mu[0], ds[1], dp[1], mu[1]);
                if (lane == 0) c_x[base + job] = dist;
            }
        } else {
            // The scan's matrices as one list of (pass, candidate) items, four to a wave
            // whatever their kind.  Items that read the same cache record sit side by side
            // in a wave, where their loads coalesce: a candidate whose left term is not
            // memoised yet (CD:84-90) contributes the pair (left, right), so its record
            // is fetched once for both.  BIC:
            //   right of the memoised candidates | [pad to an even index: the pooled window,
            //   or an idle item] | (left, right) of every new candidate | [the pooled window]
            // GLR: (right, W) of every candidate | left of the new ones.
            const long long left0 = fine ? 0 : (n_memo < count ? n_memo : count);
            const long long nnew = count - left0;
            const bool glr_kind = kind == SPKD_GLR;
            const long long padA = glr_kind ? 0 : (left0 & 1);
            const long long base2 = left0 + padA;
            const long long M = glr_kind ? 2 * count + nnew
                                         : base2 + 2 * nnew + ((pooled && !padA) ? 1 : 0);
            // item index -> (pass, candidate); false for the idle pad item
            auto decode = [&](long long it, int& pass, long long& k) -> bool {
                if (glr_kind) {
                    if (it < 2 * count) { k = it >> 1; pass = (it & 1) ? PASS_GLR : PASS_RIGHT; }
                    else { k = left0 + (it - 2 * count); pass = PASS_LEFT; }
                } else if (it < left0) {
                    pass = PASS_RIGHT; k = it;
                } else if (padA && it == left0) {
                    if (pooled) { pass = PASS_POOLED; k = 0; }
                    else { pass = PASS_RIGHT; k = it - 1; return false; }
                } else if (it < base2 + 2 * nnew) {
                    const long long j = it - base2;
                    k = left0 + (j >> 1);
                    pass = (j & 1) ? PASS_RIGHT : PASS_LEFT;
                } else {
                    pass = PASS_POOLED; k = 0;
                }
                return true;
            };
            if (tid == 0) atomicAdd(log_count + 1, (unsigned long long)M);      // work counter (spkd_last_gw_items)
#ifdef SPKD_PROFILE
            prof_items += (unsigned long long)M;
#endif
            // (with several waves per turn they take the quads of items from a counter, not by
            // stride: a wave whose records come late takes fewer -- as in k_ahc)
            for (long long q4 = wave;; q4 += GW_WAVES) {
                if constexpr (NW > 1) {
                    int qn = 0;
                    if (lane == 0) qn = atomicAdd(&S.next_q, 1);
                    q4 = __builtin_amdgcn_readfirstlane(qn);
                }
                if (4 * q4 >= M) break;
#ifdef SPKD_PROFILE
                ++prof_passes;
#endif
                long long it = 4 * q4 + L.m;
                bool valid = it < M;
                it = valid ? it : M - 1;
                int pass;
                long long k;
                if (!decode(it, pass, k)) valid = false;
                const long long slot = base + k;
                // (no candidates at all: the only item is the pooled window; record 0 is still a
                // finite record of this or an earlier epoch, or zero-initialised scratch)
                const double ik = count > 0 ? c_i[slot] : 0.0;
                // determinants; their logs are taken in phase (D), one thread per candidate
                const double v = quad_split_det(pass, glr_kind, ldsEnd, cache + slot * REC, ik, start, a, c, L, err);
                if (valid && L.t == 0) {
                    if (pass == PASS_POOLED) s_ldS = v;
                    else if (pass == PASS_RIGHT) c_x[slot] = v;
                    // left term, memoised per i inside an epoch (CD:84-90)
                    else if (pass == PASS_LEFT) c_left[slot] = v;
                    else c_w[slot] = v;
                }
            }
        }
        }
        __syncthreads();
        GW_TICK(2);
#ifdef SPKD_PROFILE
        ++prof_scans;
#endif
        // ---- (D) finish the distances and take the first-index arg-max in the same pass
        //   GLR: -(N/2) ((N1/N) log|S1| + (N2/N) log|S2| - log|W|)
        //   BIC: d = 0.5 N log|S| - c1 - 0.5 N2 log|S2| - penalty
        //        (fine scans: s_ldS still holds the pooled term of this window)
        BestD best;
        {
            // (no contraction into FMAs in this block: k_gw is compiled once per wave shape, and the
            // shapes must round a distance alike whatever code surrounds the formula -- a one-ulp
            // difference between k_gw<1> and k_gw<4> showed up when the code below was changed)
#pragma clang fp contract(off)
            const double N = (double)(c - a);
            // the window's two common terms are one thread's work (the last wave has no
            // candidates in a typical scan and runs beside the first one's logs); the others
            // pick them up behind a barrier, with their own logs already taken
            if (tid == GW_TPB - 1) {
                s_common[0] = kind == SPKD_BIC ? log(s_ldS) : 0.0;      // s_ldS, c_left, c_x, c_w hold determinants
                s_common[1] = pen_w * log(N);
            }
            const int nw = S.nw;
            best.d = fine ? S.maxd : NEG_MAXINT_M1;
            best.k = -1;
            // (a workgroup-uniform trip count: the barrier sits inside the loop)
            for (long long k0 = 0; k0 < count || k0 == 0; k0 += GW_TPB) {
                const long long k = k0 + tid;
                const bool on = k < count;
                const long long slot = base + (on ? k : 0);
                const double ik = on ? c_i[slot] : 0.0;
                const long long b = (long long)(start + ik);
                const double n1 = (double)(b - a), n2 = (double)(c - b);
                double d = on ? c_x[slot] : 1.0;
                double lgL = 0.0, lgR = 0.0, lgW = 0.0;
                if (on && kind != SPKD_KL2) {
                    lgL = log(c_left[slot]);
                    lgR = log(d);
                    if (kind == SPKD_GLR) lgW = log(c_w[slot]);
                }
                if (k0 == 0) __syncthreads();                          // s_common is there
                if (!on) continue;
                const double ldS = s_common[0], corr = s_common[1];
                if (kind == SPKD_GLR) {
                    d = -(N / 2.0) * ((n1 / N) * lgL + (n2 / N) * lgR - lgW);
                } else if (kind == SPKD_BIC) {
                    const double left = 0.5 * n1 * lgL;                   // BIC's memoised 0.5 N1 log det S1
                    d = 0.5 * N * ldS - left - 0.5 * n2 * lgR;
                    d -= corr;
                }
                if ((P.trace && !fine) || fabs(d) == __builtin_huge_val()) {
                    const long long w = fine ? (long long)(nw - 1) : (long long)nw;
                    log_cand(clog, log_cap, log_count, turn, fine ? 0 : 1,
                             (w << 32) | (fine ? 0x80000000LL : 0LL) | k, start, ik, d, b - a, c - b);
                }
                if (d > best.d && d != __builtin_huge_val()) { best.d = d; best.k = k; }   // NaN fails d > best.d
            }
#pragma unroll
            for (int s = 1; s < WAVE; s <<= 1) {
                const double d2 = __shfl_xor(best.d, s);
                const long long k2 = __shfl_xor(best.k, s);
                const bool take = (k2 >= 0) && (best.k < 0 || d2 > best.d || (d2 == best.d && k2 < best.k));
                if (take) { best.d = d2; best.k = k2; }
            }
            if (lane == 0) red[wave] = best;
            __syncthreads();
            best = red[0];
            for (int w = 1; w < GW_WAVES; ++w) {
                const BestD o = red[w];
                const bool take = (o.k >= 0) && (best.k < 0 || o.d > best.d || (o.d == best.d && o.k < best.k));
                if (take) best = o;
            }
        }
        GW_TICK(3);
        // ---- (E) thread 0 takes the decision (CD:222-284)
        if (!fine) {
            if (tid == 0) {
                const long long C = count;
                if (C > S.n_memo) S.n_memo = C;
                const bool found = best.k >= 0;
                const double maxd = best.d;
                const int nw = S.nw;
                S.maxd = maxd;
                S.maxi = found ? c_i[best.k] : 0.0;
                S.best_k = found ? best.k : 0;
                win_maxd[T.ev_off + nw] = found ? maxd : __builtin_nan("");
                win_det[T.ev_off + nw] = 0;
                S.nw = nw + 1;
                if (found && maxd > P.threshold) {
                    S.fine = 1;              // next scan: fine tune around maxi (CD:231-251)
                } else {
                    // negative: enlarge the window (CD:271-284)
                    double end = S.end, ws = S.ws, dws = S.dws;
                    if (end + ws <= fn) {
                        end += ws;
                        if (ws < winstep) { ws += dws; dws *= 2; }
                        if (ws > winstep) ws = winstep;
                        S.go = (end <= fn) ? 1 : 0;
                    } else if (end != fn) {
                        end = fn;
                        S.go = 1;
                    } else {
                        S.go = 0;
                        S.tail_in_lds = 1;   // ldsEnd = P(turn end) of this epoch
                    }
                    S.end = end; S.ws = ws; S.dws = dws;
                }
            }
        } else {
            if (tid == 0) {
                const long long C = S.C;
                double maxd = S.maxd, maxi = S.maxi;
                long long src = S.best_k;            // the coarse maximum stands unless a fine step beats it
                if (best.k >= 0) { maxd = best.d; maxi = c_i[C + best.k]; src = C + best.k; }
                const int nd = S.nd;
                det_start[T.ev_off + nd] = start;
                det_maxi[T.ev_off + nd] = maxi;
                det_d[T.ev_off + nd] = maxd;
                win_det[T.ev_off + S.nw - 1] = 1;
                S.seg_src = src;
                S.seg_dst = T.ev_off + nd;
                S.nd = nd + 1;
                S.maxd = maxd; S.maxi = maxi;
                S.fine = 0;
                S.n_memo = 0;
                S.n_written = S.n_written < C ? S.n_written : C;   // fine-scan scratch overwrote slots >= C
                S.C = 0;
                S.cur_i = minfeas;
                const double nstart = start + maxi;
                S.start = nstart;
                // new epoch: the sweep restarts at the new window start
                S.built_k = 0;
                S.built_i = minfeas;
                S.sweep_pos = (long long)nstart;
                S.fresh = 1;
                if (nstart + winsize * 2 <= fn) {
                    S.end = nstart + winsize * 2;
                    S.ws = minfeas;
                    S.dws = P.deltaws;
                    S.go = 1;
                } else {
                    S.go = 0;
                }
            }
        }
        __syncthreads();
        if (fine && seg_stats) {
            // fused mode: the detected segment's statistics = the cache record of the split
            // point (copied before the next scan's sweep, which comes after a barrier,
            // overwrites the slots of this epoch)
            const double* src = cache + S.seg_src * REC;
            double* dst = seg_stats + S.seg_dst * REC;
            for (int e = tid; e < REC; e += GW_TPB) dst[e] = src[e];
        }
    }
    if (seg_stats && !S.abort) {
        // fused mode: the tail segment [final start, turn end)
        const int nd = S.nd;
        if (nd >= T.ev_cap) {
            if (tid == 0) atomicOr(err, 4);
        } else {
            double* dst = seg_stats + (T.ev_off + nd) * REC;
            if ((long long)S.start >= n) {           // empty tail (an empty turn): all-zero moments
                for (int e = tid; e < REC; e += GW_TPB) dst[e] = 0.0;
            } else if (S.tail_in_lds) {
                for (int e = tid; e < REC; e += GW_TPB) {
                    int j, r;
                    decode_entry(e, j, r);           // packed index -> (j, r), j <= r
                    dst[e] = ldsEnd[tri_slot(r, j)];
                }
            } else {
                gw_sweep_tail<NW>(fr, (long long)S.start, n, dst, err);
            }
        }
    }
    if (tid == 0) {
        n_win[turn] = S.nw;
        final_start[turn] = S.start;
    }
#ifdef SPKD_PROFILE
    GW_TICK(1);
    if (tid == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_gw_prof[i], prof_acc[i]);
        atomicAdd(&g_gw_prof[4], prof_scans);
        atomicAdd(&g_gw_prof[5], 1ull);
        atomicAdd(&g_gw_prof[7], prof_items);
    }
    if (lane == 0) atomicAdd(&g_gw_prof[6], prof_passes);
#endif
}

}  // namespace spkd
