// K_stats: frames -> packed sufficient-statistics records.
//
// Replaces the inputs of every np.cov call of the clustering scripts
// (get_spk_features + np.cov, spk-clustering.py:46-52,88-94): instead of copying
// and re-scanning raw frames per distance, each frame is read ONCE and folded
// into the augmented second-moment record of its set.
//
// Two deterministic passes (no float atomics, so results are bit-reproducible):
//   k_chunk_stats : one workgroup per chunk of <= STATS_CHUNK consecutive frames
//                   of one range; frames staged through LDS (converted to fp64
//                   once), 4 x 4 register-blocked fp64 accumulation per lane;
//   k_reduce_sets : per set, sums its chunk partials in chunk order.
// Algorithmic bytes: 156 B per frame read + 6 560 B per set written.
#pragma once
#include "spkd_device.hpp"

namespace spkd {

struct Chunk {
    int64_t begin;      // first frame
    int32_t len;        // frames in the chunk
    int32_t set;        // owning set
};

constexpr int STATS_TPB = 256;
constexpr int STATS_WAVES = STATS_TPB / WAVE;
constexpr int STATS_TILE = 64;        // frames staged per LDS tile
constexpr int STATS_CHUNK = 1024;     // frames per chunk (host splits ranges)
constexpr int SB = 4;                 // register block: each lane owns a 4 x 4 block of entries
constexpr int SNB = DA / SB;          // 10 block rows / columns
constexpr int SBLOCKS = SNB * (SNB + 1) / 2;   // 55 upper-triangular blocks per wave

// Register-blocked accumulation: a wave covers the whole 40 x 40 upper triangle with
// 55 lanes, each holding a 4 x 4 block of fp64 accumulators; per frame a lane reads
// 4 + 4 doubles from the LDS tile (two ds_read_b128 pairs, broadcast among the lanes
// that share a block row / column) and issues 16 FMAs -- 1.5 instructions per entry
// instead of 5 for the one-entry-per-thread form.  The four waves of the workgroup
// take every fourth frame of the tile; their partial blocks are summed through LDS
// in wave order at the end (deterministic).
__global__ __launch_bounds__(STATS_TPB) void k_chunk_stats(
        const float* __restrict__ frames, const Chunk* __restrict__ chunks,
        double* __restrict__ partial) {
    __shared__ double xs[STATS_TILE][DA];                       // 20 KB
    __shared__ double part[STATS_WAVES][SBLOCKS][SB * SB];      // 28 KB
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const Chunk ch = chunks[blockIdx.x];
    // lane -> block (bi <= bj)
    int bi = 0, rem = lane < SBLOCKS ? lane : 0;
    while (rem >= SNB - bi) { rem -= SNB - bi; ++bi; }
    const int bj = bi + rem;
    double acc[SB][SB];
#pragma unroll
    for (int a = 0; a < SB; ++a)
#pragma unroll
        for (int b = 0; b < SB; ++b) acc[a][b] = 0.0;
    const float* base = frames + ch.begin * (int64_t)D;
    // software pipeline: the floats of tile k+1 are loaded into registers while tile k
    // is being accumulated, so the global latency is paid once per chunk, not per tile
    constexpr int PF = (STATS_TILE * D + STATS_TPB - 1) / STATS_TPB;      // 10 floats per thread
    int loff[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int idx = tid + STATS_TPB * k;
        const int f = idx / D;
        loff[k] = f * DA + (idx - f * D);
    }
    float pf[PF];
    auto issue = [&](int t0) {
        const int tl = min(STATS_TILE, ch.len - t0);
        const float* src = base + (int64_t)t0 * D;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int idx = tid + STATS_TPB * k;
            pf[k] = idx < tl * D ? src[idx] : 0.0f;
        }
    };
    issue(0);
    double* xsf = &xs[0][0];
    for (int t0 = 0; t0 < ch.len; t0 += STATS_TILE) {
        const int tl = min(STATS_TILE, ch.len - t0);
#pragma unroll
        for (int k = 0; k < PF; ++k)
            if (tid + STATS_TPB * k < STATS_TILE * D) xsf[loff[k]] = (double)pf[k];
        if (tid < STATS_TILE) xs[tid][D] = 1.0;
        __syncthreads();
        if (t0 + STATS_TILE < ch.len) issue(t0 + STATS_TILE);
        for (int f = wave; f < tl; f += STATS_WAVES) {
            double xi[SB], xj[SB];
#pragma unroll
            for (int a = 0; a < SB; ++a) { xi[a] = xs[f][SB * bi + a]; xj[a] = xs[f][SB * bj + a]; }
#pragma unroll
            for (int a = 0; a < SB; ++a)
#pragma unroll
                for (int b = 0; b < SB; ++b) acc[a][b] = fma(xi[a], xj[b], acc[a][b]);
        }
        __syncthreads();
    }
    if (lane < SBLOCKS) {
#pragma unroll
        for (int a = 0; a < SB; ++a)
#pragma unroll
            for (int b = 0; b < SB; ++b) part[wave][lane][a * SB + b] = acc[a][b];
    }
    __syncthreads();
    double* out = partial + (int64_t)blockIdx.x * REC;
    for (int e = tid; e < REC; e += STATS_TPB) {
        int r, c;
        decode_entry(e, r, c);
        const int pbi = r / SB, pbj = c / SB;
        const int blk = pbi * SNB - (pbi * (pbi - 1)) / 2 + (pbj - pbi);
        const int w = (r % SB) * SB + (c % SB);
        double v = part[0][blk][w];
#pragma unroll
        for (int q = 1; q < STATS_WAVES; ++q) v += part[q][blk][w];
        out[e] = v;
    }
}

// set s owns chunks [set_chunk_off[s], set_chunk_off[s+1])
__global__ __launch_bounds__(STATS_TPB) void k_reduce_sets(
        const double* __restrict__ partial, const int64_t* __restrict__ set_chunk_off,
        double* __restrict__ stats) {
    const int64_t s = blockIdx.x;
    const int64_t c0 = set_chunk_off[s], c1 = set_chunk_off[s + 1];
    for (int e = threadIdx.x; e < REC; e += STATS_TPB) {
        double acc = 0.0;
        for (int64_t c = c0; c < c1; ++c) acc += partial[c * REC + e];
        stats[s * REC + e] = acc;
    }
}

}  // namespace spkd
