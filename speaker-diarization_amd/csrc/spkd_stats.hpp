// K_stats: frames -> packed sufficient-statistics records.
//
// Replaces the inputs of every np.cov call of the clustering scripts
// (get_spk_features + np.cov, spk-clustering.py:46-52,88-94): instead of copying
// and re-scanning raw frames per distance, each frame is read ONCE and folded
// into the augmented second-moment record of its set.
//
// Two deterministic passes (no float atomics, so results are bit-reproducible):
//   k_chunk_stats : one workgroup per chunk of <= STATS_CHUNK consecutive frames
//                   of one range; frames staged through LDS, each thread owns a
//                   fixed set of (r, c) accumulators in fp64;
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
constexpr int STATS_TILE = 64;        // frames staged per LDS tile
constexpr int STATS_CHUNK = 1024;     // frames per chunk (host splits ranges)
constexpr int STATS_EPT = 4;          // entries per thread: ceil(820 / 256)

__global__ __launch_bounds__(STATS_TPB) void k_chunk_stats(
        const float* __restrict__ frames, const Chunk* __restrict__ chunks,
        double* __restrict__ partial) {
    __shared__ float xs[STATS_TILE][DA];
    const int tid = threadIdx.x;
    const Chunk ch = chunks[blockIdx.x];
    int er[STATS_EPT], ec[STATS_EPT];
    double acc[STATS_EPT];
#pragma unroll
    for (int m = 0; m < STATS_EPT; ++m) {
        int e = tid + STATS_TPB * m;
        if (e >= REC) e = REC - 1;          // clamped duplicates are never written
        decode_entry(e, er[m], ec[m]);
        acc[m] = 0.0;
    }
    const float* base = frames + ch.begin * (int64_t)D;
    for (int t0 = 0; t0 < ch.len; t0 += STATS_TILE) {
        const int tl = min(STATS_TILE, ch.len - t0);
        const float* src = base + (int64_t)t0 * D;
        for (int idx = tid; idx < tl * D; idx += STATS_TPB) {
            int f = idx / D, c = idx - f * D;
            xs[f][c] = src[idx];
        }
        if (tid < tl) xs[tid][D] = 1.0f;
        __syncthreads();
        for (int f = 0; f < tl; ++f) {
#pragma unroll
            for (int m = 0; m < STATS_EPT; ++m)
                acc[m] = fma((double)xs[f][er[m]], (double)xs[f][ec[m]], acc[m]);
        }
        __syncthreads();
    }
    double* out = partial + (int64_t)blockIdx.x * REC;
#pragma unroll
    for (int m = 0; m < STATS_EPT; ++m) {
        int e = tid + STATS_TPB * m;
        if (e < REC) out[e] = acc[m];
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
