// feacat-shaped MFCC front-end: 16-bit PCM -> float32 [T][39]
// features per the reference's fconfig.cfg (audiofile -> fft magnitude -> mel / power ->
// dct 12 -> cms +-75 -> delta, delta-delta -> normalization -> 39x39 transform).
// PARITY UNPINNED (feacat is not available): the module semantics the configuration file
// does not spell out are the choices listed in oracle/mfcc_numpy.py, which this restates.
//
//   k_mfcc_static : workgroup per 8 frames.  Pre-emphasis + Hamming window into LDS, then a
//                   direct 512-point DFT (thread k = bin k, the 8 frames share every twiddle;
//                   205 k MAC per frame is 1.5 ms per audio-hour -- no FFT needed), magnitude,
//                   mel filterbank, log, DCT, log power -> static [T][13]
//   k_mfcc_post   : workgroup per 128 frames.  The static rows it needs (+-75 for the mean,
//                   +-4 for the two delta stages) staged in LDS once; cms, deltas,
//                   normalization and the 39x39 transform from there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spkd {

constexpr int MF_WIN = 400;          // samples per window
constexpr int MF_NFFT = 512;
constexpr int MF_BINS = MF_NFFT / 2 + 1;
constexpr int MF_MEL = 21;
constexpr int MF_CEP = 12;
constexpr int MF_STATIC = MF_CEP + 1;
constexpr int MF_DIM = 3 * MF_STATIC;          // 39
constexpr int MF_FR = 8;             // frames per workgroup (static stage)
constexpr int MF_TPB = 256;
constexpr float MF_FLOOR = 1e-10f;

__global__ __launch_bounds__(MF_TPB) void k_mfcc_static(
        const int16_t* __restrict__ pcm, long long n_samples, long long n_frames, int hop, float pre_emph,
        const float* __restrict__ melfb /* [MF_MEL][MF_BINS] */, const float* __restrict__ dct /* [MF_CEP][MF_MEL] */,
        float* __restrict__ stat /* [T][13] */) {
    __shared__ float y[MF_FR][MF_WIN];
    __shared__ float tw_c[MF_NFFT], tw_s[MF_NFFT];
    __shared__ float mag[MF_FR][MF_BINS + 3];
    __shared__ float lmel[MF_FR][MF_MEL + 3];
    __shared__ float pw[MF_FR][MF_TPB / 64];
    const int tid = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * MF_FR;
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = tid; n < MF_NFFT; n += MF_TPB) {
        const double a = two_pi * (double)n / (double)MF_NFFT;
        tw_c[n] = (float)cos(a);
        tw_s[n] = (float)sin(a);
    }
    for (int e = tid; e < MF_FR * MF_WIN; e += MF_TPB) {
        const int f = e / MF_WIN, n = e - f * MF_WIN;
        const long long t = t0 + f;
        float v = 0.0f;
        if (t < n_frames) {
            long long i = t * hop - MF_WIN / 2 + n;
            long long ic = i < 0 ? 0 : (i >= n_samples ? n_samples - 1 : i);
            long long ip = i - 1 < 0 ? 0 : (i - 1 >= n_samples ? n_samples - 1 : i - 1);
            const double w = 0.54 - 0.46 * cos(two_pi * (double)n / (double)(MF_WIN - 1));
            v = (float)(((double)pcm[ic] - (double)pre_emph * (double)pcm[ip]) * w);
        }
        y[f][n] = v;
    }
    __syncthreads();
    // direct DFT: bins tid and (thread 0 only) 256
    for (int k = tid; k < MF_BINS; k += MF_TPB) {
        float re[MF_FR], im[MF_FR];
#pragma unroll
        for (int f = 0; f < MF_FR; ++f) { re[f] = 0.0f; im[f] = 0.0f; }
        int ph = 0;                                   // (k * n) mod 512
        for (int n = 0; n < MF_WIN; ++n) {
            const float c = tw_c[ph], s = tw_s[ph];
            ph = (ph + k) & (MF_NFFT - 1);
#pragma unroll
            for (int f = 0; f < MF_FR; ++f) {
                const float v = y[f][n];
                re[f] = fmaf(v, c, re[f]);
                im[f] = fmaf(-v, s, im[f]);
            }
        }
#pragma unroll
        for (int f = 0; f < MF_FR; ++f) mag[f][k] = sqrtf(re[f] * re[f] + im[f] * im[f]);
    }
    __syncthreads();
    // log power: sum of squared magnitudes (wave partials, then 4 values per frame)
    for (int f = 0; f < MF_FR; ++f) {
        float p = 0.0f;
        for (int k = tid; k < MF_BINS; k += MF_TPB) p = fmaf(mag[f][k], mag[f][k], p);
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) p += __shfl_xor(p, s);
        if ((tid & 63) == 0) pw[f][tid >> 6] = p;
    }
    // mel filterbank + log: thread (f, m)
    if (tid < MF_FR * MF_MEL) {
        const int f = tid / MF_MEL, m = tid - f * MF_MEL;
        float acc = 0.0f;
        const float* w = melfb + m * MF_BINS;
        for (int k = 0; k < MF_BINS; ++k) acc = fmaf(w[k], mag[f][k], acc);
        lmel[f][m] = logf(fmaxf(acc, MF_FLOOR));
    }
    __syncthreads();
    // DCT (cepstra 1..12) and the power column: thread (f, c)
    if (tid < MF_FR * MF_STATIC) {
        const int f = tid / MF_STATIC, c = tid - f * MF_STATIC;
        const long long t = t0 + f;
        if (t < n_frames) {
            float v;
            if (c < MF_CEP) {
                v = 0.0f;
                for (int m = 0; m < MF_MEL; ++m) v = fmaf(dct[c * MF_MEL + m], lmel[f][m], v);
            } else {
                float p = 0.0f;
                for (int w = 0; w < MF_TPB / 64; ++w) p += pw[f][w];
                v = logf(fmaxf(p, MF_FLOOR));
            }
            stat[t * MF_STATIC + c] = v;
        }
    }
}

constexpr int MP_FR = 128;           // frames per workgroup (post stage)
constexpr int MP_HALO = 4;           // two delta stages of width 2

__global__ __launch_bounds__(MF_TPB) void k_mfcc_post(
        const float* __restrict__ stat, long long n_frames, int cms_left, int cms_right,
        int w1, float norm1, int w2, float norm2,
        const float* __restrict__ mean, const float* __restrict__ scale, const float* __restrict__ transform,
        float* __restrict__ out /* [T][39] */) {
    extern __shared__ float mp_lds[];
    const int span = MP_FR + 2 * MP_HALO;                  // frames whose cms / deltas are formed here
    const int raw_n = span + cms_left + cms_right;         // static rows staged
    float* raw = mp_lds;                                   // [raw_n][13]
    float* cms = raw + raw_n * MF_STATIC;                  // [span][13]
    float* d1 = cms + span * MF_STATIC;                    // [span][13]
    float* z = d1 + span * MF_STATIC;                      // [MP_FR][39]
    float* tr = z + MP_FR * MF_DIM;                        // [39][39]
    const int tid = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * MP_FR;
    const long long first = t0 - MP_HALO;                  // global frame of span index 0
    const long long raw0 = first - cms_left;               // global frame of raw row 0
    for (int e = tid; e < raw_n * MF_STATIC; e += MF_TPB) {
        const int r = e / MF_STATIC, c = e - r * MF_STATIC;
        const long long g = raw0 + r;
        raw[e] = (g >= 0 && g < n_frames) ? stat[g * MF_STATIC + c] : 0.0f;
    }
    for (int e = tid; e < MF_DIM * MF_DIM; e += MF_TPB) tr[e] = transform[e];
    __syncthreads();
    auto clampg = [&](long long g) { return g < 0 ? 0 : (g >= n_frames ? n_frames - 1 : g); };
    // cms of span frame i = static - mean over the existing frames of [g - left, g + right]
    for (int e = tid; e < span * MF_STATIC; e += MF_TPB) {
        const int i = e / MF_STATIC, c = e - i * MF_STATIC;
        const long long g = clampg(first + i);             // frames beyond the file repeat the border frame
        long long lo = g - cms_left, hi = g + cms_right + 1;
        lo = lo < 0 ? 0 : lo;
        hi = hi > n_frames ? n_frames : hi;
        double s = 0.0;
        for (long long q = lo; q < hi; ++q) s += (double)raw[(q - raw0) * MF_STATIC + c];
        cms[e] = (float)((double)raw[(g - raw0) * MF_STATIC + c] - s / (double)(hi - lo));
    }
    __syncthreads();
    // d1 over the span (index clamps happen on GLOBAL frame numbers, like the restatement)
    for (int e = tid; e < span * MF_STATIC; e += MF_TPB) {
        const int i = e / MF_STATIC, c = e - i * MF_STATIC;
        const long long g = clampg(first + i);
        float v = 0.0f;
        for (int k = 1; k <= w1; ++k) {
            const long long a = clampg(g + k) - first, b = clampg(g - k) - first;
            const bool ok = a >= 0 && a < span && b >= 0 && b < span;
            v += ok ? (float)k * (cms[a * MF_STATIC + c] - cms[b * MF_STATIC + c]) : 0.0f;
        }
        d1[e] = v / norm1;
    }
    __syncthreads();
    for (int e = tid; e < MP_FR * MF_STATIC; e += MF_TPB) {
        const int f = e / MF_STATIC, c = e - f * MF_STATIC;
        const long long g = t0 + f;
        if (g >= n_frames) continue;
        const int i = f + MP_HALO;
        float v = 0.0f;
        for (int k = 1; k <= w2; ++k) {
            const long long a = clampg(g + k) - first, b = clampg(g - k) - first;
            v += (float)k * (d1[a * MF_STATIC + c] - d1[b * MF_STATIC + c]);
        }
        const float d2 = v / norm2;
        z[f * MF_DIM + c] = (cms[i * MF_STATIC + c] - mean[c]) * scale[c];
        z[f * MF_DIM + MF_STATIC + c] = (d1[i * MF_STATIC + c] - mean[MF_STATIC + c]) * scale[MF_STATIC + c];
        z[f * MF_DIM + 2 * MF_STATIC + c] = (d2 - mean[2 * MF_STATIC + c]) * scale[2 * MF_STATIC + c];
    }
    __syncthreads();
    for (int e = tid; e < MP_FR * MF_DIM; e += MF_TPB) {
        const int f = e / MF_DIM, r = e - f * MF_DIM;
        const long long g = t0 + f;
        if (g >= n_frames) continue;
        float v = 0.0f;
        for (int c = 0; c < MF_DIM; ++c) v = fmaf(tr[r * MF_DIM + c], z[f * MF_DIM + c], v);
        out[g * MF_DIM + r] = v;
    }
}

}  // namespace spkd
