// Micro-benchmark: one-wave-per-SIMD quad elimination (quad_pair_logdet) vs the
// column-blocked form (two waves per SIMD) on BIC pair distances from quad records.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "spkd_cluster.hpp"
#include "spkd_blocked.hpp"
#include "spkd_tri.hpp"
using namespace spkd;

// BIC union covariance of (A in LDS, C in global), column by column
struct BenchSrc {
    const double* ldsA;     // LDS quad record of A
    const double* gC;       // global quad record of C (already + lane t)
    int t;
    double f, k1;           // 1/(n-1), -(f/n)
    double sv[QS];          // sums of the union for the lane's rows
    template <int J>
    __device__ __forceinline__ void load(double (&col)[QS]) {
#pragma unroll
        for (int s = 0; s < QS; ++s) col[s] = gC[(s * DA + J) * 16];
    }
    template <int J>
    __device__ __forceinline__ void finish(double (&col)[QS]) {
        const double svj = bcast16<J % QL>(sv[J / QL]);
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const double q = ldsA[(s * DA + J) * 16 + t] + col[s];
            col[s] = fma(k1 * sv[s], svj, f * q);
        }
    }
};

#ifndef TRI_TPB
#define TRI_TPB 512
#endif
template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : (MODE == 1 ? 512 : TRI_TPB)) void k_pairs(const double* __restrict__ qr, int n_rec,
                                                                   double* __restrict__ out, int* err) {
    __shared__ double ldsA[QREC];
    __shared__ double schur[(MODE == 1 ? 8 : 1) * SCHUR_TILE];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const QuadLane L = quad_lane();
    const int a = blockIdx.x % n_rec;
    const double* A = qr + (size_t)a * QREC;
    for (int e = threadIdx.x; e < QREC; e += blockDim.x) ldsA[e] = A[e];
    __syncthreads();
    const double nA = ldsA[QREC_COUNT_AT];
    double acc = 0.0;
    for (int base = 4 * wave; base < n_rec; base += 4 * nw) {
        int rc = base + L.m; rc = rc < n_rec ? rc : n_rec - 1;
        const double* C = qr + (size_t)rc * QREC;
        if (MODE == 0) {
            // unblocked: whole matrix in registers (one wave per SIMD)
            QuadRows q; double sv[QS], svc[QS];
            int ta = L.t; asm volatile("" : "+v"(ta));
            const double n = nA + C[QREC_COUNT_AT];
            const double f = 1.0 / (n - 1.0);
            quad_load_scaled(C + L.t, 0, f, q, svc);
            __builtin_amdgcn_sched_barrier(0);
            double c1[QS];
            for (int s2 = 0; s2 < QS; ++s2) {
                for (int j = 0; j < D; ++j) q.r[s2][j] = fma(f, ldsA[(s2 * DA + j) * 16 + ta], q.r[s2][j]);
                sv[s2] = ldsA[(s2 * DA + D) * 16 + ta] + svc[s2];
                c1[s2] = -((f / n) * sv[s2]);
                __builtin_amdgcn_sched_barrier(0);
            }
            QuadRank1<0>::run(q, c1, sv);
            double det; quad_det_nopivot(q, det);
            acc += log(det);
        } else if (MODE == 2) {
            // symmetric form: lower triangle only
            QuadRows q; double sv[QS];
            int ta = L.t; asm volatile("" : "+v"(ta));
            const double n = nA + C[QREC_COUNT_AT];
            const double f = 1.0 / (n - 1.0);
            const double* Ct = C + L.t;
#pragma unroll
            for (int s2 = 0; s2 < QS; ++s2) {
#pragma unroll
                for (int j = 0; j < tri_cols(s2); ++j) q.r[s2][j] = Ct[(s2 * DA + j) * 16];
                sv[s2] = Ct[(s2 * DA + D) * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
            double c1[QS];
#pragma unroll
            for (int s2 = 0; s2 < QS; ++s2) {
#pragma unroll
                for (int j = 0; j < tri_cols(s2); ++j) q.r[s2][j] = f * (ldsA[(s2 * DA + j) * 16 + ta] + q.r[s2][j]);
                sv[s2] = ldsA[(s2 * DA + D) * 16 + ta] + sv[s2];
                c1[s2] = -((f / n) * sv[s2]);
                __builtin_amdgcn_sched_barrier(0);
            }
            TriRank1<0>::run(q, c1, sv);
            double det; tri_det_nopivot(q, det);
            acc += log(det);
        } else {
            BenchSrc src;
            int ta = L.t; asm volatile("" : "+v"(ta));
            src.ldsA = ldsA; src.gC = C + L.t; src.t = ta;
            const double n = nA + C[QREC_COUNT_AT];
            src.f = 1.0 / (n - 1.0); src.k1 = -(src.f / n);
#pragma unroll
            for (int s = 0; s < QS; ++s) src.sv[s] = ldsA[(s * DA + D) * 16 + ta] + C[(s * DA + D) * 16 + L.t];
            bool ok;
            acc += quad_logdet_blocked(src, schur + wave * SCHUR_TILE, L, ok);
        }
    }
    if (L.t == 0) out[(size_t)blockIdx.x * 64 + wave * 4 + L.m] = acc;
}

int main(int argc, char** argv) {
    int n_rec = argc > 1 ? atoi(argv[1]) : 387;
    int blocks = argc > 2 ? atoi(argv[2]) : 4096;
    std::vector<double> h((size_t)n_rec * QREC, 0.0);
    srand(2);
    for (int r = 0; r < n_rec; ++r) {
        std::vector<double> M(DA * DA, 0.0);
        const int nf = 300 + (r % 7) * 100;
        for (int f = 0; f < nf; ++f) {
            double x[DA];
            for (int i = 0; i < D; ++i) x[i] = (rand() / (double)RAND_MAX) - 0.5 + 0.01 * r;
            x[D] = 1.0;
            for (int i = 0; i < DA; ++i) for (int j = 0; j < DA; ++j) M[j * DA + i] += x[i] * x[j];
        }
        double* o = &h[(size_t)r * QREC];
        for (int i = 0; i < D; ++i) for (int j = 0; j < DA; ++j) o[qr_index(i, j)] = M[j * DA + i];
        o[QREC_COUNT_AT] = nf;
    }
    double *dE, *dO0, *dO1; int* dErr;
    hipMalloc(&dE, h.size() * 8); hipMalloc(&dO0, (size_t)blocks * 64 * 8); hipMalloc(&dO1, (size_t)blocks * 64 * 8); hipMalloc(&dErr, 4);
    hipMemcpy(dE, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dErr, 0, 4); hipMemset(dO0, 0, (size_t)blocks * 64 * 8); hipMemset(dO1, 0, (size_t)blocks * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_pairs<0>, dim3(blocks), dim3(256), 0, 0, dE, n_rec, dO0, dErr);
            else if (mode == 1) hipLaunchKernelGGL(k_pairs<1>, dim3(blocks), dim3(512), 0, 0, dE, n_rec, dO1, dErr);
            else hipLaunchKernelGGL(k_pairs<2>, dim3(blocks), dim3(TRI_TPB), 0, 0, dE, n_rec, dO1, dErr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms, %.1f M pairs/s\n", mode == 0 ? "quad    (4 waves)" : mode == 1 ? "blocked (8 waves)" : "tri", ms, (double)blocks * n_rec / ms / 1e3);
        }
    // compare per-block sums: mode 0 has 16 slots per block, mode 1 has 32; compare block totals
    std::vector<double> o0((size_t)blocks * 64), o1((size_t)blocks * 64);
    hipMemcpy(o0.data(), dO0, o0.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o1.data(), dO1, o1.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int b = 0; b < blocks; ++b) {
        double s0 = 0, s1 = 0;
        for (int i = 0; i < 64; ++i) { s0 += o0[(size_t)b * 64 + i]; s1 += o1[(size_t)b * 64 + i]; }
        worst = fmax(worst, fabs(s0 - s1) / fabs(s0));
    }
    printf("worst relative difference of block sums: %.3e\n", worst);
    return 0;
}
