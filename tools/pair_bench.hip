// Micro-benchmark of quad_pair_logdet on expanded records (mimics k_matrix's inner loop).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "spkd_cluster.hpp"
using namespace spkd;

template <int MODE>
__global__ __launch_bounds__(256) void k_pairs(const double* __restrict__ ex, int n_rec, double* __restrict__ out, int* err) {
    const int wave = threadIdx.x >> 6;
    const QuadLane L = quad_lane();
    const int a = blockIdx.x % n_rec;
    const double* A = ex + (size_t)a * EXR;
    double acc = 0.0;
    for (int base = 4 * wave; base < n_rec; base += 16) {
        const double* recs[4];
        bool selfs[4];
        for (int mi = 0; mi < 4; ++mi) { int c = base + mi; c = c < n_rec ? c : n_rec - 1; recs[mi] = ex + (size_t)c * EXR; selfs[mi] = false; }
        int rc = base + L.m; rc = rc < n_rec ? rc : n_rec - 1;
        const double* C = ex + (size_t)rc * EXR;
        if (MODE == 0) {
            acc += quad_pair_logdet(SPKD_BIC, A, C, false, L, recs, selfs, err);
        } else {
            // formation only
            QuadRows q; double sv[QS];
            const double* A2 = A; asm volatile("" : "+v"(A2));
            quad_load(A2, L.tt, q, sv);
            __builtin_amdgcn_sched_barrier(0);
            quad_add(C, L.tt, q, sv);
            if (MODE == 2) quad_cov(q, sv, ex_count(A2) + ex_count(C));
            double s = 0;
            for (int ss = 0; ss < QS; ++ss) for (int j = 0; j < D; ++j) s += q.r[ss][j];
            acc += s;
        }
    }
    if (L.t == 0) out[blockIdx.x * 16 + wave * 4 + L.m] = acc;
}

int main(int argc, char** argv) {
    int n_rec = argc > 1 ? atoi(argv[1]) : 387;
    int blocks = argc > 2 ? atoi(argv[2]) : 4096;
    std::vector<double> h((size_t)n_rec * EXR);
    srand(2);
    for (int r = 0; r < n_rec; ++r) {
        // moments of 500 random frames
        std::vector<double> M(DA * DA, 0.0);
        for (int f = 0; f < 500; ++f) {
            double x[DA];
            for (int i = 0; i < D; ++i) x[i] = (rand() / (double)RAND_MAX) - 0.5 + 0.01 * r;
            x[D] = 1.0;
            for (int i = 0; i < DA; ++i) for (int j = 0; j < DA; ++j) M[j * DA + i] += x[i] * x[j];
        }
        memcpy(&h[(size_t)r * EXR], M.data(), sizeof(double) * EXR);
    }
    double *dE, *dO; int* dErr;
    hipMalloc(&dE, h.size() * 8); hipMalloc(&dO, (size_t)blocks * 16 * 8); hipMalloc(&dErr, 4);
    hipMemcpy(dE, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dErr, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"full BIC pair", "load A+C only", "load + cov"};
    for (int mode = 0; mode < 3; ++mode)
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_pairs<0>, dim3(blocks), dim3(256), 0, 0, dE, n_rec, dO, dErr);
            if (mode == 1) hipLaunchKernelGGL(k_pairs<1>, dim3(blocks), dim3(256), 0, 0, dE, n_rec, dO, dErr);
            if (mode == 2) hipLaunchKernelGGL(k_pairs<2>, dim3(blocks), dim3(256), 0, 0, dE, n_rec, dO, dErr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double pairs = (double)blocks * n_rec;
            int herr; hipMemcpy(&herr, dErr, 4, hipMemcpyDeviceToHost);
            printf("%-16s: %.3f ms, %.1f M pairs/s, err=%d\n", names[mode], ms, pairs / ms / 1e3, herr);
        }
    return 0;
}
