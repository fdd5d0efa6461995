// Micro-benchmark: single-matrix row-per-lane elimination (v_readlane broadcast) vs
// quad elimination (4 matrices per wave, DPP row_newbcast broadcast).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I speaker-diarization_amd/csrc -o tools/lu_bench tools/lu_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "spkd_device.hpp"
#include "quad_square.hpp"
using namespace spkd;

__global__ __launch_bounds__(256) void k_single(const double* __restrict__ M, double* __restrict__ out, int n, int reps) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (wave >= n) return;
    const double* A = M + (size_t)wave * D * D;
    double acc = 0.0;
    for (int rep = 0; rep < reps; ++rep) {
        double a[DA];
        const int i = lane < D ? lane : D - 1;
#pragma unroll
        for (int j = 0; j < D; ++j) a[j] = A[j * D + i];   // symmetric: column read = row
        double det;
        det_nopivot(a, det);
        acc += log(det);
    }
    if (lane == 0) out[wave] = acc / reps;
}

__global__ __launch_bounds__(256) void k_quad(const double* __restrict__ M, double* __restrict__ out, int n, int reps) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    const int m = lane >> 4, t = lane & 15;
    const int mat = wave * 4 + m;
    if (wave * 4 >= n) return;
    const double* A = M + (size_t)(mat < n ? mat : n - 1) * D * D;
    double acc = 0.0;
    for (int rep = 0; rep < reps; ++rep) {
        QuadRows q;
        const int tt = t < QL ? t : QL - 1;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < D; ++j) q.r[s][j] = A[j * D + (QL * s + tt)];
        double det;
        quad_det_nopivot(q, det);
        acc += log(det);
    }
    if (t == 0 && mat < n) out[mat] = acc / reps;
}

int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 65536;
    int reps = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<double> h((size_t)n * D * D), ref(n);
    srand(1);
    // SPD: B B^T + 39 I  (only a few distinct matrices, tiled)
    const int distinct = 64;
    std::vector<double> B(D * D);
    for (int m = 0; m < distinct; ++m) {
        for (auto& x : B) x = (rand() / (double)RAND_MAX) - 0.5;
        double* A = &h[(size_t)m * D * D];
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double s = (i == j) ? 2.0 + m * 0.01 : 0.0;
                for (int k = 0; k < D; ++k) s += B[i * D + k] * B[j * D + k];
                A[i * D + j] = s;
            }
        // reference log det by Cholesky on the host
        std::vector<double> L(A, A + D * D);
        double ld = 0.0;
        for (int k = 0; k < D; ++k) {
            ld += log(L[k * D + k]);
            for (int i = k + 1; i < D; ++i) {
                double l = L[i * D + k] / L[k * D + k];
                for (int j = k + 1; j < D; ++j) L[i * D + j] -= l * L[k * D + j];
            }
        }
        ref[m] = ld;
    }
    for (int m = distinct; m < n; ++m) {
        memcpy(&h[(size_t)m * D * D], &h[(size_t)(m % distinct) * D * D], sizeof(double) * D * D);
        ref[m] = ref[m % distinct];
    }
    double *dM, *dO;
    hipMalloc(&dM, h.size() * 8);
    hipMalloc(&dO, n * 8);
    hipMemcpy(dM, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> o(n);
    for (int which = 0; which < 2; ++which) {
        for (int it = 0; it < 3; ++it) {
            hipMemset(dO, 0, n * 8);
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_single, dim3((n + 3) / 4), dim3(256), 0, 0, dM, dO, n, reps);
            else hipLaunchKernelGGL(k_quad, dim3((n / 4 + 3) / 4), dim3(256), 0, 0, dM, dO, n, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(o.data(), dO, n * 8, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int m = 0; m < n; ++m) worst = fmax(worst, fabs(o[m] - ref[m]) / fabs(ref[m]));
            printf("%s: %d matrices x %d reps in %.3f ms -> %.1f M det/s, %.2f ns/det, worst rel err %.2e\n",
                   which ? "quad  " : "single", n, reps, ms, n * (double)reps / ms / 1e3, ms * 1e6 / (n * (double)reps), worst);
        }
    }
    return 0;
}
