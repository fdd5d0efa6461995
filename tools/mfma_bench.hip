// Does the fp64 matrix pipe run beside the fp64 vector pipe?  (gfx950)
// One workgroup per CU; waves 0..3 (one per SIMD) issue v_mfma_f64_16x16x4_f64, waves
// 4..7 (their SIMD partners) issue v_fma_f64, each alone and both together.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_bench mfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int ITER = 4000;

// mode bit 0: the MFMA waves work; bit 1: the FMA waves work; NACC: independent MFMA accumulators
template <int NACC>
__global__ void k_mix(double* out, unsigned long long* cycles, double seed, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    double a = seed + threadIdx.x * 1e-6, b = seed - threadIdx.x * 1e-6;
    double4_t c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = double4_t{seed, seed, seed, seed};
    double acc[16], src[16];
    for (int i = 0; i < 16; ++i) { acc[i] = seed * (i + 1); src[i] = 1e-9 * (i + 1) + seed; }
    double m = seed * 1e-3;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (mf) {
        if (mode & 1) {
#pragma unroll 1
            for (int it = 0; it < ITER; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    c[u % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u % NACC], 0, 0, 0);
            }
        }
    } else {
        if (mode & 2) {
#pragma unroll 1
            for (int it = 0; it < ITER; ++it) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(src[i]), "v"(m));
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#undef X
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NACC>
void run(int mode) {
    const int blocks = 256, tpb = 512;
    double* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, (size_t)blocks * tpb * 8);
    hipMalloc(&d_cyc, (size_t)blocks * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_mix<NACC>, dim3(blocks), dim3(tpb), 0, 0, d_out, d_cyc, 1.000001, mode);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> c((size_t)blocks * 8);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double sm = 0, sf = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? sm : sf) += (double)c[b * 8 + w];
    sm /= blocks * 4; sf /= blocks * 4;
    printf("accumulators %d  mfma waves %s, fma waves %s:  %.1f cycles per mfma_f64_16x16x4, %.2f cycles per v_fma_f64\n",
           NACC, (mode & 1) ? "on " : "off", (mode & 2) ? "on " : "off",
           (mode & 1) ? sm / (ITER * 16.0) : 0.0, (mode & 2) ? sf / (ITER * 16.0) : 0.0);
    hipFree(d_out); hipFree(d_cyc);
}

int main() {
    run<4>(1); run<4>(2); run<4>(3);
    run<2>(1); run<2>(3);
    run<1>(1); run<1>(3);
    return 0;
}
