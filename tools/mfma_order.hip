// Is v_mfma_f64_16x16x4_f64 a chain of four IEEE FMAs in k order?  Compares it bit for bit
// with c = fma(a[i][k], b[k][j], c), k = 0, 1, 2, 3 (and the reverse order) on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_order mfma_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

// one wave per block; A[16][4], B[4][16], C[16][16] per block
__global__ void k_mfma(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x, blk = blockIdx.x;
    const double a = A[blk * 64 + (l & 15) * 4 + (l >> 4)];        // A[i = l & 15][k = l >> 4]
    const double b = B[blk * 64 + (l >> 4) * 16 + (l & 15)];       // B[k = l >> 4][j = l & 15]
    double4_t c;
    for (int r = 0; r < 4; ++r) c[r] = C[blk * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)];   // C[i = (l>>4)+4r][j = l&15]
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[blk * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

int main() {
    const int nb = 4096;
    std::vector<double> A(nb * 64), B(nb * 64), C(nb * 256), D(nb * 256);
    std::mt19937_64 g(7);
    std::normal_distribution<double> n(0.0, 1.0);
    for (auto& x : A) x = n(g) * 37.0;
    for (auto& x : B) x = n(g) * 0.11;
    for (auto& x : C) x = n(g) * 1e3;
    // a third of the blocks: two of the four k slices zero (the masked partial steps of the sweep)
    for (int blk = 0; blk < nb; blk += 3)
        for (int i = 0; i < 16; ++i) { A[blk * 64 + i * 4 + 2] = 0; A[blk * 64 + i * 4 + 3] = 0; }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(nb), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    long long bad_fwd = 0, bad_rev = 0, bad_tree = 0;
    for (int blk = 0; blk < nb; ++blk)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const double c0 = C[blk * 256 + i * 16 + j];
                double f = c0, r = c0;
                for (int k = 0; k < 4; ++k) f = __builtin_fma(A[blk * 64 + i * 4 + k], B[blk * 64 + k * 16 + j], f);
                for (int k = 3; k >= 0; --k) r = __builtin_fma(A[blk * 64 + i * 4 + k], B[blk * 64 + k * 16 + j], r);
                double p = 0;
                for (int k = 0; k < 4; ++k) p = __builtin_fma(A[blk * 64 + i * 4 + k], B[blk * 64 + k * 16 + j], p);
                const double t = c0 + p;
                const double d = D[blk * 256 + i * 16 + j];
                bad_fwd += std::memcmp(&d, &f, 8) != 0;
                bad_rev += std::memcmp(&d, &r, 8) != 0;
                bad_tree += std::memcmp(&d, &t, 8) != 0;
            }
    printf("%d x 256 results: differ from the k = 0..3 FMA chain: %lld, from the k = 3..0 chain: %lld, from c + (sum of products): %lld\n",
           nb, bad_fwd, bad_rev, bad_tree);
    return 0;
}
