// Accuracy of v_rcp_f64 and of its Newton refinements on gfx950 (max relative error over
// random inputs against the correctly rounded 1/x of the host).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o rcp_precision rcp_precision.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>

__global__ void k_rcp(const double* x, double* r0, double* r1, double* r2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double p = x[i];
    double r = __builtin_amdgcn_rcp(p);
    r0[i] = r;
    double e = fma(-p, r, 1.0);
    r = fma(e, r, r);
    r1[i] = r;
    e = fma(-p, r, 1.0);
    r = fma(e, r, r);
    r2[i] = r;
}

int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> m(1.0, 2.0);
    std::uniform_int_distribution<int> ex(-300, 300);
    for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), ex(g));
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rcp, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0; long long ne1 = 0, ne2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        const double tr = (double)t;
        e0 = std::fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e1 = std::fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
        ne1 += b[i] != tr; ne2 += c[i] != tr;
    }
    printf("max relative error: v_rcp_f64 %.3e, + 1 Newton step %.3e (%lld of %d not correctly rounded), + 2 steps %.3e (%lld)\n",
           e0, e1, ne1, n, e2, ne2);
    return 0;
}
