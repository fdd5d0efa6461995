// Prototype: a STREAMING ("left-looking") symmetric elimination in the quad layout that fits
// three waves per SIMD.  Column j of the matrix is loaded (packed record), formed and reduced
// against the finished columns k < j just in time; the registers hold the finished factor
// columns of the rows still alive (peak ~52 doubles) instead of the whole triangle (78), so the
// kernel fits 168 VGPRs.  The factor is the Cholesky one (one array serves as own operand and as
// DPP source); the accumulators run negated so that every update is a plain DPP fmac.
// Compared with tools/pair_bench.hip mode 2 (bare right-looking elimination, two waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../speaker-diarization_amd/csrc -o chol_bench chol_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "spkd_cluster.hpp"
#include "spkd_tri.hpp"
using namespace spkd;

constexpr int WINDOW = 387;
#ifndef CH_AHEAD
#define CH_AHEAD 2
#endif

// 1 / sqrt(x) to the last bits: v_rsq_f64 + one cubic step (e = 1 - x y^2; y (1 + e/2 + 3 e^2/8))
__device__ __forceinline__ double rsqrt_refined(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double t = x * y;
    const double e = fma(-t, y, 1.0);
    const double h = fma(0.375 * e, e, 0.5 * e);
    return fma(y, h, y);
}

// Forms column J of (minus) the covariance of the union of the LDS record A and a packed partner
// record: entries for slots s >= J / 13.  Loads are issued AHEAD columns early.
struct UnionColumns {
    const double* ldsA;          // quad record of A in LDS
    const double* rt0;           // packed partner record + t
    const double* rt1;
    double nf, nk;               // -f, -k1 (covariance scale and mean-correction weight, negated)
    double v1[QS], c1[QS];
    int ta;
    double raw[CH_AHEAD + 1][QS];

    template <int J>
    __device__ __forceinline__ void issue() {
        if constexpr (J < D) {
#pragma unroll
            for (int s = J / QL; s < QS; ++s) {
                const int e = pk_off(J) + QL * s - J;
                raw[J % (CH_AHEAD + 1)][s] = e < 512 ? rt0[e] : rt1[e - 512];
            }
        }
    }
    template <int J>
    __device__ __forceinline__ void take(double (&a)[QS]) {
        constexpr int SJ = J / QL, TJ = J % QL;
#pragma unroll
        for (int s = SJ; s < QS; ++s)
            a[s] = nf * (ldsA[(s * DA + J) * 16 + ta] + raw[J % (CH_AHEAD + 1)][s]);
        // mean correction: a[s] += nk v1_i v1_J  (v1_J broadcast from its owner lane)
        fmac_bcast16<TJ, true>(a[SJ], v1[SJ], c1[SJ]);
#pragma unroll
        for (int s = SJ + 1; s < QS; ++s) fmac_bcast16<TJ, false>(a[s], v1[SJ], c1[s]);
    }
};

template <int J, int K>
struct CholAcc {      // a[s] += G[s][K] * bcast(G[SJ][K]) for K = K .. J - 1
    static __device__ __forceinline__ void run(double (&a)[QS], const double (&G)[QS][D]) {
        if constexpr (K < J) {
            constexpr int SJ = J / QL, TJ = J % QL;
            fmac_bcast16<TJ, false>(a[SJ], G[SJ][K], G[SJ][K]);
#pragma unroll
            for (int s = SJ + 1; s < QS; ++s) fmac_bcast16<TJ, false>(a[s], G[SJ][K], G[s][K]);
            CholAcc<J, K + 1>::run(a, G);
        }
    }
};

template <int J>
struct CholCol {
    template <class Col>
    static __device__ __forceinline__ void run(Col& col, double (&G)[QS][D], double& det, int& sign) {
        if constexpr (J < D) {
            constexpr int SJ = J / QL, TJ = J % QL;
            col.template issue<J + CH_AHEAD>();
            double a[QS];
            col.template take<J>(a);                       // minus the formed entries
            __builtin_amdgcn_sched_barrier(0);
            CholAcc<J, 0>::run(a, G);                      // + sum_k G_ik G_jk  ->  -(a_ij - sum)
            const double npiv = bcast16_nop<TJ>(a[SJ]);    // minus the pivot
            __builtin_amdgcn_sched_barrier(0);
            const double piv = -npiv;
            det *= piv;
            sign |= __double2hiint(piv);
            const double nrs = -rsqrt_refined(piv);
#pragma unroll
            for (int s = SJ; s < QS; ++s) G[s][J] = a[s] * nrs;
            __builtin_amdgcn_sched_barrier(0);
            CholCol<J + 1>::run(col, G, det, sign);
        }
    }
};

template <class Col>
__device__ __forceinline__ bool chol_det(Col& col, double& det_out) {
    double G[QS][D];
    double det = 1.0;
    int sign = 0;
    [&]<int... I>(std::integer_sequence<int, I...>) { (col.template issue<I>(), ...); }(std::make_integer_sequence<int, CH_AHEAD>());
    CholCol<0>::run(col, G, det, sign);
    det_out = det;
    return (sign >= 0) && (det == det) && (det < __builtin_huge_val());
}

template <int MODE, int TPB>
__global__ __launch_bounds__(TPB, TPB / 256) void k_pairs(const double* __restrict__ pk, const double* __restrict__ qr, int n_rec,
                                                          double* __restrict__ out) {
    __shared__ double ldsA[QREC];
    const int wave = threadIdx.x >> 6;
    const QuadLane L = quad_lane();
    const int a = (int)(((long long)blockIdx.x * 97) % n_rec);
    const double* A = qr + (size_t)a * QREC;
    for (int e = threadIdx.x; e < QREC; e += blockDim.x) ldsA[e] = A[e];
    __syncthreads();
    const double nA = ldsA[QREC_COUNT_AT];
    double acc = 0.0;
    for (int base = 4 * wave; base < WINDOW; base += 4 * (TPB / 64)) {
        int w = base + L.m;
        w = w < WINDOW ? w : WINDOW - 1;
        const int rc = (a + 1 + w) % n_rec;
        const double* P = pk + (size_t)rc * REC;
        int ta = L.t; asm volatile("" : "+v"(ta));
        const int t12 = ta < QL ? ta : QL - 1;
        if (MODE == 0) {
            QuadRows q; double sv[QS];
            const double n = nA + P[REC - 1];
            const double f = 1.0 / (n - 1.0);
            const double* rt[2];
            long long o1 = 512; asm volatile("" : "+v"(o1));
            rt[0] = P + t12; rt[1] = rt[0] + o1;
#pragma unroll
            for (int s2 = 0; s2 < QS; ++s2) {
#pragma unroll
                for (int j = 0; j < tri_cols(s2); ++j) { const int e = pk_off(j) + QL * s2 - j; q.r[s2][j] = rt[e / 512][e % 512]; }
                const int c = QL * s2 + t12;
                sv[s2] = P[pk_off(c) + D - c];
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
            UnionColumns col;
            col.ldsA = ldsA; col.ta = ta;
            long long o1 = 512; asm volatile("" : "+v"(o1));
            col.rt0 = P + t12; col.rt1 = col.rt0 + o1;
            const double n = nA + P[REC - 1];
            const double f = 1.0 / (n - 1.0);
            col.nf = -f;
#pragma unroll
            for (int s2 = 0; s2 < QS; ++s2) {
                const int c = QL * s2 + t12;
                col.v1[s2] = ldsA[(s2 * DA + D) * 16 + ta] + P[pk_off(c) + D - c];
                col.c1[s2] = (f / n) * col.v1[s2];             // minus (-(f/n) v1): the accumulators run negated
            }
            double det; chol_det(col, det);
            acc += log(det);
        }
    }
    if (L.t == 0) out[(size_t)blockIdx.x * 64 + wave * 4 + L.m] = acc;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 4096;
    for (int n_rec : {387, 40000}) {
        std::vector<double> h((size_t)n_rec * QREC, 0.0);
        srand(2);
        std::vector<double> M(DA * DA);
        for (int r = 0; r < n_rec; ++r) {
            const int nf = 300 + (r % 7) * 100;
            if (r < 64) {
                std::fill(M.begin(), M.end(), 0.0);
                for (int f = 0; f < nf; ++f) {
                    double x[DA];
                    for (int i = 0; i < D; ++i) x[i] = (rand() / (double)RAND_MAX) - 0.5 + 0.01 * r;
                    x[D] = 1.0;
                    for (int i = 0; i < DA; ++i) for (int j = 0; j < DA; ++j) M[j * DA + i] += x[i] * x[j];
                }
                double* o = &h[(size_t)r * QREC];
                for (int i = 0; i < D; ++i) for (int j = 0; j < DA; ++j) o[qr_index(i, j)] = M[j * DA + i];
                o[QREC_COUNT_AT] = nf;
            } else {
                std::copy(&h[(size_t)(r % 64) * QREC], &h[(size_t)(r % 64 + 1) * QREC], &h[(size_t)r * QREC]);
            }
        }
        std::vector<double> hp((size_t)n_rec * REC, 0.0);
        for (int r = 0; r < n_rec; ++r) {
            const double* o = &h[(size_t)r * QREC];
            double* pr = &hp[(size_t)r * REC];
            for (int j = 0; j < DA; ++j) for (int i = j; i < DA; ++i) {
                double v;
                if (i < D) v = o[qr_index(i, j)];
                else if (j < D) v = o[qr_index(j, D)];
                else v = o[QREC_COUNT_AT];
                pr[pk_off(j) + (i - j)] = v;
            }
        }
        double *dE, *dO, *dO2, *dP;
        hipMalloc(&dP, hp.size() * 8); hipMemcpy(dP, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
        hipMalloc(&dE, h.size() * 8); hipMemcpy(dE, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipMalloc(&dO, (size_t)blocks * 64 * 8); hipMalloc(&dO2, (size_t)blocks * 64 * 8);
        hipMemset(dO, 0, (size_t)blocks * 64 * 8); hipMemset(dO2, 0, (size_t)blocks * 64 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 3; ++mode)
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((k_pairs<0, 512>), dim3(blocks), dim3(512), 0, 0, dP, dE, n_rec, dO);
                else if (mode == 1) hipLaunchKernelGGL((k_pairs<1, 512>), dim3(blocks), dim3(512), 0, 0, dP, dE, n_rec, dO2);
                else hipLaunchKernelGGL((k_pairs<1, 768>), dim3(blocks), dim3(768), 0, 0, dP, dE, n_rec, dO2);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it == 2) printf("n_rec %6d  %-46s %.3f ms  %.1f M pairs/s\n", n_rec,
                                    mode == 0 ? "right-looking, 2 waves/SIMD (library form)" : (mode == 1 ? "streaming Cholesky, 512 threads" : "streaming Cholesky, 768 threads (3 waves/SIMD)"), ms,
                                    (double)blocks * WINDOW / ms / 1e3);
            }
        // the two forms agree (relative difference of the summed log dets of the 512-thread runs)
        std::vector<double> o1((size_t)blocks * 64), o2((size_t)blocks * 64);
        hipLaunchKernelGGL((k_pairs<1, 512>), dim3(blocks), dim3(512), 0, 0, dP, dE, n_rec, dO2);
        hipMemcpy(o1.data(), dO, o1.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(o2.data(), dO2, o2.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0.0;
        for (size_t i = 0; i < o1.size(); ++i) if (o1[i] != 0.0) worst = std::max(worst, std::fabs(o1[i] - o2[i]) / std::fabs(o1[i]));
        printf("n_rec %6d  worst relative difference of the per-lane log-det sums: %.3g\n", n_rec, worst);
        hipFree(dP); hipFree(dE); hipFree(dO); hipFree(dO2);
    }
    return 0;
}
