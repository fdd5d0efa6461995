// Prototype for the next step of DESIGN.md §9: can the elimination run at THREE waves per
// SIMD?  The right-looking symmetric elimination holds the whole lower triangle (78 doubles
// per lane, 156 VGPRs) from its first step, which allows two.  Here the 13 x 13 block of
// slot 0 is factored on its own first (13 doubles per lane), its multiplier sources and
// pivot reciprocals go to LDS (2.9 KB per wave), its registers are given back, and only
// then the other two slots (65 doubles per lane, 130 VGPRs) are loaded: the first thirteen
// steps update them with LDS-broadcast operands for the panel columns and DPP operands for
// the trailing ones, steps 13..38 are the library's code unchanged.
//   mode 0: the library's elimination, two blocks of four waves per CU (bounds 256, 2)
//   mode 1: the split form, three blocks of four waves per CU (bounds 256, 3)
// BIC union determinants of one row record (LDS) with partner quad records (global), as in
// pair_bench.hip; the two modes must give the same determinants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../speaker-diarization_amd/csrc -o tri3_bench tri3_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "spkd_cluster.hpp"
#include "spkd_tri.hpp"
using namespace spkd;

constexpr int WINDOW = 384;
constexpr int WPB = 4;                       // waves per block

// ---- the split elimination --------------------------------------------------------------
struct Split3 {
    double* m0;          // [4][13][16]: (matrix, step k, lane J) = a[J][k] of slot 0, final
    double* rk;          // [4][16]: reciprocal pivots of steps 0..12
};

// rank-one term c v^T on the slot-0 block / on the rows of slots 1 and 2
template <int J>
__device__ __forceinline__ void rank1_head_cols(double (&b0)[QL], double c0, double v0) {
    if constexpr (J < QL) {
        fmac_bcast16<J, J == 0>(b0[J], v0, c0);
        rank1_head_cols<J + 1>(b0, c0, v0);
    }
}
__device__ __forceinline__ void rank1_head(double (&b0)[QL], double c0, double v0) { rank1_head_cols<0>(b0, c0, v0); }

template <int J>
__device__ __forceinline__ void rank1_body_cols(QuadRows& q, const double (&c)[QS], const double (&v)[QS]) {
    if constexpr (J < D) {
        constexpr int SJ = J / QL, TJ = J % QL;
        if constexpr (SJ < 2) fmac_bcast16<TJ, TJ == 0>(q.r[1][J], v[SJ], c[1]);
        fmac_bcast16<TJ, SJ == 2 && TJ == 0>(q.r[2][J], v[SJ], c[2]);
        rank1_body_cols<J + 1>(q, c, v);
    }
}
__device__ __forceinline__ void rank1_body(QuadRows& q, const double (&c)[QS], const double (&v)[QS]) { rank1_body_cols<0>(q, c, v); }

template <int K, int J>
__device__ __forceinline__ void head_cols(double (&b0)[QL], double l) {
    if constexpr (J < QL) {
        fmac_bcast16<J, false>(b0[J], b0[K], l);
        head_cols<K, J + 1>(b0, l);
    }
}

// slot 0 alone: steps 0..12 on the 13 x 13 block (row t of the DPP row in lane t)
// this lane's number, computed where it is wanted (a value kept for the whole pass is spilled:
// the registers are full during steps 0..38 of slots 1 and 2)
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <int K>
__device__ __forceinline__ void head_step(double (&b0)[QL], DetAcc& da, double* m0_wave, double* rk_wave) {
    if constexpr (K < QL) {
        double piv;
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(piv) : "v"(b0[K]), "n"(K));
        da.det *= piv;
        da.sign |= __double2hiint(piv);
        asm volatile("" : "+v"(da.det), "+v"(da.sign));      // (the product is taken here: deferred, the pivots are spilled)
        const int lane = fresh_lane();
        double* m0_lane = m0_wave + (lane >> 4) * (QL * 16) + (lane & 15);
        double* rk_row = rk_wave + (lane >> 4) * 16;
        const bool first = (lane & 15) == 0;
        double r = __builtin_amdgcn_rcp(piv);
        double e = fma(-piv, r, 1.0);
        e = fma(e, e, e);
        r = fma(r, e, r);
        m0_lane[K * 16] = b0[K];                             // a[t][K]: the broadcast source of the panel columns
        if (first) rk_row[K] = r;
        const double l = -b0[K] * r;
        __builtin_amdgcn_sched_barrier(0);
        head_cols<K, K + 1>(b0, l);
        __builtin_amdgcn_sched_barrier(0);
        head_step<K + 1>(b0, da, m0_wave, rk_wave);
    }
}

template <int K, int J>
__device__ __forceinline__ void body_cols(QuadRows& q, double l1, double l2, const double* m0k) {
    if constexpr (J < D) {
        if constexpr (J < QL) {                              // panel column: operand from LDS (same for the whole row)
            const double x = m0k[J];
            q.r[1][J] = fma(l1, x, q.r[1][J]);
            q.r[2][J] = fma(l2, x, q.r[2][J]);
        } else if constexpr (J < 2 * QL) {                   // row J lives in slot 1, lane J - 13
            fmac_bcast16<J - QL, false>(q.r[1][J], q.r[1][K], l1);
            fmac_bcast16<J - QL, false>(q.r[2][J], q.r[1][K], l2);
        } else {                                             // row J lives in slot 2
            fmac_bcast16<J - 2 * QL, false>(q.r[2][J], q.r[2][K], l2);
        }
        body_cols<K, J + 1>(q, l1, l2, m0k);
    }
}

// steps 0..12 applied to slots 1 and 2 (their rows' columns 0..12 and the trailing columns)
template <int K>
__device__ __forceinline__ void body_step(QuadRows& q, const double* m0_wave, const double* rk_wave) {
    if constexpr (K < QL) {
        const int lane = fresh_lane();
        const double* m0_row = m0_wave + (lane >> 4) * (QL * 16);
        const double* rk_row = rk_wave + (lane >> 4) * 16;
        const double r = rk_row[K];
        const double l1 = -q.r[1][K] * r, l2 = -q.r[2][K] * r;
        const double* m0k = m0_row + K * 16;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1");                             // (the column-K registers were FMA results)
        body_cols<K, K + 1>(q, l1, l2, m0k);
        __builtin_amdgcn_sched_barrier(0);
        body_step<K + 1>(q, m0_wave, rk_wave);
    }
}

template <int MODE>
__global__ __launch_bounds__(WPB * 64, MODE == 0 ? 2 : 3) void k_tri(const double* __restrict__ qr, int n_rec,
                                                                     double* __restrict__ out) {
    __shared__ double ldsA[QREC];
    __shared__ double s_m0[WPB][4 * QL * 16];
    __shared__ double s_rk[WPB][4 * 16];
    const int wave = threadIdx.x >> 6;
    const QuadLane L = quad_lane();
    const int a = (int)(((long long)blockIdx.x * 97) % n_rec);
    const double* A = qr + (size_t)a * QREC;
    for (int e = threadIdx.x; e < QREC; e += blockDim.x) ldsA[e] = A[e];
    __syncthreads();
    const double nA = ldsA[QREC_COUNT_AT];
    Split3 S;
    S.m0 = s_m0[wave];
    S.rk = s_rk[wave];
    for (int base = 4 * wave; base < WINDOW; base += 4 * WPB) {
        const int w = base + L.m;
        const int rc = (a + 1 + w) % n_rec;
        const double* C = qr + (size_t)rc * QREC;
        int ta = L.t; asm volatile("" : "+v"(ta));
        const double n = nA + C[QREC_COUNT_AT];
        const double f = 1.0 / (n - 1.0);
        const double g = -(f / n);
        const double* Ct = C + L.t;
        double det;
        if constexpr (MODE == 0) {
            QuadRows q; double sv[QS], c1[QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) {
#pragma unroll
                for (int j = 0; j < tri_cols(s); ++j) q.r[s][j] = Ct[(s * DA + j) * 16];
                sv[s] = Ct[(s * DA + D) * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < QS; ++s) {
#pragma unroll
                for (int j = 0; j < tri_cols(s); ++j) q.r[s][j] = f * (ldsA[(s * DA + j) * 16 + ta] + q.r[s][j]);
                sv[s] = ldsA[(s * DA + D) * 16 + ta] + sv[s];
                c1[s] = g * sv[s];
                __builtin_amdgcn_sched_barrier(0);
            }
            TriRank1<0>::run(q, c1, sv);
            tri_det_nopivot(q, det);
        } else {
            // ---- slot 0 on its own
            double b0[QL], sv[QS], c1[QS];
#pragma unroll
            for (int j = 0; j < QL; ++j) b0[j] = Ct[j * 16];
#pragma unroll
            for (int s = 0; s < QS; ++s) sv[s] = Ct[(s * DA + D) * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < QL; ++j) b0[j] = f * (ldsA[j * 16 + ta] + b0[j]);
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                sv[s] = ldsA[(s * DA + D) * 16 + ta] + sv[s];
                c1[s] = g * sv[s];
            }
            rank1_head(b0, c1[0], sv[0]);
            DetAcc da;
            da.det = 1.0; da.sign = 0;
            head_step<0>(b0, da, S.m0, S.rk);
            // ---- slots 1 and 2
            QuadRows q;
#pragma unroll
            for (int s = 1; s < QS; ++s) {
#pragma unroll
                for (int j = 0; j < tri_cols(s); ++j) q.r[s][j] = Ct[(s * DA + j) * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 1; s < QS; ++s) {
#pragma unroll
                for (int j = 0; j < tri_cols(s); ++j) q.r[s][j] = f * (ldsA[(s * DA + j) * 16 + ta] + q.r[s][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            rank1_body(q, c1, sv);
            body_step<0>(q, S.m0, S.rk);
            // ---- steps 13..38: the library's code on what is left
            PivotChain ch;
            double l13[QS];
            __builtin_amdgcn_sched_barrier(0);
            chain_rest<QL, 0>(q, ch, l13, da, 1.0);
            TriStepAhead<QL>::run(q, da, l13, ch.piv);
            det = da.det;
        }
        if (L.t == 0) out[(size_t)blockIdx.x * WINDOW + w] = det;
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 6144;
    for (int n_rec : {400, 40000}) {
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
        double *dE, *dO0, *dO1;
        hipMalloc(&dE, h.size() * 8);
        hipMalloc(&dO0, (size_t)blocks * WINDOW * 8);
        hipMalloc(&dO1, (size_t)blocks * WINDOW * 8);
        hipMemcpy(dE, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 2; ++mode)
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_tri<0>, dim3(blocks), dim3(WPB * 64), 0, 0, dE, n_rec, dO0);
                else hipLaunchKernelGGL(k_tri<1>, dim3(blocks), dim3(WPB * 64), 0, 0, dE, n_rec, dO1);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it == 2) printf("n_rec %6d  %-44s %.3f ms  %.1f M determinants/s\n", n_rec,
                                    mode == 0 ? "library elimination, 2 waves per SIMD" : "slot 0 split off, 3 waves per SIMD", ms,
                                    (double)blocks * WINDOW / ms / 1e3);
            }
        std::vector<double> o0((size_t)blocks * WINDOW), o1((size_t)blocks * WINDOW);
        hipMemcpy(o0.data(), dO0, o0.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(o1.data(), dO1, o1.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0; long long bad = 0;
        for (size_t i = 0; i < o0.size(); ++i) {
            const double rel = std::fabs(o0[i] - o1[i]) / std::fabs(o0[i]);
            if (!(rel < 1e-9)) ++bad;
            if (rel > worst) worst = rel;
        }
        printf("n_rec %6d  determinants of the two forms: worst relative difference %.3g, %lld beyond 1e-9 (sample %.6g vs %.6g)\n",
               n_rec, worst, bad, o0[5], o1[5]);
        hipFree(dE); hipFree(dO0); hipFree(dO1);
    }
    return 0;
}
