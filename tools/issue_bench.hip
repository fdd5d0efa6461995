// Instruction-issue micro-benchmarks for the elimination inner loop (gfx950):
// cycles per wave-instruction of the candidate forms of "a[i][j] += l_i * u_j with
// u_j broadcast inside a 16-lane row", at one and at two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o issue_bench issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 4000;
constexpr int U = 16;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int TEST>
__global__ void k_issue(double* out, unsigned long long* cycles, double seed) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = seed + i * 1e-9;
    __syncthreads();
    double acc[U], src[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { acc[i] = seed * (i + 1) + threadIdx.x; src[i] = 1e-9 * (i + 1) + seed; }
    double m = seed * 1e-3;
    // per-row broadcast address: 4 distinct 8-byte addresses per wave, one per 16-lane row
    const unsigned addr = (unsigned)(size_t)(&lds[0]) + ((threadIdx.x >> 4) & 3) * 320u;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        if constexpr (TEST == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(src[i]), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (TEST == 1) {
#define X(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(src[i]), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (TEST == 2) {
#define X(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(acc[i]) : "v"(src[i]));
            REP16(X)
#undef X
        } else if constexpr (TEST == 3) {
            float* a32 = reinterpret_cast<float*>(acc);
            float* s32 = reinterpret_cast<float*>(src);
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a32[i]) : "v"(s32[i]));
            REP16(X)
#undef X
        } else if constexpr (TEST == 4) {
#define X(i) asm volatile("s_nop 1\n\tv_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(src[i]), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (TEST == 5) {
            // 8 broadcast LDS reads feed 16 FMAs (two rows each): reads for the next
            // iteration are in flight during this iteration's FMAs
#define XR(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(src[i]) : "v"(addr), "n"((i) * 8));
#define XF(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(src[8 + ((i) & 7)]), "v"(m));
            XR(0) XR(1) XR(2) XR(3) XR(4) XR(5) XR(6) XR(7)
            REP16(XF)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef XF
#define XR2(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(src[8 + (i)]) : "v"(addr), "n"((i) * 8 + 64));
#define XF(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(src[(i) & 7]), "v"(m));
            XR2(0) XR2(1) XR2(2) XR2(3) XR2(4) XR2(5) XR2(6) XR2(7)
            REP16(XF)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef XF
#undef XR
#undef XR2
        } else if constexpr (TEST == 6) {
#define X(i) if ((i) & 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(src[i]), "v"(m)); \
             else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(src[i]), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (TEST == 7) {
#define X(i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(acc[i]) : "v"(src[i]), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (TEST == 8) {
#define X(i) asm volatile("v_add_f64 %0, %1, %2" : "=v"(acc[i]) : "v"(src[i]), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (TEST == 9) {
#define X(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(acc[i]) : "v"(src[i]));
            REP16(X)
#undef X
        } else if constexpr (TEST == 10) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(acc[i]) : "v"(addr), "n"((i) * 8));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (TEST == 11) {
            // 8 x ds_read_b128 (two doubles each)
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*reinterpret_cast<double2*>(&acc[2 * (i)])) : "v"(addr), "n"((i) * 16));
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (TEST == 12) {
            // v_fma_f64 with an SGPR-pair operand
            const double sm = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(m)),
                                               __builtin_amdgcn_readfirstlane(__double2loint(m)));
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(src[i]), "s"(sm));
            REP16(X)
#undef X
        } else if constexpr (TEST == 13) {
            // one 64-bit DPP move feeding three plain FMAs (slot-0 column shape)
#define X(i) if (((i) & 3) == 0) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(src[(i) + 1]) : "v"(src[i])); \
             else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(src[((i) & ~3) + 1]), "v"(m));
            REP16(X)
#undef X
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < U; ++i) s += acc[i] + src[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int TEST>
void run(const char* name, int per_iter) {
    for (int tpb = 256; tpb <= 512; tpb += 256) {
        const int blocks = 256;
        double* d_out; unsigned long long* d_cyc;
        hipMalloc(&d_out, (size_t)blocks * tpb * 8);
        hipMalloc(&d_cyc, (size_t)blocks * 8 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_issue<TEST>, dim3(blocks), dim3(tpb), 0, 0, d_out, d_cyc, 1.000001);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> c((size_t)blocks * (tpb / 64));
        hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; unsigned long long mx = 0;
        for (auto v : c) { sum += (double)v; mx = v > mx ? v : mx; }
        const double avg = sum / c.size();
        // s_memtime ticks = shader cycles here; per wave-instruction, and per SIMD (waves/SIMD = tpb/256)
        printf("%-44s waves/SIMD %d: %.2f cyc/instr/wave, %.2f cyc/instr/SIMD  (kernel %.3f ms, max/avg %.2f)\n",
               name, tpb / 256, avg / ((double)ITER * per_iter), avg / ((double)ITER * per_iter) / (tpb / 256), ms,
               (double)mx / avg);
        hipFree(d_out); hipFree(d_cyc);
    }
}

int main() {
    run<0>("v_fma_f64", 16);
    run<1>("v_fmac_f64_dpp row_newbcast", 16);
    run<2>("v_mov_b64_dpp row_newbcast", 16);
    run<3>("v_mov_b32_dpp row_newbcast", 16);
    run<4>("s_nop 1 + v_fma_f64 (per pair)", 16);
    run<5>("8 ds_read_b64 bcast + 16 v_fma_f64 (per instr of 24)", 48);
    run<6>("alternating v_fmac_f64_dpp / v_fma_f64", 16);
    run<7>("v_mul_f64", 16);
    run<8>("v_add_f64", 16);
    run<9>("v_rcp_f64", 16);
    run<10>("ds_read_b64 (4 addresses per wave)", 16);
    run<11>("ds_read_b128 (4 addresses per wave)", 8);
    run<12>("v_fma_f64 with SGPR operand", 16);
    run<13>("1 v_mov_b64_dpp + 3 v_fma_f64 (per instr)", 16);
    return 0;
}
