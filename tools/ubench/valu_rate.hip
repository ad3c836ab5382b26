// VALU issue-rate microbenchmark for gfx950: how many wave64 vector instructions per cycle does one SIMD sustain
// with 1, 2, 4, 8 resident waves, for the instruction kinds the traversal kernel is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {        // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 1) { // v_cvt_f32_ubyte0
                asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3\n"
                             "v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte0 %5, %5\n v_cvt_f32_ubyte0 %6, %6\n v_cvt_f32_ubyte0 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 2) { // v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                             "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 3) { // v_pk_fma_f32 on register pairs (4 instructions = 8 fmas)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&b), "v"(*(const double*)&c));
            } else if (KIND == 4) { // v_fma_f64 (4 instructions)
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&b), "v"(*(const double*)&c));
            } else if (KIND == 5) { // v_cndmask with vcc + v_cmp (pairs)
                asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_le_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             "v_cmp_le_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_le_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 6) { // dependent chain of v_fma_f32 (latency)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(b), "v"(c));
            } else if (KIND == 7) { // v_rcp_f32 (transcendental)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, int instr_per_group, float* d_out, int n_simd) {
    const int iters = 20000;
    for (int waves : {1, 2, 4, 8}) {
        const int blocks = n_simd * waves;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 100, 1.0f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)blocks * iters * 8 * instr_per_group;      // wave-instructions
        const double per_simd_per_us = instr / n_simd / (ms * 1e3);
        printf("%-22s waves/SIMD %d: %8.3f ms, %7.1f wave-instr per SIMD per us  (at 2.4 GHz: %.2f cycles per instr per SIMD)\n", name, waves, ms,
               per_simd_per_us, 2400.0 / per_simd_per_us);
    }
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int n_simd = p.multiProcessorCount * 4;
    printf("%s: %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    float* d_out;
    CHK(hipMalloc(&d_out, (size_t)n_simd * 8 * 64 * sizeof(float)));
    run<0>("v_fma_f32", 8, d_out, n_simd);
    run<1>("v_cvt_f32_ubyte0", 8, d_out, n_simd);
    run<2>("v_max3_f32", 8, d_out, n_simd);
    run<3>("v_pk_fma_f32", 4, d_out, n_simd);
    run<4>("v_fma_f64", 4, d_out, n_simd);
    run<5>("v_cmp+v_cndmask", 8, d_out, n_simd);
    run<6>("v_fma_f32 dependent", 8, d_out, n_simd);
    run<7>("v_rcp_f32", 8, d_out, n_simd);
    return 0;
}
