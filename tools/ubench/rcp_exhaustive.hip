// Exhaustive check of a short correctly-rounded reciprocal against the IEEE division the kernels use (__fdiv_rn(1.0f, x), i.e. hipcc's
// v_div_scale / v_rcp / fma chain / v_div_fmas / v_div_fixup): all 2^32 bit patterns.  Candidates:
//   A: r0 = v_rcp_f32(x); e = fma(-x, r0, 1); r = fma(r0, e, r0)
//   B: A, then once more: e = fma(-x, r, 1); r = fma(r, e, r)
// Prints, per candidate, the number of inputs inside [2^-100, 2^100] (by magnitude) whose result differs, and a few of them.
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o rcp_exhaustive rcp_exhaustive.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ float cand_a(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(r0, e, r0);
}
__device__ __forceinline__ float cand_b(float x) {
    float r = cand_a(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}
__global__ void k_check(unsigned long long* counts, uint32_t* samples) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ax = __builtin_fabsf(x);
        if (!(ax >= 0x1p-100f && ax <= 0x1p100f)) continue;
        const uint32_t want = __float_as_uint(__fdiv_rn(1.0f, x));
        const uint32_t a = __float_as_uint(cand_a(x)), b = __float_as_uint(cand_b(x));
        if (a != want) { const unsigned long long n = atomicAdd(&counts[0], 1ull); if (n < 8) samples[n] = bits; }
        if (b != want) { const unsigned long long n = atomicAdd(&counts[1], 1ull); if (n < 8) samples[8 + n] = bits; }
        atomicAdd(&counts[2], 0ull);
    }
}
int main() {
    unsigned long long* d_counts; uint32_t* d_samples;
    hipMalloc(&d_counts, 3 * sizeof(unsigned long long)); hipMalloc(&d_samples, 16 * sizeof(uint32_t));
    hipMemset(d_counts, 0, 3 * sizeof(unsigned long long)); hipMemset(d_samples, 0, 16 * sizeof(uint32_t));
    hipLaunchKernelGGL(k_check, dim3(256 * 64), dim3(256), 0, 0, d_counts, d_samples);
    unsigned long long c[3]; uint32_t s[16];
    if (hipMemcpy(c, d_counts, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    hipMemcpy(s, d_samples, sizeof s, hipMemcpyDeviceToHost);
    printf("inputs with 2^-100 <= |x| <= 2^100: candidate A differs on %llu, candidate B on %llu\n", c[0], c[1]);
    for (int k = 0; k < 2; ++k) { printf("  %c:", 'A' + k); for (int j = 0; j < 8; ++j) if (s[8 * k + j]) printf(" 0x%08x", s[8 * k + j]); printf("\n"); }
    return 0;
}
