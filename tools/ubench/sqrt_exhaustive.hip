// Exhaustive check of short correctly-rounded square roots against the one the kernels use (sqrtf -> hipcc's v_sqrt_f32 + refinement under
// -fhip-fp32-correctly-rounded-divide-sqrt): every float in [2^-100, 2^100].  Candidates:
//   A: y = v_rsq_f32(x); s = x * y; h = 0.5 * y; r = fma(-s, s, x); s = fma(r, h, s)
//   B: s = v_sqrt_f32(x); r = fma(-s, s, x); h = 0.5 * v_rcp_f32(s); s = fma(r, h, s)
//   C: A and one more step: r = fma(-s, s, x); s = fma(r, h, s)
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o sqrt_exhaustive sqrt_exhaustive.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ float cand_a(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    const float h = 0.5f * y;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float cand_b(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float r = __builtin_fmaf(-s, s, x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(s);
    return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float cand_c(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    const float h = 0.5f * y;
    float r = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(r, h, s);
    r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
__global__ void k_check(unsigned long long* counts, uint32_t* samples) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += stride) {      // positive floats
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        if (!(x >= 0x1p-100f && x <= 0x1p100f)) continue;
        const uint32_t want = __float_as_uint(__builtin_sqrtf(x));
        const uint32_t got[3] = {__float_as_uint(cand_a(x)), __float_as_uint(cand_b(x)), __float_as_uint(cand_c(x))};
        for (int k = 0; k < 3; ++k)
            if (got[k] != want) { const unsigned long long n = atomicAdd(&counts[k], 1ull); if (n < 6) samples[6 * k + n] = bits; }
    }
}
int main() {
    unsigned long long* d_counts; uint32_t* d_samples;
    if (hipMalloc(&d_counts, 3 * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&d_samples, 18 * sizeof(uint32_t)) != hipSuccess) return 1;
    (void)hipMemset(d_counts, 0, 3 * sizeof(unsigned long long)); (void)hipMemset(d_samples, 0, 18 * sizeof(uint32_t));
    hipLaunchKernelGGL(k_check, dim3(256 * 64), dim3(256), 0, 0, d_counts, d_samples);
    unsigned long long c[3]; uint32_t s[18];
    if (hipMemcpy(c, d_counts, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    (void)hipMemcpy(s, d_samples, sizeof s, hipMemcpyDeviceToHost);
    printf("floats in [2^-100, 2^100]: candidate A differs on %llu, B on %llu, C on %llu\n", c[0], c[1], c[2]);
    for (int k = 0; k < 3; ++k) { printf("  %c:", 'A' + k); for (int j = 0; j < 6; ++j) if (s[6 * k + j]) printf(" 0x%08x", s[6 * k + j]); printf("\n"); }
    return 0;
}
