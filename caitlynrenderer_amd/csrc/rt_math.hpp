// Device-side arithmetic shared by every kernel.  The operation ORDER of each helper is part
// of the parity contract with the CPU oracle (DESIGN.md "floating-point rules"); the library is
// built with -ffp-contract=off so nothing here is fused unless fmaf is written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crt {

struct vec3 { float x, y, z; };

__device__ __forceinline__ vec3 V3(float x, float y, float z) { return vec3{x, y, z}; }
__device__ __forceinline__ vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// The short forms of 1 / x and sqrt(x) below give the IEEE bits only because of THIS chip's 1-ulp v_rcp_f32 / v_rsq_f32 (checked on every
// float by tools/ubench/*_exhaustive.hip, run by the GPU suite): any other target takes the compiler's correctly rounded expansions.
#if defined(__gfx950__)
#define CRT_FAST_RCP 1
#else
#define CRT_FAST_RCP 0
#endif
// IEEE correctly rounded square root.  NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS the HIP
// headers map that to __ocml_native_sqrt_f32 (bare v_sqrt_f32, 1 ulp).  sqrtf lowers to llvm.sqrt.f32,
// which hipcc expands to v_sqrt_f32 + fma refinement under its default
// -fhip-fp32-correctly-rounded-divide-sqrt.
// For 2^-100 <= x <= 2^100 the reciprocal square root (1 ulp) and one Newton step in fma arithmetic give the same bits in five
// instructions (every such float checked: tools/ubench/sqrt_exhaustive.hip, candidate A: 0 differences); zero, denormals, infinities,
// negative numbers and NaN take sqrtf.
__device__ __forceinline__ float sqrt_ieee(float x) {
#if CRT_FAST_RCP
    if (__builtin_expect(x >= 0x1p-100f && x <= 0x1p100f, 1)) {
        const float y = __builtin_amdgcn_rsqf(x);
        const float s = x * y, h = 0.5f * y;
        const float r = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(r, h, s);
    }
#endif
    return __builtin_sqrtf(x);
}
// 1 / x, IEEE correctly rounded.  hipcc expands the division into v_div_scale x 2, v_rcp, five fma, v_div_fmas, v_div_fixup (~43 issue cycles:
// a sixth of a Moller-Trumbore test).  For 2^-100 <= |x| <= 2^100 the hardware reciprocal (1 ulp) and ONE Newton step in fma arithmetic
// give the same bits — checked on all 2^32 inputs, tools/ubench/rcp_exhaustive.hip: 0 differences — in 2 compares + 3 instructions; anything
// outside that range (denormal results, infinities, NaN) takes the division.
__device__ __forceinline__ float rcp_ieee(float x) {
#if CRT_FAST_RCP
    const float ax = __builtin_fabsf(x);
    if (__builtin_expect(ax >= 0x1p-100f && ax <= 0x1p100f, 1)) {
        const float r0 = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, r0, 1.0f);
        return __builtin_fmaf(r0, e, r0);
    }
#endif
    return __fdiv_rn(1.0f, x);
}
__device__ __forceinline__ float length(vec3 a) { return sqrt_ieee(dot(a, a)); }
// v * (1/sqrt(dot)): two correctly rounded operations then three multiplies (glm::normalize).
__device__ __forceinline__ vec3 normalize(vec3 a) {
    float inv = rcp_ieee(sqrt_ieee(dot(a, a)));
    return a * inv;
}

// ---- pinned sin/cos: Cody-Waite by pi in double, Taylor polynomial, one rounding to float.
// Same constants and the same sequence of operations as oracle/oracle.c orc_sin / orc_cos: three fused multiply-adds for the
// reduction, the two highest coefficients by a plain multiply and add, the rest of the Horner chain fused — 19 half-rate
// double-precision instructions per call instead of the 32 of the all mul + add form this replaced (+1.7 % on the Cornell
// frame, +0.5 .. 0.8 % on the 1 M-triangle frames, same floats on every fixture).  The fused form only pays with the
// coefficients as SGPR operands: left to itself hipcc emits v_fmac_f64 with every coefficient parked in a VGPR pair, and the
// kernel, at its 96-register budget, spilled 144 bytes per lane and ran 8 % SLOWER (profiles/r02_experiments.md §3).
// a * b + c as ONE v_fma_f64 whose constant operand sits in an SGPR pair (a VOP3 instruction may read one): written as inline
// assembly because hipcc, left to itself, picks v_fmac_f64 and parks every coefficient in a VGPR pair
__device__ __forceinline__ double fma_add_const(double a, double b, double c) {      // c: compile-time constant
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ double fma_mul_const(double a, double m, double c) {      // m: compile-time constant
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m), "v"(c));
    return r;
}
__device__ __forceinline__ double reduce_pi(double x, double& k) {
    k = __builtin_rint(x * 0x1.45f306dc9c883p-2);
    return fma_mul_const(k, -0x1.3198a2ep-68, fma_mul_const(k, -0x1.0b4611a6p-33, fma_mul_const(k, -0x1.921fb544p+1, x)));
}
__device__ __forceinline__ double sin_poly(double r) {
    double z = r * r;
    double p = z * (1.0 / 51090942171709440000.0) + (-1.0 / 121645100408832000.0);   // mul, add: no coefficient ever sits in a VGPR
    p = fma_add_const(p, z, 1.0 / 355687428096000.0);
    p = fma_add_const(p, z, -1.0 / 1307674368000.0);
    p = fma_add_const(p, z, 1.0 / 6227020800.0);
    p = fma_add_const(p, z, -1.0 / 39916800.0);
    p = fma_add_const(p, z, 1.0 / 362880.0);
    p = fma_add_const(p, z, -1.0 / 5040.0);
    p = fma_add_const(p, z, 1.0 / 120.0);
    p = fma_add_const(p, z, -1.0 / 6.0);
    return __builtin_fma(r * z, p, r);
}
__device__ __forceinline__ double cos_poly(double r) {
    double z = r * r;
    double p = z * (1.0 / 1124000727777607680000.0) + (-1.0 / 2432902008176640000.0);
    p = fma_add_const(p, z, 1.0 / 6402373705728000.0);
    p = fma_add_const(p, z, -1.0 / 20922789888000.0);
    p = fma_add_const(p, z, 1.0 / 87178291200.0);
    p = fma_add_const(p, z, -1.0 / 479001600.0);
    p = fma_add_const(p, z, 1.0 / 3628800.0);
    p = fma_add_const(p, z, -1.0 / 40320.0);
    p = fma_add_const(p, z, 1.0 / 720.0);
    p = fma_add_const(p, z, -1.0 / 24.0);
    p = fma_add_const(p, z, 0.5);
    return __builtin_fma(-z, p, 1.0);
}
// (Round 3 timed a FREE sine in its place — bare v_sin_f32, parity broken on purpose — to see what a cheaper definition could buy at
// most: 1 M triangles +2.7 %, 4 segments +0.8 %, Cornell +6.4 %.  Not worth redefining the oracle's RNG for.)
__device__ __forceinline__ float pinned_sin(float xf) {
    double x = (double)xf;
    if (!(__builtin_fabs(x) < 1e9)) return 0.0f;
    double k, r = reduce_pi(x, k);
    double s = sin_poly(r);
    if (((long long)k) & 1) s = -s;
    return (float)s;
}
__device__ __forceinline__ float pinned_cos(float xf) {
    double x = (double)xf;
    if (!(__builtin_fabs(x) < 1e9)) return 1.0f;
    double k, r = reduce_pi(x, k);
    double c = cos_poly(r);
    if (((long long)k) & 1) c = -c;
    return (float)c;
}

// Shader/path_trace.fs:38-42
__device__ __forceinline__ float shader_rand(float& sx, float& sy, float rv) {
    sx -= rv;
    sy -= rv;
    float d = sx * 12.9898f + sy * 78.233f;
    float v = pinned_sin(d) * 43758.5453f;
    return v - __builtin_floorf(v);
}

}  // namespace crt
