// Minimal fp32 vector helpers for the host library.
//
// The reference's host code uses glm (un-vendored, version unpinned; call sites all over
// Caitlyn/sbvh.h and Caitlyn/BBox.h).  Only plain component-wise fp32 arithmetic is used
// there, so these helpers define every operation explicitly, left to right, and the
// library is compiled with -ffp-contract=off so that no multiply-add is ever fused.
#pragma once
#include <cmath>
#include <cstdint>

namespace crt {

struct float3 {
    float x, y, z;
    float3() : x(0.f), y(0.f), z(0.f) {}
    explicit float3(float s) : x(s), y(s), z(s) {}
    float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};

inline float3 operator+(const float3& a, const float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(const float3& a, const float3& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator*(const float3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, const float3& a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator-(const float3& a) { return {-a.x, -a.y, -a.z}; }
inline float3& operator+=(float3& a, const float3& b) { a = a + b; return a; }

// dot/cross/length/normalize: evaluation order fixed here and mirrored by the kernels
// (DESIGN.md "floating-point rules").
inline float dot(const float3& a, const float3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float3 cross(const float3& a, const float3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(const float3& a) { return std::sqrt(dot(a, a)); }
// glm::normalize is v * inversesqrt(dot(v,v)) with inversesqrt = 1/sqrt: two correctly
// rounded operations, then three multiplies.  The kernels use the same form.
inline float3 normalize(const float3& a) {
    float inv = 1.0f / std::sqrt(dot(a, a));
    return {a.x * inv, a.y * inv, a.z * inv};
}

inline float fmin_(float a, float b) { return a < b ? a : b; }   // BBox.h:13-16 minf
inline float fmax_(float a, float b) { return a > b ? a : b; }   // BBox.h:8-11 maxf

inline uint32_t float_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
inline float bits_float(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

}  // namespace crt
