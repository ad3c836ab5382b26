// The link word of a FlatNode (FlatNode.h:34-40: box_min.w = left child / first triangle slot).  The reference stores it as a FLOAT whose value
// is the index (uploaded as RGBA32F, Scene.h:1057-1062), exact below 2^24 — which capped a scene at 2^23 triangles.  Arrays that never leave
// the device (crt_scene_create with CRT_BUILD_LBVH_ON_DEVICE) may hold more: an index of 2^24 or more is written as its BIT PATTERN instead
// (a float below 1.0 for every index below 0x3f800000 = 1,065,353,216), and a reader tells the two forms apart by the value: integers
// written as floats are >= 1.0 (a link never points at node 0; slot 0 is 0.0f in both forms).  Arrays handed in or out through the host
// entry points keep the reference's float form (their builders refuse 2^23 triangles or more).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define CRT_LINK_HD __host__ __device__ inline
#else
#define CRT_LINK_HD inline
#endif
namespace crt {
constexpr uint32_t kMaxLinkBits = 0x3f800000u;        // indices from here on would read as floats >= 1.0
CRT_LINK_HD float link_enc(uint32_t index) {
    if (index < (1u << 24)) return (float)index;
    float f; __builtin_memcpy(&f, &index, 4); return f;
}
CRT_LINK_HD int link_of(float w) {
    if (w >= 1.0f) return (int)w;
    int i; __builtin_memcpy(&i, &w, 4); return i;
}
}  // namespace crt
