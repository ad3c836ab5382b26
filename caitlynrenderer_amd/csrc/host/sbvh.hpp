// Host-side split-BVH builder (the tree the traversal kernels walk, via CWBVH).
//
// Mirrors the interface and the RESULT of the reference's `SBVH(trs, vertices)`
// (Caitlyn/sbvh.h:99-153): same split decisions, same leaf order, same BFS node
// numbering, so a tree built here is interchangeable with one built there.  The
// implementation is index-based (no pointer nodes) and written from the algorithm's
// description; see DESIGN.md §host builders.
#pragma once
#include <cstdint>
#include <vector>

#include "../../../include/crt.h"
#include "vecmath.hpp"

namespace crt {

struct Aabb {
    float3 lo{1e20f, 1e20f, 1e20f}, hi{-1e20f, -1e20f, -1e20f};   // BBox.h:22
    void grow(const float3& p) {
        lo = {fmin_(lo.x, p.x), fmin_(lo.y, p.y), fmin_(lo.z, p.z)};
        hi = {fmax_(hi.x, p.x), fmax_(hi.y, p.y), fmax_(hi.z, p.z)};
    }
    void grow(const Aabb& b) {
        lo = {fmin_(lo.x, b.lo.x), fmin_(lo.y, b.lo.y), fmin_(lo.z, b.lo.z)};
        hi = {fmax_(hi.x, b.hi.x), fmax_(hi.y, b.hi.y), fmax_(hi.z, b.hi.z)};
    }
    void clip(const Aabb& b) {   // BBox.h:49-53 intersect
        lo = {fmax_(lo.x, b.lo.x), fmax_(lo.y, b.lo.y), fmax_(lo.z, b.lo.z)};
        hi = {fmin_(hi.x, b.hi.x), fmin_(hi.y, b.hi.y), fmin_(hi.z, b.hi.z)};
    }
    float3 centre() const { return (lo + hi) * 0.5f; }   // BBox.h:25
    // "area" is the half surface area, unclamped (BBox.h:70-75); an inverted box can
    // give a positive value and the builder relies on exactly that expression.
    float half_area() const {
        float3 e = hi - lo;
        return (e.x * e.y + e.y * e.z) + e.z * e.x;
    }
};

// SBVH(trs, vertices), Caitlyn/sbvh.h:81-153.
struct SBVH {
    enum : uint32_t { NO_SPATIAL_SPLITS = 1u };

    std::vector<crt_flatnode> flat_nodes;        // sbvh.h:89, BFS order (sbvh.h:570-609)
    std::vector<int32_t> triangle_indices;       // sbvh.h:85, leaf slot -> original triangle
    std::vector<crt_triangle> triangles;         // the re-ordered `trs` (sbvh.h:130-139)
    int depth = 0;                               // deepest leaf level (root = 0)

    SBVH() = default;
    SBVH(const std::vector<crt_triangle>& trs, const std::vector<float3>& vertices, uint32_t flags = 0) {
        build(trs.data(), trs.size(), vertices.data(), vertices.size(), flags);
    }
    void build(const crt_triangle* trs, size_t n_trs, const float3* vertices, size_t n_vertices, uint32_t flags);
    int count_leaf() const;                      // sbvh.h:207-216 (without the prints)
};

}  // namespace crt
