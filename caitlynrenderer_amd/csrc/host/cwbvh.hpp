// BVH2 -> compressed 8-wide BVH (CWBVH) conversion on the host.
//
// Interface mirrors the reference's `CWBVH::convert(SBVH&)` (Caitlyn/cwbvh.h:51-73); the
// node layout is the 80-byte `node8` (cwbvh.h:11-25) that Shader/cwbvh.fs:484-488 fetches.
// The reference implementation of this conversion is unfinished (SURVEY.md §8a lists the
// defects); what is built here is the corrected algorithm of SURVEY.md appendix C.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/crt.h"
#include "sbvh.hpp"

namespace crt {

struct CWBVH {
    std::vector<crt_node8> nodes;             // cwbvh.h:53
    std::vector<int32_t> triangle_indices;    // cwbvh.h:54: CWBVH triangle order -> original triangle id
    std::vector<int32_t> tri_slots;           // CWBVH triangle order -> BVH2 leaf slot
    std::vector<int32_t> child_bvh2;          // 8 per node8: the BVH2 node each slot stands for, -1 = empty
    uint32_t depth = 0;                       // levels of node8 (root alone = 1)
    std::string error;                        // non-empty when convert() refused the input

    // cwbvh.h:58 — convert(SBVH&)
    bool convert(const SBVH& bvh) {
        return convert(bvh.flat_nodes.data(), bvh.flat_nodes.size(), bvh.triangle_indices.size(),
                       bvh.triangle_indices.data());
    }
    // same from raw FlatNode arrays (what Scene::gpu_data uploads, Scene.h:1057-1062);
    // slot_to_orig may be null (then triangle_indices == tri_slots).
    bool convert(const crt_flatnode* bvh2, size_t n_nodes, size_t n_slots, const int32_t* slot_to_orig);
};

// Structural validator used by the tests and by crt_scene_create for caller-supplied
// bvh8 buffers: child boxes decode inside [0,255], imask/meta agree, child and triangle
// ranges are in bounds, every triangle slot is referenced exactly once, depth <= limit.
// Returns an empty string when the structure is sound.
std::string validate_cwbvh(const crt_node8* nodes, size_t n_nodes, size_t n_tris, uint32_t max_depth,
                           uint32_t* depth_out);

}  // namespace crt
