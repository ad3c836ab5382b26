// BVH2 -> CWBVH.  Layout/intent: Caitlyn/cwbvh.h:11-411 and Shader/cwbvh.fs:355-446;
// semantics as corrected in SURVEY.md appendix C (the reference file does not compile
// and mis-handles exponents, rounding direction, slot assignment and node allocation).
#include "cwbvh.hpp"
#include "cwbvh_core.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace crt {
namespace {

using cw::Decision;
using cw::is_leaf;

struct Converter {
    const crt_flatnode* bvh2;
    size_t n2, n_slots;
    const int32_t* slot_to_orig;
    std::vector<Decision> dec;       // 7 per BVH2 node (cwbvh.h:66)
    std::vector<int32_t> nprims;     // triangles below each BVH2 node
    std::vector<uint8_t> slot_seen;
    CWBVH* out;
    uint32_t max_level = 0;
    bool root_is_leaf = false;

    Decision& D(int node, int i) { return dec[(size_t)node * 7 + i]; }

    // cwbvh.h:75-173, bottom-up.  BFS numbering puts children after parents, so a reverse
    // sweep visits every child before its parent and no recursion is needed.
    bool calculate_cost() {
        nprims.assign(n2, 0);
        dec.resize(n2 * 7);
        for (int64_t node = (int64_t)n2 - 1; node >= 0; --node) {
            const crt_flatnode& fn = bvh2[node];
            const float area = cw::half_area(fn);
            if (is_leaf(fn)) {
                const int np = (int)fn.bmax[3];
                const int start = link_of(fn.bmin[3]);
                if (np < 1 || np > 3) { out->error = "BVH2 leaf with more than 3 triangles cannot be encoded"; return false; }
                if (start < 0 || (size_t)(start + np) > n_slots) { out->error = "BVH2 leaf range outside the triangle array"; return false; }
                nprims[node] = np;
                cw::leaf_decisions(area, np, &D((int)node, 0));
                continue;
            }
            const int left = link_of(fn.bmin[3]), right = left + 1;
            if (left <= node || (size_t)right >= n2) { out->error = "BVH2 child link out of order"; return false; }
            const int np = nprims[left] + nprims[right];
            nprims[node] = np;
            cw::inner_decisions(area, np, &D(left, 0), &D(right, 0), &D((int)node, 0));
        }
        return true;
    }

    // cwbvh.h:294-411, with allocation and recursion AFTER the 8-slot loop.
    void collapse(int n2idx, uint32_t n8idx, uint32_t level) {
        max_level = std::max(max_level, level);
        int children[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        int count = 0;
        if (root_is_leaf && n8idx == 0) children[count++] = n2idx;
        else count = cw::get_children(bvh2, dec.data(), n2idx, children);
        cw::order_children(bvh2, n2idx, children, count);

        crt_node8 node;
        int n_inner = 0, n_tris = 0;
        cw::encode_node(bvh2, dec.data(), nprims.data(), n2idx, children, node, n_inner, n_tris);
        node.triangle_base_index = (uint32_t)out->tri_slots.size();
        node.child_base_index = (uint32_t)out->nodes.size();
        for (int slot = 0; slot < 8; ++slot) {          // cwbvh.h:274-292: leaf slots in slot order, left subtree first
            const int child = children[slot];
            if (child == -1 || D(child, 0).type != cw::LEAF) continue;
            int32_t slots[3];
            const int cnt = cw::collect_slots(bvh2, child, slots);
            for (int i = 0; i < cnt; ++i) {
                out->tri_slots.push_back(slots[i]);
                out->triangle_indices.push_back(slot_to_orig ? slot_to_orig[slots[i]] : slots[i]);
                slot_seen[slots[i]]++;
            }
        }
        out->nodes[n8idx] = node;
        out->nodes.resize(out->nodes.size() + n_inner);
        out->child_bvh2.resize(out->nodes.size() * 8, -1);
        for (int slot = 0; slot < 8; ++slot) out->child_bvh2[(size_t)n8idx * 8 + slot] = children[slot];
        uint32_t offset = 0;
        for (int slot = 0; slot < 8; ++slot)
            if (node.imask & (1u << slot)) collapse(children[slot], node.child_base_index + offset++, level + 1);
    }
};

}  // namespace

bool CWBVH::convert(const crt_flatnode* bvh2, size_t n_nodes, size_t n_slots, const int32_t* slot_to_orig) {
    nodes.clear();
    triangle_indices.clear();
    tri_slots.clear();
    child_bvh2.clear();
    depth = 0;
    error.clear();
    if (!bvh2 || n_nodes == 0) { error = "empty BVH2"; return false; }
    if (n_nodes >= (1u << 24)) { error = "BVH2 too large: float links are exact only below 2^24"; return false; }

    Converter c;
    c.bvh2 = bvh2;
    c.n2 = n_nodes;
    c.n_slots = n_slots;
    c.slot_to_orig = slot_to_orig;
    c.out = this;
    c.slot_seen.assign(n_slots, 0);
    if (!c.calculate_cost()) return false;
    c.root_is_leaf = (c.D(0, 0).type == cw::LEAF);

    nodes.reserve(n_nodes / 2 + 1);
    tri_slots.reserve(n_slots);
    triangle_indices.reserve(n_slots);
    nodes.emplace_back();                 // the root exists BEFORE collapse (cwbvh.h:68-72 allocates it after)
    c.collapse(0, 0, 1);
    depth = c.max_level;
    for (size_t i = 0; i < n_slots; ++i)
        if (c.slot_seen[i] != 1) { error = "BVH2 leaves do not cover the triangle array exactly once"; return false; }
    return true;
}

std::string validate_cwbvh(const crt_node8* nodes, size_t n_nodes, size_t n_tris, uint32_t max_depth,
                           uint32_t* depth_out) {
    if (!nodes || n_nodes == 0) return "no nodes";
    std::vector<uint8_t> node_seen(n_nodes, 0), tri_seen(n_tris, 0);
    struct Item { uint32_t node, level; };
    std::vector<Item> stack{{0, 1}};
    node_seen[0] = 1;
    uint32_t depth = 0;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        depth = std::max(depth, it.level);
        if (it.level > max_depth) return "CWBVH deeper than the traversal stack";
        const crt_node8& n = nodes[it.node];
        for (int k = 0; k < 3; ++k)
            if (n.e[k] == 0 || n.e[k] == 255) return "exponent byte out of range";
        uint32_t inner_rank = 0;
        for (int slot = 0; slot < 8; ++slot) {
            const uint8_t m = n.meta[slot];
            const bool inner_by_mask = (n.imask >> slot) & 1;
            if (m == 0) {
                if (inner_by_mask) return "imask set on an empty slot";
                continue;
            }
            const uint8_t* lo[3] = {n.qlo_x, n.qlo_y, n.qlo_z};
            const uint8_t* hi[3] = {n.qhi_x, n.qhi_y, n.qhi_z};
            for (int k = 0; k < 3; ++k)
                if (lo[k][slot] > hi[k][slot]) return "inverted quantised child box";
            const bool inner_by_meta = ((m & (m << 1)) & 0x10) != 0;   // cwbvh.fs:397
            if (inner_by_meta != inner_by_mask) return "imask and meta disagree";
            if (inner_by_meta) {
                if (m != (uint8_t)((24 + slot) | 0x20)) return "inner meta byte is not (24+slot)|0x20";
                const uint64_t child = (uint64_t)n.child_base_index + inner_rank++;
                if (child >= n_nodes) return "child index out of range";
                if (node_seen[child]) return "node referenced twice";
                node_seen[child] = 1;
                stack.push_back({(uint32_t)child, it.level + 1});
            } else {
                const uint32_t unary = m >> 5, off = m & 31;
                const uint32_t cnt = unary == 1 ? 1 : unary == 3 ? 2 : unary == 7 ? 3 : 0;
                if (!cnt) return "leaf meta count is not unary 1..3";
                if (off + cnt > 24) return "leaf triangle offset exceeds 24 bits of the hit mask";
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint64_t t = (uint64_t)n.triangle_base_index + off + j;
                    if (t >= n_tris) return "triangle index out of range";
                    if (tri_seen[t]) return "triangle referenced twice";
                    tri_seen[t] = 1;
                }
            }
        }
    }
    for (size_t i = 0; i < n_tris; ++i)
        if (!tri_seen[i]) return "triangle never referenced";
    if (depth_out) *depth_out = depth;
    return std::string();
}

}  // namespace crt
