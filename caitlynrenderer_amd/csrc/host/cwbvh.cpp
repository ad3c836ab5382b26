// BVH2 -> CWBVH.  Layout/intent: Caitlyn/cwbvh.h:11-411 and Shader/cwbvh.fs:355-446;
// semantics as corrected in SURVEY.md appendix C (the reference file does not compile
// and mis-handles exponents, rounding direction, slot assignment and node allocation).
#include "cwbvh.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace crt {
namespace {

constexpr float kInf = 1e20f;   // sbvh.h:13, the `inf` cwbvh.h:101,103 compares against
enum : int8_t { LEAF = 0, INTERNAL = 1, DISTRIBUTE = 2 };   // cwbvh.h:39

struct Decision {   // cwbvh.h:41-49
    float cost;
    int8_t type, dl, dr;
};

inline bool is_leaf(const crt_flatnode& n) { return n.bmax[3] != 0.0f; }   // FlatNode.h:60-63
inline float half_area(const crt_flatnode& n) {                            // FlatNode.h:41-47
    float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
    return dx * (dy + dz) + dy * dz;
}
inline float3 centre(const crt_flatnode& n) {                              // FlatNode.h:56-59
    return float3(n.bmin[0] + n.bmax[0], n.bmin[1] + n.bmax[1], n.bmin[2] + n.bmax[2]) * 0.5f;
}

struct Converter {
    const crt_flatnode* bvh2;
    size_t n2, n_slots;
    const int32_t* slot_to_orig;
    std::vector<Decision> dec;       // 7 per BVH2 node (cwbvh.h:66)
    std::vector<uint8_t> slot_seen;
    CWBVH* out;
    uint32_t max_level = 0;
    bool root_is_leaf = false;

    Decision& D(int node, int i) { return dec[(size_t)node * 7 + i]; }

    // cwbvh.h:75-173, bottom-up.  BFS numbering puts children after parents, so a reverse
    // sweep visits every child before its parent and no recursion is needed.
    bool calculate_cost() {
        std::vector<int32_t> nprims(n2);
        dec.resize(n2 * 7);
        for (int64_t node = (int64_t)n2 - 1; node >= 0; --node) {
            const crt_flatnode& fn = bvh2[node];
            const float area = half_area(fn);
            if (is_leaf(fn)) {
                const int np = (int)fn.bmax[3];
                const int start = (int)fn.bmin[3];
                if (np < 1 || np > 3) { out->error = "BVH2 leaf with more than 3 triangles cannot be encoded"; return false; }
                if (start < 0 || (size_t)(start + np) > n_slots) { out->error = "BVH2 leaf range outside the triangle array"; return false; }
                nprims[node] = np;
                const float c = area * float(np);
                for (int i = 0; i < 7; ++i) D(node, i) = Decision{c, LEAF, -1, -1};
                continue;
            }
            const int left = (int)fn.bmin[3], right = left + 1;
            if (left <= node || (size_t)right >= n2) { out->error = "BVH2 child link out of order"; return false; }
            const int np = nprims[left] + nprims[right];
            nprims[node] = np;

            const float cost_leaf = np <= 3 ? float(np) * area : kInf;
            float cost_distribute = kInf;
            int8_t dl = -1, dr = -1;
            for (int k = 0; k < 7; ++k) {
                const float c = D(left, k).cost + D(right, 6 - k).cost;
                if (c < cost_distribute) { cost_distribute = c; dl = (int8_t)k; dr = (int8_t)(6 - k); }
            }
            const float cost_internal = cost_distribute + area;
            if (cost_leaf < cost_internal) D(node, 0) = Decision{cost_leaf, LEAF, dl, dr};
            else                           D(node, 0) = Decision{cost_internal, INTERNAL, dl, dr};

            for (int i = 1; i < 7; ++i) {
                float best = D(node, i - 1).cost;
                int8_t bl = -1, br = -1;
                for (int k = 0; k < i; ++k) {
                    const float c = D(left, k).cost + D(right, i - k - 1).cost;
                    if (c < best) { best = c; bl = (int8_t)k; br = (int8_t)(i - k - 1); }
                }
                if (bl != -1) D(node, i) = Decision{best, DISTRIBUTE, bl, br};
                else          D(node, i) = D(node, i - 1);
            }
        }
        return true;
    }

    // cwbvh.h:175-204
    void get_children(int node, int i, int children[8], int& count) {
        const crt_flatnode& fn = bvh2[node];
        if (is_leaf(fn)) { children[count++] = node; return; }
        const int dl = D(node, i).dl, dr = D(node, i).dr;
        const int left = (int)fn.bmin[3], right = left + 1;
        if (D(left, dl).type == DISTRIBUTE) get_children(left, dl, children, count);
        else children[count++] = left;
        if (D(right, dr).type == DISTRIBUTE) get_children(right, dr, children, count);
        else children[count++] = right;
    }

    // cwbvh.h:206-272 with `assignment[min_index] = min_slot` (the reference line is a no-op).
    void order_children(int node, int children[8], int count) {
        const float3 p = centre(bvh2[node]);
        float cost[8][8];
        for (int c = 0; c < count; ++c) {
            const float3 rel = centre(bvh2[children[c]]) - p;
            for (int s = 0; s < 8; ++s) {
                const float3 dir((s & 4) ? -1.0f : 1.0f, (s & 2) ? -1.0f : 1.0f, (s & 1) ? -1.0f : 1.0f);
                cost[c][s] = dot(rel, dir);
            }
        }
        int assignment[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        bool filled[8] = {};
        for (;;) {
            float min_cost = kInf;
            int min_slot = -1, min_index = -1;
            for (int c = 0; c < count; ++c) {
                if (assignment[c] != -1) continue;
                for (int s = 0; s < 8; ++s)
                    if (!filled[s] && cost[c][s] < min_cost) { min_cost = cost[c][s]; min_slot = s; min_index = c; }
            }
            if (min_slot == -1) break;
            filled[min_slot] = true;
            assignment[min_index] = min_slot;
        }
        int old[8];
        for (int i = 0; i < 8; ++i) { old[i] = children[i]; children[i] = -1; }
        for (int i = 0; i < count; ++i) children[assignment[i]] = old[i];
    }

    // cwbvh.h:274-292: append the subtree's leaf slots, left first.
    int collect(int node) {
        const crt_flatnode& fn = bvh2[node];
        if (is_leaf(fn)) {
            const int start = (int)fn.bmin[3], range = (int)fn.bmax[3];
            for (int i = 0; i < range; ++i) {
                out->tri_slots.push_back(start + i);
                out->triangle_indices.push_back(slot_to_orig ? slot_to_orig[start + i] : start + i);
                slot_seen[start + i]++;
            }
            return range;
        }
        const int left = (int)fn.bmin[3];
        return collect(left) + collect(left + 1);
    }

    // Biased exponent e with scale 2^(e-127) >= extent/255 (cwbvh.h:302-321, exponent taken
    // from the float's bits, extent clamped so flat nodes do not produce log2(0)).
    static uint8_t pick_exponent(float lo, float hi) {
        const float ext = hi - lo;
        const float v = fmax_(ext, 1e-30f) * (1.0f / 255.0f);
        uint32_t e = float_bits(std::exp2(std::ceil(std::log2(v)))) >> 23;
        if (e < 1) e = 1;
        // libm's log2f may land a hair low; make sure 255 steps really reach `hi`.
        while (e < 254 && (double)lo + 255.0 * (double)bits_float(e << 23) < (double)hi) ++e;
        return (uint8_t)e;
    }

    static uint8_t quant_lo(float c, float p, float scale) {
        float q = std::floor((c - p) * (1.0f / scale));
        int qi = q < 0.f ? 0 : q > 255.f ? 255 : (int)q;
        while (qi > 0 && (double)p + (double)qi * (double)scale > (double)c) --qi;   // fp32 subtraction may round up
        return (uint8_t)qi;
    }
    static uint8_t quant_hi(float c, float p, float scale) {
        float q = std::ceil((c - p) * (1.0f / scale));   // ceil, not floor (cwbvh.h:351-353 is not conservative)
        int qi = q < 0.f ? 0 : q > 255.f ? 255 : (int)q;
        while (qi < 255 && (double)p + (double)qi * (double)scale < (double)c) ++qi;
        return (uint8_t)qi;
    }

    // cwbvh.h:294-411, with allocation and recursion AFTER the 8-slot loop.
    void collapse(int n2idx, uint32_t n8idx, uint32_t level) {
        max_level = std::max(max_level, level);
        const crt_flatnode& fn = bvh2[n2idx];
        crt_node8 node;
        std::memset(&node, 0, sizeof node);
        node.p[0] = fn.bmin[0]; node.p[1] = fn.bmin[1]; node.p[2] = fn.bmin[2];
        float scale[3];
        for (int k = 0; k < 3; ++k) {
            node.e[k] = pick_exponent(fn.bmin[k], fn.bmax[k]);
            scale[k] = bits_float((uint32_t)node.e[k] << 23);
        }

        int children[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        int count = 0;
        if (root_is_leaf && n8idx == 0) children[count++] = n2idx;
        else get_children(n2idx, 0, children, count);
        order_children(n2idx, children, count);

        node.imask = 0;
        node.triangle_base_index = (uint32_t)out->tri_slots.size();
        node.child_base_index = (uint32_t)out->nodes.size();
        int n_inner = 0, n_tris = 0;
        uint8_t* qlo[3] = {node.qlo_x, node.qlo_y, node.qlo_z};
        uint8_t* qhi[3] = {node.qhi_x, node.qhi_y, node.qhi_z};

        for (int slot = 0; slot < 8; ++slot) {
            const int child = children[slot];
            if (child == -1) continue;
            const crt_flatnode& cn = bvh2[child];
            for (int k = 0; k < 3; ++k) {
                qlo[k][slot] = quant_lo(cn.bmin[k], node.p[k], scale[k]);
                qhi[k][slot] = quant_hi(cn.bmax[k], node.p[k], scale[k]);
            }
            if (D(child, 0).type == LEAF) {
                const int cnt = collect(child);          // 1..3
                uint8_t m = 0;
                for (int j = 0; j < cnt; ++j) m |= (uint8_t)(1u << (j + 5));   // unary count in bits 7..5
                m |= (uint8_t)n_tris;                                          // offset from triangle base, 0..23
                node.meta[slot] = m;
                n_tris += cnt;
            } else {
                node.meta[slot] = (uint8_t)((slot + 24) | 0x20);
                node.imask |= (uint8_t)(1u << slot);
                ++n_inner;
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
    c.root_is_leaf = (c.D(0, 0).type == LEAF);

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
