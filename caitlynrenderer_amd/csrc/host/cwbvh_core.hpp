// The per-node pieces of the BVH2 -> CWBVH conversion, written once and compiled for the host converter
// (host/cwbvh.cpp, g++) and for the device converter (cwbvh_device.hip, gfx950), so that both produce the same
// bytes.  Layout/intent: Caitlyn/cwbvh.h:11-411; semantics as corrected in SURVEY.md appendix C.
// Everything here is plain fp32/fp64 arithmetic with a fixed operation order (both translation units are built
// with -ffp-contract=off) and no libm calls whose results could differ between host and device.
#pragma once
#include <stdint.h>

#include "../../../include/crt.h"
#include "flatnode_link.hpp"

#if defined(__HIPCC__)
#define CRT_HD __host__ __device__ inline
#else
#define CRT_HD inline
#endif

namespace crt {
namespace cw {

constexpr float kInf = 1e20f;   // sbvh.h:13, the `inf` cwbvh.h:101,103 compares against
enum : int8_t { LEAF = 0, INTERNAL = 1, DISTRIBUTE = 2 };   // cwbvh.h:39

struct Decision {   // cwbvh.h:41-49
    float cost;
    int8_t type, dl, dr;
};

CRT_HD uint32_t f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
CRT_HD float u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

CRT_HD bool is_leaf(const crt_flatnode& n) { return n.bmax[3] != 0.0f; }   // FlatNode.h:60-63
CRT_HD float half_area(const crt_flatnode& n) {                            // FlatNode.h:41-47
    const float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
    return dx * (dy + dz) + dy * dz;
}

// cwbvh.h:84-110: a BVH2 leaf costs area * nprims whatever the root budget
CRT_HD void leaf_decisions(float area, int np, Decision out[7]) {
    const float c = area * float(np);
    for (int i = 0; i < 7; ++i) out[i] = Decision{c, LEAF, -1, -1};
}

// cwbvh.h:111-173: out[i] = cheapest way to represent the subtree with at most i+1 roots
CRT_HD void inner_decisions(float area, int np, const Decision* L, const Decision* R, Decision out[7]) {
    const float cost_leaf = np <= 3 ? float(np) * area : kInf;
    float cost_distribute = kInf;
    int8_t dl = -1, dr = -1;
    for (int k = 0; k < 7; ++k) {
        const float c = L[k].cost + R[6 - k].cost;
        if (c < cost_distribute) { cost_distribute = c; dl = (int8_t)k; dr = (int8_t)(6 - k); }
    }
    const float cost_internal = cost_distribute + area;
    if (cost_leaf < cost_internal) out[0] = Decision{cost_leaf, LEAF, dl, dr};
    else                           out[0] = Decision{cost_internal, INTERNAL, dl, dr};
    for (int i = 1; i < 7; ++i) {
        float best = out[i - 1].cost;
        int8_t bl = -1, br = -1;
        for (int k = 0; k < i; ++k) {
            const float c = L[k].cost + R[i - k - 1].cost;
            if (c < best) { best = c; bl = (int8_t)k; br = (int8_t)(i - k - 1); }
        }
        if (bl != -1) out[i] = Decision{best, DISTRIBUTE, bl, br};
        else          out[i] = out[i - 1];
    }
}

// cwbvh.h:175-204: the (at most 8) BVH2 nodes that become the slots of the node8 rooted at `node`, in the
// reference's recursion order (left subtree first); iterative, at most 7 pending entries.
CRT_HD int get_children(const crt_flatnode* bvh2, const Decision* dec, int node, int children[8]) {
    int count = 0;
    if (is_leaf(bvh2[node])) { children[count++] = node; return count; }
    int st_node[8], st_i[8], sp = 0;
    st_node[sp] = node; st_i[sp] = 0; ++sp;
    while (sp > 0) {
        --sp;
        const int n = st_node[sp], i = st_i[sp];
        if (i < 0) { children[count++] = n; continue; }          // a finished child, emitted in order
        const crt_flatnode& fn = bvh2[n];
        if (is_leaf(fn)) { children[count++] = n; continue; }
        const Decision& d = dec[(size_t)n * 7 + i];
        const int left = link_of(fn.bmin[3]), right = left + 1;
        // push right first so that the left subtree is expanded (and emitted) first
        if (dec[(size_t)right * 7 + d.dr].type == DISTRIBUTE) { st_node[sp] = right; st_i[sp] = d.dr; }
        else                                                  { st_node[sp] = right; st_i[sp] = -1; }
        ++sp;
        if (dec[(size_t)left * 7 + d.dl].type == DISTRIBUTE) { st_node[sp] = left; st_i[sp] = d.dl; }
        else                                                 { st_node[sp] = left; st_i[sp] = -1; }
        ++sp;
    }
    return count;
}

// cwbvh.h:206-272 with `assignment[min_index] = min_slot` (the reference line is a no-op): greedy assignment
// of children to octant slots by dot(child centre - node centre, octant direction).
CRT_HD void order_children(const crt_flatnode* bvh2, int node, int children[8], int count) {
    const crt_flatnode& pn = bvh2[node];
    const float px = (pn.bmin[0] + pn.bmax[0]) * 0.5f, py = (pn.bmin[1] + pn.bmax[1]) * 0.5f, pz = (pn.bmin[2] + pn.bmax[2]) * 0.5f;
    float cost[8][8];
    for (int c = 0; c < count; ++c) {
        const crt_flatnode& cn = bvh2[children[c]];
        const float rx = (cn.bmin[0] + cn.bmax[0]) * 0.5f - px, ry = (cn.bmin[1] + cn.bmax[1]) * 0.5f - py,
                    rz = (cn.bmin[2] + cn.bmax[2]) * 0.5f - pz;
        for (int s = 0; s < 8; ++s) {
            const float dx = (s & 4) ? -1.0f : 1.0f, dy = (s & 2) ? -1.0f : 1.0f, dz = (s & 1) ? -1.0f : 1.0f;
            cost[c][s] = (rx * dx + ry * dy) + rz * dz;          // dot(): (x + y) + z, host/vecmath.hpp
        }
    }
    int assignment[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    bool filled[8] = {false, false, false, false, false, false, false, false};
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

// Biased exponent e with scale 2^(e-127) >= extent/255 (cwbvh.h:302-321; exponent taken from the float's bits,
// extent clamped so flat nodes do not ask for log2(0)): ceil(log2(v)) read off the bits of v, then checked in
// double that 255 steps really reach `hi` (hi - lo was rounded to fp32).
CRT_HD uint8_t pick_exponent(float lo, float hi) {
    const float ext = hi - lo;
    const float v = (ext > 1e-30f ? ext : 1e-30f) * (1.0f / 255.0f);
    const uint32_t bits = f2u(v);
    uint32_t e = (bits >> 23) & 0xffu;
    if (bits & 0x007fffffu) ++e;
    if (e < 1) e = 1;
    if (e > 254) e = 254;
    while (e < 254 && (double)lo + 255.0 * (double)u2f(e << 23) < (double)hi) ++e;
    return (uint8_t)e;
}
CRT_HD uint8_t quant_lo(float c, float p, float scale) {
    const float q = __builtin_floorf((c - p) * (1.0f / scale));
    int qi = q < 0.f ? 0 : q > 255.f ? 255 : (int)q;
    while (qi > 0 && (double)p + (double)qi * (double)scale > (double)c) --qi;   // fp32 subtraction may round up
    return (uint8_t)qi;
}
CRT_HD uint8_t quant_hi(float c, float p, float scale) {
    const float q = __builtin_ceilf((c - p) * (1.0f / scale));   // ceil, not floor (cwbvh.h:351-353 is not conservative)
    int qi = q < 0.f ? 0 : q > 255.f ? 255 : (int)q;
    while (qi < 255 && (double)p + (double)qi * (double)scale < (double)c) ++qi;
    return (uint8_t)qi;
}

// cwbvh.h:294-391 for one node8: origin, exponents, quantised child boxes, meta bytes, imask.  `children` are the
// slot-ordered BVH2 nodes (-1 = empty); nprims[c] triangles hang below a leaf-typed child.  Returns the number of
// inner children and of triangles this node references; child_base_index / triangle_base_index are the caller's.
CRT_HD void encode_node(const crt_flatnode* bvh2, const Decision* dec, const int32_t* nprims, int n2idx, const int children[8],
                        crt_node8& node, int& n_inner, int& n_tris) {
    const crt_flatnode& fn = bvh2[n2idx];
    __builtin_memset(&node, 0, sizeof node);
    node.p[0] = fn.bmin[0]; node.p[1] = fn.bmin[1]; node.p[2] = fn.bmin[2];
    float scale[3];
    for (int k = 0; k < 3; ++k) {
        node.e[k] = pick_exponent(fn.bmin[k], fn.bmax[k]);
        scale[k] = u2f((uint32_t)node.e[k] << 23);
    }
    n_inner = 0; n_tris = 0;
    for (int slot = 0; slot < 8; ++slot) {
        const int child = children[slot];
        if (child == -1) continue;
        const crt_flatnode& cn = bvh2[child];
        node.qlo_x[slot] = quant_lo(cn.bmin[0], node.p[0], scale[0]); node.qhi_x[slot] = quant_hi(cn.bmax[0], node.p[0], scale[0]);
        node.qlo_y[slot] = quant_lo(cn.bmin[1], node.p[1], scale[1]); node.qhi_y[slot] = quant_hi(cn.bmax[1], node.p[1], scale[1]);
        node.qlo_z[slot] = quant_lo(cn.bmin[2], node.p[2], scale[2]); node.qhi_z[slot] = quant_hi(cn.bmax[2], node.p[2], scale[2]);
        if (dec[(size_t)child * 7].type == LEAF) {
            const int cnt = nprims[child];                                  // 1..3
            uint8_t m = 0;
            for (int j = 0; j < cnt; ++j) m |= (uint8_t)(1u << (j + 5));    // unary count in bits 7..5
            m |= (uint8_t)n_tris;                                           // offset from the triangle base, 0..23
            node.meta[slot] = m;
            n_tris += cnt;
        } else {
            node.meta[slot] = (uint8_t)((slot + 24) | 0x20);
            node.imask |= (uint8_t)(1u << slot);
            ++n_inner;
        }
    }
}

// cwbvh.h:274-292: the leaf slots below a leaf-typed child, left subtree first (at most 3 triangles, so at most
// 3 BVH2 leaves and 2 interior nodes).  Returns how many were written.
CRT_HD int collect_slots(const crt_flatnode* bvh2, int node, int32_t out[3]) {
    int st[4], sp = 0, n = 0;
    st[sp++] = node;
    while (sp > 0) {
        const crt_flatnode& fn = bvh2[st[--sp]];
        if (is_leaf(fn)) {
            const int start = link_of(fn.bmin[3]), range = (int)fn.bmax[3];
            for (int i = 0; i < range && n < 3; ++i) out[n++] = start + i;
        } else {
            const int left = link_of(fn.bmin[3]);
            if (sp + 2 <= 4) { st[sp++] = left + 1; st[sp++] = left; }
        }
    }
    return n;
}

}  // namespace cw
}  // namespace crt
