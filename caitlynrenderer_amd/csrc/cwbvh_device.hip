// BVH2 -> CWBVH conversion on the device (crt_cwbvh_convert_device): the step between the BVH builders and
// crt_scene_create, SURVEY.md §8f rank 1.  Same algorithm and the same bytes as the host converter
// (host/cwbvh.cpp; layout/intent Caitlyn/cwbvh.h:58-411, semantics SURVEY.md appendix C): the per-node
// arithmetic is the shared code of host/cwbvh_core.hpp, and what the host does with a reverse sweep and a
// depth-first recursion is re-expressed as data-parallel passes:
//   1. k_parents        parent links of the BFS-ordered FlatNode array
//   2. k_costs_level    the 7-entry cost/decision table of every BVH2 node, bottom-up, one launch per level of the
//                       BFS-ordered array (k_depths / k_level_starts find the levels): plain loads and stores, the
//                       kernel boundary orders a level after its children's
//   3. k_discover       the node8 tree level by level: slot-ordered children of every node8, inner children
//                       appended to the next level
//   4. k_sizes          subtree sizes (node8 count, triangle count), deepest level first
//   5. k_place          the host converter numbers nodes and triangles in depth-first pre-order; with the subtree
//                       sizes known, every node's index / child base / triangle base follow from its parent's
//   6. k_emit           quantised 80-byte nodes, triangle slots, debug child map, slot coverage
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <new>
#include <vector>

#include "crt_error.hpp"
#include "crt_handles.hpp"
#include "device_build.hpp"
#include "host/cwbvh_core.hpp"

#define CW_HIPCHK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) { cleanup(); return fail(CRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

namespace {

using namespace crt::cw;
using crt::fail;

enum : uint32_t { ERR_LEAF_SIZE = 1u, ERR_LEAF_RANGE = 2u, ERR_LINK = 4u, ERR_COVER = 8u };

static_assert(sizeof(Decision) == 8, "7 decisions per node are stored as 7 x 8 bytes");

__global__ void k_parents(const crt_flatnode* __restrict__ bvh2, uint32_t n2, int32_t* __restrict__ parent, uint32_t* flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const crt_flatnode fn = bvh2[i];
    if (is_leaf(fn)) return;
    const int left = crt::link_of(fn.bmin[3]);
    if (left <= (int)i || (uint32_t)left + 1u >= n2) { atomicOr(flags, ERR_LINK); return; }   // children follow parents (BFS order)
    parent[left] = (int32_t)i;
    parent[left + 1] = (int32_t)i;
}

// depth of every BVH2 node (root 0) by walking the parent links; in a BFS-ordered array it never decreases with the index
__global__ void k_depths(const int32_t* __restrict__ parent, uint32_t n2, uint32_t* __restrict__ depth) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    uint32_t d = 0;
    for (int p = parent[i]; p >= 0; p = parent[p]) ++d;
    depth[i] = d;
}
__global__ void k_level_starts(const uint32_t* __restrict__ depth, uint32_t n2, uint32_t* __restrict__ level_start, uint32_t cap, uint32_t* flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const uint32_t d = depth[i];
    if (i > 0u && depth[i - 1] > d) atomicOr(flags, ERR_LINK);          // not breadth-first
    if ((i == 0u || depth[i - 1] != d) && d < cap) level_start[d] = i;
}
// The 7-entry cost/decision table of every node of one level (cwbvh.h:75-173); the children's tables are in the next
// level, computed by the previous launch.  (First version: one thread per leaf climbing with an agent-scope acq_rel
// arrival counter per node — 3.6 ms of the 4.6 ms conversion at 1 M triangles.)
__global__ void k_costs_level(const crt_flatnode* __restrict__ bvh2, uint32_t begin, uint32_t end, uint32_t n2, uint32_t n_slots,
                              Decision* __restrict__ dec, int32_t* __restrict__ nprims, uint32_t* flags) {
    const uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    const crt_flatnode fn = bvh2[i];
    Decision d[7];
    int np;
    if (is_leaf(fn)) {
        np = (int)fn.bmax[3];
        const int start = crt::link_of(fn.bmin[3]);
        if (np < 1 || np > 3) { atomicOr(flags, ERR_LEAF_SIZE); return; }
        if (start < 0 || (uint32_t)(start + np) > n_slots) { atomicOr(flags, ERR_LEAF_RANGE); return; }
        leaf_decisions(half_area(fn), np, d);
    } else {
        const int left = crt::link_of(fn.bmin[3]);
        // the host bails out on ERR_LINK before the first cost pass; this check keeps the kernel in bounds on its own
        if (left <= (int)i || (uint32_t)left + 1u >= n2) { atomicOr(flags, ERR_LINK); return; }
        np = nprims[left] + nprims[left + 1];
        inner_decisions(half_area(fn), np, dec + (size_t)left * 7, dec + (size_t)(left + 1) * 7, d);
    }
    for (int k = 0; k < 7; ++k) dec[(size_t)i * 7 + k] = d[k];
    nprims[i] = np;
}

// per node8 (temporary id = discovery order, level by level)
struct Tmp {
    int32_t* bvh2;       // BVH2 node this node8 stands for
    int32_t* children;   // 8 slot-ordered BVH2 nodes, -1 = empty
    int32_t* first;      // temporary id of the first inner child (inner children are consecutive, in slot order)
    uint8_t* n_inner;
    uint8_t* n_tris;
    uint32_t* S;         // node8 nodes in the subtree (self included)
    uint32_t* T;         // triangles referenced in the subtree
    uint32_t* idx;       // final node index
    uint32_t* A;         // child_base_index
    uint32_t* B;         // triangle_base_index
};

// One returning atomic per WAVE on the list counter (exclusive prefix of the lanes' child counts): one per thread on a single
// address runs at ~90 per microsecond, which made the deepest level — 100 k node8 at 1 M triangles — 1 ms of the conversion.
// The order of the next level's list is free: the final numbering comes from k_sizes / k_place.
__global__ void k_discover(const crt_flatnode* __restrict__ bvh2, const Decision* __restrict__ dec, const int32_t* __restrict__ nprims,
                           Tmp t, uint32_t begin, uint32_t end, uint32_t* n_tmp, int root_is_leaf) {
    const uint32_t id = begin + blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = id < end;
    int children[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int n_inner = 0, n_tris = 0;
    if (live) {
        const int node = t.bvh2[id];
        int count = 0;
        if (root_is_leaf && id == 0u) children[count++] = node;
        else count = get_children(bvh2, dec, node, children);
        order_children(bvh2, node, children, count);
        for (int s = 0; s < 8; ++s) {
            if (children[s] == -1) continue;
            if (dec[(size_t)children[s] * 7].type == LEAF) n_tris += nprims[children[s]];
            else ++n_inner;
        }
    }
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t incl = (uint32_t)n_inner;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += up; }
    const uint32_t total = __shfl(incl, 63);
    uint32_t wave_base = 0u;
    if (lane == 63u && total) wave_base = atomicAdd(n_tmp, total);
    wave_base = __shfl(wave_base, 63);
    if (!live) return;
    const uint32_t base = n_inner ? wave_base + incl - (uint32_t)n_inner : 0u;
    uint32_t k = 0;
    for (int s = 0; s < 8; ++s) {
        t.children[(size_t)id * 8 + s] = children[s];
        if (children[s] != -1 && dec[(size_t)children[s] * 7].type != LEAF) t.bvh2[base + k++] = children[s];
    }
    t.first[id] = (int32_t)base;
    t.n_inner[id] = (uint8_t)n_inner;
    t.n_tris[id] = (uint8_t)n_tris;
}

__global__ void k_sizes(Tmp t, uint32_t begin, uint32_t end) {
    const uint32_t id = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= end) return;
    uint32_t S = 1u, T = t.n_tris[id];
    const uint32_t first = (uint32_t)t.first[id];
    for (uint32_t k = 0; k < t.n_inner[id]; ++k) { S += t.S[first + k]; T += t.T[first + k]; }
    t.S[id] = S; t.T[id] = T;
}

// The host converter (collapse) visits a node, appends its inner children to the node array and its triangles to
// the triangle array, then recurses into the inner children in slot order.  So child j sits at child_base + j, and
// when the recursion reaches it the arrays have grown by the node's own children/triangles and by the complete
// subtrees of its earlier siblings.
__global__ void k_place(Tmp t, uint32_t begin, uint32_t end) {
    const uint32_t id = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= end) return;
    const uint32_t A = t.A[id], first = (uint32_t)t.first[id];
    uint32_t a = A + t.n_inner[id], b = t.B[id] + t.n_tris[id];
    for (uint32_t k = 0; k < t.n_inner[id]; ++k) {
        const uint32_t c = first + k;
        t.idx[c] = A + k;
        t.A[c] = a;
        t.B[c] = b;
        a += t.S[c] - 1u;
        b += t.T[c];
    }
}

__global__ void k_emit(const crt_flatnode* __restrict__ bvh2, const Decision* __restrict__ dec, const int32_t* __restrict__ nprims, Tmp t,
                       uint32_t n8, uint32_t n_slots, crt_node8* __restrict__ nodes, int32_t* __restrict__ tri_slots,
                       int32_t* __restrict__ child_bvh2, uint32_t* __restrict__ seen, uint32_t* flags) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n8) return;
    int children[8];
    for (int s = 0; s < 8; ++s) children[s] = t.children[(size_t)id * 8 + s];
    crt_node8 node;
    int n_inner, n_tris;
    encode_node(bvh2, dec, nprims, t.bvh2[id], children, node, n_inner, n_tris);
    node.child_base_index = t.A[id];
    node.triangle_base_index = t.B[id];
    const uint32_t idx = t.idx[id];
    nodes[idx] = node;
    uint32_t off = 0;
    for (int s = 0; s < 8; ++s) {
        if (child_bvh2) child_bvh2[(size_t)idx * 8 + s] = children[s];
        if (children[s] == -1 || dec[(size_t)children[s] * 7].type != LEAF) continue;
        int32_t slots[3];
        const int cnt = collect_slots(bvh2, children[s], slots);
        for (int i = 0; i < cnt; ++i) {
            const uint32_t at = t.B[id] + off + (uint32_t)i;
            if (at < n_slots && (uint32_t)slots[i] < n_slots) { tri_slots[at] = slots[i]; atomicAdd(&seen[slots[i]], 1u); }
            else atomicOr(flags, ERR_COVER);
        }
        off += (uint32_t)cnt;
    }
}

__global__ void k_cover(const uint32_t* __restrict__ seen, uint32_t n_slots, uint32_t* flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots && seen[i] != 1u) atomicOr(flags, ERR_COVER);
}

constexpr size_t kMaxLevels = 4096;
thread_local float g_device_ms = 0.f, g_total_ms = 0.f;

inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + 255) / 256 ? (n + 255) / 256 : 1)); }

}  // namespace

namespace crt {

size_t cwbvh_tmp_bytes(size_t n2, size_t ns) {
    auto P = DeviceArena::padded;
    const size_t cap8 = n2 / 2 + 1;
    return 3 * P(n2 * 4) + P(n2 * 7 * 8) + 2 * P(4) + P(kMaxLevels * 4) + P(ns * 4) + 7 * P(cap8 * 4) + P(cap8 * 8 * 4) + 2 * P(cap8) + 4096;
}

int cwbvh_convert_on_device(const crt_flatnode* d_bvh2, uint32_t n2, uint32_t ns, DeviceArena& tmp, int32_t* d_tri_slots,
                            crt_node8** d_nodes_out, int32_t** d_child_bvh2_out, uint32_t* n8_out, uint32_t* depth_out, float* device_ms,
                            hipStream_t st) {
    *d_nodes_out = nullptr;
    if (d_child_bvh2_out) *d_child_bvh2_out = nullptr;
    const uint32_t cap8 = n2 / 2u + 1u;            // every node8 stands for a distinct interior BVH2 node (or the root)
    crt_node8* d_nodes = nullptr; int32_t* d_child_bvh2 = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        if (d_nodes) (void)hipFree(d_nodes);
        if (d_child_bvh2) (void)hipFree(d_child_bvh2);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    };
    int32_t* d_parent = tmp.take<int32_t>(n2);
    uint32_t* d_depth = tmp.take<uint32_t>(n2);
    unsigned long long* d_dec = tmp.take<unsigned long long>((size_t)n2 * 7);
    int32_t* d_nprims = tmp.take<int32_t>(n2);
    uint32_t* d_flags = tmp.take<uint32_t>(1);
    uint32_t* d_levels = tmp.take<uint32_t>(kMaxLevels);
    uint32_t* d_ntmp = tmp.take<uint32_t>(1);
    uint32_t* d_seen = tmp.take<uint32_t>(ns);
    Tmp t{};
    t.bvh2 = tmp.take<int32_t>(cap8);
    t.children = tmp.take<int32_t>((size_t)cap8 * 8);
    t.first = tmp.take<int32_t>(cap8);
    t.n_inner = tmp.take<uint8_t>(cap8);
    t.n_tris = tmp.take<uint8_t>(cap8);
    t.S = tmp.take<uint32_t>(cap8);
    t.T = tmp.take<uint32_t>(cap8);
    t.idx = tmp.take<uint32_t>(cap8);
    t.A = tmp.take<uint32_t>(cap8);
    t.B = tmp.take<uint32_t>(cap8);
    if (!d_parent || !d_depth || !d_dec || !d_nprims || !d_flags || !d_levels || !d_ntmp || !d_seen || !t.bvh2 || !t.children || !t.first ||
        !t.n_inner || !t.n_tris || !t.S || !t.T || !t.idx || !t.A || !t.B)
        return fail(CRT_ERR_NOMEM, "cwbvh: temporary arena too small");
    CW_HIPCHK(hipEventCreate(&ev0));
    CW_HIPCHK(hipEventCreate(&ev1));

    CW_HIPCHK(hipEventRecord(ev0, st));
    CW_HIPCHK(hipMemsetAsync(d_parent, 0xff, (size_t)n2 * 4, st));
    CW_HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
    CW_HIPCHK(hipMemsetAsync(d_seen, 0, (size_t)ns * 4, st));
    hipLaunchKernelGGL(k_parents, grid_for(n2), dim3(256), 0, st, d_bvh2, n2, d_parent, d_flags);
    // levels of the BFS-ordered array, then the cost tables deepest level first
    hipLaunchKernelGGL(k_depths, grid_for(n2), dim3(256), 0, st, d_parent, n2, d_depth);
    hipLaunchKernelGGL(k_level_starts, grid_for(n2), dim3(256), 0, st, d_depth, n2, d_levels, (uint32_t)kMaxLevels, d_flags);
    uint32_t deepest = 0;
    uint32_t flags = 0;
    CW_HIPCHK(hipMemcpyAsync(&deepest, d_depth + (n2 - 1), 4, hipMemcpyDeviceToHost, st));
    // a link that is out of order, out of range, negative or NaN was flagged by k_parents / k_level_starts: stop before
    // any pass follows the links (host/cwbvh.cpp returns CRT_ERR_INVALID at the same point)
    CW_HIPCHK(hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, st));
    CW_HIPCHK(hipStreamSynchronize(st));
    if (flags) {
        cleanup();
        return fail(CRT_ERR_INVALID, "crt_cwbvh_convert_device: BVH2 child link out of order");
    }
    if (deepest + 1u > kMaxLevels) { cleanup(); return fail(CRT_ERR_LIMIT, "crt_cwbvh_convert_device: BVH2 deeper than 4096 levels"); }
    std::vector<uint32_t> lv(deepest + 2u);
    CW_HIPCHK(hipMemcpyAsync(lv.data(), d_levels, (deepest + 1u) * 4, hipMemcpyDeviceToHost, st));
    CW_HIPCHK(hipStreamSynchronize(st));
    lv[deepest + 1u] = n2;
    for (uint32_t l = deepest + 1u; l-- > 0;)
        hipLaunchKernelGGL(k_costs_level, grid_for(lv[l + 1] - lv[l]), dim3(256), 0, st, d_bvh2, lv[l], lv[l + 1], n2, ns,
                           reinterpret_cast<Decision*>(d_dec), d_nprims, d_flags);

    Decision root0;
    CW_HIPCHK(hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, st));
    CW_HIPCHK(hipMemcpyAsync(&root0, d_dec, 8, hipMemcpyDeviceToHost, st));
    CW_HIPCHK(hipStreamSynchronize(st));
    auto input_error = [&](uint32_t f) -> int {
        cleanup();
        const char* msg = (f & ERR_LINK) ? "BVH2 child link out of order"
                        : (f & ERR_LEAF_SIZE) ? "BVH2 leaf with more than 3 triangles cannot be encoded"
                        : (f & ERR_LEAF_RANGE) ? "BVH2 leaf range outside the triangle array"
                        : "BVH2 leaves do not cover the triangle array exactly once";
        return fail(CRT_ERR_INVALID, std::string("crt_cwbvh_convert_device: ") + msg);
    };
    if (flags) return input_error(flags);
    const int root_is_leaf = root0.type == LEAF;
    const Decision* dec = reinterpret_cast<const Decision*>(d_dec);

    // node8 tree, level by level (the root stands for BVH2 node 0)
    const int32_t zero = 0; const uint32_t one = 1;
    CW_HIPCHK(hipMemcpyAsync(t.bvh2, &zero, 4, hipMemcpyHostToDevice, st));
    CW_HIPCHK(hipMemcpyAsync(d_ntmp, &one, 4, hipMemcpyHostToDevice, st));
    std::vector<uint32_t> level_begin{0u};
    uint32_t begin = 0, end = 1;
    while (begin < end) {
        if (level_begin.size() > 64) { cleanup(); return fail(CRT_ERR_LIMIT, "crt_cwbvh_convert_device: CWBVH deeper than 64 levels"); }
        hipLaunchKernelGGL(k_discover, grid_for(end - begin), dim3(256), 0, st, d_bvh2, dec, d_nprims, t, begin, end, d_ntmp, root_is_leaf);
        uint32_t n_tmp = 0;
        CW_HIPCHK(hipMemcpyAsync(&n_tmp, d_ntmp, 4, hipMemcpyDeviceToHost, st));
        CW_HIPCHK(hipStreamSynchronize(st));
        if (n_tmp > cap8) { cleanup(); return fail(CRT_ERR_LIMIT, "crt_cwbvh_convert_device: node8 count exceeds its bound"); }
        begin = end; end = n_tmp;
        level_begin.push_back(begin);
    }
    const uint32_t n8 = end;
    const uint32_t depth = (uint32_t)level_begin.size() - 1u;      // level_begin = starts of levels 0..depth-1, then n8
    for (uint32_t l = depth; l-- > 0;)
        hipLaunchKernelGGL(k_sizes, grid_for(level_begin[l + 1] - level_begin[l]), dim3(256), 0, st, t, level_begin[l], level_begin[l + 1]);
    const uint32_t root_place[3] = {0u, 1u, 0u};
    CW_HIPCHK(hipMemcpyAsync(t.idx, &root_place[0], 4, hipMemcpyHostToDevice, st));
    CW_HIPCHK(hipMemcpyAsync(t.A, &root_place[1], 4, hipMemcpyHostToDevice, st));
    CW_HIPCHK(hipMemcpyAsync(t.B, &root_place[2], 4, hipMemcpyHostToDevice, st));
    for (uint32_t l = 0; l < depth; ++l)
        hipLaunchKernelGGL(k_place, grid_for(level_begin[l + 1] - level_begin[l]), dim3(256), 0, st, t, level_begin[l], level_begin[l + 1]);
    CW_HIPCHK(hipMalloc(&d_nodes, (size_t)n8 * sizeof(crt_node8)));
    if (d_child_bvh2_out) CW_HIPCHK(hipMalloc(&d_child_bvh2, (size_t)n8 * 8 * 4));
    hipLaunchKernelGGL(k_emit, grid_for(n8), dim3(256), 0, st, d_bvh2, dec, d_nprims, t, n8, ns, d_nodes, d_tri_slots, d_child_bvh2, d_seen, d_flags);
    hipLaunchKernelGGL(k_cover, grid_for(ns), dim3(256), 0, st, d_seen, ns, d_flags);
    CW_HIPCHK(hipEventRecord(ev1, st));
    CW_HIPCHK(hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, st));
    CW_HIPCHK(hipStreamSynchronize(st));
    CW_HIPCHK(hipGetLastError());
    float ms = 0.f;
    CW_HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    if (flags) return input_error(flags);
    if (device_ms) *device_ms = ms;
    *d_nodes_out = d_nodes; d_nodes = nullptr;
    if (d_child_bvh2_out) { *d_child_bvh2_out = d_child_bvh2; d_child_bvh2 = nullptr; }
    *n8_out = n8;
    *depth_out = depth;
    cleanup();
    return CRT_OK;
}

// crt_warmup: load this translation unit's code object on the current device (device_build.hpp)
int warm_cwbvh_kernels() {
    hipFuncAttributes a;
    hipError_t e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_parents))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_depths))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_level_starts))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_costs_level))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_discover))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sizes))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_place))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_emit))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_cover))) != hipSuccess) return (int)e;
    return 0;
}

}  // namespace crt

extern "C" {

int crt_cwbvh_convert_device(const crt_flatnode* bvh2, size_t n_nodes, size_t n_slots, crt_cwbvh** out) {
    if (!out) return fail(CRT_ERR_INVALID, "crt_cwbvh_convert_device: null out");
    *out = nullptr;
    if (!bvh2 || n_nodes == 0) return fail(CRT_ERR_INVALID, "crt_cwbvh_convert_device: empty BVH2");
    if (n_nodes >= (1u << 24)) return fail(CRT_ERR_LIMIT, "crt_cwbvh_convert_device: BVH2 too large: float links are exact only below 2^24");
    if (n_slots == 0 || n_slots >= (1u << 24)) return fail(CRT_ERR_INVALID, "crt_cwbvh_convert_device: bad triangle slot count");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(CRT_ERR_NO_DEVICE, "crt_cwbvh_convert_device: no HIP device visible");

    const auto t_begin = std::chrono::steady_clock::now();
    const uint32_t n2 = (uint32_t)n_nodes, ns = (uint32_t)n_slots;
    crt::DeviceArena arena;
    crt_node8* d_nodes = nullptr; int32_t* d_child_bvh2 = nullptr;
    auto cleanup = [&]() {
        arena.release();
        if (d_nodes) (void)hipFree(d_nodes);
        if (d_child_bvh2) (void)hipFree(d_child_bvh2);
    };
    auto P = crt::DeviceArena::padded;
    CW_HIPCHK(arena.reserve(crt::cwbvh_tmp_bytes(n2, ns) + P((size_t)n2 * sizeof(crt_flatnode)) + P((size_t)ns * 4)));
    crt_flatnode* d_bvh2 = arena.take<crt_flatnode>(n2);
    int32_t* d_tri_slots = arena.take<int32_t>(ns);
    if (!d_bvh2 || !d_tri_slots) { cleanup(); return fail(CRT_ERR_NOMEM, "crt_cwbvh_convert_device: arena"); }
    CW_HIPCHK(hipMemcpy(d_bvh2, bvh2, (size_t)n2 * sizeof(crt_flatnode), hipMemcpyHostToDevice));
    uint32_t n8 = 0, depth = 0;
    int rc = crt::cwbvh_convert_on_device(d_bvh2, n2, ns, arena, d_tri_slots, &d_nodes, &d_child_bvh2, &n8, &depth, &g_device_ms, (hipStream_t)0);
    if (rc) { cleanup(); return rc; }

    crt_cwbvh* h = new (std::nothrow) crt_cwbvh;
    if (!h) { cleanup(); return fail(CRT_ERR_NOMEM, "crt_cwbvh_convert_device: out of memory"); }
    try {
        h->bvh.nodes.resize(n8);
        h->bvh.tri_slots.resize(ns);
        h->bvh.child_bvh2.resize((size_t)n8 * 8);
    } catch (const std::exception& e) {
        delete h; cleanup();
        return fail(CRT_ERR_NOMEM, std::string("crt_cwbvh_convert_device: ") + e.what());
    }
    if (hipMemcpy(h->bvh.nodes.data(), d_nodes, (size_t)n8 * sizeof(crt_node8), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(h->bvh.tri_slots.data(), d_tri_slots, (size_t)ns * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(h->bvh.child_bvh2.data(), d_child_bvh2, (size_t)n8 * 8 * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        delete h; cleanup();
        return fail(CRT_ERR_HIP, "crt_cwbvh_convert_device: copy back failed");
    }
    h->bvh.triangle_indices = h->bvh.tri_slots;       // no slot -> original map at this entry point (as crt_cwbvh_convert)
    h->bvh.depth = depth;
    cleanup();
    g_total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *out = h;
    return CRT_OK;
}

void crt_cwbvh_last_convert_ms(float* device_ms, float* total_ms) {
    if (device_ms) *device_ms = g_device_ms;
    if (total_ms) *total_ms = g_total_ms;
}

}  // extern "C"
