// GPU BVH construction (the step immediately before the hot path).
//
// A linear BVH in the FlatNode layout the path consumes (Caitlyn/FlatNode.h:34-40, BFS order, children
// adjacent, one triangle per leaf like sbvh.h:285-324 leaves them): 30-bit Morton codes of the triangle
// centroids, a device radix sort (rocPRIM), Karras' parallel binary radix tree, and a bottom-up refit in
// which the second child to arrive at a node computes its box.  Topology and boxes are built on the
// device; the BFS renumbering into FlatNode order is a single O(n) sweep on the host because the result
// is handed back as host arrays anyway (crt_sbvh handle, interchangeable with crt_sbvh_build's).
// This is NOT the reference's SBVH: no SAH, no spatial splits, hence a different (lower quality, ~100x
// faster to build) tree; closest hits are identical by construction.
#include <cstring>   // rocPRIM's texture_cache_iterator.hpp calls memset without including it

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <chrono>
#include <deque>
#include <string>
#include <vector>

#include "../../include/crt.h"
#include "crt_error.hpp"
#include "crt_handles.hpp"

using crt::fail;

namespace {

#define LB_HIPCHK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) { cleanup(); return fail(CRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

// order-preserving float <-> uint mapping for atomicMin/atomicMax on floats
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void k_tri_bounds(const int32_t* __restrict__ vidx, const float* __restrict__ verts, uint32_t n, float* __restrict__ leaf_box,
                             uint32_t* __restrict__ scene_box) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    if (i < n) {
        for (int k = 0; k < 3; ++k) {
            const float* p = verts + 3 * (size_t)vidx[3 * (size_t)i + k];
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
        }
        for (int a = 0; a < 3; ++a) { leaf_box[6 * (size_t)i + a] = lo[a]; leaf_box[6 * (size_t)i + 3 + a] = hi[a]; }
    }
    // wave reduction of the centroid bounds, one atomic per wave and component
    for (int a = 0; a < 3; ++a) {
        float c = i < n ? 0.5f * (lo[a] + hi[a]) : 0.f;
        float mn = i < n ? c : 1e30f, mx = i < n ? c : -1e30f;
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_down(mn, off)); mx = fmaxf(mx, __shfl_down(mx, off)); }
        if ((threadIdx.x & 63u) == 0) { atomicMin(&scene_box[a], f2ord(mn)); atomicMax(&scene_box[3 + a], f2ord(mx)); }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// key = morton30 << 32 | triangle index: unique keys, ties broken by index (Karras 2012, section 4)
__global__ void k_morton(const float* __restrict__ leaf_box, const uint32_t* __restrict__ scene_box, uint32_t n, unsigned long long* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t code = 0;
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(scene_box[a]), hi = ord2f(scene_box[3 + a]);
        const float c = 0.5f * (leaf_box[6 * (size_t)i + a] + leaf_box[6 * (size_t)i + 3 + a]);
        const float ext = hi - lo;
        float q = ext > 0.f ? (c - lo) / ext * 1024.0f : 0.f;
        q = fminf(fmaxf(q, 0.f), 1023.0f);
        code |= spread10((uint32_t)q) << (2 - a);
    }
    keys[i] = ((unsigned long long)code << 32) | i;
}

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __builtin_clzll(k[i] ^ k[j]);
}

// Karras' binary radix tree: internal node i owns a key range; children >= n-1 encode leaves (n-1 + leaf).
__global__ void k_radix_tree(const unsigned long long* __restrict__ keys, int n, int2* __restrict__ child, int* __restrict__ parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = lo == gamma ? (n - 1) + gamma : gamma;
    const int right = hi == gamma + 1 ? (n - 1) + gamma + 1 : gamma + 1;
    child[i] = make_int2(left, right);
    parent[left] = i;
    parent[right] = i;
    if (i == 0) parent[0] = -1;
}

// Bottom-up refit.  One thread per leaf climbs; at every internal node the first arrival stops and the
// second one — which is therefore ordered after both children's boxes — writes the union.  The arrival
// counter is an agent-scope acq_rel RMW: it releases this thread's box stores and acquires the sibling's
// (workgroups on different XCDs do not share an L2; see the cross-XCD rules in the CDNA4 notes).
__global__ void k_refit(const unsigned long long* __restrict__ keys, const float* __restrict__ leaf_box, int n, const int2* __restrict__ child,
                        const int* __restrict__ parent, float* __restrict__ node_box, uint32_t* __restrict__ arrivals) {
    const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    const uint32_t tri = (uint32_t)(keys[leaf] & 0xffffffffull);
    float box[6];
    for (int a = 0; a < 6; ++a) box[a] = leaf_box[6 * (size_t)tri + a];
    float* mine = node_box + 6 * (size_t)((n - 1) + leaf);
    for (int a = 0; a < 6; ++a) mine[a] = box[a];
    int node = parent[(n - 1) + leaf];
    while (node >= 0) {
        const uint32_t prev = __hip_atomic_fetch_add(&arrivals[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0u) return;
        const int2 c = child[node];
        const float* a0 = node_box + 6 * (size_t)c.x;
        const float* a1 = node_box + 6 * (size_t)c.y;
        for (int a = 0; a < 3; ++a) {
            box[a] = fminf(__hip_atomic_load(&a0[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&a1[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            box[3 + a] = fmaxf(__hip_atomic_load(&a0[3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&a1[3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        float* out = node_box + 6 * (size_t)node;
        for (int a = 0; a < 6; ++a) __hip_atomic_store(&out[a], box[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        node = parent[node];
    }
}

thread_local float g_last_device_ms = 0.f, g_last_total_ms = 0.f;

}  // namespace

extern "C" {

int crt_lbvh_build(const crt_triangle* tris, size_t n_tris, const float* vertices, size_t n_vertices, uint32_t /*flags*/, crt_sbvh** out) {
    if (!out) return fail(CRT_ERR_INVALID, "crt_lbvh_build: null out");
    *out = nullptr;
    if (!tris || !vertices || n_tris == 0) return fail(CRT_ERR_INVALID, "crt_lbvh_build: empty input");
    if (n_tris >= (1u << 21)) return fail(CRT_ERR_LIMIT, "crt_lbvh_build: more than 2^21 triangles (FlatNode.h:24 start field)");
    for (size_t i = 0; i < n_tris; ++i)
        for (int j = 0; j < 3; ++j)
            if (tris[i].v[j] < 0 || (size_t)tris[i].v[j] >= n_vertices) return fail(CRT_ERR_INVALID, "crt_lbvh_build: vertex index out of range");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(CRT_ERR_NO_DEVICE, "crt_lbvh_build: no HIP device visible");

    const auto t_begin = std::chrono::steady_clock::now();
    const int n = (int)n_tris;
    int32_t* d_vidx = nullptr; float* d_verts = nullptr; float* d_leaf_box = nullptr; uint32_t* d_scene = nullptr;
    unsigned long long *d_keys = nullptr, *d_sorted = nullptr; void* d_tmp = nullptr;
    int2* d_child = nullptr; int* d_parent = nullptr; float* d_node_box = nullptr; uint32_t* d_arrivals = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        void* ptrs[] = {d_vidx, d_verts, d_leaf_box, d_scene, d_keys, d_sorted, d_tmp, d_child, d_parent, d_node_box, d_arrivals};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    };

    std::vector<int32_t> vidx(3 * n_tris);
    for (size_t i = 0; i < n_tris; ++i) for (int j = 0; j < 3; ++j) vidx[3 * i + j] = tris[i].v[j];
    const size_t n_nodes = 2 * n_tris - 1;
    LB_HIPCHK(hipMalloc(&d_vidx, vidx.size() * 4));
    LB_HIPCHK(hipMalloc(&d_verts, n_vertices * 12));
    LB_HIPCHK(hipMalloc(&d_leaf_box, n_tris * 24));
    LB_HIPCHK(hipMalloc(&d_scene, 6 * 4));
    LB_HIPCHK(hipMalloc(&d_keys, n_tris * 8));
    LB_HIPCHK(hipMalloc(&d_sorted, n_tris * 8));
    LB_HIPCHK(hipMalloc(&d_child, std::max<size_t>(n_tris - 1, 1) * sizeof(int2)));
    LB_HIPCHK(hipMalloc(&d_parent, n_nodes * 4));
    LB_HIPCHK(hipMalloc(&d_node_box, n_nodes * 24));
    LB_HIPCHK(hipMalloc(&d_arrivals, std::max<size_t>(n_tris - 1, 1) * 4));
    LB_HIPCHK(hipMemcpy(d_vidx, vidx.data(), vidx.size() * 4, hipMemcpyHostToDevice));
    LB_HIPCHK(hipMemcpy(d_verts, vertices, n_vertices * 12, hipMemcpyHostToDevice));
    const uint32_t scene_init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    LB_HIPCHK(hipMemcpy(d_scene, scene_init, sizeof scene_init, hipMemcpyHostToDevice));
    LB_HIPCHK(hipMemset(d_arrivals, 0, std::max<size_t>(n_tris - 1, 1) * 4));
    size_t tmp_bytes = 0;
    LB_HIPCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys, d_sorted, n_tris, 0, 64, (hipStream_t)0));
    LB_HIPCHK(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
    LB_HIPCHK(hipEventCreate(&ev0));
    LB_HIPCHK(hipEventCreate(&ev1));

    const uint32_t g = (uint32_t)((n_tris + 255) / 256);
    LB_HIPCHK(hipEventRecord(ev0, 0));
    hipLaunchKernelGGL(k_tri_bounds, dim3(g), dim3(256), 0, 0, d_vidx, d_verts, (uint32_t)n, d_leaf_box, d_scene);
    hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, 0, d_leaf_box, d_scene, (uint32_t)n, d_keys);
    LB_HIPCHK(rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_keys, d_sorted, n_tris, 0, 64, (hipStream_t)0));
    if (n > 1) hipLaunchKernelGGL(k_radix_tree, dim3((uint32_t)((n_tris - 1 + 255) / 256)), dim3(256), 0, 0, d_sorted, n, d_child, d_parent);
    else { const int minus1 = -1; LB_HIPCHK(hipMemcpyAsync(d_parent, &minus1, 4, hipMemcpyHostToDevice, 0)); }
    hipLaunchKernelGGL(k_refit, dim3(g), dim3(256), 0, 0, d_sorted, d_leaf_box, n, d_child, d_parent, d_node_box, d_arrivals);
    LB_HIPCHK(hipEventRecord(ev1, 0));
    LB_HIPCHK(hipDeviceSynchronize());
    LB_HIPCHK(hipGetLastError());
    LB_HIPCHK(hipEventElapsedTime(&g_last_device_ms, ev0, ev1));

    std::vector<unsigned long long> sorted(n_tris);
    std::vector<int2> child(std::max<size_t>(n_tris - 1, 1));
    std::vector<float> node_box(6 * n_nodes);
    LB_HIPCHK(hipMemcpy(sorted.data(), d_sorted, n_tris * 8, hipMemcpyDeviceToHost));
    if (n > 1) LB_HIPCHK(hipMemcpy(child.data(), d_child, (n_tris - 1) * sizeof(int2), hipMemcpyDeviceToHost));
    LB_HIPCHK(hipMemcpy(node_box.data(), d_node_box, node_box.size() * 4, hipMemcpyDeviceToHost));
    cleanup();
    ev0 = ev1 = nullptr; d_vidx = nullptr;   // (cleanup already ran; nothing below touches the device)

    crt_sbvh* h = new (std::nothrow) crt_sbvh;
    if (!h) return fail(CRT_ERR_NOMEM, "crt_lbvh_build: out of memory");
    crt::SBVH& b = h->bvh;
    // leaf slot j <-> j-th triangle in Morton order
    b.triangle_indices.resize(n_tris);
    b.triangles.resize(n_tris);
    for (size_t jx = 0; jx < n_tris; ++jx) {
        b.triangle_indices[jx] = (int32_t)(sorted[jx] & 0xffffffffull);
        b.triangles[jx] = tris[b.triangle_indices[jx]];
    }
    // BFS renumbering into FlatNode order: an interior node's children are adjacent (sbvh.h:570-609)
    b.flat_nodes.reserve(n_nodes);
    std::deque<std::pair<int, int>> queue;   // (radix-tree node id, level); ids >= n-1 are leaves
    queue.emplace_back(n > 1 ? 0 : (n - 1), 0);
    int next_child = 0;
    b.depth = 0;
    while (!queue.empty()) {
        const auto [id, level] = queue.front();
        queue.pop_front();
        crt_flatnode f;
        const float* bx = node_box.data() + 6 * (size_t)id;
        f.bmin[0] = bx[0]; f.bmin[1] = bx[1]; f.bmin[2] = bx[2];
        f.bmax[0] = bx[3]; f.bmax[1] = bx[4]; f.bmax[2] = bx[5];
        if (id >= n - 1) {
            f.bmin[3] = (float)(id - (n - 1));
            f.bmax[3] = 1.0f;
            b.depth = std::max(b.depth, level);
        } else {
            f.bmin[3] = (float)(next_child + 1);
            f.bmax[3] = 0.0f;
            next_child += 2;
            queue.emplace_back(child[id].x, level + 1);
            queue.emplace_back(child[id].y, level + 1);
        }
        b.flat_nodes.push_back(f);
    }
    g_last_total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *out = h;
    return CRT_OK;
}

void crt_lbvh_last_build_ms(float* device_ms, float* total_ms) {
    if (device_ms) *device_ms = g_last_device_ms;
    if (total_ms) *total_ms = g_last_total_ms;
}

}  // extern "C"
