// GPU BVH construction (the step immediately before the hot path).
//
// A linear BVH in the FlatNode layout the path consumes (Caitlyn/FlatNode.h:34-40, BFS order, children
// adjacent, one triangle per leaf like sbvh.h:285-324 leaves them): 30-bit Morton codes of the triangle
// centroids, a device radix sort (rocPRIM), Karras' parallel binary radix tree, the renumbering into FlatNode BFS
// order as one more radix sort of the nodes by (depth, first key of the node's range), and a bottom-up refit
// of that array, one small launch per level.  The result is handed back
// as host arrays (crt_sbvh handle, interchangeable with crt_sbvh_build's).
// This is NOT the reference's SBVH: no SAH, no spatial splits, hence a different (lower quality, ~100x
// faster to build) tree; closest hits are identical by construction.
#include <cstring>   // rocPRIM's texture_cache_iterator.hpp calls memset without including it

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <string>
#include <vector>

#include "../../include/crt.h"
#include "crt_error.hpp"
#include "crt_handles.hpp"
#include "device_build.hpp"
#include "host/flatnode_link.hpp"

using crt::fail;

namespace {

#define LB_HIPCHK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) { cleanup(); return fail(CRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

// leaf slot j <-> j-th triangle in Morton order: the low word of the sorted key
__global__ void k_tri_order(const unsigned long long* __restrict__ sorted, uint32_t n, uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = (uint32_t)(sorted[i] & 0xffffffffull);
}

// order-preserving float <-> uint mapping for atomicMin/atomicMax on floats
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// Largest |coordinate| a builder accepts: box extents stay below 2e18, so half-areas (sums of products of two extents) stay
// below 1.2e37 and finite in fp32.
#define CRT_MAX_COORD 1.0e18f

// Triangle boxes + the bounds of their centroids.  scene_box[0..5] = centroid bounds (ordered-uint form), [6] = 1 once a triangle
// with a non-finite or over-range vertex was seen.  Grid-stride over the triangles with at most 1024 workgroups, the
// centroid bounds reduced per lane, per wave and per workgroup before they touch the six global words: one atomic
// per wave and component (94 k atomics on one cache line) made this kernel 1.07 ms of a 5 ms build.
__global__ void __launch_bounds__(256) k_tri_bounds(const int32_t* __restrict__ vidx, uint32_t stride, const float* __restrict__ verts, uint32_t n,
                                                    float* __restrict__ leaf_box, uint32_t* __restrict__ scene_box) {
    __shared__ float s_red[4][6];
    float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
            const float* p = verts + 3 * (size_t)vidx[(size_t)stride * i + k];
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]);
                ok = ok && fabsf(p[a]) <= CRT_MAX_COORD;             // false for NaN and +-inf as well
            }
        }
        if (!ok) {
            // The build is refused on the host once this flag comes back (CRT_ERR_INVALID); until then every later kernel sees a
            // finite box, so no surface area overflows to inf / NaN (which is what let a PLOC cluster go without a neighbour)
            for (int a = 0; a < 3; ++a) lo[a] = hi[a] = 0.f;
            atomicOr(&scene_box[6], 1u);
        }
        for (int a = 0; a < 3; ++a) {
            leaf_box[6 * (size_t)i + a] = lo[a]; leaf_box[6 * (size_t)i + 3 + a] = hi[a];
            const float c = 0.5f * (lo[a] + hi[a]);
            cmn[a] = fminf(cmn[a], c); cmx[a] = fmaxf(cmx[a], c);
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; ++a) {
        float mn = cmn[a], mx = cmx[a];
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_down(mn, off)); mx = fmaxf(mx, __shfl_down(mx, off)); }
        if (lane == 0) { s_red[wave][a] = mn; s_red[wave][3 + a] = mx; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = (int)threadIdx.x;
        float v = s_red[0][a];
        for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, s_red[w][a]) : fmaxf(v, s_red[w][a]);
        if (a < 3) { if (v < 1e30f) atomicMin(&scene_box[a], f2ord(v)); }
        else if (v > -1e30f) atomicMax(&scene_box[a], f2ord(v));
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
__global__ void k_radix_tree(const unsigned long long* __restrict__ keys, int n, int2* __restrict__ child, int* __restrict__ parent,
                             int* __restrict__ range_first) {
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
    range_first[i] = lo;
    parent[left] = i;
    parent[right] = i;
    if (i == 0) parent[0] = -1;
}

// BFS renumbering into FlatNode order (sbvh.h:570-609: children adjacent, parents before children) without a
// queue: nodes of one level own disjoint key ranges, so sorting all nodes by (depth, first key of the range) IS the
// breadth-first order a queue would produce (left child before right, parents in order).
__global__ void k_bfs_keys(const int* __restrict__ parent, const int* __restrict__ range_first, int n, unsigned long long* __restrict__ keys,
                           uint32_t* __restrict__ ids) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2 * n - 1) return;
    uint32_t depth = 0;
    for (int p = parent[id]; p >= 0; p = parent[p]) ++depth;
    const uint32_t first = id >= n - 1 ? (uint32_t)(id - (n - 1)) : (uint32_t)range_first[id];
    keys[id] = ((unsigned long long)depth << 32) | first;
    ids[id] = (uint32_t)id;
}
__global__ void k_bfs_pos(const uint32_t* __restrict__ order, uint32_t total, uint32_t* __restrict__ pos) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < total) pos[order[p]] = p;
}
// FlatNode array in BFS order: links for every node, boxes for the leaves (slot j = j-th triangle in Morton order).
__global__ void k_flatten(const uint32_t* __restrict__ order, const uint32_t* __restrict__ pos, const int2* __restrict__ child,
                          const unsigned long long* __restrict__ sorted, const float* __restrict__ leaf_box, int n,
                          crt_flatnode* __restrict__ flat, uint32_t* __restrict__ bad) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (uint32_t)(2 * n - 1)) return;
    const int id = (int)order[p];
    crt_flatnode f;
    if (id >= n - 1) {
        const int leaf = id - (n - 1);
        const float* bx = leaf_box + 6 * (size_t)(uint32_t)(sorted[leaf] & 0xffffffffull);
        f.bmin[0] = bx[0]; f.bmin[1] = bx[1]; f.bmin[2] = bx[2];
        f.bmax[0] = bx[3]; f.bmax[1] = bx[4]; f.bmax[2] = bx[5];
        f.bmin[3] = crt::link_enc((uint32_t)leaf);                        // leaf slot = position in Morton order
        f.bmax[3] = 1.0f;
    } else {
        const int2 c = child[id];
        const uint32_t l = pos[c.x], r = pos[c.y];
        if (r != l + 1u || l <= p) atomicOr(bad, 1u);   // children adjacent and after their parent
        f.bmin[0] = f.bmin[1] = f.bmin[2] = 0.f; f.bmax[0] = f.bmax[1] = f.bmax[2] = 0.f;   // set by k_refit_level
        f.bmin[3] = crt::link_enc(l);
        f.bmax[3] = 0.0f;
    }
    flat[p] = f;
}
// level_start[d] = first BFS position of depth d (the sorted keys carry the depth in their high word)
__global__ void k_level_starts(const unsigned long long* __restrict__ sorted_keys, uint32_t total, uint32_t* __restrict__ level_start, uint32_t cap,
                               uint32_t shift = 32u) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const uint32_t d = (uint32_t)(sorted_keys[p] >> shift);
    if ((p == 0u || (uint32_t)(sorted_keys[p - 1] >> shift) != d) && d < cap) level_start[d] = p;
}
// Bottom-up refit, one launch per level of the BFS-ordered array, deepest first: an interior node's children lie in
// the next level, already final.  Plain loads and stores — the kernel boundary orders them.  (The first version
// climbed from the leaves with an agent-scope acq_rel arrival counter per node: 3.3 ms of a 5 ms build at 1 M
// triangles, against ~0.3 ms for ~30 small launches.)
__global__ void k_refit_level(crt_flatnode* __restrict__ flat, uint32_t begin, uint32_t end) {
    const uint32_t p = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    crt_flatnode f = flat[p];
    if (f.bmax[3] != 0.0f) return;                      // leaf
    const uint32_t l = (uint32_t)crt::link_of(f.bmin[3]);
    const crt_flatnode a = flat[l], b = flat[l + 1];
    for (int k = 0; k < 3; ++k) { f.bmin[k] = fminf(a.bmin[k], b.bmin[k]); f.bmax[k] = fmaxf(a.bmax[k], b.bmax[k]); }
    flat[p] = f;
}


// ------------------------------------------------------------------ PLOC -------------
// Parallel locally-ordered clustering (Meister & Bittner 2018): bottom-up agglomeration over the Morton-sorted cluster array.
// Every iteration each cluster looks `radius` positions to either side for the partner that gives the smallest merged box,
// mutual nearest neighbours merge, the array is compacted; ~35 % of the clusters disappear per iteration.  The tree quality
// is that of a SAH sweep builder rather than of a spatial-median split (LBVH): VERDICT r1 item 7.  Everything is
// deterministic: node numbers and compacted positions come from one prefix sum, ties go to the nearer position, buddy first.

struct PlocNodes { float4* lo; float4* hi; int* parent2; };      // lo.w = left child (int bits, -1 = leaf), hi.w = right child / leaf slot; parent2 = 2 * parent + side

__device__ __forceinline__ float half_area_union(const float* a_lo, const float* a_hi, const float* b_lo, const float* b_hi) {
    const float dx = fmaxf(a_hi[0], b_hi[0]) - fminf(a_lo[0], b_lo[0]);
    const float dy = fmaxf(a_hi[1], b_hi[1]) - fminf(a_lo[1], b_lo[1]);
    const float dz = fmaxf(a_hi[2], b_hi[2]) - fminf(a_lo[2], b_lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_ploc_init(const unsigned long long* __restrict__ sorted, const float* __restrict__ leaf_box, uint32_t n, PlocNodes nd, int* __restrict__ C) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* bx = leaf_box + 6 * (size_t)(uint32_t)(sorted[i] & 0xffffffffull);
    nd.lo[i] = make_float4(bx[0], bx[1], bx[2], __int_as_float(-1));
    nd.hi[i] = make_float4(bx[3], bx[4], bx[5], __int_as_float((int)i));
    C[i] = (int)i;
}

#define PLOC_MAX_RADIUS 64
// nearest neighbour of every cluster inside the window: one workgroup = 256 consecutive clusters, their boxes (+ the halo)
// staged in LDS
__global__ void __launch_bounds__(256) k_ploc_nn(const int* __restrict__ C, uint32_t m, PlocNodes nd, int radius, int* __restrict__ nn) {
    __shared__ float s_box[(256 + 2 * PLOC_MAX_RADIUS) * 6];
    const int base = (int)(blockIdx.x * 256u);
    for (int t = (int)threadIdx.x; t < 256 + 2 * radius; t += 256) {
        const int g = base - radius + t;
        if (g >= 0 && g < (int)m) {
            const int c = C[g];
            const float4 lo = nd.lo[c], hi = nd.hi[c];
            float* b = s_box + 6 * t;
            b[0] = lo.x; b[1] = lo.y; b[2] = lo.z; b[3] = hi.x; b[4] = hi.y; b[5] = hi.z;
        }
    }
    __syncthreads();
    const int i = base + (int)threadIdx.x;
    if (i >= (int)m) return;
    const float* me = s_box + 6 * ((int)threadIdx.x + radius);
    float best = 3.0e38f;
    int bj = -1;
    // candidates nearest first, the "buddy" (i ^ 1) before the other neighbour: on ties (coincident or grid-regular boxes) the
    // array pairs up (0,1)(2,3)... instead of everyone pointing at the lowest position, which merged ONE pair per iteration
    const int first = (i & 1) ? -1 : 1;
    for (int dist = 1; dist <= radius; ++dist)
        for (int side = 0; side < 2; ++side) {
            const int dj = side == 0 ? first * dist : -first * dist;
            const int j = i + dj;
            if (j < 0 || j >= (int)m) continue;
            const float* o = s_box + 6 * ((int)threadIdx.x + radius + dj);
            const float a = half_area_union(me, me + 3, o, o + 3);
            if (a < best || bj < 0) { best = a < 3.0e38f ? a : 3.0e38f; bj = j; }   // ties keep the earlier candidate; every cluster gets a neighbour
        }
    nn[i] = bj;
}

// keep | merge << 32 per cluster: a mutual pair survives at its lower position as the merged node
__global__ void k_ploc_flags(const int* __restrict__ nn, uint32_t m, unsigned long long* __restrict__ f) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == (int)i;
    const bool keep = !(mutual && (int)i > j), merge = mutual && (int)i < j;
    f[i] = (unsigned long long)(keep ? 1u : 0u) | ((unsigned long long)(merge ? 1u : 0u) << 32);
}

__global__ void k_ploc_apply(const int* __restrict__ C, const int* __restrict__ nn, const unsigned long long* __restrict__ f,
                             const unsigned long long* __restrict__ scan, uint32_t m, uint32_t node_base, PlocNodes nd, int* __restrict__ Cn,
                             uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const unsigned long long fi = f[i], si = scan[i];
    if (fi & 1ull) {
        const uint32_t pos = (uint32_t)(si & 0xffffffffull);
        int c = C[i];
        if (fi >> 32) {
            const int k = (int)(node_base + (uint32_t)(si >> 32));
            const int a = c, b = C[nn[i]];
            const float4 alo = nd.lo[a], ahi = nd.hi[a], blo = nd.lo[b], bhi = nd.hi[b];
            nd.lo[k] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), __int_as_float(a));
            nd.hi[k] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), __int_as_float(b));
            nd.parent2[a] = 2 * k;
            nd.parent2[b] = 2 * k + 1;
            c = k;
        }
        Cn[pos] = c;
    }
    if (i == m - 1u) {
        const unsigned long long tot = si + fi;
        counts[0] = (uint32_t)(tot & 0xffffffffull);
        counts[1] = node_base + (uint32_t)(tot >> 32);
    }
}

// the last <= 1024 clusters: every remaining iteration inside one workgroup (boxes and cluster ids in LDS)
__global__ void __launch_bounds__(1024) k_ploc_tail(const int* __restrict__ C, uint32_t m0, uint32_t node_base, int radius, PlocNodes nd,
                                                    uint32_t* __restrict__ counts) {
    __shared__ float s_box[2][1024 * 6];
    __shared__ int s_c[2][1024];
    __shared__ int s_nn[1024];
    __shared__ uint32_t s_wave[2][16];
    __shared__ uint32_t s_tot[2];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cur = 0;
    uint32_t m = m0, nodes = node_base;
    if (tid < (int)m) {
        const int c = C[tid];
        const float4 lo = nd.lo[c], hi = nd.hi[c];
        float* b = s_box[0] + 6 * tid;
        b[0] = lo.x; b[1] = lo.y; b[2] = lo.z; b[3] = hi.x; b[4] = hi.y; b[5] = hi.z;
        s_c[0][tid] = c;
    }
    __syncthreads();
    while (m > 1u) {
        int bj = -1;
        if (tid < (int)m) {
            const float* me = s_box[cur] + 6 * tid;
            float best = 3.0e38f;
            const int first = (tid & 1) ? -1 : 1;
            for (int dist = 1; dist <= radius; ++dist)
                for (int side = 0; side < 2; ++side) {
                    const int j = tid + (side == 0 ? first * dist : -first * dist);
                    if (j < 0 || j >= (int)m) continue;
                    const float* o = s_box[cur] + 6 * j;
                    const float a = half_area_union(me, me + 3, o, o + 3);
                    if (a < best || bj < 0) { best = a < 3.0e38f ? a : 3.0e38f; bj = j; }
                }
            s_nn[tid] = bj;
        }
        __syncthreads();
        // Progress guard: an iteration in which no two clusters chose each other (possible only when areas tie in a cycle)
        // merges positions 0 and 1 instead, so the loop always ends after at most m - 1 iterations.
        bool keep = false, merge = false;
        uint32_t pk = 0, pm = 0, tk = 0, tm = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            keep = false; merge = false;
            if (tid < (int)m) {
                if (attempt == 1) { if (tid == 0) bj = 1; else if (tid == 1) bj = 0; else bj = -1; }
                const bool mutual = attempt == 1 ? tid < 2 : (bj >= 0 && s_nn[bj] == tid);
                keep = !(mutual && tid > bj);
                merge = mutual && tid < bj;
            }
            // exclusive prefix sums of keep and merge over the workgroup: wave ballots + per-wave totals
            const unsigned long long bk = __ballot(keep), bm = __ballot(merge);
            const unsigned long long lt = (1ull << lane) - 1ull;
            pk = (uint32_t)__builtin_popcountll(bk & lt); pm = (uint32_t)__builtin_popcountll(bm & lt);
            if (lane == 0) { s_wave[0][wave] = (uint32_t)__builtin_popcountll(bk); s_wave[1][wave] = (uint32_t)__builtin_popcountll(bm); }
            __syncthreads();
            tk = 0; tm = 0;
            for (int w = 0; w < 16; ++w) {
                const uint32_t a = s_wave[0][w], b = s_wave[1][w];
                if (w < wave) { pk += a; pm += b; }
                tk += a; tm += b;
            }
            if (tm != 0u) break;                 // uniform: every thread sums the same LDS words
            __syncthreads();                     // s_wave is rewritten by the forced attempt
        }
        if (keep) {
            int c = s_c[cur][tid];
            const float* me = s_box[cur] + 6 * tid;
            float* out = s_box[cur ^ 1] + 6 * pk;
            if (merge) {
                const int k = (int)(nodes + pm);
                const int a = c, b = s_c[cur][bj];
                const float* o = s_box[cur] + 6 * bj;
                const float lo0 = fminf(me[0], o[0]), lo1 = fminf(me[1], o[1]), lo2 = fminf(me[2], o[2]);
                const float hi0 = fmaxf(me[3], o[3]), hi1 = fmaxf(me[4], o[4]), hi2 = fmaxf(me[5], o[5]);
                nd.lo[k] = make_float4(lo0, lo1, lo2, __int_as_float(a));
                nd.hi[k] = make_float4(hi0, hi1, hi2, __int_as_float(b));
                nd.parent2[a] = 2 * k;
                nd.parent2[b] = 2 * k + 1;
                out[0] = lo0; out[1] = lo1; out[2] = lo2; out[3] = hi0; out[4] = hi1; out[5] = hi2;
                c = k;
            } else {
                for (int q = 0; q < 6; ++q) out[q] = me[q];
            }
            s_c[cur ^ 1][pk] = c;
        }
        if (tid == 0) { s_tot[0] = tk; s_tot[1] = tm; }
        __syncthreads();
        m = s_tot[0];
        nodes += s_tot[1];
        cur ^= 1;
        __syncthreads();
    }
    if (tid == 0) {
        nd.parent2[s_c[cur][0]] = -1;                  // the root
        counts[0] = 1u;
        counts[1] = nodes;
        counts[2] = (uint32_t)s_c[cur][0];
    }
}

// BFS key of every node: (depth, root-to-node path) — within one level the breadth-first queue order is the
// lexicographic order of the paths, and a node's two children are neighbours in it
__global__ void k_ploc_bfs_keys(const int* __restrict__ parent2, uint32_t total, unsigned long long* __restrict__ keys, uint32_t* __restrict__ ids,
                                uint32_t* __restrict__ bad) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    unsigned long long path = 0;
    uint32_t depth = 0;
    for (int p = parent2[id]; p >= 0;) {
        if (depth < 56u) path |= (unsigned long long)(p & 1) << depth;
        ++depth;
        // stop where the key ends (such a tree is refused on the host) and never follow a link that points outside the array
        if (depth > 56u) break;
        if ((uint32_t)(p >> 1) >= total) { atomicOr(bad, 1u); break; }
        p = parent2[p >> 1];
    }
    if (depth > 56u) { atomicOr(bad, 2u); depth = 56u; }
    keys[id] = ((unsigned long long)depth << 56) | path;
    ids[id] = id;
}
__global__ void k_ploc_flatten(const uint32_t* __restrict__ order, const uint32_t* __restrict__ pos, PlocNodes nd, uint32_t total,
                               crt_flatnode* __restrict__ flat, uint32_t* __restrict__ bad) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const uint32_t id = order[p];
    const float4 lo = nd.lo[id], hi = nd.hi[id];
    crt_flatnode f;
    f.bmin[0] = lo.x; f.bmin[1] = lo.y; f.bmin[2] = lo.z;
    f.bmax[0] = hi.x; f.bmax[1] = hi.y; f.bmax[2] = hi.z;
    const int left = __float_as_int(lo.w);
    if (left < 0) {
        f.bmin[3] = crt::link_enc((uint32_t)__float_as_int(hi.w));        // leaf slot = position in Morton order
        f.bmax[3] = 1.0f;
    } else {
        const uint32_t l = pos[left], r = pos[__float_as_int(hi.w)];
        if (r != l + 1u || l <= p) atomicOr(bad, 1u);
        f.bmin[3] = crt::link_enc(l);
        f.bmax[3] = 0.0f;
    }
    flat[p] = f;
}

// ------------------------------------------------------------------ binned SAH -------------
// Top-down surface-area-heuristic build, the GPU counterpart of the reference's sweep (sbvh.h:338-378) without spatial
// splits: a numpy prototype of exactly this algorithm gave 5.445 node8 per primary ray on the 1,004,672-triangle mesh against
// 5.419 for the host SBVH and 5.709 for the LBVH (PLOC: 5.99-6.58).  Two phases:
//   A  breadth-first over the nodes with more than SAH_SMALL triangles: centroid bounds, 16 bins per axis (count + box),
//      45 candidate planes per node, stable partition by one prefix sum.  A workgroup keeps the bins of the node its first
//      triangle belongs to in LDS (upper levels: the only node it sees) and falls back to global atomics for the others.
//   B  every node of at most SAH_SMALL triangles is finished by ONE thread with the exact sweep over all three axes.
// Node numbers come from an atomic counter (any order): the final numbering is the canonical breadth-first one of
// k_ploc_bfs_keys (depth, root-to-node path), so the output is deterministic; boxes come from the level-by-level refit.
#define SAH_BINS 16
#define SAH_SMALL 32                     // capacity of the per-thread arrays of phase B; the threshold itself is a parameter (default 8)
#define SAH_NONE 0xffffffffu
#define SAH_BIN_WORDS (3 * SAH_BINS * 7)

struct SahWork {
    uint32_t node, beg, end;
    uint32_t cb[6];                       // centroid bounds: [0..2] = ~ord(min), [3..5] = ord(max) — all grown by atomicMax, 0 = empty
    int axis, plane;                      // chosen split: bin(axis) < plane goes left; axis < 0: split by position (degenerate)
    uint32_t n_left, left_work, right_work;
};

__device__ __forceinline__ void sah_centroid(const float* __restrict__ leaf_box, uint32_t tri, float c[3]) {
    const float* b = leaf_box + 6 * (size_t)tri;
    c[0] = 0.5f * (b[0] + b[3]); c[1] = 0.5f * (b[1] + b[4]); c[2] = 0.5f * (b[2] + b[5]);
}
__device__ __forceinline__ int sah_bin(float c, float cmin, float cmax) {
    const float ext = cmax - cmin;
    if (!(ext > 0.f)) return 0;
    int b = (int)((c - cmin) * ((float)SAH_BINS / ext));
    return b < 0 ? 0 : b > SAH_BINS - 1 ? SAH_BINS - 1 : b;
}

__global__ void k_sah_init(uint32_t n, uint32_t* __restrict__ idx, uint32_t* __restrict__ pwork, SahWork* __restrict__ work, int* __restrict__ parent2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { idx[i] = i; pwork[i] = 0u; }
    if (i == 0u) {
        SahWork w{};
        w.node = 0u; w.beg = 0u; w.end = n;
        work[0] = w;
        parent2[0] = -1;
    }
}

// A workgroup covers 256 consecutive positions.  Triangles are sorted by node and an active node owns more than `small`
// (>= 8) of them, so a workgroup sees at most SAH_LOCAL distinct active nodes: each gets a slot in LDS (slot = number of
// node boundaries before the position), everything is accumulated there and flushed with one global atomic per touched word.
#define SAH_LOCAL 34
__device__ __forceinline__ uint32_t sah_local_slot(uint32_t w, uint32_t w_prev, bool in_range, uint32_t* s_scan, uint32_t* s_node) {
    // boundary = first position of the chunk, or a different work item than the previous position
    const uint32_t tid = threadIdx.x;
    const bool valid = in_range && w != SAH_NONE;
    const bool boundary = valid && (tid == 0u || w_prev != w);
    const unsigned long long b = __ballot(boundary);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    if (lane == 0u) s_scan[wave] = (uint32_t)__builtin_popcountll(b);
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t k = 0; k < wave; ++k) before += s_scan[k];
    const uint32_t slot = before + (uint32_t)__builtin_popcountll(b & ((2ull << lane) - 1ull)) - 1u;     // inclusive count - 1
    if (boundary && slot < SAH_LOCAL) s_node[slot] = w;
    return valid ? slot : SAH_NONE;
}

// A workgroup walks SAH_CHUNKS consecutive chunks of 256 positions.  While the chunks lie inside ONE node (the upper
// levels: node ranges of thousands of positions) its LDS accumulators are carried from chunk to chunk and flushed once:
// at level 0 every workgroup flushes into the same words, and ~90 atomics per microsecond per address made the flush —
// not the binning — the cost of the upper levels (cbounds 51 us, bins 102 us at 4096 workgroups).
#define SAH_CHUNKS 4
__global__ void __launch_bounds__(256) k_sah_cbounds(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ pwork, uint32_t n,
                                                     const float* __restrict__ leaf_box, SahWork* __restrict__ work) {
    __shared__ uint32_t s_cb[SAH_LOCAL * 6];
    __shared__ uint32_t s_node[SAH_LOCAL];
    __shared__ uint32_t s_scan[4];
    // all four chunks' loads are issued before the first barrier: the kernel is a chain of dependent loads (work item ->
    // triangle -> box) at three workgroups per CU, and was bound by that latency, not by the atomics
    uint32_t wv[SAH_CHUNKS], wp[SAH_CHUNKS], tri[SAH_CHUNKS];
    float cen[SAH_CHUNKS][3];
#pragma unroll
    for (int c = 0; c < SAH_CHUNKS; ++c) {
        const uint32_t pos = (blockIdx.x * SAH_CHUNKS + c) * 256u + threadIdx.x;
        wv[c] = pos < n ? pwork[pos] : SAH_NONE;
        wp[c] = (pos < n && threadIdx.x != 0u) ? pwork[pos - 1u] : SAH_NONE;
    }
#pragma unroll
    for (int c = 0; c < SAH_CHUNKS; ++c) {
        const uint32_t pos = (blockIdx.x * SAH_CHUNKS + c) * 256u + threadIdx.x;
        tri[c] = wv[c] != SAH_NONE ? idx[pos] : 0u;
    }
#pragma unroll
    for (int c = 0; c < SAH_CHUNKS; ++c) {
        cen[c][0] = cen[c][1] = cen[c][2] = 0.f;
        if (wv[c] != SAH_NONE) sah_centroid(leaf_box, tri[c], cen[c]);
    }
    uint32_t carried = SAH_NONE;                                     // node whose bounds are live in slot 0 (uniform)
#pragma unroll
    for (int chunk = 0; chunk < SAH_CHUNKS; ++chunk) {
        const uint32_t pos = (blockIdx.x * SAH_CHUNKS + chunk) * 256u + threadIdx.x;
        __syncthreads();
        if (threadIdx.x < SAH_LOCAL) s_node[threadIdx.x] = SAH_NONE;
        __syncthreads();
        const uint32_t w = wv[chunk];
        const uint32_t slot = sah_local_slot(w, wp[chunk], pos < n, s_scan, s_node);
        __syncthreads();
        uint32_t n_slots = 0;
        for (uint32_t k = 0; k < 4u; ++k) n_slots += s_scan[k];
        if (n_slots > SAH_LOCAL) n_slots = SAH_LOCAL;
        const bool single = n_slots == 1u;
        const uint32_t node0 = s_node[0];
        if (carried != SAH_NONE && !(single && node0 == carried)) {
            if (threadIdx.x < 6u) { const uint32_t v = s_cb[threadIdx.x]; if (v != 0u) atomicMax(&work[carried].cb[threadIdx.x], v); }
            carried = SAH_NONE;
            __syncthreads();
        }
        if (carried == SAH_NONE) {
            for (uint32_t t = threadIdx.x; t < n_slots * 6u; t += 256u) s_cb[t] = 0u;
            __syncthreads();
        }
        if (slot != SAH_NONE) {
            uint32_t* dst = slot < SAH_LOCAL ? s_cb + 6 * slot : work[w].cb;
            // a bound only grows: most triangles leave it unchanged, and a plain read (same-address reads broadcast) spares the atomic
            for (int k = 0; k < 3; ++k) {
                const uint32_t o = f2ord(cen[chunk][k]);
                if (~o > *(volatile uint32_t*)&dst[k]) atomicMax(&dst[k], ~o);
                if (o > *(volatile uint32_t*)&dst[3 + k]) atomicMax(&dst[3 + k], o);
            }
        }
        __syncthreads();
        if (single) { carried = node0; continue; }
        for (uint32_t t = threadIdx.x; t < n_slots * 6u; t += 256u) {
            const uint32_t v = s_cb[t], node = s_node[t / 6u];
            if (v != 0u && node != SAH_NONE) atomicMax(&work[node].cb[t % 6u], v);
        }
    }
    if (carried != SAH_NONE && threadIdx.x < 6u) { const uint32_t v = s_cb[threadIdx.x]; if (v != 0u) atomicMax(&work[carried].cb[threadIdx.x], v); }
}

// bins[w][axis][bin] = {count, ~ord(lo.xyz), ord(hi.xyz)}: every field grows by atomicAdd / atomicMax, so all-zero = empty
__global__ void __launch_bounds__(256) k_sah_bin(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ pwork, uint32_t n,
                                                 const float* __restrict__ leaf_box, const SahWork* __restrict__ work, uint32_t* __restrict__ bins) {
    __shared__ uint32_t s_bins[SAH_LOCAL * SAH_BIN_WORDS];        // 45.7 KB
    __shared__ uint32_t s_node[SAH_LOCAL];
    __shared__ uint32_t s_scan[4];
    auto flush = [&](uint32_t slots) {
        for (uint32_t t = threadIdx.x; t < slots * SAH_BIN_WORDS; t += 256u) {
            const uint32_t v = s_bins[t];
            if (v == 0u) continue;
            uint32_t* g = bins + (size_t)s_node[t / SAH_BIN_WORDS] * SAH_BIN_WORDS + t % SAH_BIN_WORDS;
            if (t % 7u == 0u) atomicAdd(g, v); else atomicMax(g, v);
        }
    };
    // everything a position needs — work item, triangle box as ordered integers, its three bin numbers — for all four
    // chunks before the first barrier (see k_sah_cbounds)
    uint32_t wv[SAH_CHUNKS], wp[SAH_CHUNKS], tri[SAH_CHUNKS], ob[SAH_CHUNKS][6];
    int bi[SAH_CHUNKS][3];
#pragma unroll
    for (int c = 0; c < SAH_CHUNKS; ++c) {
        const uint32_t pos = (blockIdx.x * SAH_CHUNKS + c) * 256u + threadIdx.x;
        wv[c] = pos < n ? pwork[pos] : SAH_NONE;
        wp[c] = (pos < n && threadIdx.x != 0u) ? pwork[pos - 1u] : SAH_NONE;
    }
#pragma unroll
    for (int c = 0; c < SAH_CHUNKS; ++c) {
        const uint32_t pos = (blockIdx.x * SAH_CHUNKS + c) * 256u + threadIdx.x;
        tri[c] = wv[c] != SAH_NONE ? idx[pos] : 0u;
    }
#pragma unroll
    for (int c = 0; c < SAH_CHUNKS; ++c) {
        for (int k = 0; k < 6; ++k) ob[c][k] = 0u;
        bi[c][0] = bi[c][1] = bi[c][2] = 0;
        if (wv[c] == SAH_NONE) continue;
        const float* b = leaf_box + 6 * (size_t)tri[c];
        const float bx[6] = {b[0], b[1], b[2], b[3], b[4], b[5]};
        const uint32_t* cbw = work[wv[c]].cb;
        for (int k = 0; k < 3; ++k) {
            ob[c][k] = ~f2ord(bx[k]); ob[c][3 + k] = f2ord(bx[3 + k]);
            bi[c][k] = sah_bin(0.5f * (bx[k] + bx[3 + k]), ord2f(~cbw[k]), ord2f(cbw[3 + k]));
        }
    }
    auto accumulate = [&](uint32_t* base, int c) {
        for (int ax = 0; ax < 3; ++ax) {
            uint32_t* d = base + (ax * SAH_BINS + bi[c][ax]) * 7;
            atomicAdd(&d[0], 1u);
            for (int k = 0; k < 6; ++k) if (ob[c][k] > *(volatile uint32_t*)&d[1 + k]) atomicMax(&d[1 + k], ob[c][k]);   // bounds only grow
        }
    };
    uint32_t carried = SAH_NONE;                                     // node whose bins are live in slot 0 (uniform)
#pragma unroll
    for (int chunk = 0; chunk < SAH_CHUNKS; ++chunk) {
        const uint32_t first = (blockIdx.x * SAH_CHUNKS + chunk) * 256u, last = first + 255u, pos = first + threadIdx.x;
        const uint32_t w = wv[chunk];
        // the chunk lies wholly inside node x  <=>  its first and last position belong to x (positions are sorted by node)
        const uint32_t w_last = last < n ? pwork[last] : SAH_NONE;
        __syncthreads();
        if (carried != SAH_NONE) {
            if (w_last == carried && pwork[first] == carried) { accumulate(s_bins, chunk); continue; }
            flush(1u);                                               // s_node[0] is still the carried node
            carried = SAH_NONE;
            __syncthreads();
        }
        if (threadIdx.x < SAH_LOCAL) s_node[threadIdx.x] = SAH_NONE;
        __syncthreads();
        const uint32_t slot = sah_local_slot(w, wp[chunk], pos < n, s_scan, s_node);
        __syncthreads();
        uint32_t n_slots = 0;
        for (uint32_t k = 0; k < 4u; ++k) n_slots += s_scan[k];
        if (n_slots > SAH_LOCAL) n_slots = SAH_LOCAL;
        for (uint32_t t = threadIdx.x; t < n_slots * SAH_BIN_WORDS; t += 256u) s_bins[t] = 0u;
        __syncthreads();
        if (slot != SAH_NONE) accumulate(slot < SAH_LOCAL ? s_bins + (size_t)slot * SAH_BIN_WORDS : bins + (size_t)w * SAH_BIN_WORDS, chunk);
        __syncthreads();
        // a chunk that lies wholly inside one node keeps its bins for the next chunk
        if (n_slots == 1u && w_last != SAH_NONE && w_last == s_node[0] && pwork[first] == w_last && chunk + 1 < SAH_CHUNKS) { carried = w_last; continue; }
        flush(n_slots);
    }
    if (carried != SAH_NONE) { __syncthreads(); flush(1u); }
}

struct SahLists { uint32_t* counters; SahWork* next; uint32_t* small; uint32_t cap_next, cap_small; uint32_t* bad; };   // counters: [0] nodes so far, [1] next-level work items, [2] small nodes (node, beg, end triples), [3] this level's work items

// between two levels: the children of this level's m nodes are numbered, the next list becomes the current one
__global__ void k_sah_advance(uint32_t* __restrict__ counters) {
    counters[0] += 2u * counters[3];
    counters[3] = counters[1];
    counters[1] = 0u;
}

__device__ __forceinline__ float sah_half_area(const float lo[3], const float hi[3]) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// One wave per active node, four nodes per workgroup: 45 candidate planes (lane = axis * 16 + plane) evaluated from the node's
// bins staged in LDS, argmin, children.  Child node numbers are base + 2 * (position in the level's list) — every active node
// splits in two — and the appends to the next level's list / the small-node list are aggregated per workgroup: a returning
// atomic per node on ONE counter was 4 of the builder's 10 ms (the same ~90 atomics per microsecond per address that capped
// round 1's queue appends).
// The number of active nodes m and the first free node number live on the device (counters[3], counters[0]): the host
// launches an upper bound of workgroups and does not wait for a level to finish.  The bins are zeroed again as they are
// staged, so the array is cleared once per build, not once per level.
__global__ void __launch_bounds__(256) k_sah_sweep(SahWork* __restrict__ work, uint32_t* __restrict__ bins, PlocNodes nd, SahLists out, uint32_t small) {
    __shared__ uint32_t s_b[4][SAH_BIN_WORDS];
    __shared__ uint32_t s_cnt[4][2];          // per wave: entries for the next-level list, for the small list
    __shared__ uint32_t s_base[2];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t m = out.counters[3], node_base = out.counters[0];
    if (blockIdx.x * 4u >= m) return;                               // whole workgroup beyond the list (uniform)
    const uint32_t w = blockIdx.x * 4u + wave;
    const bool live = w < m;
    SahWork wk{};
    uint32_t n_left = 0, cnt = 0;
    float best = 3.0e38f; uint32_t bl = 0;
    if (live) {
        wk = work[w];
        cnt = wk.end - wk.beg;
        uint32_t* B = bins + (size_t)w * SAH_BIN_WORDS;
        for (uint32_t t = lane; t < SAH_BIN_WORDS; t += 64u) { const uint32_t v = B[t]; s_b[wave][t] = v; if (v) B[t] = 0u; }
    }
    __builtin_amdgcn_wave_barrier();
    if (live) {
        // lane = axis * 16 + bin: the box and count left of plane p are an inclusive prefix over the 16 lanes of an axis
        // taken from lane p - 1, those right of it an inclusive suffix — min / max / integer sums, so the order of the
        // combination does not change a bit of the result (same tree as the O(16) loop per lane it replaces; at the deep levels
        // the kernel is bound by the 1.3 KB of bins per node it reads, not by this arithmetic)
        const int ax = (int)(lane >> 4), p = (int)(lane & 15u);
        float cost = 3.0e38f;
        uint32_t nl = 0;
        {
            const uint32_t* d = s_b[wave] + ((ax < 3 ? ax : 0) * SAH_BINS + p) * 7;
            const uint32_t c0 = ax < 3 ? d[0] : 0u;
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) { lo[k] = c0 ? ord2f(~d[1 + k]) : 3e38f; hi[k] = c0 ? ord2f(d[4 + k]) : -3e38f; }
            uint32_t cl = c0, cr = c0;
            float llo[3] = {lo[0], lo[1], lo[2]}, lhi[3] = {hi[0], hi[1], hi[2]}, rlo[3] = {lo[0], lo[1], lo[2]}, rhi[3] = {hi[0], hi[1], hi[2]};
            for (int dlt = 1; dlt < SAH_BINS; dlt <<= 1) {
                const uint32_t ucl = __shfl_up(cl, dlt, SAH_BINS), ucr = __shfl_down(cr, dlt, SAH_BINS);
                float ul[3], uh[3], dl[3], dh[3];
                for (int k = 0; k < 3; ++k) {
                    ul[k] = __shfl_up(llo[k], dlt, SAH_BINS); uh[k] = __shfl_up(lhi[k], dlt, SAH_BINS);
                    dl[k] = __shfl_down(rlo[k], dlt, SAH_BINS); dh[k] = __shfl_down(rhi[k], dlt, SAH_BINS);
                }
                if (p >= dlt) { cl += ucl; for (int k = 0; k < 3; ++k) { llo[k] = fminf(llo[k], ul[k]); lhi[k] = fmaxf(lhi[k], uh[k]); } }
                if (p + dlt < SAH_BINS) { cr += ucr; for (int k = 0; k < 3; ++k) { rlo[k] = fminf(rlo[k], dl[k]); rhi[k] = fmaxf(rhi[k], dh[k]); } }
            }
            // left of plane p = bins [0, p): the inclusive prefix of lane p - 1
            const uint32_t pl = __shfl_up(cl, 1, SAH_BINS);
            float pll[3], plh[3];
            for (int k = 0; k < 3; ++k) { pll[k] = __shfl_up(llo[k], 1, SAH_BINS); plh[k] = __shfl_up(lhi[k], 1, SAH_BINS); }
            if (ax < 3 && p >= 1) {
                nl = pl;
                const uint32_t nr = cr;
                if (nl != 0u && nr != 0u) cost = sah_half_area(pll, plh) * (float)nl + sah_half_area(rlo, rhi) * (float)nr;
            }
        }
        // argmin over the wave, ties to the lower lane (axis, then plane): deterministic
        best = cost; bl = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const float oc = __shfl_down(best, off); const uint32_t ol = __shfl_down(bl, off);
            if (oc < best || (oc == best && ol < bl)) { best = oc; bl = ol; }
        }
        best = __shfl(best, 0); bl = __shfl(bl, 0);
        n_left = best < 3.0e38f ? __shfl(nl, (int)bl) : cnt / 2u;
    }
    // classify the two children: 0 leaf, 1 small, 2 active
    uint32_t cls[2] = {0u, 0u};
    if (live && lane == 0u) {
        const uint32_t cc[2] = {n_left, cnt - n_left};
        for (int side = 0; side < 2; ++side) cls[side] = cc[side] == 1u ? 0u : cc[side] <= small ? 1u : 2u;
        s_cnt[wave][0] = (cls[0] == 2u) + (cls[1] == 2u);
        s_cnt[wave][1] = (cls[0] == 1u) + (cls[1] == 1u);
    } else if (lane == 0u) { s_cnt[wave][0] = 0u; s_cnt[wave][1] = 0u; }
    __syncthreads();
    if (threadIdx.x < 2u) {
        const uint32_t tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        s_base[threadIdx.x] = tot ? atomicAdd(&out.counters[1u + threadIdx.x], tot) : 0u;
    }
    __syncthreads();
    if (!live || lane != 0u) return;
    uint32_t at[2] = {s_base[0], s_base[1]};
    for (uint32_t k = 0; k < wave; ++k) { at[0] += s_cnt[k][0]; at[1] += s_cnt[k][1]; }
    wk.axis = best < 3.0e38f ? (int)(bl >> 4) : -1;
    wk.plane = (int)(bl & 15u);
    wk.n_left = n_left;
    const uint32_t kids = node_base + 2u * w;
    nd.lo[wk.node].w = __int_as_float((int)kids);
    nd.hi[wk.node].w = __int_as_float((int)kids + 1);
    nd.parent2[kids] = 2 * (int)wk.node;
    nd.parent2[kids + 1u] = 2 * (int)wk.node + 1;
    uint32_t cw[2] = {SAH_NONE, SAH_NONE};
    for (int side = 0; side < 2; ++side) {
        const uint32_t cb = side == 0 ? wk.beg : wk.beg + n_left, ce = side == 0 ? wk.beg + n_left : wk.end, child = kids + (uint32_t)side;
        if (cls[side] == 0u) {
            nd.lo[child].w = __int_as_float(-1);
            nd.hi[child].w = __int_as_float((int)cb);              // leaf slot = final position of its triangle
        } else if (cls[side] == 1u) {
            // small nodes own disjoint ranges of >= 2 triangles (at most n / 2 of them over the whole build), active ones disjoint
            // ranges of more than `small` (at most n / small per level): the lists are sized for that, and an append that would
            // fall outside anyway is dropped and flagged instead of written
            const uint32_t k = at[1]++;
            if (k < out.cap_small) { out.small[3 * (size_t)k] = child; out.small[3 * (size_t)k + 1] = cb; out.small[3 * (size_t)k + 2] = ce; }
            else atomicOr(out.bad, 4u);
        } else {
            const uint32_t k = at[0]++;
            SahWork c{};
            c.node = child; c.beg = cb; c.end = ce;
            if (k < out.cap_next) { out.next[k] = c; cw[side] = k; }
            else atomicOr(out.bad, 4u);
        }
    }
    wk.left_work = cw[0]; wk.right_work = cw[1];
    work[w] = wk;
}

__device__ __forceinline__ bool sah_goes_left(const SahWork& wk, uint32_t pos, uint32_t tri, const float* __restrict__ leaf_box) {
    if (wk.axis < 0) return pos - wk.beg < wk.n_left;
    float c[3];
    sah_centroid(leaf_box, tri, c);
    return sah_bin(c[wk.axis], ord2f(~wk.cb[wk.axis]), ord2f(wk.cb[3 + wk.axis])) < wk.plane;
}
__global__ void k_sah_flags(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ pwork, uint32_t n, const float* __restrict__ leaf_box,
                            const SahWork* __restrict__ work, uint32_t* __restrict__ flags) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const uint32_t w = pwork[pos];
    flags[pos] = w != SAH_NONE && sah_goes_left(work[w], pos, idx[pos], leaf_box) ? 1u : 0u;
}
__global__ void k_sah_scatter(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ pwork, uint32_t n, const SahWork* __restrict__ work,
                              const uint32_t* __restrict__ flags, const uint32_t* __restrict__ scan, uint32_t* __restrict__ idx2, uint32_t* __restrict__ pwork2) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const uint32_t w = pwork[pos];
    if (w == SAH_NONE) { idx2[pos] = idx[pos]; pwork2[pos] = SAH_NONE; return; }
    const SahWork wk = work[w];
    const uint32_t left_before = scan[pos] - scan[wk.beg];
    const bool gl = flags[pos] != 0u;
    const uint32_t np = gl ? wk.beg + left_before : wk.beg + wk.n_left + (pos - wk.beg - left_before);
    idx2[np] = idx[pos];
    pwork2[np] = gl ? wk.left_work : wk.right_work;
}

// phase B: one thread finishes a node of <= SAH_SMALL triangles with the exact sweep (all split positions of all three axes)
// 2 (c - 1) new nodes per small node of c triangles; their exclusive prefix gives every thread its own block of node numbers
__global__ void k_sah_small_counts(const uint32_t* __restrict__ small, uint32_t n_small, uint32_t* __restrict__ counts) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_small) counts[t] = 2u * (small[3 * (size_t)t + 2] - small[3 * (size_t)t + 1] - 1u);
}
// The per-thread work arrays (triangles, boxes, centroids, sort order, suffix areas, stack) are columns of LDS arrays
// [entry][lane] — dynamic indexing into private arrays lives in scratch memory (0.92 ms of the 5.3 ms build at 1 M triangles
// with arrays of 32 entries); CAP = 8 / 16 / 32 is the smallest capacity that holds the threshold.
template <int CAP>
__global__ void __launch_bounds__(64) k_sah_small(const uint32_t* __restrict__ small, uint32_t n_small, uint32_t* __restrict__ idx,
                                                  const float* __restrict__ leaf_box, PlocNodes nd, const uint32_t* __restrict__ id_offset, uint32_t node_base) {
    constexpr bool kBoxes = CAP <= 16;                              // the boxes fit as well (24 KB at CAP = 16)
    __shared__ uint32_t s_tri[CAP][64];
    __shared__ float s_cen[3][CAP][64];
    __shared__ float s_box[kBoxes ? 6 : 1][kBoxes ? CAP : 1][64];
    __shared__ float s_rarea[CAP][64];
    __shared__ uint8_t s_ord[CAP][64], s_tmp[CAP][64];
    __shared__ uint32_t s_stack_node[CAP][64];
    __shared__ uint16_t s_stack_be[CAP][64];
    const uint32_t lane = threadIdx.x;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_small) return;
    uint32_t next_id = node_base + id_offset[t];
    const uint32_t root = small[3 * (size_t)t], beg = small[3 * (size_t)t + 1], end = small[3 * (size_t)t + 2];
    const int cnt = (int)(end - beg);
    for (int i = 0; i < cnt; ++i) {
        const uint32_t tr = idx[beg + i];
        const float* bx = leaf_box + 6 * (size_t)tr;
        s_tri[i][lane] = tr;
        for (int k = 0; k < 3; ++k) s_cen[k][i][lane] = 0.5f * (bx[k] + bx[3 + k]);
        if constexpr (kBoxes) for (int k = 0; k < 6; ++k) s_box[k][i][lane] = bx[k];
        s_ord[i][lane] = (uint8_t)i;
    }
    auto grow = [&](int item, float lo[3], float hi[3]) {
        if constexpr (kBoxes) {
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], s_box[k][item][lane]); hi[k] = fmaxf(hi[k], s_box[3 + k][item][lane]); }
        } else {
            const float* bx = leaf_box + 6 * (size_t)s_tri[item][lane];
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], bx[k]); hi[k] = fmaxf(hi[k], bx[3 + k]); }
        }
    };
    int sp = 0;
    s_stack_node[0][lane] = root; s_stack_be[0][lane] = (uint16_t)(cnt << 8); sp = 1;
    while (sp > 0) {
        --sp;
        const uint32_t node = s_stack_node[sp][lane];
        const int b = s_stack_be[sp][lane] & 255, e = s_stack_be[sp][lane] >> 8, c = e - b;
        if (c == 1) {
            nd.lo[node].w = __int_as_float(-1);
            nd.hi[node].w = __int_as_float((int)(beg + (uint32_t)b));
            continue;
        }
        float best = 3.0e38f; int best_ax = 0, best_i = c / 2;
        for (int ax = 0; ax < 3; ++ax) {
            // insertion sort of ord[b..e) by (centroid[ax], triangle index): a strict total order
            for (int i = b + 1; i < e; ++i) {
                const uint8_t v = s_ord[i][lane];
                const float cv = s_cen[ax][v][lane]; const uint32_t tv = s_tri[v][lane];
                int j = i - 1;
                while (j >= b) {
                    const uint8_t o = s_ord[j][lane];
                    const float co = s_cen[ax][o][lane];
                    if (!(co > cv || (co == cv && s_tri[o][lane] > tv))) break;
                    s_ord[j + 1][lane] = o; --j;
                }
                s_ord[j + 1][lane] = v;
            }
            float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
            for (int i = e - 1; i > b; --i) {
                grow(s_ord[i][lane], lo, hi);
                s_rarea[i][lane] = sah_half_area(lo, hi);
            }
            for (int k = 0; k < 3; ++k) { lo[k] = 3e38f; hi[k] = -3e38f; }
            for (int i = b + 1; i < e; ++i) {
                grow(s_ord[i - 1][lane], lo, hi);
                const float cost = sah_half_area(lo, hi) * (float)(i - b) + s_rarea[i][lane] * (float)(e - i);
                if (cost < best) { best = cost; best_ax = ax; best_i = i - b; }
            }
            if (ax == best_ax) for (int i = b; i < e; ++i) s_tmp[i][lane] = s_ord[i][lane];     // remember the winning order
        }
        for (int i = b; i < e; ++i) s_ord[i][lane] = s_tmp[i][lane];
        const uint32_t kids = next_id;
        next_id += 2u;
        nd.lo[node].w = __int_as_float((int)kids);
        nd.hi[node].w = __int_as_float((int)kids + 1);
        nd.parent2[kids] = 2 * (int)node;
        nd.parent2[kids + 1u] = 2 * (int)node + 1;
        s_stack_node[sp][lane] = kids + 1u; s_stack_be[sp][lane] = (uint16_t)((b + best_i) | (e << 8)); ++sp;
        s_stack_node[sp][lane] = kids;      s_stack_be[sp][lane] = (uint16_t)(b | ((b + best_i) << 8)); ++sp;
    }
    for (int i = 0; i < cnt; ++i) idx[beg + i] = s_tri[s_ord[i][lane]][lane];
}

// FlatNode array in the canonical BFS order: links for every node, boxes for the leaves (slot j holds triangle order[j])
__global__ void k_sah_flatten(const uint32_t* __restrict__ order, const uint32_t* __restrict__ pos, PlocNodes nd, const uint32_t* __restrict__ tri_order,
                              const float* __restrict__ leaf_box, uint32_t total, crt_flatnode* __restrict__ flat, uint32_t* __restrict__ bad) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const uint32_t id = order[p];
    const int left = __float_as_int(nd.lo[id].w), right = __float_as_int(nd.hi[id].w);
    crt_flatnode f;
    if (left < 0) {
        const float* bx = leaf_box + 6 * (size_t)tri_order[right];
        f.bmin[0] = bx[0]; f.bmin[1] = bx[1]; f.bmin[2] = bx[2];
        f.bmax[0] = bx[3]; f.bmax[1] = bx[4]; f.bmax[2] = bx[5];
        f.bmin[3] = crt::link_enc((uint32_t)right);
        f.bmax[3] = 1.0f;
    } else {
        const uint32_t l = pos[left], r = pos[right];
        if (r != l + 1u || l <= p) atomicOr(bad, 1u);
        f.bmin[0] = f.bmin[1] = f.bmin[2] = 0.f; f.bmax[0] = f.bmax[1] = f.bmax[2] = 0.f;   // set by k_refit_level
        f.bmin[3] = crt::link_enc(l);
        f.bmax[3] = 0.0f;
    }
    flat[p] = f;
}

constexpr size_t kMaxLevels = 4096;
thread_local float g_last_device_ms = 0.f, g_last_total_ms = 0.f;

}  // namespace

namespace crt {

static size_t sah_tmp_bytes(size_t n, uint32_t flags);
size_t lbvh_tmp_bytes(size_t n_tris, uint32_t flags) {
    const size_t n_nodes = 2 * n_tris - 1;
    if (flags & CRT_GPU_BUILD_SAH) return sah_tmp_bytes(n_tris, flags);
    if (flags & CRT_GPU_BUILD_PLOC) {
        size_t sort1 = 0, sort2 = 0, scan = 0;
        (void)rocprim::radix_sort_keys(nullptr, sort1, (unsigned long long*)nullptr, (unsigned long long*)nullptr, n_tris, 0, 64, (hipStream_t)0);
        (void)rocprim::radix_sort_pairs(nullptr, sort2, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                        (uint32_t*)nullptr, n_nodes, 0, 64, (hipStream_t)0);
        (void)rocprim::exclusive_scan(nullptr, scan, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull, n_tris, rocprim::plus<unsigned long long>(), (hipStream_t)0);
        auto P = DeviceArena::padded;
        return P(n_tris * 24) + P(32) + 2 * P(n_tris * 8) + 2 * P(n_nodes * 16) + P(n_nodes * 4) + 3 * P(n_tris * 4) + 2 * P(n_tris * 8) + P(16) +
               2 * P(n_nodes * 8) + 3 * P(n_nodes * 4) + P(4) + P(std::max<size_t>(sort1, 16)) + P(std::max<size_t>(sort2, 16)) + P(std::max<size_t>(scan, 16)) + 4096;
    }
    size_t sort1 = 0, sort2 = 0;
    (void)rocprim::radix_sort_keys(nullptr, sort1, (unsigned long long*)nullptr, (unsigned long long*)nullptr, n_tris, 0, 64, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, sort2, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                    (uint32_t*)nullptr, n_nodes, 0, 64, (hipStream_t)0);
    auto P = DeviceArena::padded;
    return P(n_tris * 24) + P(32) + 2 * P(n_tris * 8) + P(std::max<size_t>(n_tris - 1, 1) * sizeof(int2)) + P(n_nodes * 4) + P(kMaxLevels * 4) +
           P(std::max<size_t>(n_tris - 1, 1) * 4) + 2 * P(n_nodes * 8) + 3 * P(n_nodes * 4) + P(4) + P(std::max<size_t>(sort1, 16)) +
           P(std::max<size_t>(sort2, 16)) + 4096;
}

static int ploc_build_on_device(const int32_t* d_vidx, uint32_t stride, const float* d_verts, uint32_t n_tris_u, uint32_t flags, DeviceArena& tmp,
                                crt_flatnode* d_flat, uint32_t* d_tri_order, uint32_t* depth_out, float* device_ms, hipStream_t stream) {
    const size_t n_tris = n_tris_u, n_nodes = 2 * n_tris - 1;
    const uint32_t n = n_tris_u;
    int radius = (int)((flags >> 8) & 0xffu);
    if (radius == 0) radius = 16;
    if (radius > PLOC_MAX_RADIUS) radius = PLOC_MAX_RADIUS;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    };
    size_t sort1 = 0, sort2 = 0, scan_bytes = 0;
    LB_HIPCHK(rocprim::radix_sort_keys(nullptr, sort1, (unsigned long long*)nullptr, (unsigned long long*)nullptr, n_tris, 0, 64, stream));
    LB_HIPCHK(rocprim::radix_sort_pairs(nullptr, sort2, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                        (uint32_t*)nullptr, n_nodes, 0, 64, stream));
    LB_HIPCHK(rocprim::exclusive_scan(nullptr, scan_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull, n_tris,
                                      rocprim::plus<unsigned long long>(), stream));
    float* d_leaf_box = tmp.take<float>(n_tris * 6);
    uint32_t* d_scene = tmp.take<uint32_t>(8);
    unsigned long long* d_keys = tmp.take<unsigned long long>(n_tris);
    unsigned long long* d_sorted = tmp.take<unsigned long long>(n_tris);
    PlocNodes nd;
    nd.lo = tmp.take<float4>(n_nodes);
    nd.hi = tmp.take<float4>(n_nodes);
    nd.parent2 = tmp.take<int>(n_nodes);
    int* d_c0 = tmp.take<int>(n_tris);
    int* d_c1 = tmp.take<int>(n_tris);
    int* d_nn = tmp.take<int>(n_tris);
    unsigned long long* d_f = tmp.take<unsigned long long>(n_tris);
    unsigned long long* d_scan = tmp.take<unsigned long long>(n_tris);
    uint32_t* d_counts = tmp.take<uint32_t>(4);
    unsigned long long* d_bkeys = tmp.take<unsigned long long>(n_nodes);
    unsigned long long* d_bkeys2 = tmp.take<unsigned long long>(n_nodes);
    uint32_t* d_ids = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_order = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_pos = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_bad = tmp.take<uint32_t>(1);
    void* d_tmp = tmp.take<char>(std::max<size_t>(sort1, 16));
    void* d_tmp2 = tmp.take<char>(std::max<size_t>(sort2, 16));
    void* d_tmp3 = tmp.take<char>(std::max<size_t>(scan_bytes, 16));
    if (!d_leaf_box || !d_scene || !d_keys || !d_sorted || !nd.lo || !nd.hi || !nd.parent2 || !d_c0 || !d_c1 || !d_nn || !d_f || !d_scan || !d_counts ||
        !d_bkeys || !d_bkeys2 || !d_ids || !d_order || !d_pos || !d_bad || !d_tmp || !d_tmp2 || !d_tmp3)
        return fail(CRT_ERR_NOMEM, "ploc: temporary arena too small");
    const uint32_t scene_init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    LB_HIPCHK(hipEventCreate(&ev0));
    LB_HIPCHK(hipEventCreate(&ev1));
    LB_HIPCHK(hipMemcpyAsync(d_scene, scene_init, sizeof scene_init, hipMemcpyHostToDevice, stream));
    LB_HIPCHK(hipMemsetAsync(d_bad, 0, 4, stream));
    const uint32_t g = (uint32_t)((n_tris + 255) / 256);
    LB_HIPCHK(hipEventRecord(ev0, stream));
    hipLaunchKernelGGL(k_tri_bounds, dim3(std::min<uint32_t>(g, 1024u)), dim3(256), 0, stream, d_vidx, stride, d_verts, n, d_leaf_box, d_scene);
    hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, stream, d_leaf_box, d_scene, n, d_keys);
    LB_HIPCHK(rocprim::radix_sort_keys(d_tmp, sort1, d_keys, d_sorted, n_tris, 0, 64, stream));
    hipLaunchKernelGGL(k_tri_order, dim3(g), dim3(256), 0, stream, d_sorted, n, d_tri_order);
    hipLaunchKernelGGL(k_ploc_init, dim3(g), dim3(256), 0, stream, d_sorted, d_leaf_box, n, nd, d_c0);
    uint32_t m = n, nodes = n, iterations = 0;
    int* C = d_c0; int* Cn = d_c1;
    while (m > 1024u) {
        const dim3 gm((m + 255u) / 256u);
        hipLaunchKernelGGL(k_ploc_nn, gm, dim3(256), 0, stream, C, m, nd, radius, d_nn);
        hipLaunchKernelGGL(k_ploc_flags, gm, dim3(256), 0, stream, d_nn, m, d_f);
        LB_HIPCHK(rocprim::exclusive_scan(d_tmp3, scan_bytes, d_f, d_scan, 0ull, (size_t)m, rocprim::plus<unsigned long long>(), stream));
        hipLaunchKernelGGL(k_ploc_apply, gm, dim3(256), 0, stream, C, d_nn, d_f, d_scan, m, nodes, nd, Cn, d_counts);
        uint32_t counts[2] = {0, 0};
        LB_HIPCHK(hipMemcpyAsync(counts, d_counts, 8, hipMemcpyDeviceToHost, stream));
        LB_HIPCHK(hipStreamSynchronize(stream));
        if (counts[0] >= m || counts[0] == 0u) { cleanup(); return fail(CRT_ERR_INVALID, "ploc: an iteration merged nothing (cyclic ties between cluster areas)"); }
        m = counts[0]; nodes = counts[1];
        std::swap(C, Cn);
        if (++iterations > 4096u) { cleanup(); return fail(CRT_ERR_HIP, "ploc: did not converge"); }
    }
    hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(1024), 0, stream, C, m, nodes, radius, nd, d_counts);
    const dim3 gn((uint32_t)((n_nodes + 255) / 256));
    hipLaunchKernelGGL(k_ploc_bfs_keys, gn, dim3(256), 0, stream, nd.parent2, (uint32_t)n_nodes, d_bkeys, d_ids, d_bad);
    LB_HIPCHK(rocprim::radix_sort_pairs(d_tmp2, sort2, d_bkeys, d_bkeys2, d_ids, d_order, n_nodes, 0, 64, stream));
    hipLaunchKernelGGL(k_bfs_pos, gn, dim3(256), 0, stream, d_order, (uint32_t)n_nodes, d_pos);
    hipLaunchKernelGGL(k_ploc_flatten, gn, dim3(256), 0, stream, d_order, d_pos, nd, (uint32_t)n_nodes, d_flat, d_bad);
    LB_HIPCHK(hipEventRecord(ev1, stream));
    uint32_t bad = 0, tail[3] = {0, 0, 0};
    unsigned long long deepest_key = 0;
    uint32_t bad_vertex = 0;
    LB_HIPCHK(hipMemcpyAsync(&bad_vertex, d_scene + 6, 4, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipMemcpyAsync(tail, d_counts, 12, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipMemcpyAsync(&deepest_key, d_bkeys2 + (n_nodes - 1), 8, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipStreamSynchronize(stream));
    LB_HIPCHK(hipGetLastError());
    float ms = 0.f;
    LB_HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    cleanup();
    if (bad_vertex) return fail(CRT_ERR_INVALID, "ploc: a vertex coordinate is not finite or exceeds 1e18");
    if (tail[1] != (uint32_t)n_nodes) return fail(CRT_ERR_HIP, "ploc: node count is not 2n - 1");
    if (bad & 2u) return fail(CRT_ERR_LIMIT, "ploc: tree deeper than 56 levels");
    if (bad) return fail(CRT_ERR_HIP, "ploc: breadth-first renumbering is inconsistent");
    if (device_ms) *device_ms = ms;
    if (depth_out) *depth_out = (uint32_t)(deepest_key >> 56);
    return CRT_OK;
}

static uint32_t sah_small_of(uint32_t flags) {
    uint32_t s = (flags >> 8) & 0xffu;
    if (s == 0u) s = 8u;                                             // measured at 1 M triangles: 3.8 / 4.0 / 6.2 ms for 8 / 16 / 32, same tree quality
    return s < 8u ? 8u : s > SAH_SMALL ? SAH_SMALL : s;            // >= 8 keeps a workgroup's active nodes within SAH_LOCAL
}
static size_t sah_tmp_bytes(size_t n, uint32_t flags) {
    const size_t n_nodes = 2 * n - 1, cap = n / sah_small_of(flags) + 4;
    size_t sort2 = 0, scan = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort2, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    n_nodes, 0, 64, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, scan, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)0);
    auto P = DeviceArena::padded;
    return P(n * 24) + P(32) + 2 * P(n_nodes * 16) + P(n_nodes * 4) + 5 * P(n * 4) + 2 * P(cap * sizeof(SahWork)) + P(cap * SAH_BIN_WORDS * 4) +
           P((n / 2 + 2) * 12) + P(16) + 2 * P(n_nodes * 8) + 3 * P(n_nodes * 4) + P(4) + P(kMaxLevels * 4) + P(std::max<size_t>(sort2, 16)) +
           P(std::max<size_t>(scan, 16)) + 4096;
}

static int sah_build_on_device(const int32_t* d_vidx, uint32_t stride, const float* d_verts, uint32_t n, uint32_t flags, DeviceArena& tmp, crt_flatnode* d_flat,
                               uint32_t* d_tri_order, uint32_t* depth_out, float* device_ms, hipStream_t stream) {
    const uint32_t small = sah_small_of(flags);
    const size_t n_nodes = 2 * (size_t)n - 1, cap = (size_t)n / small + 4;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    };
    size_t sort2 = 0, scan_bytes = 0;
    LB_HIPCHK(rocprim::radix_sort_pairs(nullptr, sort2, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                        (uint32_t*)nullptr, n_nodes, 0, 64, stream));
    LB_HIPCHK(rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    float* d_leaf_box = tmp.take<float>((size_t)n * 6);
    uint32_t* d_scene = tmp.take<uint32_t>(8);
    PlocNodes nd;
    nd.lo = tmp.take<float4>(n_nodes);
    nd.hi = tmp.take<float4>(n_nodes);
    nd.parent2 = tmp.take<int>(n_nodes);
    uint32_t* d_idx2 = tmp.take<uint32_t>(n);
    uint32_t* d_pw0 = tmp.take<uint32_t>(n);
    uint32_t* d_pw1 = tmp.take<uint32_t>(n);
    uint32_t* d_fl = tmp.take<uint32_t>(n);
    uint32_t* d_scan = tmp.take<uint32_t>(n);
    SahWork* d_w0 = tmp.take<SahWork>(cap);
    SahWork* d_w1 = tmp.take<SahWork>(cap);
    uint32_t* d_bins = tmp.take<uint32_t>(cap * SAH_BIN_WORDS);
    uint32_t* d_small = tmp.take<uint32_t>(((size_t)n / 2 + 2) * 3);
    uint32_t* d_counters = tmp.take<uint32_t>(4);
    unsigned long long* d_bkeys = tmp.take<unsigned long long>(n_nodes);
    unsigned long long* d_bkeys2 = tmp.take<unsigned long long>(n_nodes);
    uint32_t* d_ids = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_order = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_pos = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_bad = tmp.take<uint32_t>(1);
    uint32_t* d_levels = tmp.take<uint32_t>(kMaxLevels);
    void* d_tmp2 = tmp.take<char>(std::max<size_t>(sort2, 16));
    void* d_tmp3 = tmp.take<char>(std::max<size_t>(scan_bytes, 16));
    if (!d_leaf_box || !d_scene || !nd.lo || !nd.hi || !nd.parent2 || !d_idx2 || !d_pw0 || !d_pw1 || !d_fl || !d_scan || !d_w0 || !d_w1 || !d_bins ||
        !d_small || !d_counters || !d_bkeys || !d_bkeys2 || !d_ids || !d_order || !d_pos || !d_bad || !d_levels || !d_tmp2 || !d_tmp3)
        return fail(CRT_ERR_NOMEM, "sah: temporary arena too small");
    const uint32_t scene_init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    const uint32_t counters_init[4] = {1u, 0u, 0u, 1u};          // node 0 is the root, and the first level's only work item
    LB_HIPCHK(hipEventCreate(&ev0));
    LB_HIPCHK(hipEventCreate(&ev1));
    LB_HIPCHK(hipMemcpyAsync(d_scene, scene_init, sizeof scene_init, hipMemcpyHostToDevice, stream));
    LB_HIPCHK(hipMemcpyAsync(d_counters, counters_init, sizeof counters_init, hipMemcpyHostToDevice, stream));
    LB_HIPCHK(hipMemsetAsync(d_bad, 0, 4, stream));
    const dim3 gt((n + 255u) / 256u);
    LB_HIPCHK(hipEventRecord(ev0, stream));
    hipLaunchKernelGGL(k_tri_bounds, dim3(std::min<uint32_t>(gt.x, 1024u)), dim3(256), 0, stream, d_vidx, stride, d_verts, n, d_leaf_box, d_scene);
    uint32_t* idx = d_tri_order; uint32_t* idx2 = d_idx2;
    uint32_t* pw = d_pw0; uint32_t* pw2 = d_pw1;
    SahWork* work = d_w0; SahWork* next = d_w1;
    hipLaunchKernelGGL(k_sah_init, gt, dim3(256), 0, stream, n, idx, pw, work, nd.parent2);
    uint32_t m = 1, n_small = 0, levels = 0, nodes = 1;              // node 0 is the root
    if (n <= small) {                                                // the root itself is a small node
        const uint32_t root_small[3] = {0u, 0u, n};
        LB_HIPCHK(hipMemcpyAsync(d_small, root_small, sizeof root_small, hipMemcpyHostToDevice, stream));
        LB_HIPCHK(hipStreamSynchronize(stream));
        m = 0; n_small = 1;
    }
    if (m > 0) LB_HIPCHK(hipMemsetAsync(d_bins, 0, cap * SAH_BIN_WORDS * 4, stream));
    // a level's list is at most twice the previous one and never longer than n / small (an active node owns more than
    // `small` triangles): that bound sizes the sweep's grid, and the host looks at the real count only from the level on at
    // which a balanced tree runs out of active nodes
    const dim3 gc((n + 256u * SAH_CHUNKS - 1u) / (256u * SAH_CHUNKS));
    uint32_t bound = 1, sync_from = 0;
    while (((size_t)small << sync_from) < (size_t)n) ++sync_from;
    while (m > 0) {
        hipLaunchKernelGGL(k_sah_cbounds, gc, dim3(256), 0, stream, idx, pw, n, d_leaf_box, work);
        hipLaunchKernelGGL(k_sah_bin, gc, dim3(256), 0, stream, idx, pw, n, d_leaf_box, work, d_bins);
        SahLists lists{d_counters, next, d_small, (uint32_t)cap, n / 2u + 2u, d_bad};
        hipLaunchKernelGGL(k_sah_sweep, dim3((bound + 3u) / 4u), dim3(256), 0, stream, work, d_bins, nd, lists, small);
        hipLaunchKernelGGL(k_sah_advance, dim3(1), dim3(1), 0, stream, d_counters);
        hipLaunchKernelGGL(k_sah_flags, gt, dim3(256), 0, stream, idx, pw, n, d_leaf_box, work, d_fl);
        LB_HIPCHK(rocprim::exclusive_scan(d_tmp3, scan_bytes, d_fl, d_scan, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_sah_scatter, gt, dim3(256), 0, stream, idx, pw, n, work, d_fl, d_scan, idx2, pw2);
        std::swap(idx, idx2); std::swap(pw, pw2); std::swap(work, next);
        bound = (uint32_t)std::min<size_t>(2 * (size_t)bound, cap);
        if (++levels > 512u) { cleanup(); return fail(CRT_ERR_HIP, "sah: did not converge"); }
        if (levels >= sync_from) {
            uint32_t c[4] = {0, 0, 0, 0}, flags_now = 0;
            LB_HIPCHK(hipMemcpyAsync(c, d_counters, 16, hipMemcpyDeviceToHost, stream));
            LB_HIPCHK(hipMemcpyAsync(&flags_now, d_bad, 4, hipMemcpyDeviceToHost, stream));
            LB_HIPCHK(hipStreamSynchronize(stream));
            if ((flags_now & 4u) || c[3] > cap || c[2] > n / 2u + 1u || c[0] > n_nodes) { cleanup(); return fail(CRT_ERR_HIP, "sah: work list overflow"); }
            nodes = c[0]; n_small = c[2]; m = c[3];
        }
    }
    if (n_small) {
        // phase B node numbers: exclusive prefix of 2 (c - 1) over the small nodes (d_fl / d_scan are free again)
        hipLaunchKernelGGL(k_sah_small_counts, dim3((n_small + 255u) / 256u), dim3(256), 0, stream, d_small, n_small, d_fl);
        LB_HIPCHK(rocprim::exclusive_scan(d_tmp3, scan_bytes, d_fl, d_scan, 0u, (size_t)n_small, rocprim::plus<uint32_t>(), stream));
        const dim3 gs((n_small + 63u) / 64u);
        if (small <= 8u) hipLaunchKernelGGL(k_sah_small<8>, gs, dim3(64), 0, stream, d_small, n_small, idx, d_leaf_box, nd, d_scan, nodes);
        else if (small <= 16u) hipLaunchKernelGGL(k_sah_small<16>, gs, dim3(64), 0, stream, d_small, n_small, idx, d_leaf_box, nd, d_scan, nodes);
        else hipLaunchKernelGGL(k_sah_small<SAH_SMALL>, gs, dim3(64), 0, stream, d_small, n_small, idx, d_leaf_box, nd, d_scan, nodes);
    }
    if (idx != d_tri_order) LB_HIPCHK(hipMemcpyAsync(d_tri_order, idx, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    const dim3 gn((uint32_t)((n_nodes + 255) / 256));
    hipLaunchKernelGGL(k_ploc_bfs_keys, gn, dim3(256), 0, stream, nd.parent2, (uint32_t)n_nodes, d_bkeys, d_ids, d_bad);
    LB_HIPCHK(rocprim::radix_sort_pairs(d_tmp2, sort2, d_bkeys, d_bkeys2, d_ids, d_order, n_nodes, 0, 64, stream));
    hipLaunchKernelGGL(k_bfs_pos, gn, dim3(256), 0, stream, d_order, (uint32_t)n_nodes, d_pos);
    hipLaunchKernelGGL(k_sah_flatten, gn, dim3(256), 0, stream, d_order, d_pos, nd, d_tri_order, d_leaf_box, (uint32_t)n_nodes, d_flat, d_bad);
    hipLaunchKernelGGL(k_level_starts, gn, dim3(256), 0, stream, d_bkeys2, (uint32_t)n_nodes, d_levels, (uint32_t)kMaxLevels, 56u);
    uint32_t bad = 0;
    unsigned long long deepest_key = 0;
    LB_HIPCHK(hipMemcpyAsync(&deepest_key, d_bkeys2 + (n_nodes - 1), 8, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipStreamSynchronize(stream));
    const uint32_t n_levels = (uint32_t)(deepest_key >> 56) + 1u;
    std::vector<uint32_t> level_start(n_levels + 1);
    LB_HIPCHK(hipMemcpyAsync(level_start.data(), d_levels, n_levels * 4, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipStreamSynchronize(stream));
    level_start[n_levels] = (uint32_t)n_nodes;
    for (uint32_t l = n_levels; l-- > 0;) {
        const uint32_t cnt = level_start[l + 1] - level_start[l];
        hipLaunchKernelGGL(k_refit_level, dim3((cnt + 255) / 256), dim3(256), 0, stream, d_flat, level_start[l], level_start[l + 1]);
    }
    LB_HIPCHK(hipEventRecord(ev1, stream));
    uint32_t bad_vertex = 0;
    LB_HIPCHK(hipMemcpyAsync(&bad_vertex, d_scene + 6, 4, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipStreamSynchronize(stream));
    LB_HIPCHK(hipGetLastError());
    float ms = 0.f;
    LB_HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    cleanup();
    if (bad_vertex) return fail(CRT_ERR_INVALID, "sah: a vertex coordinate is not finite or exceeds 1e18");
    if (bad & 4u) return fail(CRT_ERR_HIP, "sah: work list overflow");
    if (bad & 2u) return fail(CRT_ERR_LIMIT, "sah: tree deeper than 56 levels");
    if (bad) return fail(CRT_ERR_HIP, "sah: breadth-first renumbering is inconsistent");
    if (device_ms) *device_ms = ms;
    if (depth_out) *depth_out = n_levels - 1u;
    return CRT_OK;
}

int lbvh_build_on_device(const int32_t* d_vidx, uint32_t stride, const float* d_verts, uint32_t n_tris_u, uint32_t flags, DeviceArena& tmp,
                         crt_flatnode* d_flat, uint32_t* d_tri_order, uint32_t* depth_out, float* device_ms, hipStream_t stream) {
    if ((flags & CRT_GPU_BUILD_SAH) && n_tris_u > 1u)
        return sah_build_on_device(d_vidx, stride, d_verts, n_tris_u, flags, tmp, d_flat, d_tri_order, depth_out, device_ms, stream);
    if ((flags & CRT_GPU_BUILD_PLOC) && n_tris_u > 1u)
        return ploc_build_on_device(d_vidx, stride, d_verts, n_tris_u, flags, tmp, d_flat, d_tri_order, depth_out, device_ms, stream);
    const size_t n_tris = n_tris_u;
    const int n = (int)n_tris;
    const size_t n_nodes = 2 * n_tris - 1;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    };
    size_t tmp_bytes = 0, tmp2_bytes = 0;
    LB_HIPCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, n_tris, 0, 64, stream));
    LB_HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp2_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                        (uint32_t*)nullptr, n_nodes, 0, 64, stream));
    float* d_leaf_box = tmp.take<float>(n_tris * 6);
    uint32_t* d_scene = tmp.take<uint32_t>(8);
    unsigned long long* d_keys = tmp.take<unsigned long long>(n_tris);
    unsigned long long* d_sorted = tmp.take<unsigned long long>(n_tris);
    int2* d_child = tmp.take<int2>(std::max<size_t>(n_tris - 1, 1));
    int* d_parent = tmp.take<int>(n_nodes);
    uint32_t* d_levels = tmp.take<uint32_t>(kMaxLevels);
    int* d_first = tmp.take<int>(std::max<size_t>(n_tris - 1, 1));
    unsigned long long* d_bkeys = tmp.take<unsigned long long>(n_nodes);
    unsigned long long* d_bkeys2 = tmp.take<unsigned long long>(n_nodes);
    uint32_t* d_ids = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_order = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_pos = tmp.take<uint32_t>(n_nodes);
    uint32_t* d_bad = tmp.take<uint32_t>(1);
    void* d_tmp = tmp.take<char>(std::max<size_t>(tmp_bytes, 16));
    void* d_tmp2 = tmp.take<char>(std::max<size_t>(tmp2_bytes, 16));
    if (!d_leaf_box || !d_scene || !d_keys || !d_sorted || !d_child || !d_parent || !d_levels || !d_first || !d_bkeys || !d_bkeys2 || !d_ids ||
        !d_order || !d_pos || !d_bad || !d_tmp || !d_tmp2)
        return fail(CRT_ERR_NOMEM, "lbvh: temporary arena too small");
    const uint32_t scene_init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    LB_HIPCHK(hipEventCreate(&ev0));
    LB_HIPCHK(hipEventCreate(&ev1));
    LB_HIPCHK(hipMemcpyAsync(d_scene, scene_init, sizeof scene_init, hipMemcpyHostToDevice, stream));
    LB_HIPCHK(hipMemsetAsync(d_bad, 0, 4, stream));

    const uint32_t g = (uint32_t)((n_tris + 255) / 256);
    LB_HIPCHK(hipEventRecord(ev0, stream));
    hipLaunchKernelGGL(k_tri_bounds, dim3(std::min<uint32_t>(g, 1024u)), dim3(256), 0, stream, d_vidx, stride, d_verts, (uint32_t)n, d_leaf_box, d_scene);
    hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, stream, d_leaf_box, d_scene, (uint32_t)n, d_keys);
    LB_HIPCHK(rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_keys, d_sorted, n_tris, 0, 64, stream));
    hipLaunchKernelGGL(k_tri_order, dim3(g), dim3(256), 0, stream, d_sorted, (uint32_t)n, d_tri_order);
    if (n > 1) hipLaunchKernelGGL(k_radix_tree, dim3((uint32_t)((n_tris - 1 + 255) / 256)), dim3(256), 0, stream, d_sorted, n, d_child, d_parent, d_first);
    else { const int minus1 = -1; LB_HIPCHK(hipMemcpyAsync(d_parent, &minus1, 4, hipMemcpyHostToDevice, stream)); LB_HIPCHK(hipStreamSynchronize(stream)); }
    const dim3 gn((uint32_t)((n_nodes + 255) / 256));
    hipLaunchKernelGGL(k_bfs_keys, gn, dim3(256), 0, stream, d_parent, d_first, n, d_bkeys, d_ids);
    LB_HIPCHK(rocprim::radix_sort_pairs(d_tmp2, tmp2_bytes, d_bkeys, d_bkeys2, d_ids, d_order, n_nodes, 0, 64, stream));
    hipLaunchKernelGGL(k_bfs_pos, gn, dim3(256), 0, stream, d_order, (uint32_t)n_nodes, d_pos);
    hipLaunchKernelGGL(k_flatten, gn, dim3(256), 0, stream, d_order, d_pos, d_child, d_sorted, d_leaf_box, n, d_flat, d_bad);
    hipLaunchKernelGGL(k_level_starts, gn, dim3(256), 0, stream, d_bkeys2, (uint32_t)n_nodes, d_levels, (uint32_t)kMaxLevels);
    unsigned long long deepest_key = 0;
    std::vector<uint32_t> level_start;
    LB_HIPCHK(hipMemcpyAsync(&deepest_key, d_bkeys2 + (n_nodes - 1), 8, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipStreamSynchronize(stream));
    const uint32_t n_levels = (uint32_t)(deepest_key >> 32) + 1u;
    if (n_levels > kMaxLevels) { cleanup(); return fail(CRT_ERR_LIMIT, "crt_lbvh_build: tree deeper than 4096 levels"); }
    level_start.resize(n_levels + 1);
    LB_HIPCHK(hipMemcpyAsync(level_start.data(), d_levels, n_levels * 4, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipStreamSynchronize(stream));
    level_start[n_levels] = (uint32_t)n_nodes;
    for (uint32_t l = n_levels; l-- > 0;) {
        const uint32_t cnt = level_start[l + 1] - level_start[l];
        hipLaunchKernelGGL(k_refit_level, dim3((cnt + 255) / 256), dim3(256), 0, stream, d_flat, level_start[l], level_start[l + 1]);
    }
    LB_HIPCHK(hipEventRecord(ev1, stream));
    uint32_t bad = 0, bad_vertex = 0;
    LB_HIPCHK(hipMemcpyAsync(&bad_vertex, d_scene + 6, 4, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream));
    LB_HIPCHK(hipStreamSynchronize(stream));
    LB_HIPCHK(hipGetLastError());
    float ms = 0.f;
    LB_HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    cleanup();
    if (bad_vertex) return fail(CRT_ERR_INVALID, "crt_lbvh_build: a vertex coordinate is not finite or exceeds 1e18");
    if (bad) return fail(CRT_ERR_HIP, "crt_lbvh_build: breadth-first renumbering is inconsistent");
    if (device_ms) *device_ms = ms;
    if (depth_out) *depth_out = n_levels - 1u;           // the deepest level holds leaves only
    return CRT_OK;
}

// crt_warmup: load this translation unit's code object on the current device (device_build.hpp)
int warm_lbvh_kernels() {
    hipFuncAttributes a;
    hipError_t e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_tri_order))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_tri_bounds))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_morton))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_radix_tree))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_bfs_keys))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_bfs_pos))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_flatten))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_level_starts))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_refit_level))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ploc_init))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ploc_nn))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ploc_flags))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ploc_apply))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ploc_tail))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ploc_bfs_keys))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ploc_flatten))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_init))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_cbounds))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_bin))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_advance))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_sweep))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_flags))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_scatter))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_small_counts))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_small<8>))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_small<16>))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_small<SAH_SMALL>))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sah_flatten))) != hipSuccess) return (int)e;
    return 0;
}

}  // namespace crt

extern "C" {

int crt_lbvh_build(const crt_triangle* tris, size_t n_tris, const float* vertices, size_t n_vertices, uint32_t flags, crt_sbvh** out) {
    if (!out) return fail(CRT_ERR_INVALID, "crt_lbvh_build: null out");
    *out = nullptr;
    if (!tris || !vertices || n_tris == 0) return fail(CRT_ERR_INVALID, "crt_lbvh_build: empty input");
    if (n_tris >= (1u << 23)) return fail(CRT_ERR_LIMIT, "crt_lbvh_build: more than 2^23 triangles (the 2 n - 1 FlatNode links are floats, exact below 2^24)");
    if (flags & ~(uint32_t)(CRT_GPU_BUILD_PLOC | CRT_GPU_BUILD_SAH | 0xff00u)) return fail(CRT_ERR_INVALID, "crt_lbvh_build: unknown flags");
    for (size_t i = 0; i < n_tris; ++i)
        for (int j = 0; j < 3; ++j)
            if (tris[i].v[j] < 0 || (size_t)tris[i].v[j] >= n_vertices) return fail(CRT_ERR_INVALID, "crt_lbvh_build: vertex index out of range");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(CRT_ERR_NO_DEVICE, "crt_lbvh_build: no HIP device visible");

    const auto t_begin = std::chrono::steady_clock::now();
    const size_t n_nodes = 2 * n_tris - 1;
    crt::DeviceArena arena;
    auto cleanup = [&]() { arena.release(); };
    auto P = crt::DeviceArena::padded;
    LB_HIPCHK(arena.reserve(crt::lbvh_tmp_bytes(n_tris, flags) + P(n_tris * 12) + P(n_vertices * 12) + P(n_nodes * sizeof(crt_flatnode)) + P(n_tris * 4)));
    std::vector<int32_t> vidx;
    crt_sbvh* h = nullptr;
    try {
        vidx.resize(3 * n_tris);
        h = new crt_sbvh;
        h->bvh.flat_nodes.resize(n_nodes);
        h->bvh.triangle_indices.resize(n_tris);
        h->bvh.triangles.resize(n_tris);
    } catch (const std::exception& e) {
        delete h; cleanup();
        return fail(CRT_ERR_NOMEM, std::string("crt_lbvh_build: ") + e.what());
    }
    for (size_t i = 0; i < n_tris; ++i) for (int j = 0; j < 3; ++j) vidx[3 * i + j] = tris[i].v[j];
    int32_t* d_vidx = arena.take<int32_t>(3 * n_tris);
    float* d_verts = arena.take<float>(3 * n_vertices);
    crt_flatnode* d_flat = arena.take<crt_flatnode>(n_nodes);
    uint32_t* d_tri_order = arena.take<uint32_t>(n_tris);
    if (!d_vidx || !d_verts || !d_flat || !d_tri_order) { delete h; cleanup(); return fail(CRT_ERR_NOMEM, "crt_lbvh_build: arena"); }
    hipError_t e = hipMemcpy(d_vidx, vidx.data(), vidx.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_verts, vertices, n_vertices * 12, hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete h; cleanup(); return fail(CRT_ERR_HIP, std::string("crt_lbvh_build: upload: ") + hipGetErrorString(e)); }
    uint32_t depth = 0;
    int rc = crt::lbvh_build_on_device(d_vidx, 3, d_verts, (uint32_t)n_tris, flags, arena, d_flat, d_tri_order, &depth, &g_last_device_ms, (hipStream_t)0);
    if (rc) { delete h; cleanup(); return rc; }
    crt::SBVH& b = h->bvh;
    static_assert(sizeof(int32_t) == sizeof(uint32_t), "");
    if (hipMemcpy(b.flat_nodes.data(), d_flat, n_nodes * sizeof(crt_flatnode), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(b.triangle_indices.data(), d_tri_order, n_tris * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        delete h; cleanup();
        return fail(CRT_ERR_HIP, "crt_lbvh_build: copy back failed");
    }
    cleanup();
    b.depth = (int)depth;
    for (size_t jx = 0; jx < n_tris; ++jx) b.triangles[jx] = tris[b.triangle_indices[jx]];
    g_last_total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *out = h;
    return CRT_OK;
}

void crt_lbvh_last_build_ms(float* device_ms, float* total_ms) {
    if (device_ms) *device_ms = g_last_device_ms;
    if (total_ms) *total_ms = g_last_total_ms;
}

}  // extern "C"
