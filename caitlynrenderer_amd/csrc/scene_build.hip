// Scene assembly kernels of crt_scene_create's build-on-device path: the steps the host does with loops over the
// triangle array for host-built scenes (index validation, re-ordering into leaf order, pre-gathered intersection
// records), done where the builders left their output.  The record arithmetic is the same two fp32 subtractions per edge
// as the host loop in crt_device.cpp (path_trace.fs:337-338), so the records are bit-identical to an uploaded scene's.
#include <hip/hip_runtime.h>

#include "device_build.hpp"

namespace crt {
namespace {

__global__ void k_validate_triangles(const crt_triangle* __restrict__ tris, uint32_t n, uint32_t n_vertices, uint32_t n_materials, uint32_t n_normals,
                                     uint32_t n_texcoords, const float* __restrict__ materials, int have_tex, uint32_t* flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const crt_triangle t = tris[i];
    uint32_t bad = 0;
    for (int j = 0; j < 3; ++j) if (t.v[j] < 0 || (uint32_t)t.v[j] >= n_vertices) bad |= 1u;
    if (t.v[3] < 0 || (uint32_t)t.v[3] >= n_materials) bad |= 2u;
    if (t.vn[3] != 0) for (int j = 0; j < 3; ++j) if (t.vn[j] < 0 || (uint32_t)t.vn[j] >= n_normals) bad |= 4u;
    if (have_tex && !(bad & 2u) && materials[16 * (size_t)t.v[3] + 12] != -1.0f)
        for (int j = 0; j < 3; ++j) if (t.vt[j] < 0 || (uint32_t)t.vt[j] >= n_texcoords) bad |= 8u;
    if (bad) atomicOr(flag, bad);
}

__device__ __forceinline__ void make_record(const crt_triangle& t, const float* __restrict__ verts, int32_t id, int32_t slot, float4& a, float4& b, float4& c) {
    const float* v0 = verts + 3 * (size_t)t.v[0];
    const float* v1 = verts + 3 * (size_t)t.v[1];
    const float* v2 = verts + 3 * (size_t)t.v[2];
    a = make_float4(v0[0], v0[1], v0[2], __int_as_float(id));
    b = make_float4(v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2], __int_as_float(slot));
    c = make_float4(v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2], __int_as_float(t.v[3]));
}

__global__ void k_gather_slots(const crt_triangle* __restrict__ in, const uint32_t* __restrict__ order, const float* __restrict__ verts, uint32_t n,
                               crt_triangle* __restrict__ out, float4* __restrict__ recs2) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const uint32_t src = order[slot];
    const crt_triangle t = in[src];
    out[slot] = t;
    if (recs2) {
        float4 a, b, c;
        make_record(t, verts, (int32_t)src, (int32_t)slot, a, b, c);
        recs2[3 * (size_t)slot] = a; recs2[3 * (size_t)slot + 1] = b; recs2[3 * (size_t)slot + 2] = c;
    }
}

__global__ void k_gather_records(const crt_triangle* __restrict__ in, const uint32_t* __restrict__ order, const int32_t* __restrict__ tri_slots,
                                 const float* __restrict__ verts, uint32_t n, float4* __restrict__ recs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t slot = tri_slots[i];
    const uint32_t src = order[slot];
    const crt_triangle t = in[src];
    float4 a, b, c;
    make_record(t, verts, (int32_t)src, slot, a, b, c);
    recs[3 * (size_t)i] = a; recs[3 * (size_t)i + 1] = b; recs[3 * (size_t)i + 2] = c;
}

// dst[i * rows_out + r] = src[i * rows_in + r] for r < rows_in; the padding rows are zeroed (never read)
__global__ void k_restride(const uint4* __restrict__ src, uint32_t rows_in, uint4* __restrict__ dst, uint32_t rows_out, uint64_t n_items) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t / rows_out;
    const uint32_t r = (uint32_t)(t % rows_out);
    if (i >= n_items) return;
    dst[t] = r < rows_in ? src[i * rows_in + r] : make_uint4(0u, 0u, 0u, 0u);
}

// The child planes of every node as floats, for the uniform node steps of the traversal (rt_kernels.hip): 12 rows of float4 per node,
// row = half * 6 + axis * 2 + side (side 0 = lo planes, 1 = hi planes), the four children 4 * half .. 4 * half + 3 of that row.
// (float)byte is exact, so the traversal's fma sees the operand v_cvt_f32_ubyte would have produced.
__global__ void k_expand_planes(const uint4* __restrict__ nodes, uint32_t node_rows, float4* __restrict__ planes, uint64_t n_nodes) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_nodes * 12u) return;
    const uint64_t i = t / 12u;
    const uint32_t r = (uint32_t)(t - i * 12u), half = r / 6u, axis = (r % 6u) >> 1, side = r & 1u;
    const uint4 row = nodes[i * node_rows + 2u + axis];       // (lo[0..3], lo[4..7], hi[0..3], hi[4..7])
    const uint32_t w = side ? (half ? row.w : row.z) : (half ? row.y : row.x);
    planes[t] = make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24));
}

inline dim3 grid_for(uint32_t n) { return dim3((n + 255u) / 256u ? (n + 255u) / 256u : 1u); }

}  // namespace

void launch_validate_triangles(const crt_triangle* d_tris, uint32_t n, uint32_t n_vertices, uint32_t n_materials, uint32_t n_normals,
                               uint32_t n_texcoords, const float* d_materials, int have_tex, uint32_t* d_flag, hipStream_t stream) {
    hipLaunchKernelGGL(k_validate_triangles, grid_for(n), dim3(256), 0, stream, d_tris, n, n_vertices, n_materials, n_normals, n_texcoords, d_materials,
                       have_tex, d_flag);
}
void launch_restride(const void* d_src, uint32_t rows_in, void* d_dst, uint32_t rows_out, uint64_t n_items, hipStream_t stream) {
    const uint64_t threads = n_items * rows_out;
    hipLaunchKernelGGL(k_restride, dim3((uint32_t)((threads + 255u) / 256u)), dim3(256), 0, stream, static_cast<const uint4*>(d_src), rows_in,
                       static_cast<uint4*>(d_dst), rows_out, n_items);
}
void launch_expand_planes(const void* d_nodes, uint32_t node_rows, void* d_planes, uint64_t n_nodes, hipStream_t stream) {
    const uint64_t threads = n_nodes * 12u;
    hipLaunchKernelGGL(k_expand_planes, dim3((uint32_t)((threads + 255u) / 256u)), dim3(256), 0, stream, static_cast<const uint4*>(d_nodes), node_rows,
                       static_cast<float4*>(d_planes), n_nodes);
}
void launch_gather_slots(const crt_triangle* d_in, const uint32_t* d_tri_order, const float* d_verts, uint32_t n_slots, crt_triangle* d_slot_tris,
                         float4* d_recs2, hipStream_t stream) {
    hipLaunchKernelGGL(k_gather_slots, grid_for(n_slots), dim3(256), 0, stream, d_in, d_tri_order, d_verts, n_slots, d_slot_tris, d_recs2);
}
void launch_gather_records(const crt_triangle* d_in, const uint32_t* d_tri_order, const int32_t* d_tri_slots, const float* d_verts, uint32_t n_tris8,
                           float4* d_recs, hipStream_t stream) {
    hipLaunchKernelGGL(k_gather_records, grid_for(n_tris8), dim3(256), 0, stream, d_in, d_tri_order, d_tri_slots, d_verts, n_tris8, d_recs);
}

// crt_warmup: load this translation unit's code object on the current device (device_build.hpp)
int warm_scene_build_kernels() {
    hipFuncAttributes a;
    hipError_t e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_validate_triangles))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_gather_slots))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_gather_records))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_restride))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_expand_planes))) != hipSuccess) return (int)e;
    return 0;
}

}  // namespace crt
