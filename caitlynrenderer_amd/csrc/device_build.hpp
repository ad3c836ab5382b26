// Device-resident halves of the GPU builders, shared by the host-array entry points
// (crt_lbvh_build, crt_cwbvh_convert_device) and by crt_scene_create's build-on-device path, where the BVH2, the CWBVH
// and the intersection records are produced in HBM and never cross PCIe (the reference's own flow is
// build on the host, Scene.h:929-958, then one upload, :1000-1062).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crt.h"

namespace crt {

// One hipMalloc carved into 256-byte aligned pieces and freed as a whole: the builders need ~25 temporaries, and a
// hipMalloc / hipFree pair per temporary was most of the wall time of a call.
struct DeviceArena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    DeviceArena() = default;
    DeviceArena(const DeviceArena&) = delete;
    DeviceArena& operator=(const DeviceArena&) = delete;
    ~DeviceArena() { release(); }
    hipError_t reserve(size_t bytes) {
        release();
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), bytes ? bytes : 256);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (base) (void)hipFree(base); base = nullptr; cap = used = 0; }
    void reset() { used = 0; }
    template <typename T> T* take(size_t count) {          // nullptr when the arena is exhausted
        const size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        if (used + bytes > cap) return nullptr;
        T* p = reinterpret_cast<T*>(base + used);
        used += bytes;
        return p;
    }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
};

// ---- linear BVH (lbvh.hip) ----
// d_vidx: vertex indices of triangle i at d_vidx[stride * i + 0..2] (stride 3 = packed, 12 = the crt_triangle array itself).
// d_flat (2n - 1 nodes, BFS order, one triangle per leaf) and d_tri_order (leaf slot -> input triangle) are caller-owned
// device buffers; temporaries come from `tmp` (lbvh_tmp_bytes).  Synchronises the stream twice (level table).
// flags: CRT_GPU_BUILD_* of include/crt.h (0 = linear BVH; CRT_GPU_BUILD_PLOC | radius << 8 = PLOC, whose boxes need no refit).
size_t lbvh_tmp_bytes(size_t n_tris, uint32_t flags);
int lbvh_build_on_device(const int32_t* d_vidx, uint32_t stride, const float* d_verts, uint32_t n_tris, uint32_t flags, DeviceArena& tmp,
                         crt_flatnode* d_flat, uint32_t* d_tri_order, uint32_t* depth_out, float* device_ms, hipStream_t stream);

// ---- BVH2 -> CWBVH (cwbvh_device.hip) ----
// d_tri_slots (n_slots) is caller-owned; *d_nodes_out is hipMalloc'ed here once the node count is known (caller frees);
// d_child_bvh2_out may be null (the debug child map is then not produced).  Same bytes as the host converter.
size_t cwbvh_tmp_bytes(size_t n_bvh2_nodes, size_t n_slots);
int cwbvh_convert_on_device(const crt_flatnode* d_bvh2, uint32_t n_nodes, uint32_t n_slots, DeviceArena& tmp, int32_t* d_tri_slots,
                            crt_node8** d_nodes_out, int32_t** d_child_bvh2_out, uint32_t* n8_out, uint32_t* depth_out, float* device_ms,
                            hipStream_t stream);

// ---- scene assembly on the device (scene_build.hip) ----
// Index validation of the uploaded crt_triangle array (what crt_scene_create checks on the host for host-built scenes):
// *d_flag |= 1 vertex index, 2 material index, 4 normal index, 8 texcoord index of a textured material out of range.
void launch_validate_triangles(const crt_triangle* d_tris, uint32_t n, uint32_t n_vertices, uint32_t n_materials, uint32_t n_normals,
                               uint32_t n_texcoords, const float* d_materials /* 16 floats each */, int have_tex, uint32_t* d_flag, hipStream_t stream);
// items of rows_in 16-byte rows -> items of rows_out rows (padding zeroed): the line-aligned device copies of nodes and records
void launch_restride(const void* d_src, uint32_t rows_in, void* d_dst, uint32_t rows_out, uint64_t n_items, hipStream_t stream);
// 12 float4 rows per node: the child planes as floats (uniform node steps; layout in scene_build.hip k_expand_planes)
void launch_expand_planes(const void* d_nodes, uint32_t node_rows, void* d_planes, uint64_t n_nodes, hipStream_t stream);
// d_slot_tris[slot] = d_in[tri_order[slot]] (the triangle array in BVH2 leaf order, as sbvh.h:130-139 leaves it) and the
// slot-ordered intersection records (v0 | original id) (e1 | slot) (e2 | material) of the BVH2 walk (may be null).
void launch_gather_slots(const crt_triangle* d_in, const uint32_t* d_tri_order, const float* d_verts, uint32_t n_slots, crt_triangle* d_slot_tris,
                         float4* d_recs2, hipStream_t stream);
// CWBVH-ordered intersection records: record i describes slot d_tri_slots[i].
void launch_gather_records(const crt_triangle* d_in, const uint32_t* d_tri_order, const int32_t* d_tri_slots, const float* d_verts, uint32_t n_tris8,
                           float4* d_recs, hipStream_t stream);

// ---- code-object warm-up (crt_warmup) ----
// HIP loads a code object the first time one of its kernels is looked up; each of these asks for the attributes of its
// translation unit's kernels, which loads that unit's code object on the current device (~ms each) without launching anything.
int warm_lbvh_kernels();
int warm_cwbvh_kernels();
int warm_scene_build_kernels();

}  // namespace crt
