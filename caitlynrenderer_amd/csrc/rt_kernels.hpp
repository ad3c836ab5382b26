// Kernel argument blocks and launcher prototypes shared by rt_kernels.hip and crt_device.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crt {

// Traversal stack entries per ray, kept in LDS (reference LOCAL_STACK_SIZE, cwbvh.fs:374).
// crt_scene_create refuses a CWBVH deeper than this.
#define CRT_STACK_ENTRIES 16
#define CRT_TRACE_BLOCK 256

struct TraceArgs {
    const uint4* nodes;        // 5 x 16 B per node8 (cwbvh.fs:484-488)
    const float4* tris;        // 3 x 16 B per CWBVH triangle: (v0|orig_id) (e1|bvh2 slot) (e2|material)
    const float4* rays;        // crt_ray: (o, tmax) (d, payload)
    float4* hits;              // crt_hit: (t, u, v, tri)
    uint32_t* stats;           // optional: nodes | tris << 16
    const uint32_t* count_ptr; // device-side ray count, or null
    uint32_t n;
    uint32_t out_orig_id;      // 1: hit.tri = original triangle id, 0: CWBVH triangle index
};

struct PathBuffers {           // indexed by local pixel
    float4* L;                 // radiance so far, prev_pdf
    float4* T;                 // throughput, is_specular
    float2* seed;              // shader RNG state (path_trace.fs:27)
    float4* C;                 // pending NEE contribution
};

struct FrameArgs {
    const uint2* tile_xy;      // local tile -> (tile x, tile y)
    uint32_t n_local_pixels;   // n_local_tiles * tile * tile
    uint32_t tile, width, height;
    uint32_t jitter;
    float rv;                  // randomVector.x * randomVector.y (path_trace.fs:40)
    float tan_fov, aspect_tan; // tan(fov/2), W/H*tan(fov/2) (path_trace.fs:1041-1043)
    float cam_pos[3], cam_right[3], cam_up[3], cam_forward[3];
};

struct ShadeArgs {
    const float4* rays_in;
    const float4* hits;
    const uint32_t* count_in;
    float4* rays_next;   uint32_t* count_next;
    float4* rays_shadow; uint32_t* count_shadow;
    const float4* tris;
    const int4* triangles;     // 3 x int4 per BVH2-ordered triangle (v, vn, vt)
    const float* normals;
    const float4* materials;
    const float* lights;
    int32_t n_lights;
    float rv;
    uint32_t last_segment;
};

void launch_trace(const TraceArgs& a, int mode, bool stats, uint32_t grid, hipStream_t stream);
void launch_reduce_stats(const uint32_t* stats, const uint32_t* count_ptr, uint32_t n, unsigned long long* out, uint32_t grid,
                         hipStream_t stream);
void launch_raygen(const FrameArgs& f, const PathBuffers& pb, float4* rays, uint32_t grid, hipStream_t stream);
void launch_shade(const ShadeArgs& a, const PathBuffers& pb, uint32_t grid, hipStream_t stream);
void launch_shadow_resolve(const float4* rays_shadow, const float4* hits, const uint32_t* count, const PathBuffers& pb,
                           uint32_t grid, hipStream_t stream);
void launch_accumulate(float* sum, const PathBuffers& pb, uint32_t n, uint32_t grid, hipStream_t stream);
void launch_untile(const FrameArgs& f, const float* packed, float* linear, uint32_t grid, hipStream_t stream);
void launch_resolve(const float* linear, uint32_t n_pixels, float inv_count, uint8_t* rgba, uint32_t grid, hipStream_t stream);

}  // namespace crt
